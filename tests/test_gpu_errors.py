"""The C boundary under failure (include/rslf_hip.h: 'never throws across the boundary'): an exception raised inside a
device worker of rslf_multi_*, a std::bad_alloc there, and a std::thread that cannot be started all come back as a
status (or as a completed call), leave nothing running, and the very next call on the same object is bit-identical to
a clean run.  Armed through rslf_debug_inject (process-wide, off by default)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PLANES = ("edge_confidence", "edge_mask", "disp_confidence", "depth", "rbar", "depth_idx", "score", "depth_raw")


@pytest.fixture(scope="module")
def rs():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from remotesensingproject_amd import depth
    return depth


def _epis(V=19, S=7, U=66):
    from remotesensingproject_amd.synth import make_lightfield
    vol, _ = make_lightfield(U, V, S, 1, seed=3, dmin=-1.0, dmax=2.0, band=3)
    return [vol[v, :, :, 0] for v in range(V)]


@pytest.mark.parametrize("site,status", [("worker", -6), ("alloc", -5)])
def test_a_throwing_worker_becomes_a_status(rs, site, status):
    from remotesensingproject_amd import _lib
    L = _lib.lib()
    epis = _epis()
    m = rs.MultiDevice([0, 0, 0])
    m.set_chunk_rows(4)
    clean = m.depth1d_pile(epis, -1.0, 2.0, 10, epi_scale_factor=1.0)
    assert L.rslf_debug_inject(site.encode(), 2) == 0     # two of the three workers fail, one completes
    with pytest.raises(_lib.RslfError) as ei:
        m.depth1d_pile(epis, -1.0, 2.0, 10, epi_scale_factor=1.0)
    assert ei.value.status == status
    assert ("injected" in str(ei.value)) or ("bad_alloc" in str(ei.value))
    assert L.rslf_debug_inject(site.encode(), 0) == 0
    again = m.depth1d_pile(epis, -1.0, 2.0, 10, epi_scale_factor=1.0)   # the object is intact
    for k in PLANES:
        assert np.array_equal(again[k], clean[k]), k
    m.close()


def test_without_threads_the_caller_does_the_work(rs):
    """std::thread's constructor throwing std::system_error (the GPU boxes cap a process's threads) must neither
    terminate the process nor lose work: the device workers, the pinned gather and the host maximum then run on the
    calling thread."""
    from remotesensingproject_amd import _lib
    L = _lib.lib()
    epis = _epis()
    raw = [np.ascontiguousarray(e * np.float32(37.0)) for e in epis]   # separate heap blocks + the default scale (host max)
    m = rs.MultiDevice([0, 0])
    m.set_chunk_rows(5)
    clean = m.depth1d_pile(raw, -1.0, 2.0, 10, epi_scale_factor=-1.0)
    assert L.rslf_debug_inject(b"thread_create", 1000) == 0
    try:
        got = m.depth1d_pile(raw, -1.0, 2.0, 10, epi_scale_factor=-1.0)
    finally:
        assert L.rslf_debug_inject(b"thread_create", 0) == 0
    for k in PLANES:
        assert np.array_equal(got[k], clean[k]), k
    assert m.stats.pixels_scanned > 0
    m.close()


def test_peer_access_matrix_of_one_gpu(rs):
    m = rs.MultiDevice([0, 0])
    assert m.peer_access() == [[1, 1], [1, 1]]     # two workers on one GPU reach each other's memory by definition
    from remotesensingproject_amd import _lib
    assert _lib.lib().rslf_multi_peer_access(m._h, 0, 5) == -1
    m.close()
