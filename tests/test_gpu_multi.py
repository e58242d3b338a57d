"""rslf_multi: the host-pointer pile path cut into scanline blocks (one per device) and pipelined chunks must give, bit
for bit, the planes of the one-volume run -- default normalisation (max over ALL EPIs) included."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rs():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from remotesensingproject_amd import depth
    return depth


def _field(V, S, U, C, seed):
    from remotesensingproject_amd.synth import make_lightfield
    vol, _ = make_lightfield(U, V, S, C, seed=seed, dmin=-1.0, dmax=2.0, band=3)
    return vol


PLANES = ("edge_confidence", "edge_mask", "disp_confidence", "depth", "rbar", "depth_idx", "score", "depth_raw")


@pytest.mark.parametrize("devices,chunk,C_,dtype", [([0], 0, 1, "f32"), ([0, 0], 5, 1, "f32"), ([0, 0, 0], 4, 3, "u8"),
                                                   ([0, 0], 1, 1, "f32max"), ([0], 7, 3, "f32max")])
def test_multi_equals_single_volume(rs, oracle_mod, devices, chunk, C_, dtype):
    V, S, U, D = 23, 9, 70, 12
    vol = _field(V, S, U, C_, 11 + len(devices))
    if dtype == "u8":
        raw = np.round(vol * 255.0).astype(np.uint8)
        epis = [raw[v] if C_ == 3 else raw[v, :, :, 0] for v in range(V)]
        scale = -1.0
    elif dtype == "f32max":
        raw = (vol * np.float32(173.0)).astype(np.float32)
        raw[: V // 2] *= np.float32(0.4)     # a block-local maximum would rescale the first half
        epis = [raw[v] if C_ == 3 else raw[v, :, :, 0] for v in range(V)]
        scale = -1.0
    else:
        epis = [vol[v] if C_ == 3 else vol[v, :, :, 0] for v in range(V)]
        scale = 1.0
    m = rs.MultiDevice(devices)
    assert m.device_count() == len(devices)
    m.set_chunk_rows(chunk)
    got = m.depth1d_pile(epis, -1.0, 2.0, D, epi_scale_factor=scale)
    comp = rs.Depth1DComputer_pile(epis, -1.0, 2.0, D, epi_scale_factor=scale)
    comp.run()
    ref = comp.results()
    for k in PLANES:
        assert np.array_equal(got[k], ref[k]), k
    assert m.stats.pixels_scanned == comp.stats.pixels_scanned
    assert m.stats.units == comp.stats.units
    if dtype == "f32max":
        assert m.scale_used == float(raw.max()) == comp.m_epis.scale_used
    # and the oracle agrees (float scale given, or the same normalisation restated)
    if dtype == "f32":
        o = oracle_mod.depth1d_pile_run(vol, -1.0, 2.0, D)
        assert np.array_equal(got["depth_idx"], o.depth_idx) and np.array_equal(got["edge_mask"], o.edge_mask)
    m.close()


def test_multi_with_opened_mask_and_wide_median(rs):
    """The recomputed halo covers the median's rows plus twice the opening radius (core.hpp:686, :759-768)."""
    V, S, U, D = 31, 5, 80, 8
    rng = np.random.default_rng(5)
    vol = rng.uniform(0.0, 1.0, size=(V, S, U)).astype(np.float32)
    epis = [vol[v] for v in range(V)]
    p = rs.Depth1DParameters()
    p.par_median_filter_size = 7
    p.par_edge_confidence_opening_type, p.par_edge_confidence_opening_size = 2, 5
    m = rs.MultiDevice([0, 0])
    m.set_chunk_rows(3)
    got = m.depth1d_pile(epis, -1.0, 1.0, D, epi_scale_factor=1.0, parameters=p)
    comp = rs.Depth1DComputer_pile(epis, -1.0, 1.0, D, epi_scale_factor=1.0, parameters=p)
    comp.run()
    ref = comp.results()
    for k in PLANES:
        assert np.array_equal(got[k], ref[k]), k


def test_multi_rejects_bad_arguments(rs):
    from remotesensingproject_amd import _lib
    with pytest.raises(_lib.RslfError):
        rs.MultiDevice([99])
    m = rs.MultiDevice()
    with pytest.raises(TypeError):
        m.depth1d_pile([np.zeros((3, 8), np.float64)], 0.0, 1.0, 4)


def test_multi_device_out(rs):
    """Device-out: the planes stay in HBM on the chosen device, each worker's rows arriving by peer copy."""
    V, S, U, D = 27, 7, 90, 10
    vol = _field(V, S, U, 1, 21)
    epis = [vol[v, :, :, 0] for v in range(V)]
    m = rs.MultiDevice([0, 0, 0])
    m.set_chunk_rows(4)
    got = m.depth1d_pile_device_out(epis, -1.0, 2.0, D, out_device=0, epi_scale_factor=1.0)
    comp = rs.Depth1DComputer_pile(epis, -1.0, 2.0, D, epi_scale_factor=1.0)
    comp.run()
    ref = comp.results()
    for k in PLANES:
        assert got[k].is_cuda
        assert np.array_equal(got[k].cpu().numpy(), ref[k]), k
    assert m.stats.pixels_scanned == comp.stats.pixels_scanned


@pytest.mark.parametrize("C_,dtype", [(1, "f32"), (3, "u8")])
def test_multi_scattered_epis_and_graded_chunks(rs, C_, dtype):
    """A Vec<Mat>-like input: every EPI its own allocation (they are gathered into pinned memory by a few host threads
    before they go up), 150 scanlines so that the automatic chunk plan has several chunks of different heights."""
    V, S, U, D = 150, 7, 96, 10
    vol = _field(V, S, U, C_, 31)
    src = np.round(vol * 255.0).astype(np.uint8) if dtype == "u8" else vol
    epis = [np.array(src[v] if C_ == 3 else src[v, :, :, 0], copy=True, order="C") for v in range(V)]
    pad = [np.empty(1000 + 37 * v, np.uint8) for v in range(V)]          # keeps the allocations apart
    assert any(epis[v + 1].ctypes.data != epis[v].ctypes.data + epis[v].nbytes for v in range(V - 1))
    m = rs.MultiDevice([0])
    got = m.depth1d_pile(epis, -1.0, 2.0, D, epi_scale_factor=-1.0 if dtype == "u8" else 1.0)
    comp = rs.Depth1DComputer_pile(epis, -1.0, 2.0, D, epi_scale_factor=-1.0 if dtype == "u8" else 1.0)
    comp.run()
    ref = comp.results()
    for k in PLANES:
        assert np.array_equal(got[k], ref[k]), k
    assert m.stats.pixels_scanned == comp.stats.pixels_scanned
    m.close()
    del pad


@pytest.mark.parametrize("devices,C_,dtype,V", [([0], 1, "f32", 21), ([0, 0], 1, "f32max", 24), ([0, 0, 0], 3, "u8", 31), ([0, 0, 0, 0], 1, "f32", 40)])
def test_multi_depth2d_equals_single_volume(rs, devices, C_, dtype, V):
    """The 2-D sweep behind the C-ABI over several devices (here: several contexts on the one GPU): scanline blocks with
    the median's halo, the neighbours' boundary rows fetched by copy on every visit between the scan and the median,
    events for the order -- every plane of every view equals Depth2DComputer's on one volume, bit for bit."""
    S, U, D = 5, 70, 12
    vol = _field(V, S, U, C_, 40 + V)
    rng = np.random.default_rng(V)
    vol[V // 3:V // 3 + 4] = rng.uniform(0.0, 1.0, size=vol[V // 3:V // 3 + 4].shape).astype(np.float32)   # noise across a cut
    if dtype == "u8":
        raw = np.round(vol * 255.0).astype(np.uint8)
        scale = -1.0
    elif dtype == "f32max":
        raw = (vol * np.float32(97.0)).astype(np.float32)
        raw[: V // 2] *= np.float32(0.5)
        scale = -1.0
    else:
        raw, scale = vol, 1.0
    epis = [raw[v] if C_ == 3 else raw[v, :, :, 0] for v in range(V)]
    m = rs.MultiDevice(devices)
    got = m.depth2d(epis, -1.0, 2.0, D, epi_scale_factor=scale)
    comp = rs.Depth2DComputer(epis, -1.0, 2.0, D, epi_scale_factor=scale)
    comp.run()
    ref = comp.results()
    for k in ref:
        assert np.array_equal(got[k], ref[k]), k
    assert m.stats.pixels_scanned == comp.stats.pixels_scanned
    if dtype == "f32max":
        assert m.scale_used == float(raw.max())
    # the context is fit for the pile path afterwards (streams and sweep state restored)
    again = m.depth1d_pile(epis, -1.0, 2.0, D, epi_scale_factor=scale)
    pile = rs.Depth1DComputer_pile(epis, -1.0, 2.0, D, epi_scale_factor=scale)
    pile.run()
    assert np.array_equal(again["depth"], pile.results()["depth"])
    m.close()


@pytest.mark.parametrize("devices,C_,dtype,V", [([0], 1, "f32", 44), ([0, 0], 1, "f32max", 44), ([0, 0, 0], 3, "u8", 61), ([0, 0, 0, 0], 1, "u8", 90)])
def test_multi_fine_to_coarse_equals_single_device(rs, devices, C_, dtype, V):
    """FineToCoarse behind the C-ABI over several devices: every level's sweep sharded by scanline (the coarsest levels fall
    back to fewer blocks by themselves), pyramid, tightening and fusion on the first device -- the fused map and its
    validity equal the single-context run bit for bit, and so does the count of scanned pixels."""
    from remotesensingproject_amd.synth import make_lightfield
    S, U, D = 5, 64, 9
    vol, _ = make_lightfield(U, V, S, C_, seed=2, dmin=-1, dmax=1, band=8)
    if dtype == "u8":
        raw = np.round(vol * 255.0).astype(np.uint8)
    else:
        raw = (vol * 200 + 3).astype(np.float32)
        if dtype == "f32max":
            raw[V // 2:] *= np.float32(0.5)
    scale = 1.0 if dtype == "f32" else -1.0
    if dtype == "f32":
        raw = vol
    epis = [raw[v] if C_ == 3 else raw[v, :, :, 0] for v in range(V)]
    f = rs.FineToCoarse(raw, -1.0, 1.0, D, epi_scale_factor=scale)
    f.run()
    want_map, want_valid = f.get_results()
    m = rs.MultiDevice(devices)
    got_map, got_valid, levels = m.fine_to_coarse(epis, -1.0, 1.0, D, epi_scale_factor=scale)
    assert levels == len(f.m_computers)
    assert np.array_equal(got_map, want_map.cpu().numpy())
    assert np.array_equal(got_valid, want_valid.cpu().numpy())
    assert m.stats.pixels_scanned == sum(int(c.stats.pixels_scanned) for c in f.m_computers)
    m.close()


def _need_devices(n):
    from remotesensingproject_amd import _lib
    have = int(_lib.lib().rslf_device_count())
    if have < n:
        pytest.skip("needs %d GPUs, this box has %d: the copies between DIFFERENT devices (hipMemcpyPeerAsync of the device-out "
                    "pile path, the sweep's boundary-row fetch, the fine-to-coarse row transfers) have never run" % (n, have))


@pytest.mark.parametrize("devices", [[0, 1], [0, 1, 2, 3]])
def test_on_distinct_devices_everything_equals_one_device(rs, devices):
    """the cross-device branches.  Skipped on the one-GPU development boxes; on a multi-GPU node this and
    tools/multi_gpu_selftest.py are their first execution.  Pile path (host-out and device-out), sharded sweep,
    fine-to-coarse -- each bit-identical to the one-device run -- and the peer-access matrix is printed."""
    _need_devices(len(devices))
    V, S, U, D = 64, 7, 140, 12
    vol = _field(V, S, U, 1, 77)
    epis = [vol[v, :, :, 0] for v in range(V)]
    m = rs.MultiDevice(devices)
    print("peer access:", m.peer_access())
    comp = rs.Depth1DComputer_pile(epis, -1.0, 2.0, D, epi_scale_factor=1.0)
    comp.run()
    ref = comp.results()
    got = m.depth1d_pile(epis, -1.0, 2.0, D, epi_scale_factor=1.0)
    for k in PLANES:
        assert np.array_equal(got[k], ref[k]), k
    dev_out = m.depth1d_pile_device_out(epis, -1.0, 2.0, D, out_device=devices[-1], epi_scale_factor=1.0)
    for k in PLANES:
        assert dev_out[k].device.index == devices[-1]
        assert np.array_equal(dev_out[k].cpu().numpy(), ref[k]), k
    c2 = rs.Depth2DComputer(epis, -1.0, 2.0, D, epi_scale_factor=1.0)
    c2.run()
    r2 = c2.results()
    g2 = m.depth2d(epis, -1.0, 2.0, D, epi_scale_factor=1.0)
    for k in r2:
        assert np.array_equal(g2[k], r2[k]), k
    f = rs.FineToCoarse(vol[..., 0], -1.0, 1.0, 9, epi_scale_factor=1.0)
    f.run()
    wm, wv = f.get_results()
    gm, gv, _ = m.fine_to_coarse(epis, -1.0, 1.0, 9, epi_scale_factor=1.0)
    assert np.array_equal(gm, wm.cpu().numpy()) and np.array_equal(gv, wv.cpu().numpy())
    m.close()
