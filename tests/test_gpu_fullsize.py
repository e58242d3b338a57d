"""BASELINE.json full-size configurations on the GPU, checked through size-independent
properties (the oracle cannot run these sizes in test time) plus an oracle spot check."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _planes(comp):
    return comp.results()


@pytest.fixture(scope="module")
def rs():
    from remotesensingproject_amd import depth
    return depth


def _expected_idx(delta, c):
    step = (c["dmax"] - c["dmin"]) / (c["D"] - 1)
    return np.round((delta - c["dmin"]) / step).astype(np.int32)


def _check_known_answer(a, delta, c):
    """Every confident pixel scores exactly 1.0 (its true line is constant).  Where the whole
    line family stays inside the image the argmax is the true disparity; within `reach` of the
    left/right border a line keeps only the views on one side, and if two neighbouring texels
    happen to agree to ~1e-5 an earlier hypothesis also reaches exactly 1.0 and -- first maximum
    wins, cv::minMaxLoc -- takes the index (seen on 2 of 262144 pixels of c2; the oracle agrees)."""
    m = a["edge_mask"] > 0
    U = m.shape[1]
    want = np.broadcast_to(_expected_idx(delta, c)[:, None], m.shape)
    assert (a["score"][m] == 1.0).all()
    assert (a["depth_idx"][m] <= want[m]).all()
    reach = int(np.ceil(max(abs(c["dmin"]), abs(c["dmax"])) * (c["S"] // 2))) + 1
    inner = np.zeros_like(m)
    inner[:, reach:U - reach] = True
    sel = m & inner
    assert (a["depth_idx"][sel] == want[sel]).all()
    assert (a["depth_raw"][sel] == np.broadcast_to(delta[:, None], m.shape)[sel]).all()
    assert (a["depth_idx"][m] != want[m]).mean() < 1e-4
    return m


def test_c2_full_known_answer_determinism_and_sharding(rs, oracle_mod):
    """512x512x33, 128 hypotheses (configs[1]): (1) known answer on every pixel, (2) two runs are
    bit-identical, (3) three scanline shards with recomputed halos stitch to the unsharded planes,
    (4) a block of scanlines matches the oracle bit for bit."""
    import torch
    from remotesensingproject_amd import sharding
    from remotesensingproject_amd.synth import make_config
    vol, delta, c = make_config("c2")
    comp = rs.Depth1DComputer_pile(vol, c["dmin"], c["dmax"], c["D"], epi_scale_factor=1.0)
    comp.run()
    a = _planes(comp)
    m = _check_known_answer(a, delta, c)
    assert m.mean() > 0.999
    assert comp.stats.units == int(m.sum()) * c["D"]
    # (2)
    comp.run()
    b = _planes(comp)
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    # (3)
    V, U = c["V"], c["U"]
    parts = sharding.row_partition(V, 3)
    max_rows = max(q - p for p, q in parts)
    bufs = []
    for r in range(3):
        sh = sharding.make_shard(V, r, 3, 5)
        cs = rs.Depth1DComputer_pile(np.ascontiguousarray(vol[sh.rows]), c["dmin"], c["dmax"], c["D"], epi_scale_factor=1.0)
        cs.run()
        pl = dict(edge_confidence=cs.m_edge_confidence_v_u, disp_confidence=cs.m_disp_confidence_v_u, depth=cs.m_best_depth_v_u,
                  depth_raw=cs.m_depth_raw_v_u, score=cs.m_score_v_u, depth_idx=cs.m_depth_idx_v_u, rbar=cs.m_rbar_v_u,
                  edge_mask=cs.m_edge_confidence_mask_v_u)
        bufs.append(sharding.pack_planes(pl, sh.interior, max_rows, U, 1))
    st = sharding.unpack_planes(bufs, parts, max_rows, U, 1)
    torch.cuda.synchronize()
    for k in a:
        assert np.array_equal(st[k].cpu().numpy(), a[k]), k
    # (4)
    blk = slice(30, 38)   # spans a band boundary (bands are 32 rows)
    ref = oracle_mod.depth1d_pile_run(np.ascontiguousarray(vol[blk]), c["dmin"], c["dmax"], c["D"])
    inner = slice(2, 6)   # rows whose 5x5 median window lies inside the block
    for k in ("depth_idx", "score", "rbar", "depth_raw", "edge_confidence", "edge_mask", "depth"):
        assert np.array_equal(a[k][blk][inner], getattr(ref, k)[inner]), k


def test_c3_full_known_answer(rs, oracle_mod):
    """1920x1080x101, 256 hypotheses (configs[2], the bench workload): known answer everywhere,
    units as BASELINE.md section 4 states, and an oracle spot check on two scanlines."""
    from remotesensingproject_amd.synth import make_config
    vol, delta, c = make_config("c3")
    comp = rs.Depth1DComputer_pile(vol, c["dmin"], c["dmax"], c["D"], epi_scale_factor=1.0)
    comp.run()
    a = _planes(comp)
    m = _check_known_answer(a, delta, c)
    assert m.sum() >= 0.9999 * m.size
    assert comp.stats.units == int(m.sum()) * 256
    assert comp.stats.scan_kernel == 1 and comp.stats.s_pad == 104
    rows = [0, 1, 2, 3, 4]   # top edge of the image: the median window is clipped there
    ref = oracle_mod.depth1d_pile_run(np.ascontiguousarray(vol[rows]), c["dmin"], c["dmax"], c["D"])
    for k in ("depth_idx", "score", "rbar", "depth_raw", "edge_confidence", "edge_mask"):
        assert np.array_equal(a[k][rows], getattr(ref, k)), k
    assert np.array_equal(a["depth"][:3], ref.depth[:3])   # rows 0..2 see only rows 0..4


def test_c3_full_size_8_way_partition_stitches(rs):
    """BASELINE.json configs[3] (c4): the c3 sweep cut into eight blocks of 135 scanlines with their recomputed 2-row
    halos (what eight ranks compute), run one after the other on the one GPU and stitched through the same packed-plane
    path the RCCL gather uses -- bit-identical to the unsharded planes, scanned-pixel counts adding up.  RCCL itself has
    not run on hardware (one-GPU boxes); tools/multi_gpu_selftest.py is the one command for the first multi-GPU lease."""
    import torch
    from remotesensingproject_amd import sharding
    from remotesensingproject_amd.synth import make_config
    vol, _, c = make_config("c3")
    V, U, D = c["V"], c["U"], c["D"]
    dev = torch.from_numpy(vol).cuda()
    comp = rs.Depth1DComputer_pile(rs.Volume.from_dense(dev, 1.0), c["dmin"], c["dmax"], D)
    comp.run()
    full = dict(edge_confidence=comp.m_edge_confidence_v_u, disp_confidence=comp.m_disp_confidence_v_u, depth=comp.m_best_depth_v_u,
                depth_raw=comp.m_depth_raw_v_u, score=comp.m_score_v_u, depth_idx=comp.m_depth_idx_v_u, rbar=comp.m_rbar_v_u,
                edge_mask=comp.m_edge_confidence_mask_v_u)
    parts = sharding.row_partition(V, 8)
    assert all(b - a == 135 for a, b in parts)
    bufs, scanned = [], 0
    for r in range(8):
        sh = sharding.make_shard(V, r, 8, 5)
        assert (sh.v1 - sh.v0, sh.hi - sh.lo) == (135, 137 if r in (0, 7) else 139)
        cs = rs.Depth1DComputer_pile(rs.Volume.from_dense(dev[sh.rows], 1.0), c["dmin"], c["dmax"], D)
        cs.run(want_stats=False)
        pl = dict(edge_confidence=cs.m_edge_confidence_v_u, disp_confidence=cs.m_disp_confidence_v_u, depth=cs.m_best_depth_v_u,
                  depth_raw=cs.m_depth_raw_v_u, score=cs.m_score_v_u, depth_idx=cs.m_depth_idx_v_u, rbar=cs.m_rbar_v_u,
                  edge_mask=cs.m_edge_confidence_mask_v_u)
        scanned += int((pl["edge_mask"][sh.interior] > 0).sum().item())
        bufs.append(sharding.pack_planes(pl, sh.interior, 135, U, 1))
        del cs, pl
    st = sharding.unpack_planes(bufs, parts, 135, U, 1)
    torch.cuda.synchronize()
    for k in full:
        assert torch.equal(st[k], full[k]), k
    assert scanned == comp.stats.pixels_scanned
