"""The 2-D sweep (compute_2D_edge_confidence + compute_2D_depth_epi + Depth2DComputer,
core.hpp:901-1133, dc.hpp:651-805) on the GPU vs the oracle.  Everything is bit-exact except C_d
(double arithmetic with a free summation order: 1e-5)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _check(got, ref, label):
    for k in ("edge_mask", "scan_mask"):
        assert np.array_equal(got[k], getattr(ref, k)), (label, k)
    for k, r in (("edge_confidence", ref.edge_confidence), ("depth", ref.depth), ("rbar", ref.rbar)):
        bad = np.flatnonzero(got[k].reshape(-1) != r.reshape(-1))
        assert bad.size == 0, (label, k, bad.size, np.unravel_index(bad[0], r.shape))
    assert np.abs(got["disp_confidence"] - ref.disp_confidence).max() <= 1e-5, label


@pytest.mark.parametrize("C,S,U,V,D,kind", [(1, 7, 80, 5, 12, "struct"), (1, 9, 140, 4, 10, "noise"), (3, 5, 70, 4, 8, "struct"),
                                            (1, 6, 64, 3, 9, "struct"), (1, 13, 200, 3, 16, "mixed"),
                                            # D >= 32: the later visits run as grouped launches (k2_scan_combine)
                                            (1, 7, 100, 4, 40, "mixed"), (3, 5, 70, 3, 70, "struct"), (1, 9, 90, 3, 64, "noise")])
def test_depth2d_matches_oracle(oracle_mod, C, S, U, V, D, kind):
    from remotesensingproject_amd import depth as rs
    from remotesensingproject_amd.synth import make_lightfield
    rng = np.random.default_rng(100 + S)
    vol, _ = make_lightfield(U, V, S, C, seed=200 + S, dmin=-1.0, dmax=1.0, band=2)
    if kind == "noise":
        vol = rng.uniform(0.0, 1.0, size=vol.shape).astype(np.float32)
    elif kind == "mixed":
        vol[V // 2:] = rng.uniform(0.0, 1.0, size=vol[V // 2:].shape).astype(np.float32)
    ref = oracle_mod.depth2d_run(vol, -1.0, 1.0, D)
    comp = rs.Depth2DComputer(vol, -1.0, 1.0, D, epi_scale_factor=1.0)
    comp.run()
    _check(comp.results(), ref, "%s_C%d_S%d" % (kind, C, S))
    # every visit scans only what earlier visits left in the mask: far fewer units than S full sweeps
    assert 0 < comp.stats.pixels_scanned <= int((ref.edge_confidence > 0).sum()) + S * V * U
    assert comp.stats.pixels_scanned < S * V * U
    # propagation did paint: pixels left the running mask without having been rejected by the scan
    assert (ref.scan_mask == 0).sum() > (ref.edge_mask == 0).sum() or kind == "noise"


def test_propagation_fills_the_whole_field_on_a_clean_scene(oracle_mod):
    """Integer disparities, exact copies between views: the centre view's scan explains every view, so after the
    sweep every confident pixel of every view carries its band's disparity."""
    from remotesensingproject_amd import depth as rs
    from remotesensingproject_amd.synth import make_lightfield
    U, V, S, D = 160, 6, 9, 17
    deltas = np.full(6, 1.0, np.float32)   # one plane: the 5x5 selective median then cannot mix disparities
    vol, _ = make_lightfield(U, V, S, 1, seed=9, deltas=deltas)
    comp = rs.Depth2DComputer(vol, -2.0, 2.0, D, epi_scale_factor=1.0)
    comp.run()
    got = comp.results()
    ref = oracle_mod.depth2d_run(vol, -2.0, 2.0, D)
    _check(got, ref, "clean")
    inner = slice(8, U - 8)
    for v in range(V):
        m = got["edge_mask"][:, v, inner] > 0
        assert (got["depth"][:, v, inner][m] == deltas[v]).mean() > 0.95
    valid = comp.get_valid_depths_mask_s_v_u().cpu().numpy()
    assert np.array_equal(valid > 0, got["edge_confidence"] > np.float32(0.02))


def test_per_view_ranges_and_entry_points(oracle_mod):
    """compute_2D_edge_confidence / compute_2D_depth_epi called separately, scalar ranges."""
    import torch
    from remotesensingproject_amd import depth as rs
    rng = np.random.default_rng(3)
    V, S, U, D = 3, 5, 90, 8
    vol = rng.uniform(0.0, 1.0, size=(V, S, U, 1)).astype(np.float32)
    ref = oracle_mod.depth2d_run(vol, -0.5, 1.5, D)
    v = rs.Volume.from_dense(vol)
    z = lambda *s, dt=torch.float32: torch.zeros(s, dtype=dt, device="cuda")
    Ce = z(S, V, U)
    cm = rs.compute_2D_edge_confidence(v, Ce)
    Cd, depth, rbar, sm = z(S, V, U), z(S, V, U), z(S, V, U, 1), z(S, V, U, dt=torch.uint8)
    st = rs.compute_2D_depth_epi(v, -0.5, 1.5, D, Ce, cm, Cd, depth, rbar, scan_mask_s_v_u=sm, want_stats=True)
    torch.cuda.synchronize()
    got = dict(edge_confidence=Ce.cpu().numpy(), edge_mask=cm.cpu().numpy(), disp_confidence=Cd.cpu().numpy(),
               depth=depth.cpu().numpy(), rbar=rbar.cpu().numpy(), scan_mask=sm.cpu().numpy())
    _check(got, ref, "entry_points")
    assert st.units == st.pixels_scanned * D


@pytest.mark.parametrize("C_,dtype", [(1, np.float32), (3, np.uint8)])
def test_single_epi_computer(oracle_mod, C_, dtype):
    """rslf::Depth1DComputer (dc.hpp:256-371): one EPI, edge confidence + scan, NO median; constructor
    normalisation included (max of the EPI for float input, 1/255 for uchar)."""
    from remotesensingproject_amd import depth as rs
    rng = np.random.default_rng(17)
    S, U, D = 11, 130, 14
    if dtype == np.uint8:
        raw = rng.integers(0, 256, size=(S, U, C_), dtype=np.uint8)
        epi_n = oracle_mod.normalize_u8(raw)
    else:
        raw = rng.uniform(5.0, 200.0, size=(S, U)).astype(np.float32)
        epi_n, _ = oracle_mod.normalize_f32(raw[..., None], -1.0)
    epi_n = np.ascontiguousarray(epi_n.reshape(S, U, C_))
    Ce, cm = oracle_mod.edge_confidence_pile(epi_n[None], S // 2)
    ref = oracle_mod.depth_epi(epi_n, np.full(U, -1.0, np.float32), np.full(U, 2.0, np.float32), D, S // 2, Ce[0], cm[0])
    comp = rs.Depth1DComputer(raw, -1.0, 2.0, D)
    comp.run()
    got = comp.results()
    assert comp.get_s_hat() == S // 2 if hasattr(comp, "get_s_hat") else comp.m_s_hat == S // 2
    assert np.array_equal(got["edge_mask"], ref["Ce_mask"])
    assert np.array_equal(got["depth_idx"], ref["idx"])
    assert np.array_equal(got["edge_confidence"], ref["Ce"])
    assert np.array_equal(got["score"], ref["score"])
    assert np.array_equal(got["depth"], ref["depth"])          # raw arg-max depths: no median in this class
    assert np.array_equal(got["rbar"], ref["rbar"])
    assert np.abs(got["disp_confidence"] - ref["Cd"]).max() <= 1e-5


def test_scratch_reuse_across_shapes(oracle_mod):
    """One context, volumes whose S*V*U shrinks while V*U grows: the per-view scratch (S*V*U entries) and the
    per-plane scratch (V*U) have separate capacities.  A shared one once let the median plane overrun its
    buffer and corrupt a neighbouring volume (found by tools/fuzz_sweep.py)."""
    from remotesensingproject_amd import depth as rs
    rng = np.random.default_rng(11)
    for S, V, U in [(13, 2, 100), (5, 4, 110), (3, 16, 40), (2, 20, 45), (1, 30, 50)]:
        vol = rng.uniform(0.0, 1.0, size=(V, S, U, 1)).astype(np.float32)
        ref = oracle_mod.depth2d_run(vol, -1.0, 1.0, 12)
        comp = rs.Depth2DComputer(vol, -1.0, 1.0, 12, epi_scale_factor=1.0)
        comp.run()
        _check(comp.results(), ref, "reuse_S%d_V%d_U%d" % (S, V, U))


@pytest.mark.parametrize("C_,thr", [(1, 0.01), (3, 0.01), (1, 0.2)])
def test_sweep_with_the_disp_confidence_gate(oracle_mod, C_, thr):
    """par_use_disp_confidence_score: the reference's _USE_DISP_CONFIDENCE_SCORE build (core.hpp:35, :1097-1098) --
    propagation gated by C_d > par_disp_score_threshold instead of the edge mask."""
    from remotesensingproject_amd import depth as rs
    from remotesensingproject_amd.synth import make_lightfield
    rng = np.random.default_rng(40 + C_)
    vol, _ = make_lightfield(90, 5, 7, C_, seed=12, deltas=np.array([0, 1, -1, 1, 0], np.float32))
    vol[2:] = (vol[2:] + rng.normal(0, 0.05, size=vol[2:].shape)).clip(0, 1).astype(np.float32)
    po = oracle_mod.default_params()
    po.use_disp_confidence_score, po.disp_score_threshold = 1, thr
    pr = rs.Depth1DParameters(par_use_disp_confidence_score=True, par_disp_score_threshold=thr)
    ref = oracle_mod.depth2d_run(vol, -1.0, 1.0, 12, params=po)
    comp = rs.Depth2DComputer(vol, -1.0, 1.0, 12, epi_scale_factor=1.0, parameters=pr)
    comp.run()
    _check(comp.results(), ref, "disp_gate_C%d_%g" % (C_, thr))
    if thr > 0.1:   # a threshold most confidences miss: fewer pixels paint than under the edge-mask gate
        plain = oracle_mod.depth2d_run(vol, -1.0, 1.0, 12)
        assert not np.array_equal(plain.scan_mask, ref.scan_mask) or not np.array_equal(plain.depth, ref.depth)


@pytest.fixture
def rs():
    from remotesensingproject_amd import depth
    return depth


def _sharded_vs_unsharded(rs, vol_np, world, D, params=None, exchange=True):
    """Run Depth2DComputer::run unsharded and as `world` scanline shards with a halo exchange per visit; returns the
    two sets of planes (sharded ones stitched)."""
    import torch
    from remotesensingproject_amd import sharding
    V = vol_np.shape[0]
    p = params or rs.Depth1DParameters()
    full = rs.Depth2DComputer(rs.Volume.from_dense(torch.from_numpy(vol_np).cuda(), 1.0), -1.0, 1.5, D, parameters=p)
    full.run()
    ref = full.results()
    shards = []
    for r in range(world):
        sh = sharding.make_shard(V, r, world, p.par_median_filter_size, p.par_edge_confidence_opening_size)
        ctx = rs.Context(0)                                        # one context per "rank", as in one process per GPU
        v = rs.Volume.from_dense(torch.from_numpy(np.ascontiguousarray(vol_np[sh.rows])).cuda(), 1.0, ctx)
        shards.append(sharding.ShardedDepth2D(v, sh, -1.0, 1.5, D, p))
    sharding.run_lockstep_sweep(shards, exchange=exchange)
    torch.cuda.synchronize()
    got = {k: np.concatenate([s.own_planes()[k].cpu().numpy() for s in shards], axis=1) for k in ref}
    scanned = sum(int(s.stats.pixels_scanned) for s in shards)
    return ref, got, scanned, int(full.stats.pixels_scanned)


@pytest.mark.parametrize("world,V,S,U,C_", [(2, 24, 5, 70, 1), (3, 31, 7, 64, 1), (2, 17, 4, 90, 3), (4, 40, 3, 50, 1)])
def test_sweep_sharded_by_scanline_with_halo_exchange(rs, world, V, S, U, C_):
    """The 2-D sweep over scanline shards: scan and propagation are row-local, the median of every visit reads 2 rows of
    the neighbour's freshly written disparities and mask -- exchanged between rslf_sweep_visit_scan and _finish.  Every
    plane of every view must equal the unsharded sweep bit for bit (lock-step harness: the ranks' steps in one process)."""
    from remotesensingproject_amd.synth import make_lightfield
    vol, _ = make_lightfield(U, V, S, C_, seed=40 + V, dmin=-1.0, dmax=1.5, band=3)
    rng = np.random.default_rng(V)
    vol[V // 3:V // 3 + 4] = rng.uniform(0.0, 1.0, size=vol[V // 3:V // 3 + 4].shape).astype(np.float32)   # a band of noise across a cut
    ref, got, scanned, scanned_full = _sharded_vs_unsharded(rs, vol, world, 12)
    for k in ref:
        assert np.array_equal(got[k], ref[k]), (k, world)
    assert scanned == scanned_full
    if world == 2 and C_ == 1:                                       # the exchange is load-bearing: stale halo rows change the result
        _, stale, _, _ = _sharded_vs_unsharded(rs, vol, world, 12, exchange=False)
        assert not np.array_equal(stale["depth"], ref["depth"])


def test_sweep_sharded_wide_median(rs):
    p = rs.Depth1DParameters()
    p.par_median_filter_size = 7
    rng = np.random.default_rng(3)
    vol = rng.uniform(0.0, 1.0, size=(26, 3, 60, 1)).astype(np.float32)
    ref, got, scanned, scanned_full = _sharded_vs_unsharded(rs, vol, 2, 8, params=p)
    for k in ref:
        assert np.array_equal(got[k], ref[k]), k
    assert scanned == scanned_full


def _sweep_rank(rank, world, port, V, S, U, D, out_dir):
    import os
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from remotesensingproject_amd import depth as rs
        from remotesensingproject_amd import sharding
        from remotesensingproject_amd.synth import make_lightfield
        torch.cuda.set_device(0)
        sh = sharding.make_shard(V, rank, world, 5)
        vol, _ = make_lightfield(U, V, S, 1, seed=77, dmin=-1.0, dmax=1.5, band=3, rows=sh.rows)
        v = rs.Volume.from_dense(torch.from_numpy(vol).cuda(), 1.0)
        sw = sharding.ShardedDepth2D(v, sh, -1.0, 1.5, D)
        sw.run()                                                    # visits + neighbour exchange over torch.distributed
        torch.cuda.synchronize()
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), scanned=int(sw.stats.pixels_scanned),
                 **{k: t.cpu().numpy() for k, t in sw.own_planes().items()})
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_sweep_sharded_over_torch_distributed(rs, tmp_path):
    """The same through ShardedDepth2D.run(): two rank processes on the one GPU, the halo rows travelling over
    torch.distributed (gloo here -- a rehearsal of the transport; RCCL on a multi-GPU node)."""
    import socket
    import torch
    import torch.multiprocessing as mp
    from remotesensingproject_amd.synth import make_lightfield
    V, S, U, D, world = 22, 5, 64, 10, 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_sweep_rank, args=(world, port, V, S, U, D, str(tmp_path)), nprocs=world, join=True)
    vol, _ = make_lightfield(U, V, S, 1, seed=77, dmin=-1.0, dmax=1.5, band=3)
    full = rs.Depth2DComputer(rs.Volume.from_dense(torch.from_numpy(vol).cuda(), 1.0), -1.0, 1.5, D)
    full.run()
    ref = full.results()
    parts = [np.load(tmp_path / ("rank%d.npz" % r)) for r in range(world)]
    for k in ref:
        assert np.array_equal(np.concatenate([q[k] for q in parts], axis=1), ref[k]), k
    assert sum(int(q["scanned"]) for q in parts) == int(full.stats.pixels_scanned)


def test_sweep_steps_enforce_the_reference_order_and_recover(rs):
    """rslf_sweep_visit_scan takes the views in the reference's order only (each visit's apply pass has already listed the
    next view's pixels); a sweep abandoned half-way leaves the context fit for the next one."""
    import torch
    from remotesensingproject_amd import sharding
    from remotesensingproject_amd.synth import make_lightfield
    vol_np, _ = make_lightfield(64, 12, 5, 1, seed=9, dmin=-1.0, dmax=1.5, band=3)
    ctx = rs.Context(0)
    mk = lambda: rs.Volume.from_dense(torch.from_numpy(vol_np).cuda(), 1.0, ctx)
    sw = sharding.ShardedDepth2D(mk(), sharding.make_shard(12, 0, 1), -1.0, 1.5, 10)
    sw.prepare()
    with pytest.raises(RuntimeError, match="visits view 2 next"):
        sw.visit_scan(3)
    sw.visit_scan(2)
    with pytest.raises(RuntimeError, match="open visit is view 2"):
        sw.visit_finish(3)
    sw.visit_finish(2)
    sw.visit_scan(3)                      # ... and the caller gives up here
    sw.finish(ok=False)
    with pytest.raises(RuntimeError, match="without rslf_sweep_begin"):
        sw.visit_finish(3)
    again = rs.Depth2DComputer(mk(), -1.0, 1.5, 10, ctx=ctx)           # same context, after the abandoned sweep
    again.run()
    fresh = rs.Depth2DComputer(rs.Volume.from_dense(torch.from_numpy(vol_np).cuda(), 1.0), -1.0, 1.5, 10)
    fresh.run()
    a, b = again.results(), fresh.results()
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    assert int(again.stats.pixels_scanned) == int(fresh.stats.pixels_scanned)


def test_sweep_group_merge_is_exact_at_scale(rs):
    """The sparse visits at scale, in their three forms: lanes that own hypotheses (k2_scan_reg_px, what a sweep takes since
    round 3), and the pixel-per-lane kernel whose hypothesis groups hand their records from workgroup to workgroup across
    XCDs (agent-coherent stores and loads, a ticket per tile, the group count settled on the device) -- on a c2-sized
    sweep whose views disagree enough for ~10^5 pixels to be scanned on sparse visits, every plane must equal the run that
    never splits a tile's hypotheses (px = 0, force_groups = 1: no records, no merge)."""
    import torch
    from remotesensingproject_amd.synth import make_lightfield
    vol, _ = make_lightfield(512, 256, 17, 1, seed=123, dmin=-2.0, dmax=2.0, band=16)
    rng = np.random.default_rng(5)
    vol = (vol + rng.normal(0.0, 0.04, size=vol.shape)).clip(0.0, 1.0).astype(np.float32)   # propagation fails for many pixels
    out = []
    for hooks, kernel in (({}, 4), (dict(px=0), 1), (dict(px=0, force_groups=1), 1)):
        ctx = rs.Context(0)
        ctx.set_debug(**hooks)
        comp = rs.Depth2DComputer(rs.Volume.from_dense(torch.from_numpy(vol).cuda(), 1.0, ctx), -2.0, 2.0, 128, ctx=ctx)
        comp.run()
        assert comp.stats.scan_kernel == kernel, (hooks, comp.stats.scan_kernel)      # the kernel of the LAST visit's scan
        out.append((comp.results(), int(comp.stats.pixels_scanned)))
        ctx.reset_debug()
    (a, na), (b, nb), (c, nc) = out
    assert na == nb == nc and na > 256 * 512 + 50000, na          # well beyond the dense first visit
    for k in a:
        assert np.array_equal(a[k], b[k]), k
        assert np.array_equal(a[k], c[k]), k


@pytest.mark.parametrize("case", ["plain", "gate", "nan_range"])
def test_claims_with_and_without_view_skipping_agree(rs, hooks, case):
    """rslf_ctx_set_debug("claim_skip", 0 / 1): the claims' skip of views with nothing left to paint within reach is a
    pure shortcut -- every plane of every view identical with and without it, also under the disparity-
    confidence gate and with NaN / huge disparities among the sources (per-pixel ranges holding NaN and 1e12: the
    workgroup's disparity range then falls back to +-1e9, i.e. "every segment")."""
    import torch
    rng = np.random.default_rng(321)
    V, S, U, D = 18, 9, 600, 12
    vol_np = rng.uniform(0.0, 1.0, size=(V, S, U, 1)).astype(np.float32)
    vol_np[:, :, 100:400] = np.round(vol_np[:, :, 100:400] * 2) / 2
    p = rs.Depth1DParameters()
    if case == "gate":
        p.par_use_disp_confidence_score = True
        p.par_disp_score_threshold = 0.02
    out = {}
    for skip in (1, 0):
        hooks(claim_skip=skip)
        vol = rs.Volume.from_dense(torch.from_numpy(vol_np).cuda(), 1.0)
        Ce = torch.zeros((S, V, U), dtype=torch.float32, device="cuda")
        cem = rs.compute_2D_edge_confidence(vol, Ce, p)
        Cd, depth = torch.zeros_like(Ce), torch.zeros_like(Ce)
        rbar = torch.zeros((S, V, U, 1), dtype=torch.float32, device="cuda")
        scan = torch.empty((S, V, U), dtype=torch.uint8, device="cuda")
        if case == "nan_range":
            dmin = torch.full((S, V, U), -1.0, dtype=torch.float32, device="cuda")
            dmax = torch.full((S, V, U), 1.0, dtype=torch.float32, device="cuda")
            dmin[:, 3, 50:60] = float("nan")
            dmin[:, 7, 300:310], dmax[:, 7, 300:310] = 1.0e12, 2.0e12
            rs.compute_2D_depth_epi(vol, dmin, dmax, D, Ce, cem, Cd, depth, rbar, p, scan_mask_s_v_u=scan)
        else:
            rs.compute_2D_depth_epi(vol, -1.0, 1.0, D, Ce, cem, Cd, depth, rbar, p, scan_mask_s_v_u=scan)
        torch.cuda.synchronize()
        out[skip] = dict(depth=depth.cpu().numpy(), Cd=Cd.cpu().numpy(), rbar=rbar.cpu().numpy(), scan=scan.cpu().numpy(),
                         cem=cem.cpu().numpy(), Ce=Ce.cpu().numpy())
    for k in out[1]:
        assert np.array_equal(out[1][k], out[0][k], equal_nan=True), (case, k)
    if case == "nan_range":   # (a NaN range scores 0 and drops its pixel; the 1e12 range leaves sources past the +-1e9 guard)
        assert (np.abs(out[1]["depth"]) > 1.0e9).any()
