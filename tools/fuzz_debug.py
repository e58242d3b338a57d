"""Developer tool: replay a fuzz_parity plane-form case and print the planes at the first mismatching pixel.
    python tools/fuzz_debug.py SEED CASE"""
import importlib.util, os, sys
import numpy as np, torch
spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz_parity.py"))
fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
rs, oracle = fz.rs, fz.oracle
seed, case = int(sys.argv[1]), int(sys.argv[2])
for share, force in ((1, "stream"), (0, "stream"), (1, "generic")):
    rng = np.random.default_rng([seed, case])
    c = fz.draw_case(rng)
    c["force"] = force
    ctx = rs.default_context(0)
    ctx.set_debug(stream_share=share)
    try:
        fz.run_case(case, c, rng)
        print("share", share, force, "ok")
    except AssertionError as e:
        print("share", share, force, "FAIL", str(e)[:160])
ctx.reset_debug()
# hand replay of the plane form to look at the scores
rng = np.random.default_rng([seed, case])
c = fz.draw_case(rng)
vol = fz.make_volume(c, rng)
V, S, U, C = vol.shape
po = oracle.default_params(); pr = rs.Depth1DParameters()
for name, val in (("slope_factor", c["slope"]), ("mean_shift_max_iter", c["iters"]), ("kernel_bandwidth", c["h"]),
                  ("edge_score_threshold", c["thr"]), ("raw_score_threshold", c["raw_thr"]), ("median_filter_size", c["median"]),
                  ("cut_shadows", int(c["shadows"]))):
    setattr(po, name, val); setattr(pr, "par_" + name, val)
sh = S // 2
Ce, cm = oracle.edge_confidence_pile(vol, sh, params=po)
dmin = np.full((V, U), c["dmin"], np.float32); dmax = np.full((V, U), c["dmax"], np.float32)
ref = oracle.depth_epi_pile(vol, dmin, dmax, c["D"], sh, Ce, cm, params=po, mask_vu=None)
v = rs.Volume.from_dense(vol)
for force in ("stream", "generic"):
    ctx.set_debug(force_scan=force)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a).copy()).cuda()
    tCe, tcm = t(Ce), t(cm)
    tCd = torch.zeros((V, U), device="cuda"); td = torch.zeros((V, U), device="cuda"); trb = torch.zeros((V, U, C), device="cuda")
    tidx = torch.empty((V, U), dtype=torch.int32, device="cuda"); tsc = torch.empty((V, U), device="cuda")
    rs.compute_1D_depth_epi_pile(v, c["dmin"], c["dmax"], c["D"], sh, tCe, tcm, tCd, td, trb, pr, None, idx_v_u=tidx, score_v_u=tsc)
    torch.cuda.synchronize()
    gi, gs = tidx.cpu().numpy(), tsc.cpu().numpy()
    bad = np.argwhere(gi != ref.depth_idx)
    print(force, "no scan mask: idx mismatches", len(bad), "score mismatches", int((gs.view(np.uint32) != ref.score.view(np.uint32)).sum()))
    for (y, x) in bad[:4]:
        print("   at", (int(y), int(x)), "gpu idx", gi[y, x], "score %.9g" % gs[y, x], "oracle idx", ref.depth_idx[y, x], "score %.9g" % ref.score[y, x])
ctx.reset_debug()
print("U", U, "S", S, "D", c["D"], "slope", c["slope"], "dmin/dmax", c["dmin"], c["dmax"])
