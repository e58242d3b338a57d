"""Per-visit anatomy of a 2-D sweep (developer probe): how many pixels each visit scans, how they lie on the scanlines
(share in runs of >= 63 consecutive pixels), and how long the visit's scan takes.
    python tools/probe_visits.py [config] [max_visits]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from remotesensingproject_amd import depth as rs, sharding
from remotesensingproject_amd.synth import CONFIGS, make_lightfield

cfg = CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "mansion_lr"]
max_visits = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
U, V, S, C, D = cfg["U"], cfg["V"], cfg["S"], cfg["C"], cfg["D"]
host, _ = make_lightfield(U, V, S, C, seed=cfg["seed"], dmin=cfg["dmin"], dmax=cfg["dmax"])
vol = rs.Volume.from_dense(torch.from_numpy(host).cuda(), 1.0)
sw = sharding.ShardedDepth2D(vol, sharding.make_shard(V, 0, 1), cfg["dmin"], cfg["dmax"], D)
sw.prepare()
order = sharding.sweep_order(S)
tot_px, tot_ms = 0, 0.0
for i, s in enumerate(order[:max_visits]):
    m = (sw.scan_mask[s] > 0) & (sw.cem[s] > 0)
    n = int(m.sum())
    # pixels in runs of >= 63 consecutive mask pixels of a scanline
    mm = m.cpu().numpy()
    long_px = 0
    for v in range(0, V, max(1, V // 40)):          # a sample of the scanlines
        row = np.flatnonzero(np.diff(np.concatenate(([0], mm[v].astype(np.int8), [0]))))
        runs = row[1::2] - row[0::2]
        long_px += int(runs[runs >= 63].sum())
    sampled = int(mm[::max(1, V // 40)].sum())
    if i == 1 and os.environ.get("FORMS"):   # the launch forms of tools/probe_density.py on THIS visit's own pixel list
        ctx = vol.ctx
        per_row = mm.sum(axis=1)
        print("visit 1: rows with 0 px %d, 1-63 px %d, >= 64 px %d; px in rows >= 64: %d" % (
            int((per_row == 0).sum()), int(((per_row > 0) & (per_row < 64)).sum()), int((per_row >= 64).sum()), int(per_row[per_row >= 64].sum())))
        for name, hooks in (("rows", dict(force_packed=0)), ("px", dict(force_packed=1, px=1)), ("packed", dict(force_packed=1, px=0))):
            ctx.reset_debug()
            ctx.set_debug(**hooks)
            Ce1 = sw.Ce[s].clone(); cem1 = sw.cem[s].clone(); Cd1 = torch.zeros_like(Ce1); d1 = torch.zeros_like(Ce1)
            rb1 = torch.zeros((V, U, C), dtype=torch.float32, device="cuda")
            ms = []
            for _ in range(3):
                mk = m.to(torch.uint8) * 255
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                st = rs.compute_1D_depth_epi(vol, cfg["dmin"], cfg["dmax"], D, s, Ce1, cem1, Cd1, d1, rb1, None, mk, want_stats=True)
                torch.cuda.synchronize()
                ms.append((time.perf_counter() - t0) * 1e3)
            print("   %-16s %.2f ms wall (kernel %d)" % (name, min(ms), st.scan_kernel), flush=True)
        ctx.reset_debug()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sw.visit_scan(s)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    sw.visit_finish(s)
    tot_px += n
    tot_ms += dt
    if i < 12 or i % 10 == 0:
        print("visit %3d view %3d: %7d px (%.1f %% of the plane), %4.1f %% of them in runs >= 63, scan %.2f ms, %.2f G units/s" % (
            i, s, n, 100.0 * n / (U * V), 100.0 * long_px / max(sampled, 1), dt, n * D / dt / 1e6 if dt > 0 else 0), flush=True)
sw.finish()
print("total: %d px scanned (%.2f views' worth), %.1f ms of scans" % (tot_px, tot_px / (U * V), tot_ms))
