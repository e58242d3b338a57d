"""Developer probe: the floor of a sparse (packed) scan launch -- c2-sized volume, N pixels to scan, by groups."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from remotesensingproject_amd import depth as rs
from remotesensingproject_amd.synth import make_lightfield

U = V = 512; S = 33; D = 128
if len(sys.argv) > 4:      # python tools/probe_sparse.py U V S D [pixel counts ...]
    U, V, S, D = [int(x) for x in sys.argv[1:5]]
vol_np, _ = make_lightfield(U, V, S, 1, seed=1, dmin=-2, dmax=2, band=8)
vol = rs.Volume.from_dense(torch.from_numpy(vol_np).cuda(), 1.0)
ctx = vol.ctx
dev = "cuda"
Ce = torch.ones((V, U), device=dev); cm = torch.full((V, U), 255, dtype=torch.uint8, device=dev)
Cd = torch.zeros((V, U), device=dev); depth = torch.zeros((V, U), device=dev); rbar = torch.zeros((V, U, 1), device=dev)
rng = np.random.default_rng(1)
for npx in ([int(x) for x in sys.argv[5:]] or (64, 1000, 8000, 64000)):
    m = np.zeros(V * U, np.uint8); m[rng.choice(V * U, npx, replace=False)] = 255
    mask0 = torch.from_numpy(m.reshape(V, U)).to(dev)
    for groups in ((1, 2, 4, 8, 16, 32, 64) if len(sys.argv) <= 4 else (8, 16, 32, 64)):
        ctx.set_debug(force_packed=1, force_groups=groups)
        ts = []
        for rep in range(6):
            mask = mask0.clone(); cmc = cm.clone()
            rs.compute_1D_depth_epi_pile(vol, -2.0, 2.0, D, S // 2, Ce, cmc, Cd, depth, rbar, None, mask)
            ts.append(ctx.last_scan_kernel_ms() * 1e3)
        print("pixels %6d groups %2d: K2 %7.1f us (min of 5); %.2f G units/s" % (npx, groups, min(ts[1:]), npx * D / min(ts[1:]) / 1e3), flush=True)
ctx.reset_debug()
