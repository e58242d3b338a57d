"""Which launch form suits which list density (developer probe): the scan of a random subset of a field's pixels (a caller's
scan mask of density p) as row tiles (per-row lists), as a packed list with a pixel per lane, and with a pixel per wave.
    python tools/probe_density.py [config] [densities...]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from remotesensingproject_amd import depth as rs
from remotesensingproject_amd.synth import CONFIGS, make_lightfield

cfg = CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "mansion_lr"]
if os.environ.get("SHAPE"):   # U,V,S,C,D,dmin,dmax instead of a named config
    f = os.environ["SHAPE"].split(",")
    cfg = dict(U=int(f[0]), V=int(f[1]), S=int(f[2]), C=int(f[3]), D=int(f[4]), dmin=float(f[5]), dmax=float(f[6]), seed=20260099)
dens = [float(x) for x in sys.argv[2:]] or [1.0, 0.5, 0.2, 0.05, 0.01]
U, V, S, C, D = cfg["U"], cfg["V"], cfg["S"], cfg["C"], cfg["D"]
V = int(os.environ.get("ROWS", V))
host, _ = make_lightfield(U, V, S, C, seed=cfg["seed"], dmin=cfg["dmin"], dmax=cfg["dmax"])
vol = rs.Volume.from_dense(torch.from_numpy(host).cuda(), 1.0)
ctx = vol.ctx
p = rs.Depth1DParameters()
rng = np.random.default_rng(5)
forms = [("rows", dict(force_packed=0)), ("rows noshare", dict(force_packed=0, stream_share=0)), ("packed", dict(force_packed=1, px=0)),
         ("px", dict(force_packed=1, px=1))]
for dn in dens:
    m_np = rng.uniform(size=(V, U)) < dn
    zone = os.environ.get("ZONE", "")          # "border": only the columns a sample line can leave the row from; "interior": the rest
    if zone:
        reach = int(max(S // 2, S - 1 - S // 2) * max(abs(cfg["dmin"]), abs(cfg["dmax"]))) + 2
        cols = np.arange(U)
        edge = (cols < reach) | (cols > U - 1 - reach)
        m_np &= (edge if zone == "border" else ~edge)[None, :]
    if os.environ.get("BANDS"):                # as the synthetic sweeps' lists: bands of 32 scanlines, every fifth band empty, the others denser
        m_np &= (((np.arange(V) // 32) % 5) != 0)[:, None]
    mask = torch.from_numpy((m_np * 255).astype(np.uint8)).cuda()
    out = []
    for name, hooks in forms:
        ctx.reset_debug()
        ctx.set_debug(**hooks)
        Ce = torch.zeros((V, U), dtype=torch.float32, device="cuda")
        cem = rs.compute_1D_edge_confidence_pile(vol, S // 2, Ce, p)
        Cd, depth = torch.zeros_like(Ce), torch.zeros_like(Ce)
        rbar = torch.zeros((V, U, C), dtype=torch.float32, device="cuda")
        ms = []
        for _ in range(3):
            st = rs.compute_1D_depth_epi(vol, cfg["dmin"], cfg["dmax"], D, S // 2, Ce, cem, Cd, depth, rbar, p, mask.clone(), want_stats=True)
            ms.append(ctx.last_scan_kernel_ms())
        n = int(st.pixels_scanned)
        out.append("%s %.2f ms (%.2f G/s, k%d)" % (name, min(ms), n * D / min(ms) / 1e6, st.scan_kernel))
    print("density %.2f, %d px: " % (dn, n) + "; ".join(out), flush=True)
