"""Developer probe: gaps between dependent launches of the library's own kernels (run under rocprofv3 --kernel-trace)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from remotesensingproject_amd import _lib
from remotesensingproject_amd import depth as rs
from remotesensingproject_amd.synth import make_lightfield

vol_np, _ = make_lightfield(512, 512, 9, 1, seed=1, dmin=-1, dmax=1, band=8)
vol = rs.Volume.from_dense(torch.from_numpy(vol_np).cuda(), 1.0)
L = _lib.lib()
ctx = vol.ctx
src = torch.rand((512, 512), device="cuda")
dst = torch.empty_like(src)
mask = (torch.rand((512, 512), device="cuda") > 0.5).to(torch.uint8) * 255
Ce = torch.zeros((512, 512), device="cuda")
m2 = torch.empty((512, 512), dtype=torch.uint8, device="cuda")
p = rs.Depth1DParameters().to_c()
ctx.use_current_stream()
vp = C.c_void_p
torch.cuda.synchronize()
for size in (5, 1, 3):
    for _ in range(20):
        L.rslf_selective_median(ctx._h, vol._h, vp(src.data_ptr()), vp(dst.data_ptr()), 4, size, vp(mask.data_ptr()), C.c_float(0.1))
torch.cuda.synchronize()
for _ in range(20):
    L.rslf_edge_confidence_pile(ctx._h, vol._h, 4, C.byref(p), vp(Ce.data_ptr()), vp(m2.data_ptr()))
torch.cuda.synchronize()
for _ in range(20):
    dst.copy_(src)
torch.cuda.synchronize()
