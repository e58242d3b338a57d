"""Time rslf_selective_median on a c3-sized plane (1920 x 1080) for a list of window sizes, one and three channels.
    python tools/probe_median.py [sizes...]        (RSLF_LIBRARY=ab/librslf_x.so for another build)"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from remotesensingproject_amd import depth as rs

sizes = [int(a) for a in sys.argv[1:]] or [3, 5, 7, 9, 11, 15, 31]
U, V, S = 1920, 1080, 3
rng = np.random.default_rng(1)
for C_ in (1, 3):
    vol = rng.uniform(0.2, 1.0, size=(V, S, U, C_)).astype(np.float32)
    vol[:, 1] = np.round(vol[:, 1] * 8) / 8
    v = rs.Volume.from_dense(torch.from_numpy(vol).cuda(), 1.0)
    src = torch.from_numpy((np.round(rng.uniform(-2, 6, size=(V, U)) * 32) / 32).astype(np.float32)).cuda()
    mask = torch.full((V, U), 255, dtype=torch.uint8, device="cuda")
    for size in sizes:
        try:
            for _ in range(3):
                rs.selective_median_filter(src, v, 1, size, mask, 0.1)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 20 if size <= 11 else 5
            e0.record()
            for _ in range(n):
                rs.selective_median_filter(src, v, 1, size, mask, 0.1)
            e1.record()
            torch.cuda.synchronize()
            print("C=%d size %2d: %8.1f us" % (C_, size, e0.elapsed_time(e1) / n * 1e3), flush=True)
        except Exception as ex:  # noqa: BLE001
            print("C=%d size %2d: %s" % (C_, size, str(ex)[:80]), flush=True)
