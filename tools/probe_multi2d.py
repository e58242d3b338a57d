"""Developer probe: the 2-D sweep and fine-to-coarse behind the C-ABI over 1 / 2 / 4 contexts on the ONE GPU of the box --
what the per-visit exchange and the host-side round trips cost when the devices do not add compute (c2 volume)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from remotesensingproject_amd import depth as rs
from remotesensingproject_amd.synth import CONFIGS, make_lightfield

cfg = dict(CONFIGS["c2"])
host, _ = make_lightfield(cfg["U"], cfg["V"], cfg["S"], cfg["C"], seed=cfg["seed"], dmin=cfg["dmin"], dmax=cfg["dmax"])
epis = list(host[..., 0])
for nd in (1, 2, 4):
    m = rs.MultiDevice([0] * nd)
    for name, f in (("sweep", lambda: m.depth2d(epis, cfg["dmin"], cfg["dmax"], cfg["D"], epi_scale_factor=1.0)),
                    ("f2c", lambda: m.fine_to_coarse(epis, cfg["dmin"], cfg["dmax"], cfg["D"], epi_scale_factor=1.0))):
        ts = []
        for _ in range(4):
            t0 = time.perf_counter(); f(); ts.append((time.perf_counter() - t0) * 1e3)
        print("%d context(s) %-5s host in -> host out: %s ms" % (nd, name, " ".join("%.1f" % t for t in ts)), flush=True)
    m.close()
