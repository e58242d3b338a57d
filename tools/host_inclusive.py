"""PCIe-inclusive rate of the host-pointer path (DESIGN.md "Measurement"): host EPIs in,
host planes out, for a synthetic config.  Developer tool; bench.py's `value` never includes PCIe."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from remotesensingproject_amd import depth as rs
from remotesensingproject_amd.synth import make_config, CONFIGS

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
vol, _, c = make_config(name)
epis = list(vol[..., 0]) if c["C"] == 1 else list(vol)
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    v = rs.Volume.from_epis(epis, 1.0)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    comp = rs.Depth1DComputer_pile(v, c["dmin"], c["dmax"], c["D"])
    comp.run(want_stats=False)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    res = comp.results()
    t3 = time.perf_counter()
    units = c["U"] * c["V"] * c["D"]
    print("%s rep %d: upload+pack %.1f ms (%.2f GB/s), kernels %.1f ms, download %.1f ms, total %.1f ms -> %.0f M units/s PCIe-inclusive" % (
        name, rep, (t1 - t0) * 1e3, vol.nbytes / (t1 - t0) / 1e9, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t3 - t0) * 1e3, units / (t3 - t0) / 1e6), flush=True)
    del comp, v
