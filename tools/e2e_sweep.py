"""Host-in / host-out time of the pipelined path (rslf_multi_*) against the chunk size and the number of workers on
the one GPU (developer tool).
    WORKERS=2 python tools/e2e_sweep.py [config] [chunk_rows ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from remotesensingproject_amd import depth as rs
from remotesensingproject_amd.synth import make_config

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
chunks = [int(x) for x in sys.argv[2:]] or [0, 270, 135, 90, 68, 45, 34]
vol, _, c = make_config(name)
epis = list(vol[..., 0]) if c["C"] == 1 else list(vol)
workers = int(os.environ.get("WORKERS", "1"))
m = rs.MultiDevice([0] * workers)
for ch in chunks:
    m.set_chunk_rows(ch)
    ts = []
    for _ in range(4):
        t0 = time.perf_counter()
        m.depth1d_pile(epis, c["dmin"], c["dmax"], c["D"], epi_scale_factor=1.0)
        ts.append(time.perf_counter() - t0)
    print("%s workers %d chunk_rows %4d: %s ms" % (name, workers, ch, " ".join("%.1f" % (t * 1e3) for t in ts)), flush=True)
