"""Replay one fuzz_parity case under variations of the launch shape (developer tool).
    python tools/fuzz_replay.py SEED CASE"""
import importlib.util, os, sys
import numpy as np
spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz_parity.py"))
fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
seed, case = int(sys.argv[1]), int(sys.argv[2])
for groups, force in ((None, None), (1, None), (2, None), (4, None), (None, "generic"), (None, None)):
    rng = np.random.default_rng([seed, case])
    c = fz.draw_case(rng)
    if groups is not None:
        c["groups"] = groups
    if force is not None:
        c["force"] = force
    try:
        k, n = fz.run_case(case, c, rng)
        print("groups", c["groups"], "force", c["force"], "packed", c["packed"], "-> ok (kernel %d, %d px)" % (k, n), flush=True)
    except AssertionError as e:
        print("groups", c["groups"], "force", c["force"], "packed", c["packed"], "-> FAIL", str(e)[:200], flush=True)
