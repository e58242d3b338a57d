"""bench.py's contract on the GPU box: one JSON line with the roofline / e2e objects, and `--gpus N` typed as is
(the form the driver uses) starting its own rank processes."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None, timeout=600):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=e, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_bench_line_single_gpu_small():
    j = _run(["--config", "c2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["unit"] == "Mpixel*hyp/s" and j["value"] > 0
    assert j["config"]["workload"].startswith("c2:") and "model" not in j["config"]
    rf = j["roofline"]
    assert 0 < rf["frac"] < 1 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and rf["kernel_ms"] > 0
    assert j["e2e"]["ms"] > j["ms_per_step"] * 0.5 and j["e2e"]["value"] > 0      # host in -> host out, PCIe included
    assert j["vs_baseline"] is None and j["dtype"] == "f32" and j["data"] == "synthetic"


def test_bench_gpus_2_as_typed_starts_its_own_ranks():
    """`python bench.py --gpus 2 ...` with no launcher around it: the parent starts two rank processes
    through torch.distributed.run and relays rank 0's line.  On a one-GPU box the ranks share cuda:0 and talk over gloo
    (RSLF_ONE_DEVICE / RSLF_DIST_BACKEND: a rehearsal of the code path, never a result -- the line says so)."""
    j = _run(["--gpus", "2", "--rows", "64", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
             env={"RSLF_DIST_BACKEND": "gloo", "RSLF_ONE_DEVICE": "1"})
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["value"] > 0
    assert "REHEARSAL" in j["data"]
    assert j["config"]["pixels_scanned"] >= 64 * 1920 - 64 and "halo" in j["config"]["sharding"]


def test_bench_path_e2e_one_process_two_workers():
    """`--path e2e --gpus N`: ONE process, N devices behind rslf_multi_create.  On a one-GPU box
    RSLF_E2E_DEVICES="0,0" puts both workers on cuda:0: a rehearsal of the code path, never a result."""
    j = _run(["--path", "e2e", "--gpus", "2", "--config", "c2", "--steps", "2"], env={"RSLF_E2E_DEVICES": "0,0"})
    assert j["path"] == "e2e" and j["n_gpus"] == 2 and j["value"] is None
    assert j["config"]["devices"] == [0, 0] and j["config"]["peer_access"] == [[1, 1], [1, 1]]
    assert j["e2e"]["value"] > 0 and len(j["e2e"]["calls_ms"]) == 3
    one = _run(["--path", "e2e", "--config", "c2", "--steps", "2"])
    assert one["n_gpus"] == 1 and one["e2e"]["value"] > 0


@pytest.mark.parametrize("path", ["sweep2d", "f2c"])
def test_rows_around_the_path_carry_roofline_and_cpu_baseline(path):
    """`--path sweep2d` / `--path f2c`: the line carries `roofline` -- units x algorithmic flops over the
    SUMMED K2 time of one run (every scan launch timed by its own pair of HIP events, a separate instrumented pass) and the
    share of a step's wall time that is K2 -- and a `cpu_baseline` from the oracle's restatement of the same row."""
    j = _run(["--path", path, "--config", "c2", "--rows", "96", "--steps", "2", "--warmup", "1"], env={"RSLF_CPU_THREADS": "8"})
    assert j["config"]["path"] == path and j["value"] > 0
    rf = j["roofline"]
    assert 0 < rf["frac"] < 1 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert rf["kernel_ms"] > 0 and 0 < rf["k2_share"] <= 1.05 and rf["scan_launches"] >= 33
    cb = j["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] == 8 and "scanlines" in cb["sample"]


def test_c1_config_is_the_real_frame():
    j = _run(["--config", "c1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-e2e"])
    assert j["config"]["pixels_scanned"] == 223271 and "000.tif" in j["data"]
    assert j["config"]["scan_kernel"].startswith("k2_scan_reg<16,1>")
