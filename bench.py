#!/usr/bin/env python3
"""Throughput of the EPI depth scan on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c3]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the hot path (K1 edge confidence, compaction, K2 scan, K3
selective median; for N > 1 also the RCCL gather that reassembles the depth map
on rank 0) over one synthetic light field that is already resident in HBM.
N = 1 runs BASELINE.json configs[2] (1920x1080, 101 views, 256 hypotheses); for
N > 1 the SAME sweep is sharded by scanline (configs[3]): strong scaling.
Prints one JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_VECTOR_TFLOPS = 157.3   # MI355X_MICROARCH.md "Peak FP32 (vector)"
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md "HBM3E peak BW"


def algorithmic_flops_per_unit(S: int, C: int) -> int:
    """each add/sub/mul/div/max/floor/ceil of the reference's
    arithmetic counted once per (pixel, hypothesis)."""
    return 95 * S + 21 if C == 1 else 230 * S + 40


def algorithmic_bytes_per_pixel(S: int, C: int) -> int:
    """every input voxel read once, every output written once."""
    return 4 * S * C + 8 + 13 + 4 * C


def measured_traffic(key):
    """roofline.traffic = HBM bytes per launch from the PMC counters of THIS code (profiles/k2_traffic.json, written by
    tools/summarize_profiles.py from a rocprofv3 --pmc run of this very command).  Every entry carries the hash of the
    sources it was measured on; an entry measured on other code is not quoted (traffic: null, and the source says why)."""
    tpath = os.path.join(ROOT, "profiles", "k2_traffic.json")
    if key is None:
        return None, "none: a developer run with --rows has no committed counter run"
    try:
        from remotesensingproject_amd import build as hb
        have = hb.source_hash()
        e = json.load(open(tpath)).get(key)
    except Exception as ex:  # noqa: BLE001
        return None, "none: %s" % ex
    if not e:
        return None, "none: profiles/k2_traffic.json has no entry %s" % key
    if e.get("source_hash") != have:
        return None, "stale: profiles/k2_traffic.json[%s] was measured on sources %s, this tree is %s (%.4g bytes then)" % (
            key, e.get("source_hash"), have, e.get("hbm_bytes_per_launch", float("nan")))
    return e["hbm_bytes_per_launch"], "profiles/k2_traffic.json[%s]: rocprofv3 --pmc FETCH_SIZE (calibrated on k0_pack) + WRITE_SIZE, %s, sources %s" % (
        key, e.get("collected", "undated"), have)


def cpu_baseline(cfg: dict, budget_s: float = 12.0) -> dict:
    """The CPU oracle (a port of the reference's arithmetic, OpenMP over scanlines like
    core.hpp:799) timed on this host on a bounded sample: whole scanlines of the same workload."""
    import oracle
    from remotesensingproject_amd.synth import make_lightfield
    # threads = the CPU share this box gives us (16 of the host's cores for one GPU), capped for the report
    threads = int(os.environ.get("RSLF_CPU_THREADS", "0")) or min(oracle.usable_cpus(), 16)
    oracle.set_num_threads(threads)
    rows = max(2 * threads, 8)
    if cfg["seed"] is None:   # c1: scanlines 400.. of data/000.tif (the committed frame), 9 identical views (see main)
        img = np.load(os.path.join(ROOT, "tests", "golden", "c1_000tif_960x540_f32.npz"))["image"]
        tif_max = json.load(open(os.path.join(ROOT, "tests", "golden", "c1_anchor.json")))["tif_max"]
        plane = img[(400 + np.arange(rows)) % img.shape[0]] * np.float32(1.0 / float(tif_max))
        vol = np.ascontiguousarray(np.repeat(plane[:, None, :, None], cfg["S"], axis=1), np.float32)
    else:
        vol, _ = make_lightfield(cfg["U"], rows, cfg["S"], cfg["C"], seed=cfg["seed"], dmin=cfg["dmin"], dmax=cfg["dmax"])
    # units = pixels whose mask is set on entry to the scan (core.hpp:515-527) x hypotheses
    pixels = int((oracle.edge_confidence_pile(vol, cfg["S"] // 2)[1] > 0).sum())
    oracle.depth1d_pile_run(vol[:threads], cfg["dmin"], cfg["dmax"], cfg["D"])   # untimed: thread pool, page faults
    units, elapsed, reps = 0, 0.0, 0
    while elapsed < budget_s and reps < 64:
        t0 = time.perf_counter()
        oracle.depth1d_pile_run(vol, cfg["dmin"], cfg["dmax"], cfg["D"])
        elapsed += time.perf_counter() - t0
        units += pixels * cfg["D"]
        reps += 1
    return {
        "value": units / elapsed / 1e6,
        "unit": "Mpixel*hyp/s",
        "cores": threads,
        "kind": "port",
        "sample": "%d scanlines x %d px x %d views x %d hypotheses of the same field, %d passes, %.1f s" % (
            rows, cfg["U"], cfg["S"], cfg["D"], reps, elapsed),
    }


def end_to_end(rs, host: np.ndarray, cfg: dict, units: int, device: int) -> dict:
    """BASELINE.md section 3's second figure: the drop-in signature is host buffers in, host planes out
    (dc.hpp:97-122).  One call of the pipelined host path (rslf_multi_*: scanline chunks, upload of chunk k+1 and
    download of chunk k-1 behind the kernels of chunk k), PCIe included, pageable memory on both sides.  Never `value`."""
    epis = list(host[..., 0]) if cfg["C"] == 1 else list(host)
    m = rs.MultiDevice([device])
    ts = []
    calls = 5 if host.nbytes <= (4 << 30) else 3     # the first call sizes the device buffers; c5 (21 GB in) takes seconds per call
    for _ in range(calls):
        t0 = time.perf_counter()
        m.depth1d_pile(epis, cfg["dmin"], cfg["dmax"], cfg["D"], epi_scale_factor=1.0)
        ts.append(time.perf_counter() - t0)
    m.close()
    print("e2e calls (ms): " + " ".join("%.1f" % (x * 1e3) for x in ts), file=sys.stderr)
    t = sorted(ts[1:])[(calls - 1) // 2]              # median of the calls after the first
    # the upload alone (Depth1DComputer_pile's constructor on the same host EPIs: copy + normalise + pack, dc.hpp:425-477):
    # what the PCIe link delivers from pageable memory, for the reader who wants e2e minus kernels explained
    t0 = time.perf_counter()
    v = rs.Volume.from_epis(epis, 1.0, rs.default_context(device))
    up = time.perf_counter() - t0
    v.close()
    return {"ms": t * 1e3, "value": units / t / 1e6, "unit": "Mpixel*hyp/s",
            "what": "host EPIs (Vec<Mat>-style, pageable) in -> host planes out, one call; upload, kernels and download "
                    "overlap in scanline chunks with a recomputed halo (rslf_multi_depth1d_pile_f32); median of the %d calls "
                    "after the first" % (calls - 1),
            "input_gb": host.nbytes / 1e9, "upload_alone_ms": up * 1e3, "upload_alone_gbs": host.nbytes / 1e9 / up}


PEAK_FP32_TFLOPS = 157.3   # MI355X fp32 vector peak (MI355X_MICROARCH.md), FMA counted as two


def flops_per_unit(S: int, C: int) -> int:
    """algorithmic flops of one (pixel, hypothesis) unit."""
    return 95 * S + 21 if C == 1 else 230 * S + 40


def rows_roofline(ctx, run_once, units: int, S: int, C: int, wall_ms: float) -> dict:
    """`roofline` of the rows around the path (2-D sweep, fine-to-coarse): one more, instrumented pass -- every scan launch
    bracketed by its own pair of HIP events (rslf_ctx_set_debug "time_all") -- gives the SUMMED K2 time of the run; achieved =
    units x algorithmic flops / that time.  `k2_share` = summed K2 time / wall time of an (un-instrumented) step."""
    ctx.set_debug(time_all=1)
    ctx.scan_time_total_ms()
    run_once()
    k2_ms, launches = ctx.scan_time_total_ms()
    ctx.set_debug(time_all=0)
    flops = units * flops_per_unit(S, C)
    achieved = flops / (k2_ms * 1e-3) / 1e12
    return {"bound": "fp32 VALU (non-MFMA)", "achieved": achieved, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_TFLOPS,
            "traffic": None, "kernel_ms": k2_ms, "scan_launches": launches, "k2_share": k2_ms / wall_ms,
            "what": "units x %d algorithmic flop/unit (SURVEY 8d) / summed K2 time of one run (HIP events around every scan launch, "
                    "a separate instrumented pass); k2_share = that time / wall time of a step" % flops_per_unit(S, C)}


def rows_cpu_baseline(kind: str, cfg: dict, budget_s: float = 12.0) -> dict:
    """`cpu_baseline` of the same rows: the oracle's Depth2DComputer / FineToCoarse restatement on a bounded sample of the
    same field (a block of whole scanlines, all views), OpenMP over scanlines as the reference (core.hpp:799)."""
    import oracle
    from remotesensingproject_amd.synth import make_lightfield
    threads = int(os.environ.get("RSLF_CPU_THREADS", "0")) or min(oracle.usable_cpus(), 16)
    oracle.set_num_threads(threads)
    rows = max(2 * threads, 24) if kind == "sweep2d" else max(4 * threads, 48)     # f2c: enough rows for a few pyramid levels
    rows = min(rows, cfg["V"])
    vol, _ = make_lightfield(cfg["U"], rows, cfg["S"], cfg["C"], seed=cfg["seed"], dmin=cfg["dmin"], dmax=cfg["dmax"])
    units, elapsed, reps = 0, 0.0, 0
    while elapsed < budget_s and reps < 16:
        oracle.sweep_pixels_scanned(reset=True)
        t0 = time.perf_counter()
        if kind == "sweep2d":
            oracle.depth2d_run(vol, cfg["dmin"], cfg["dmax"], cfg["D"])
        else:
            oracle.fine_to_coarse_run((vol * 200.0 + 3.0).astype(np.float32), cfg["dmin"], cfg["dmax"], cfg["D"])
        elapsed += time.perf_counter() - t0
        scanned = oracle.sweep_pixels_scanned(reset=True)
        if not scanned:
            raise RuntimeError("the oracle did not report its scanned pixels")
        units += scanned * cfg["D"]
        reps += 1
    return {"value": units / elapsed / 1e6, "unit": "Mpixel*hyp/s", "cores": threads, "kind": "port",
            "sample": "%s of %d scanlines x %d px x %d views x %d ch x %d hypotheses of the same field (units = pixels the oracle "
                      "scanned x hypotheses), %d passes, %.1f s" % (
                          "Depth2DComputer::run" if kind == "sweep2d" else "FineToCoarse ctor + run + get_results",
                          rows, cfg["U"], cfg["S"], cfg["C"], cfg["D"], reps, elapsed)}


def apply_bench_hooks(ctx) -> None:
    """RSLF_BENCH_HOOKS="px=0,share=2,groups=8,lds=64": rslf_ctx_set_debug hooks for developer A/B runs (never a result)."""
    if os.environ.get("RSLF_BENCH_HOOKS"):
        names = {"share": "stream_share", "groups": "stream_groups", "lds": "stream_lds_kib"}
        ctx.set_debug(**{names.get(k, k): int(v) for k, v in (kv.split("=") for kv in os.environ["RSLF_BENCH_HOOKS"].split(",")) if not k.startswith("_")})


def pick_config(args, default_for_c3: str | None = None) -> tuple[dict, str]:
    """--config, or --shape U,V,S,C,D,dmin,dmax for a synthetic field of another size (context lines)."""
    from remotesensingproject_amd.synth import CONFIGS
    if args.shape:
        U, V, S, C_, D = [int(x) for x in args.shape.split(",")[:5]]
        dmin, dmax = [float(x) for x in args.shape.split(",")[5:7]]
        return dict(U=U, V=V, S=S, C=C_, D=D, dmin=dmin, dmax=dmax, seed=20260099), "shape " + args.shape
    name = args.config if not (default_for_c3 and args.config == "c3" and not args.config_given) else default_for_c3
    return dict(CONFIGS[name]), name


def bench_sweep2d(args) -> None:
    """Depth2DComputer::run() (dc.hpp:748-805) on a synthetic field: a step = edge confidence of every view
    + one visit per view (scan on the running mask, median, propagation).  Units = pixels actually scanned,
    summed over the visits, x hypotheses."""
    import torch
    from remotesensingproject_amd import depth as rs
    from remotesensingproject_amd.synth import CONFIGS, make_lightfield
    if args.gpus != 1:
        raise SystemExit("the 2-D sweep bench is single-GPU")
    cfg, cfg_name = pick_config(args, "c2")
    if args.rows:
        cfg["V"] = args.rows
    U, V, S, C, D = cfg["U"], cfg["V"], cfg["S"], cfg["C"], cfg["D"]
    torch.cuda.set_device(0)
    host, _ = make_lightfield(U, V, S, C, seed=cfg["seed"], dmin=cfg["dmin"], dmax=cfg["dmax"])
    ctx = rs.default_context(0)
    apply_bench_hooks(ctx)
    vol = rs.Volume.from_dense(torch.from_numpy(host).cuda(), 1.0, ctx)
    comp = rs.Depth2DComputer(vol, cfg["dmin"], cfg["dmax"], D)
    for _ in range(max(args.warmup, 1)):
        comp.run(want_stats=True)
    torch.cuda.synchronize()
    units = int(comp.stats.units)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        comp.run(want_stats=False)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    roof = rows_roofline(ctx, lambda: (comp.run(want_stats=False), torch.cuda.synchronize()), units, S, C, elapsed / args.steps * 1e3)
    line = {
        "metric": "Mpixel*disparity-hypotheses/s", "value": units / (elapsed / args.steps) / 1e6, "unit": "Mpixel*hyp/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "2-D sweep (Depth2DComputer::run) over %s: %dx%d px x %d views x %d ch, %d hypotheses; %d visits, "
                               "%d pixels scanned in total (%.2f views' worth)" % (
                                   cfg_name, U, V, S, C, D, S, units // D, units / D / (U * V)),
                   "path": "sweep2d"},
        "roofline": roof,
    }
    if not args.no_cpu_baseline:
        line["cpu_baseline"] = rows_cpu_baseline("sweep2d", cfg)
    print(json.dumps(line), flush=True)


def bench_f2c(args) -> None:
    """FineToCoarse constructor + run() + get_results() (rslf_fine_to_coarse.hpp) on a synthetic field, raw
    float input in [3, 203): a step = pyramid construction, every level's 2-D sweep, bound tightening and fusion."""
    import torch
    from remotesensingproject_amd import depth as rs
    from remotesensingproject_amd.synth import CONFIGS, make_lightfield
    cfg, cfg_name = pick_config(args, "c2")
    if args.rows:
        cfg["V"] = args.rows
    U, V, S, C, D = cfg["U"], cfg["V"], cfg["S"], cfg["C"], cfg["D"]
    torch.cuda.set_device(0)
    host, _ = make_lightfield(U, V, S, C, seed=cfg["seed"], dmin=cfg["dmin"], dmax=cfg["dmax"])
    raw = (host * 200.0 + 3.0).astype(np.float32)
    apply_bench_hooks(rs.default_context())

    def once():
        f = rs.FineToCoarse(raw, cfg["dmin"], cfg["dmax"], D)
        f.run()
        m, v = f.get_results()
        torch.cuda.synchronize()
        return f

    f = None
    for _ in range(max(args.warmup, 1)):
        f = once()
    units = sum(int(c.stats.units) for c in f.m_computers)
    dims = [(c.m_epis.V, c.m_epis.U) for c in f.m_computers]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        once()
    elapsed = time.perf_counter() - t0
    roof = rows_roofline(rs.default_context(), once, units, S, C, elapsed / args.steps * 1e3)
    line = {
        "metric": "Mpixel*disparity-hypotheses/s", "value": units / (elapsed / args.steps) / 1e6, "unit": "Mpixel*hyp/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "fine-to-coarse (FineToCoarse ctor + run + get_results, host upload included) over %s: %dx%d px x %d views "
                               "x %d ch, %d hypotheses; pyramid %s" % (cfg_name, U, V, S, C, D, dims),
                   "path": "f2c", "pixels_scanned": units // D},
        "roofline": roof,
    }
    if not args.no_cpu_baseline:
        line["cpu_baseline"] = rows_cpu_baseline("f2c", cfg)
    print(json.dumps(line), flush=True)


def bench_e2e(args) -> None:
    """`--path e2e [--gpus N]`: the pipelined host path alone -- host EPIs in, host planes out (dc.hpp:97-122) -- in ONE
    process that drives N devices through rslf_multi_create([0..N-1]) (one host thread per device, scanline blocks, no
    collective).  RSLF_E2E_DEVICES="0,0" rehearses two workers on one GPU (never a result).  PCIe-inclusive: the line
    carries no `value`; the figure is `e2e`.  N > 1 has not run on hardware here (one-GPU boxes)."""
    import torch
    from remotesensingproject_amd import depth as rs
    from remotesensingproject_amd.synth import make_lightfield
    cfg, cfg_name = pick_config(args)
    if args.rows:
        cfg["V"] = args.rows
    if cfg["seed"] is None:
        raise SystemExit("--path e2e takes the synthetic configs (c2, c3, c5) or --shape")
    devices = [int(x) for x in os.environ["RSLF_E2E_DEVICES"].split(",")] if os.environ.get("RSLF_E2E_DEVICES") else list(range(args.gpus))
    if max(devices) >= torch.cuda.device_count():
        raise SystemExit("--gpus %d but %d device(s) visible" % (args.gpus, torch.cuda.device_count()))
    U, V, S, C, D = cfg["U"], cfg["V"], cfg["S"], cfg["C"], cfg["D"]
    host, _ = make_lightfield(U, V, S, C, seed=cfg["seed"], dmin=cfg["dmin"], dmax=cfg["dmax"])
    epis = list(host[..., 0]) if C == 1 else list(host)
    m = rs.MultiDevice(devices)
    peer = m.peer_access()
    ts, units = [], 0
    calls = max(2, args.steps + 1) if host.nbytes <= (4 << 30) else 3
    for _ in range(calls):
        t0 = time.perf_counter()
        out = m.depth1d_pile(epis, cfg["dmin"], cfg["dmax"], D, epi_scale_factor=1.0)
        ts.append(time.perf_counter() - t0)
    units = int((out["edge_mask"] > 0).sum()) * D
    m.close()
    t = sorted(ts[1:])[(calls - 1) // 2]
    print(json.dumps({
        "metric": "Mpixel*disparity-hypotheses/s", "value": None, "unit": "Mpixel*hyp/s", "n_gpus": len(devices), "path": "e2e",
        "higher_is_better": True, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s: %dx%d px x %d views x %d ch, %d hypotheses in [%g, %g], seed %d" % (cfg_name, U, V, S, C, D, cfg["dmin"], cfg["dmax"], cfg["seed"]),
                   "devices": devices, "peer_access": peer},
        "e2e": {"ms": t * 1e3, "value": units / t / 1e6, "unit": "Mpixel*hyp/s", "calls_ms": [x * 1e3 for x in ts], "input_gb": host.nbytes / 1e9,
                "what": "host EPIs (pageable) in -> host planes out through rslf_multi_depth1d_pile_f32 on %d device(s) of one process; "
                        "median of the calls after the first; PCIe-inclusive, never `value`" % len(devices)},
    }), flush=True)


def launch_ranks(n: int) -> None:
    """`python bench.py --gpus N` typed as is: this process never touches the GPU; it starts N fresh rank
    processes (one per GPU) through torch.distributed.run with the same arguments, lets rank 0's JSON line
    through, and exits with their return code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    sys.exit(subprocess.run(cmd, env=env).returncode)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default=None, help="synthetic config of BASELINE.md section 4 (c1, c2, c3, c5; default c3), or a published shape of the reference's "
                                                    "report for the sweep2d / f2c paths (skysat_lr, mansion_lr)")
    ap.add_argument("--shape", default="", help="U,V,S,C,D,dmin,dmax: a synthetic field of another size (sweep2d / f2c paths)")
    ap.add_argument("--rows", type=int, default=0, help="override the number of scanlines (developer runs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the host-in / host-out figure (the `e2e` key)")
    ap.add_argument("--path", default="pile", choices=["pile", "sweep2d", "f2c", "e2e"],
                    help="pile = Depth1DComputer_pile::run (the headline path); sweep2d = Depth2DComputer::run, the 'next' row "
                         "(all views, centre outwards, with propagation), 1 GPU only")
    args = ap.parse_args()
    args.config_given = args.config is not None
    args.config = args.config or "c3"
    if args.path == "e2e":      # one process, N devices behind the C-ABI (no ranks)
        return bench_e2e(args)
    if args.gpus > 1 and "RANK" not in os.environ:
        return launch_ranks(args.gpus)
    if args.path == "sweep2d":
        return bench_sweep2d(args)
    if args.path == "f2c":
        return bench_f2c(args)

    import torch
    import torch.distributed as dist
    from remotesensingproject_amd import depth as rs
    from remotesensingproject_amd import sharding
    from remotesensingproject_amd.synth import CONFIGS, make_lightfield

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # RSLF_DIST_BACKEND=gloo + RSLF_ONE_DEVICE=1 rehearse the N > 1 path on a one-GPU box (never a result)
    backend = os.environ.get("RSLF_DIST_BACKEND", "nccl")
    if os.environ.get("RSLF_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)   # RCCL over xGMI
        else:
            dist.init_process_group(backend)

    cfg = dict(CONFIGS[args.config])
    if args.rows:
        cfg["V"] = args.rows
    U, V, S, C, D = cfg["U"], cfg["V"], cfg["S"], cfg["C"], cfg["D"]
    params = rs.Depth1DParameters()
    shard = sharding.make_shard(V, rank, world, params.par_median_filter_size, params.par_edge_confidence_opening_size)

    if cfg["seed"] is None:
        # c1 (BASELINE.json configs[0]): data/000.tif replicated into 9 identical views.  The decoded frame travels as
        # a fixture (tests/golden/c1_000tif_960x540_f32.npz: data, 960 scanlines x 540 columns) and is normalised by its
        # max as the constructor does (dc.hpp:474); --rows beyond 960 wrap around.
        img = np.load(os.path.join(ROOT, "tests", "golden", "c1_000tif_960x540_f32.npz"))["image"]
        tif_max = json.load(open(os.path.join(ROOT, "tests", "golden", "c1_anchor.json")))["tif_max"]
        rows_v = np.arange(V)[shard.rows] % img.shape[0]
        plane = img[rows_v] * np.float32(1.0 / float(tif_max))
        host = np.ascontiguousarray(np.repeat(plane[:, None, :, None], S, axis=1), np.float32)
    else:
        # synthetic light field: every rank draws the same textures and keeps its scanlines (+halo)
        host, _ = make_lightfield(U, V, S, C, seed=cfg["seed"], dmin=cfg["dmin"], dmax=cfg["dmax"], rows=shard.rows)
    ctx = rs.default_context(dev)
    apply_bench_hooks(ctx)
    vol = rs.Volume.from_dense(torch.from_numpy(host).to(dev), 1.0, ctx)
    want_e2e = world == 1 and not args.no_e2e
    if not want_e2e:
        del host
    comp = rs.Depth1DComputer_pile(vol, cfg["dmin"], cfg["dmax"], D, parameters=params)

    def planes():
        return dict(edge_confidence=comp.m_edge_confidence_v_u, disp_confidence=comp.m_disp_confidence_v_u,
                    depth=comp.m_best_depth_v_u, depth_raw=comp.m_depth_raw_v_u, score=comp.m_score_v_u,
                    depth_idx=comp.m_depth_idx_v_u, rbar=comp.m_rbar_v_u, edge_mask=comp.m_edge_confidence_mask_v_u)

    k2_ms, gather_ms = [], []
    gatherer = sharding.PlaneGatherer(shard, U, C, dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)] if world > 1 else None

    def step(record: bool):
        comp.run(want_stats=False)
        if ev:
            ev[0].record()
        out = gatherer(planes())   # N > 1: RCCL gather of the owned rows onto rank 0
        if ev:
            ev[1].record()
        if record:
            k2_ms.append(ctx.last_scan_kernel_ms())   # HIP events on the launching stream
            if ev:
                ev[1].synchronize()
                gather_ms.append(ev[0].elapsed_time(ev[1]))   # enqueue-to-done of the gather on this rank
        return out

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step(True)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # units = mask-selected pixels actually scanned x hypotheses, owned rows only:
    # the scan mask on entry is the edge mask K1 produces (core.hpp:513)
    ce0 = torch.zeros((vol.V, U), dtype=torch.float32, device=dev)
    m0 = rs.compute_1D_edge_confidence_pile(vol, comp.get_s_hat(), ce0, params)
    scanned_local = int((m0[shard.interior] > 0).sum().item())
    comp.run(want_stats=True)
    torch.cuda.synchronize(dev)
    n = torch.tensor([scanned_local], dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(n)
    pixels = int(n.item())
    # per-rank figures for the N > 1 line: a slow gather and a short grid look the same in ms_per_step alone
    per_rank = None
    if world > 1:
        mine = torch.tensor([float(np.mean(k2_ms)), float(np.mean(gather_ms)) if gather_ms else 0.0, float(shard.hi - shard.lo),
                             float(shard.v1 - shard.v0)],
                            dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [[float(x) for x in t.tolist()] for t in allr]
    units_per_step = pixels * D
    ms_per_step = elapsed / args.steps * 1e3
    value = units_per_step / (elapsed / args.steps) / 1e6

    if rank == 0:
        k2_avg_ms = float(np.mean(k2_ms))
        units_launch = int(comp.stats.units)          # what rank 0's K2 launch processed (incl. halo rows)
        flops = algorithmic_flops_per_unit(S, C)
        achieved_tflops = units_launch * flops / (k2_avg_ms * 1e-3) / 1e12
        traffic, traffic_source = measured_traffic("%s_n%d" % (args.config, world) if not args.rows else None)
        px_launch = units_launch // D
        hbm_bytes = px_launch * algorithmic_bytes_per_pixel(S, C)
        line = {
            "metric": "Mpixel*disparity-hypotheses/s",
            "value": value,
            "unit": "Mpixel*hyp/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": ("synthetic" if cfg["seed"] is not None else "the reference's data/000.tif (decoded frame, a committed fixture), 9 identical views") if backend == "nccl" else "synthetic (REHEARSAL: %s backend, not a result)" % backend,
            "config": {
                "workload": "%s: %dx%d px x %d views x %d ch, %d hypotheses in [%g, %g], %s" % (
                    args.config, U, V, S, C, D, cfg["dmin"], cfg["dmax"],
                    "seed %d, all pixels confident" % cfg["seed"] if cfg["seed"] is not None else
                    "data/000.tif, %d scanlines, %d identical views, real edge mask" % (V, S)),
                "sharding": "none" if world == 1 else "scanline blocks + %d-row recomputed halo, RCCL gather of the output planes per step" % sharding.halo_rows(
                    params.par_median_filter_size, params.par_edge_confidence_opening_size),
                "scan_kernel": {1: "k2_scan_reg<%d,%d>" % (comp.stats.s_pad, C), 2: "k2_scan_stream<%d>" % C, 3: "k2_scan_chip",
                                4: "k2_scan_reg_px<%d,%d>" % (comp.stats.s_pad, C), 5: "k2_scan_stream_px<%d>" % C}.get(
                    comp.stats.scan_kernel, "k2_scan_generic<%d>" % C),
                "pixels_scanned": pixels,
            },
            "roofline": {
                "bound": "valu_fp32",
                "kernel": "k2_scan",
                "achieved": achieved_tflops,
                "peak": PEAK_FP32_VECTOR_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved_tflops / PEAK_FP32_VECTOR_TFLOPS,
                "traffic": traffic,
                "traffic_source": traffic_source,
                "flops_per_unit": flops,
                "units_per_launch": units_launch,
                "kernel_ms": k2_avg_ms,
                "note": "algorithmic flops (SURVEY 8d) / K2 HIP-event time; the reference's arithmetic forbids FMA, "
                        "so the reachable ceiling is the non-FMA issue rate (DESIGN.md)",
            },
            "roofline_hbm": {
                "bound": "hbm",
                "achieved": hbm_bytes / (k2_avg_ms * 1e-3) / 1e9,
                "peak": PEAK_HBM_GBS,
                "unit": "GB/s",
                "frac": hbm_bytes / (k2_avg_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                "bytes_per_unit": algorithmic_bytes_per_pixel(S, C) / D,
                "note": "compulsory-byte model; this path is VALU-bound, not HBM-bound (SURVEY 8d)",
            },
        }
        if per_rank:
            line["per_rank"] = {"k2_ms": [r[0] for r in per_rank], "gather_ms": [r[1] for r in per_rank],
                                "rows_held": [int(r[2]) for r in per_rank], "rows_owned": [int(r[3]) for r in per_rank],
                                "note": "K2 HIP-event time and enqueue-to-done time of the plane gather on every rank, means over the steps"}
        if want_e2e:
            line["e2e"] = end_to_end(rs, host, cfg, units_per_step, local_rank)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg)
        print(json.dumps(line), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
