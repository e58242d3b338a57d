#!/usr/bin/env python3
"""One command for the first multi-GPU lease: does every multi-device path give, on N real MI355X devices, bit for bit what
one device gives?

    python tools/multi_gpu_selftest.py [--gpus N]        (default: every visible device; N = 1 still runs, as a rehearsal)

None of this has run on more than one physical GPU before (the development boxes have one): the branches that copy between
DIFFERENT devices -- hipMemcpyPeerAsync of the device-out pile path, the sweep's per-visit boundary-row fetch, the
fine-to-coarse row transfers, and RCCL itself -- are exercised here for the first time.  What is checked, on c2-sized
fields (512 x 512, 33 views, 128 hypotheses; core.hpp:743-757, :799-854, :933-1133; f2c.hpp:103-299):

  1. peer access: hipDeviceCanAccessPeer / enabled, per pair of workers (rslf_multi_peer_access) -- printed as a matrix;
  2. rslf_multi_depth1d_pile_f32 (host planes out) and _f32_dev (planes left on device 0, peer copies) over N devices
     == the one-device pile run;
  3. rslf_multi_depth2d_run_f32 (sharded sweep, peer-copy exchange per visit) == the one-device sweep;
  4. rslf_multi_fine_to_coarse_run_host over N devices == the one-device run;
  5. one process per GPU over torch.distributed / RCCL (started here with torch.distributed.run): PlaneGatherer on the
     pile path, ShardedDepth2D with its RCCL neighbour exchange, ShardedFineToCoarse -- each == the one-device result
     computed on rank 0.

Exit code 0 = every comparison was bit-identical.  Prints one line per check and the time each form took.
"""
from __future__ import annotations

import argparse
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PILE = ("edge_confidence", "edge_mask", "disp_confidence", "depth", "rbar", "depth_idx", "score", "depth_raw")
SWEEP = ("edge_confidence", "edge_mask", "disp_confidence", "depth", "rbar")


def field(V=512, U=512, S=33, C=1, seed=20260001):
    from remotesensingproject_amd.synth import make_lightfield
    vol, _ = make_lightfield(U, V, S, C, seed=seed, dmin=-1.0, dmax=2.96875)
    return vol


def same(tag, got, ref, keys, t):
    bad = [k for k in keys if not np.array_equal(np.asarray(got[k]), np.asarray(ref[k]))]
    print("%-72s %s   %.1f ms" % (tag, "ok" if not bad else "MISMATCH in " + ", ".join(bad), t * 1e3), flush=True)
    return not bad


def one_process(n: int) -> bool:
    import torch
    from remotesensingproject_amd import depth as rs
    ok = True
    devs = list(range(n))
    vol = field()
    epis = [vol[v, :, :, 0] for v in range(vol.shape[0])]
    dmin, dmax, D = -1.0, 2.96875, 128
    m = rs.MultiDevice(devs)
    pa = m.peer_access()
    print("peer access between workers (1 = direct over xGMI or same GPU, 0 = staged through the host):")
    for i, row in enumerate(pa):
        print("   worker %d (device %d): %s" % (i, devs[i], " ".join(str(x) for x in row)))
    if n > 1 and any(0 in row for row in pa):
        print("   NOTE: some pairs have no peer access: their copies stage through the host (correct, slower)")
    # one-device references
    comp = rs.Depth1DComputer_pile(epis, dmin, dmax, D, epi_scale_factor=1.0)
    comp.run()
    ref_pile = comp.results()
    c2 = rs.Depth2DComputer(epis, dmin, dmax, D, epi_scale_factor=1.0)
    c2.run()
    ref_sweep = c2.results()
    # 2. pile over N devices
    t0 = time.perf_counter()
    got = m.depth1d_pile(epis, dmin, dmax, D, epi_scale_factor=1.0)
    ok &= same("rslf_multi_depth1d_pile_f32 over %d device(s), host planes out" % n, got, ref_pile, PILE, time.perf_counter() - t0)
    if True:
        t0 = time.perf_counter()
        got = m.depth1d_pile_device_out(epis, dmin, dmax, D, out_device=0, epi_scale_factor=1.0)
        got = {k: v.cpu().numpy() for k, v in got.items()}
        ok &= same("rslf_multi_depth1d_pile_f32_dev: planes left on device 0 by peer copies", got, ref_pile, PILE, time.perf_counter() - t0)
    # 3. sharded sweep
    t0 = time.perf_counter()
    got = m.depth2d(epis, dmin, dmax, D, epi_scale_factor=1.0)
    ok &= same("rslf_multi_depth2d_run_f32: sharded sweep, per-visit boundary rows by peer copy", got, ref_sweep, SWEEP, time.perf_counter() - t0)
    # 4. fine-to-coarse
    f1 = rs.FineToCoarse(epis, dmin, dmax, D, epi_scale_factor=1.0)
    f1.run()
    r1 = f1.get_results()
    t0 = time.perf_counter()
    rN = m.fine_to_coarse(epis, dmin, dmax, D, epi_scale_factor=1.0)
    ok &= same("rslf_multi_fine_to_coarse_run_host over %d device(s)" % n, {"map": rN[0], "valid": rN[1]},
               {"map": r1[0].cpu().numpy(), "valid": r1[1].cpu().numpy()}, ("map", "valid"), time.perf_counter() - t0)
    m.close()
    return ok


def rank_main() -> int:
    """One process per GPU (started by torch.distributed.run): RCCL gather on the pile path, RCCL neighbour exchange in the sweep."""
    import torch
    import torch.distributed as dist
    from remotesensingproject_amd import depth as rs, sharding
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    backend = os.environ.get("RSLF_DIST_BACKEND", "nccl")
    if os.environ.get("RSLF_ONE_DEVICE") == "1":
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group(backend, device_id=dev) if backend == "nccl" else dist.init_process_group(backend)
    vol = field()
    V, S, U, C_ = vol.shape
    dmin, dmax, D = -1.0, 2.96875, 128
    p = rs.Depth1DParameters()
    ok = True
    # pile: own rows + halo, gather on rank 0
    sh = sharding.make_shard(V, rank, world, p.par_median_filter_size, p.par_edge_confidence_opening_size)
    ctx = rs.default_context(dev)
    comp = rs.Depth1DComputer_pile(rs.Volume.from_dense(torch.from_numpy(vol[sh.rows]).to(dev), 1.0, ctx), dmin, dmax, D, parameters=p)
    comp.run()
    planes = dict(edge_confidence=comp.m_edge_confidence_v_u, disp_confidence=comp.m_disp_confidence_v_u, depth=comp.m_best_depth_v_u,
                  depth_raw=comp.m_depth_raw_v_u, score=comp.m_score_v_u, depth_idx=comp.m_depth_idx_v_u, rbar=comp.m_rbar_v_u,
                  edge_mask=comp.m_edge_confidence_mask_v_u)
    t0 = time.perf_counter()
    out = sharding.PlaneGatherer(sh, U, C_, dev)(planes)
    torch.cuda.synchronize()
    tg = time.perf_counter() - t0
    # sweep: RCCL neighbour exchange per visit
    sd = sharding.ShardedDepth2D(rs.Volume.from_dense(torch.from_numpy(vol[sh.rows]).to(dev), 1.0, ctx), sh, dmin, dmax, D, parameters=p)
    t0 = time.perf_counter()
    sd.run()
    torch.cuda.synchronize()
    ts = time.perf_counter() - t0
    mine = torch.stack([sd.depth[:, sd.own], sd.Cd[:, sd.own]]).contiguous()
    parts = sharding.row_partition(V, world)
    equal = len({b - a for a, b in parts}) == 1
    gathered = None
    if equal:
        gathered = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
        dist.gather(mine, gathered, dst=0)
    if rank == 0:
        full = rs.Depth1DComputer_pile(rs.Volume.from_dense(torch.from_numpy(vol).to(dev), 1.0, ctx), dmin, dmax, D, parameters=p)
        full.run()
        ref = full.results()
        got = {k: v.cpu().numpy() for k, v in out.items()}
        ok &= same("one process per GPU (%s, %d ranks): PlaneGatherer == one-device pile run" % (backend, world), got, ref, PILE, tg)
        if equal:
            c2 = rs.Depth2DComputer(rs.Volume.from_dense(torch.from_numpy(vol).to(dev), 1.0, ctx), dmin, dmax, D, parameters=p)
            c2.run()
            r2 = c2.results()
            st = torch.cat(gathered, dim=2).cpu().numpy()
            ok &= same("one process per GPU: ShardedDepth2D with its %s neighbour exchange == one-device sweep" % backend,
                       {"depth": st[0], "disp_confidence": st[1]}, r2, ("depth", "disp_confidence"), ts)
    # fine-to-coarse: every level's sweep on the rank's rows, one all-gather of two planes per level
    epis = [vol[v, :, :, 0] for v in range(V)]
    t0 = time.perf_counter()
    sf = sharding.ShardedFineToCoarse(epis, dmin, dmax, D, rank, world, epi_scale_factor=1.0, ctx=ctx)
    sf.run()
    fm, fv = sf.get_results()
    torch.cuda.synchronize()
    tf = time.perf_counter() - t0
    if rank == 0:
        f1 = rs.FineToCoarse(epis, dmin, dmax, D, epi_scale_factor=1.0, ctx=ctx)
        f1.run()
        r1 = f1.get_results()
        ok &= same("one process per GPU: ShardedFineToCoarse == one-device fine-to-coarse", {"map": fm.cpu().numpy(), "valid": fv.cpu().numpy()},
                   {"map": r1[0].cpu().numpy(), "valid": r1[1].cpu().numpy()}, ("map", "valid"), tf)
    flag = torch.tensor([1 if ok else 0], device=dev if backend == "nccl" else "cpu")
    dist.broadcast(flag, 0)
    dist.barrier()
    dist.destroy_process_group()
    return 0 if int(flag.item()) else 1


def main() -> int:
    if "RANK" in os.environ and os.environ.get("RSLF_SELFTEST_RANK") == "1":
        return rank_main()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=0)
    ap.add_argument("--port", type=int, default=29611)
    args = ap.parse_args()
    import torch
    n = args.gpus or torch.cuda.device_count()      # (device_count does not initialise the GPU)
    if n < 1:
        print("no GPU visible")
        return 2
    print("multi-GPU self-test on %d device(s)%s" % (n, "" if n > 1 else " -- a REHEARSAL: one device exercises no cross-device branch"))
    # the one-process forms run in a child so that this process never initialises the GPU before starting the ranks
    rc1 = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); import tools.multi_gpu_selftest as t; sys.exit(0 if t.one_process(%d) else 1)" % (ROOT, n)]).returncode
    rc2 = 0
    ranks = max(n, 2) if n == 1 else n
    env = dict(os.environ, RSLF_SELFTEST_RANK="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if n == 1:      # rehearsal: two ranks on the one GPU over gloo
        env.update(RSLF_DIST_BACKEND="gloo", RSLF_ONE_DEVICE="1")
    rc2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
                          "--master-port", str(args.port), os.path.abspath(__file__)], env=env).returncode
    print("self-test %s" % ("PASSED" if rc1 == 0 and rc2 == 0 else "FAILED (one-process forms rc %d, one-process-per-GPU forms rc %d)" % (rc1, rc2)))
    return 0 if rc1 == 0 and rc2 == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
