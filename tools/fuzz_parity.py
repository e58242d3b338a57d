"""Randomised parity campaign: HIP path vs the CPU oracle on many small random configurations.

    python tools/fuzz_parity.py [cases] [seed]

Each case draws a shape (V, S, U, C, D), a hypothesis range, a slope factor, an input kind (noise, structured
scene, mixture, with dark / flat rows), parameters (thresholds, iteration count, bandwidth, median size), optional
per-pixel [dmin, dmax] planes and caller mask, a forced kernel variant (register / stream / generic) and a launch
shape (row tiles or packed tiles x hypothesis groups), runs both sides and requires bit-identical planes
(C_d within 1e-5).  Developer tool: the committed tests hold a fixed subset of such cases; this one is for
sweeping many seeds on the GPU box.  Prints one line per failing case and a summary; exit code 1 on any failure.
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import oracle
from remotesensingproject_amd import depth as rs
from remotesensingproject_amd.synth import make_lightfield
from tests.util import assert_pile_parity

S_CHOICES_1 = [1, 2, 3, 5, 8, 9, 15, 16, 17, 24, 31, 33, 40, 47, 56, 64, 65, 72, 90, 101, 104, 105, 120, 129, 150, 192, 201, 209, 256, 257, 300]
S_CHOICES_3 = [1, 2, 3, 5, 8, 9, 16, 17, 24, 31, 40, 48, 49, 56, 57, 64, 70, 88, 100, 104, 105, 120,
               199, 200, 201, 201, 202, 203, 209, 220, 230, 256]   # 201 .. 220 views, dense launch: the on-chip kernel (k2_chip.hpp)


ONLY_CHIP = os.environ.get("FUZZ_ONLY_CHIP") == "1"   # RGB, 123..220 views, dense launches: what k2_scan_chip takes (every rung, padded and exact)


def draw_case(rng):
    C = int(rng.choice([1, 1, 3]))
    S = int(rng.choice(S_CHOICES_1 if C == 1 else S_CHOICES_3))
    if ONLY_CHIP:
        C, S = 3, int(rng.choice([201, 201, 202, 203, 204, 205, 206, 211, 217, 220]) if rng.uniform() < 0.4 else rng.integers(123, 202))
    budget = 600000 if C == 1 else 250000          # oracle work ~ V*U*D*S
    U = int(rng.choice([1, 2, 7, 33, 63, 64, 65, 100, 129, 200, 260, 513, 700]))
    V = int(rng.integers(1, 13))
    D = int(rng.choice([2, 3, 5, 8, 13, 16, 17, 31, 32, 40, 64, 70, 120]))
    while V * U * D * S > budget * 64 and D > 2:
        D = max(2, D // 2)
    while V * U * D * S > budget * 64 and U > 8:
        U = max(8, U // 2)
    lo = float(rng.choice([-2.0, -1.0, -0.5, 0.0, 0.25]))
    hi = lo + float(rng.choice([0.0, 0.5, 1.0, 2.5, 4.0]))
    kind = str(rng.choice(["noise", "struct", "mixed", "dark"]))
    if ONLY_CHIP:   # no per-pixel planes, no forced variant, no packed lists, non-negative: the launch shape the kernel takes
        return dict(C=C, S=S, U=U, V=V, D=D, dmin=lo, dmax=hi, kind=str(rng.choice(["noise", "struct", "mixed", "dark"])),
                    slope=float(rng.choice([1.0, 1.0, 0.5, 0.25, 1.5])), s_hat=int(rng.integers(0, S)) if rng.uniform() < 0.3 else -1,
                    planes=False, mask=bool(rng.uniform() < 0.3), force="", packed=0, groups=int(rng.choice([1, 1, 2, 4, 8])),
                    iters=float(rng.choice([10.0, 10.0, 1.0, 3.5, 12.0])), h=float(rng.choice([0.2, 0.2, 0.1, 0.5])),
                    thr=float(rng.choice([0.02, 0.02, 0.0, 0.2])), raw_thr=float(rng.choice([0.0, 0.0, 0.3])),
                    median=int(rng.choice([5, 5, 3, 7, 1, 0, 2, 4, 6, 9, 11, 13, 15, 21, 33])), shadows=bool(rng.uniform() < 0.8), negative=False, interp=0,
                    opening=(2, 1), form=str(rng.choice(["dense", "dense", "epis_f32", "epis_u8"])))
    return dict(C=C, S=S, U=U, V=V, D=D, dmin=lo, dmax=hi, kind=kind,
                slope=float(rng.choice([1.0, 1.0, 0.5, 0.25, 1.5])),
                s_hat=int(rng.integers(0, S)) if rng.uniform() < 0.3 else -1,
                planes=bool(rng.uniform() < 0.3), mask=bool(rng.uniform() < 0.3),
                force=str(rng.choice(["", "", "stream", "generic"])),
                packed=int(rng.uniform() < 0.4), groups=int(rng.choice([1, 1, 2, 4, 8])),
                iters=float(rng.choice([10.0, 10.0, 1.0, 3.5, 12.0])), h=float(rng.choice([0.2, 0.2, 0.1, 0.5])),
                thr=float(rng.choice([0.02, 0.02, 0.0, 0.2])), raw_thr=float(rng.choice([0.0, 0.0, 0.3])),
                median=int(rng.choice([5, 5, 3, 7, 1, 0, 2, 4, 6, 9, 11, 13, 15, 21, 33])), shadows=bool(rng.uniform() < 0.8),
                negative=bool(rng.uniform() < 0.1), interp=int(rng.choice([0, 0, 0, 0, 1, 2])),
                opening=(int(rng.choice([0, 1, 2])), int(rng.choice([2, 3, 4, 5, 7]))) if rng.uniform() < 0.15 else (2, 1),
                # how the light field reaches the device: dense device tensor, host EPIs (f32, scale 1), host EPIs
                # normalised by their max (dc.hpp:442-460), uint8 EPIs (x/255, dc.hpp:470), uint8 image stack (io.cpp:194-227)
                form=str(rng.choice(["dense", "dense", "epis_f32", "epis_max", "epis_u8", "images_u8"])),
                # packed launches of a register / streaming kernel: lanes own hypotheses (k2_scan_reg_px, k2_scan_stream_px) -- automatic / never / whenever it can run
                px=int(rng.choice([-1, -1, 0, 1])),
                # the streaming kernel's shared-tap tiles: automatic (by the tail's length) / never / always
                share=int(rng.choice([1, 1, 0, 2])),
                # packed launches of stream-class volumes: rows with >= N pixels as row tiles of the list (1: the default 64; 0: off)
                row_split=int(rng.choice([1, 1, 0, 4, 20])))


def make_volume(c, rng):
    V, S, U, C = c["V"], c["S"], c["U"], c["C"]
    if c["kind"] == "struct" and U >= 8:
        vol = make_lightfield(U, V, S, C, seed=int(rng.integers(1 << 30)), deltas=rng.integers(-1, 2, size=V).astype(np.float32))[0]
    else:
        vol = rng.uniform(0.0, 1.0, size=(V, S, U, C)).astype(np.float32)
    if c["kind"] == "mixed" and V > 1:
        vol[V // 2:] = rng.uniform(0.0, 1.0, size=vol[V // 2:].shape).astype(np.float32) ** 3
    if c["kind"] == "dark":
        vol[0] *= np.float32(0.04)                      # below the shadow level
        if V > 1:
            vol[-1] = np.float32(0.5)                   # flat: C_e = 0
    if c["negative"]:
        vol = vol - np.float32(0.3)                     # negative radiances: generic kernel only
        if rng.uniform() < 0.3 and vol.size > 4:        # and now and then a NaN (generic kernel as well)
            flat = vol.reshape(-1)
            flat[rng.integers(0, flat.size)] = np.nan
    return np.ascontiguousarray(vol, np.float32)


def run_case(i, c, rng):
    # per-context hooks (rslf_ctx_set_debug)
    rs.default_context(0).set_debug(force_scan=c["force"], force_packed=c["packed"], force_groups=c["groups"], px=c.get("px", -1),
                                    stream_share=c.get("share", 1), row_split=c.get("row_split", 1))
    vol = make_volume(c, rng)
    V, S, U, C = vol.shape
    po = oracle.default_params()
    pr = rs.Depth1DParameters()
    for name, val in (("slope_factor", c["slope"]), ("mean_shift_max_iter", c["iters"]), ("kernel_bandwidth", c["h"]),
                      ("edge_score_threshold", c["thr"]), ("raw_score_threshold", c["raw_thr"]),
                      ("median_filter_size", c["median"]), ("cut_shadows", int(c["shadows"]))):
        setattr(po, name, val)
        setattr(pr, "par_" + name, val)
    po.interpolation = c["interp"]
    pr.par_interpolation_class = c["interp"]
    po.edge_confidence_opening_type, po.edge_confidence_opening_size = c["opening"]
    pr.par_edge_confidence_opening_type, pr.par_edge_confidence_opening_size = c["opening"]
    s_hat = c["s_hat"]
    if not (c["planes"] or c["mask"]):
        form = c["form"] if not c["negative"] else "dense"
        sq = (lambda a: np.ascontiguousarray(a[..., 0])) if C == 1 else np.ascontiguousarray   # [S,U,1] -> [S,U]
        if form == "dense":
            src, scale = torch.from_numpy(vol).cuda(), 1.0
        elif form == "epis_f32":
            src, scale = [sq(vol[v]) for v in range(V)], 1.0
        elif form == "epis_max":
            raw = (vol * np.float32(rng.uniform(2.0, 300.0))).astype(np.float32)
            src, scale = [sq(raw[v]) for v in range(V)], -1.0
            vol, _ = oracle.normalize_f32(raw, -1.0)
        else:
            raw = (vol.clip(0.0, 1.0) * 255.0).astype(np.uint8)
            vol = oracle.normalize_u8(raw)
            if form == "epis_u8":
                src, scale = [sq(raw[v]) for v in range(V)], 1.0
            else:
                src, scale = rs.Volume.from_images([sq(raw[:, s_]) for s_ in range(S)]), 1.0
        ref = oracle.depth1d_pile_run(vol, c["dmin"], c["dmax"], c["D"], s_hat, params=po)
        if isinstance(src, torch.Tensor):
            src = rs.Volume.from_dense(src, 1.0)
        comp = rs.Depth1DComputer_pile(src, c["dmin"], c["dmax"], c["D"], s_hat, scale, pr)
        comp.run()
        got = comp.results()
        assert_pile_parity(got, ref, label="case%d" % i)
        return comp.stats.scan_kernel, int((ref.depth_idx >= 0).sum())
    # the compute_1D_depth_epi_pile form: caller planes for C_e / mask, per-pixel ranges and a scan mask
    sh = s_hat if s_hat >= 0 else S // 2
    Ce, cm = oracle.edge_confidence_pile(vol, sh, params=po)
    if c["planes"]:
        dmin = rng.uniform(c["dmin"] - 0.5, c["dmin"] + 0.5, size=(V, U)).astype(np.float32)
        dmax = (dmin + rng.uniform(0.0, max(c["dmax"] - c["dmin"], 0.25), size=(V, U))).astype(np.float32)
    else:
        dmin = np.full((V, U), c["dmin"], np.float32)
        dmax = np.full((V, U), c["dmax"], np.float32)
    mask = None
    if c["mask"]:
        mask = (rng.uniform(size=(V, U)) < rng.choice([0.03, 0.3, 0.9])).astype(np.uint8) * 255
    ref = oracle.depth_epi_pile(vol, dmin, dmax, c["D"], sh, Ce, cm, params=po, mask_vu=mask)
    v = rs.Volume.from_dense(vol)
    dev = "cuda"
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a).copy()).to(dev)
    tCe, tcm = t(Ce), t(cm)
    tmask = t(mask) if mask is not None else None
    tCd = torch.zeros((V, U), device=dev); tdepth = torch.zeros((V, U), device=dev); trbar = torch.zeros((V, U, C), device=dev)
    tidx = torch.empty((V, U), dtype=torch.int32, device=dev); tsc = torch.empty((V, U), device=dev); traw = torch.zeros((V, U), device=dev)
    rs.compute_1D_depth_epi_pile(v, t(dmin) if c["planes"] else c["dmin"], t(dmax) if c["planes"] else c["dmax"], c["D"], sh,
                                 tCe, tcm, tCd, tdepth, trbar, pr, tmask, idx_v_u=tidx, score_v_u=tsc, depth_raw_v_u=traw)
    torch.cuda.synchronize()
    got = dict(edge_confidence=tCe.cpu().numpy(), edge_mask=tcm.cpu().numpy(), disp_confidence=tCd.cpu().numpy(),
               depth=tdepth.cpu().numpy(), rbar=trbar.cpu().numpy(), depth_idx=tidx.cpu().numpy(), score=tsc.cpu().numpy(),
               depth_raw=traw.cpu().numpy())
    assert_pile_parity(got, ref, label="case%d" % i)
    return -1, int((ref.depth_idx >= 0).sum())


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    oracle.set_num_threads(min(oracle.usable_cpus(), 16))
    only = int(os.environ.get("FUZZ_ONLY", "-1"))     # replay one case: every case has its own generator
    bad, t0, kernels, pixels = 0, time.time(), {}, 0
    for i in range(cases):
        if only >= 0 and i != only:
            continue
        rng = np.random.default_rng([seed, i])
        c = draw_case(rng)
        try:
            k, npx = run_case(i, c, rng)
            kernels[k] = kernels.get(k, 0) + 1
            pixels += npx
        except AssertionError as e:
            bad += 1
            print("FAIL case %d %s\n     %s" % (i, c, str(e)[:300]), flush=True)
        except Exception as e:  # noqa: BLE001
            bad += 1
            print("ERROR case %d %s\n     %r" % (i, c, e), flush=True)
        if (i + 1) % 50 == 0:
            print("... %d cases, %d failures, %.0f s" % (i + 1, bad, time.time() - t0), flush=True)
    print("fuzz_parity: %d cases (seed %d), %d failures, %.0f s; %d pixels with a disparity compared; kernel variants of the "
          "Depth1DComputer_pile cases (0 generic, 1 register, 2 stream, 3 on-chip, 4 register with hypotheses in the lanes; -1 = plane form) %s" % (
              cases, seed, bad, time.time() - t0, pixels, dict(sorted(kernels.items()))))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
