#!/usr/bin/env python3
"""How much would each unverifiable OpenCV 3.x reading move?  (DESIGN.md "Oracle": parity unpinned.)

The oracle restates OpenCV primitives from knowledge of 3.4; the reference holds no vectors and
cannot be built here, so nothing pins those readings.  This script re-runs the oracle with each reading replaced by
its plausible alternative (oracle.set_assumptions) on the golden cases, a slice of c2, an RGB field and two shapes
whose hypothesis count leaves a SIMD tail, and counts what changes: arg-max indices, best scores, edge-mask pixels,
filtered depths.  CPU only (the oracle); the product is not involved.

    python tools/blast_radius.py [--out profiles/r03_blast_radius.md] [--rows scan|f2c|all]

Round 3: the same for the fine-to-coarse row (rslf_fine_to_coarse_core.cpp:22-41, :69-135) -- Gaussian row / column tap
order, the area mean's pairing, cvRound level sizes, the upscaling's coordinate precision: the whole oracle pipeline
(pyramid, per-level sweeps with tightened bounds, fusion) is re-run with one reading replaced, on the shapes
tests/test_gpu_f2c.py drives, and the fused map's pixels that change are counted.
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from remotesensingproject_amd.synth import make_lightfield  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
ALTS = [
    ("(i) RGB channel sum (q0+q1)+q2", oracle.ASSUME_RGB_SUM_IN_ORDER, "cv::reduce over the 3 columns in index order instead of reduceC_'s two accumulators (q0+q2)+q1 (kernels.cpp:48)"),
    ("(ii) multiply as scale*(a*b)", oracle.ASSUME_MUL_SCALE_LAST, "cv::multiply(a, b, dst, scale) associating scale*(a*b) instead of (scale*a)*b (kernels.cpp:21,43)"),
    ("(iii) x/0 = IEEE inf/NaN", oracle.ASSUME_DIV0_IEEE, "cv::divide by zero as in OpenCV 4.x instead of 3.x's 0 (core.hpp:620, core.cpp:42,50)"),
    ("(iv) NaN survives cv::max tail", oracle.ASSUME_MAX_NAN_TAIL, "cv::max(x, 0) keeping a NaN in the scalar tail (last n%8 elements) instead of sending every NaN to 0 (core.hpp:580,609,622; kernels.cpp:25,53)"),
]


def cases():
    meta = lambda n: json.loads(str(np.load(os.path.join(GOLD, n + ".npz"))["meta"]))
    crop = np.load(os.path.join(GOLD, "c1_crop_000tif_rows400_424.npy"))
    m = meta("c1crop")
    norm, _ = oracle.normalize_f32(crop, m["tif_max"])
    yield "golden c1crop (24 rows of data/000.tif, 9 views, 64 hyp)", np.ascontiguousarray(np.repeat(norm[:, None, :, None], 9, axis=1)), m
    for n, label in (("rand1", "golden rand1 (96x12, 33 views, 48 hyp)"), ("rgb", "golden rgb (64x8, 17 views RGB, 24 hyp)"),
                     ("edge", "golden edge (dark / flat rows, dmin == dmax)")):
        yield label, np.load(os.path.join(GOLD, n + "_input.npy")), meta(n)
    vol, _ = make_lightfield(512, 6, 33, 1, seed=20260001, dmin=-1.0, dmax=2.96875)
    yield "c2 slice (512 px x 6 rows, 33 views, 128 hyp)", vol, dict(dmin=-1.0, dmax=2.96875, D=128, s_hat=-1)
    rng = np.random.default_rng(20261010)
    yield "random RGB (120 px x 6 rows, 21 views, 37 hyp: SIMD tail)", rng.uniform(0, 1, (6, 21, 120, 3)).astype(np.float32), dict(dmin=-1.5, dmax=2.5, D=37, s_hat=-1)
    yield "random 1-ch (150 px x 6 rows, 13 views, 20 hyp: SIMD tail)", rng.uniform(0, 1, (6, 13, 150, 1)).astype(np.float32), dict(dmin=-1.5, dmax=2.0, D=20, s_hat=-1)
    v3, _ = make_lightfield(160, 6, 25, 3, seed=7, dmin=-1.0, dmax=1.0, band=2)
    yield "structured RGB (160 px x 6 rows, 25 views, 33 hyp)", v3, dict(dmin=-1.0, dmax=1.0, D=33, s_hat=-1)


F2C_ALTS = [
    ("(v) Gaussian ROW filter in the symmetric form", oracle.ASSUME_GAUSS_ROW_SYMM,
     "cv::GaussianBlur's row pass as k[c]*x[c] + sum_j k[c+j]*(x[c+j] + x[c-j]) (what its column pass does) instead of the seven taps accumulated left to right (fine_to_coarse_core.cpp:33)"),
    ("(vi) Gaussian COLUMN filter tap by tap", oracle.ASSUME_GAUSS_COL_ORDER,
     "the column pass accumulating its seven taps top to bottom instead of the symmetric form (fine_to_coarse_core.cpp:33)"),
    ("(vii) area mean as ((S00+S01)+S10)+S11", oracle.ASSUME_AREA_SCALAR,
     "cv::resize(0.5, 0.5)'s 2x2 mean in resizeAreaFast's scalar order -- what the last W2 % 4 columns of a SIMD build take -- applied to EVERY pixel (an upper bound) instead of the SIMD form (S00+S10)+(S01+S11) (fine_to_coarse_core.cpp:37)"),
    ("(viii) level sizes floor(n/2)", oracle.ASSUME_SIZE_FLOOR,
     "Size(cols*0.5, rows*0.5) truncated instead of cvRound (ties to even): differs only where a side is odd (fine_to_coarse_core.cpp:37)"),
    ("(ix) upscaling coordinates in float", oracle.ASSUME_RESIZE_FLOAT,
     "cv::resize(INTER_LINEAR)'s source coordinates (dx+0.5)*scale-0.5 evaluated in float instead of double-then-cast (fine_to_coarse_core.cpp:104)"),
]


def f2c_cases():
    """The shapes of tests/test_gpu_f2c.py (44 x 64 px, 5 views) and a larger, odd-sized field whose levels have odd sides."""
    for C_, u8 in ((1, False), (3, False), (3, True), (1, True)):
        vol, _ = make_lightfield(64, 44, 5, C_, seed=2, dmin=-1, dmax=1, band=8)
        raw = np.round(vol * 255.0).astype(np.float32) if u8 else (vol * 200 + 3).astype(np.float32)
        yield "test_gpu_f2c case: 64 x 44 px, 5 views, %d ch, %s, 9 hyp" % (C_, "uchar" if u8 else "float"), raw, dict(dmin=-1.0, dmax=1.0, D=9, u8=u8)
    vol, _ = make_lightfield(203, 91, 7, 1, seed=11, dmin=-1, dmax=2, band=8)
    yield "203 x 91 px (odd sides at every level), 7 views, float, 16 hyp", (vol * 180 + 5).astype(np.float32), dict(dmin=-1.0, dmax=2.0, D=16, u8=False)
    rng = np.random.default_rng(5)
    yield "noise 150 x 70 px, 5 views, float, 12 hyp", rng.uniform(0, 255, (70, 5, 150, 1)).astype(np.float32), dict(dmin=-1.0, dmax=1.0, D=12, u8=False)


def f2c_section(lines):
    lines += ["", "# Blast radius of the fine-to-coarse row's readings (round 3)", "",
              "`python tools/blast_radius.py --rows f2c` — CPU oracle only: `FineToCoarse` constructor + `run()` + `get_results()`",
              "(pyramid, per-level sweeps with tightened bounds, fusion) with ONE reading replaced.  `pyramid` = values of the first halved",
              "volume that differ (bitwise: the reading's direct effect, one ulp each); `levels` = pyramid levels whose size",
              "changes; `lvl px` = pixels of the per-level disparity planes that differ (bitwise, summed over the levels of equal size);",
              "`fused` = pixels of the fused map that differ (bitwise), `valid` = pixels of the fused validity mask that differ,",
              "`max |d|` = largest change of a fused disparity.  uchar fields blur and halve in integer arithmetic (exact: tap order",
              "and pairing cannot matter there), so (v)-(vii) can only move float fields.", ""]
    for label, flag, what in F2C_ALTS:
        lines += ["## %s" % label, "", what + ".", "", "| case | fused px | pyramid | levels | lvl px | fused | valid | max \\|d\\| |", "|---|---|---|---|---|---|---|---|"]
        tot = moved = 0
        for name, raw, m in f2c_cases():
            down = oracle.downsample_epis_u8 if m["u8"] else oracle.downsample_epis
            oracle.set_assumptions(0)
            a = oracle.fine_to_coarse_run(raw, m["dmin"], m["dmax"], m["D"], is_u8=m["u8"])
            pa = down(raw)
            oracle.set_assumptions(flag)
            b = oracle.fine_to_coarse_run(raw, m["dmin"], m["dmax"], m["D"], is_u8=m["u8"])
            pb = down(raw)
            oracle.set_assumptions(0)
            pyr = "%.1f %%" % (100.0 * float((pa.view(np.uint32) != pb.view(np.uint32)).mean())) if pa.shape == pb.shape else "resized"
            n = a["fused_map"].size
            dl = sum(1 for x, y in zip(a["dims"], b["dims"]) if x != y) + abs(len(a["dims"]) - len(b["dims"]))
            lvl = sum(int((la.depth.view(np.uint32) != lb.depth.view(np.uint32)).sum())
                      for la, lb, x, y in zip(a["levels"], b["levels"], a["dims"], b["dims"]) if x == y)
            same_shape = a["fused_map"].shape == b["fused_map"].shape
            df = int((a["fused_map"].view(np.uint32) != b["fused_map"].view(np.uint32)).sum()) if same_shape else n
            dv = int((a["fused_valid"] != b["fused_valid"]).sum()) if same_shape else n
            with np.errstate(invalid="ignore"):
                mx = float(np.nanmax(np.abs(a["fused_map"].astype(np.float64) - b["fused_map"].astype(np.float64)))) if same_shape else float("nan")
            lines.append("| %s | %d | %s | %d | %d | %d | %d | %.3g |" % (name, n, pyr, dl, lvl, df, dv, mx))
            tot += n
            moved += df
        lines += ["", "Total: %d of %d fused disparities change (%.3f %%)." % (moved, tot, 100.0 * moved / max(tot, 1)), ""]
        print(label, "->", moved, "of", tot, "fused pixels", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03_blast_radius.md"))
    ap.add_argument("--rows", default="all", choices=["scan", "f2c", "all"])
    args = ap.parse_args()
    oracle.build()
    if args.rows == "f2c":
        lines = []
        f2c_section(lines)
        open(args.out, "w").write("\n".join(lines).lstrip("\n") + "\n")
        print("wrote", args.out)
        return
    lines = ["# Blast radius of the unverifiable OpenCV 3.x readings: the scan (round 2 table, re-run)", "",
             "`python tools/blast_radius.py` — CPU oracle only.  Each row re-runs `Depth1DComputer_pile::run` with ONE reading of",
             "SURVEY.md App. B replaced by its alternative and counts what differs from the reading of record (the one the HIP",
             "path is bit-identical to).  `idx` = arg-max indices among scanned pixels, `score` = best scores (bitwise),",
             "`mask` = edge-mask pixels after the scan, `depth` = filtered depths (bitwise), `max |d score|` = largest score change.", ""]
    for label, flag, what in ALTS:
        lines += ["## %s" % label, "", what + ".", "", "| case | scanned px | idx | score | mask | depth | max \\|d score\\| |", "|---|---|---|---|---|---|---|"]
        tot_px = tot_idx = 0
        for name, vol, m in cases():
            oracle.set_assumptions(0)
            a = oracle.depth1d_pile_run(vol, m["dmin"], m["dmax"], m["D"], m.get("s_hat", -1))
            oracle.set_assumptions(flag)
            b = oracle.depth1d_pile_run(vol, m["dmin"], m["dmax"], m["D"], m.get("s_hat", -1))
            oracle.set_assumptions(0)
            scanned = a.depth_idx >= 0
            n = int(scanned.sum())
            d_idx = int((a.depth_idx != b.depth_idx).sum())
            d_sc = int((a.score.view(np.uint32) != b.score.view(np.uint32)).sum())
            d_mask = int((a.edge_mask != b.edge_mask).sum())
            d_depth = int((a.depth.view(np.uint32) != b.depth.view(np.uint32)).sum())
            with np.errstate(invalid="ignore"):
                mx = float(np.nanmax(np.abs(a.score.astype(np.float64) - b.score.astype(np.float64)))) if n else 0.0
            lines.append("| %s | %d | %d | %d | %d | %d | %.3g |" % (name, n, d_idx, d_sc, d_mask, d_depth, mx))
            tot_px += n
            tot_idx += d_idx
        lines += ["", "Total: %d of %d arg-max indices move (%.4f %%)." % (tot_idx, tot_px, 100.0 * tot_idx / max(tot_px, 1)), ""]
        print(label, "->", tot_idx, "of", tot_px, "indices", flush=True)
    if args.rows == "all":
        f2c_section(lines)
    open(args.out, "w").write("\n".join(lines) + "\n")
    print("wrote", args.out)


if __name__ == "__main__":
    main()
