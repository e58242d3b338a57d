#!/usr/bin/env python3
"""How much would each unverifiable OpenCV 3.x reading move?  (DESIGN.md "Oracle": parity unpinned.)

The oracle restates OpenCV primitives from knowledge of 3.4; the reference holds no vectors and
cannot be built here, so nothing pins those readings.  This script re-runs the oracle with each reading replaced by
its plausible alternative (oracle.set_assumptions) on the golden cases, a slice of c2, an RGB field and two shapes
whose hypothesis count leaves a SIMD tail, and counts what changes: arg-max indices, best scores, edge-mask pixels,
filtered depths.  CPU only (the oracle); the product is not involved.

    python tools/blast_radius.py [--out profiles/r02_blast_radius.md]
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r02_blast_radius.md"))
    args = ap.parse_args()
    oracle.build()
    lines = ["# Blast radius of the unverifiable OpenCV 3.x readings (round 2)", "",
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
    open(args.out, "w").write("\n".join(lines) + "\n")
    print("wrote", args.out)


if __name__ == "__main__":
    main()
