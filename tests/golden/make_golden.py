#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/.

    python tests/golden/make_golden.py          (run in the build container)

What is pinned and by what
--------------------------
The reference (14chanwa/remotesensingProject) holds NO golden vectors or
assertions for this path and cannot be built here (OpenCV 3.x is absent), so
these fixtures are outputs of OUR oracle (oracle/rslf_oracle.c), cross-checked
at generation time against the independent numpy restatement
(oracle/oracle_np.py).  They pin the oracle against regressions and give the
GPU tests fixed expected values; they do not pin the oracle to the reference
("parity unpinned").

Inputs: seeded synthetic volumes, and one data file of the reference:
/root/reference/data/000.tif (960x540 float32, the input of BASELINE.json
configs[0]) -- decoded, as data, whole (c1_000tif_960x540_f32.npz) and as a
24-row crop -- normalised by the max of the WHOLE image as
Depth1DComputer_pile's constructor would (dc.hpp:442-477).
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from oracle import oracle_np as onp  # noqa: E402
from remotesensingproject_amd.synth import make_lightfield  # noqa: E402

TIF = "/root/reference/data/000.tif"
OUT_KEYS = ("edge_confidence", "edge_mask", "disp_confidence", "depth", "rbar", "depth_idx", "score", "depth_raw")


def run_and_save(name: str, vol: np.ndarray, dmin: float, dmax: float, D: int, s_hat: int = -1, extra=None, np_rows=2):
    r = oracle.depth1d_pile_run(vol, dmin, dmax, D, s_hat)
    # cross-check a couple of scanlines against the numpy restatement before committing anything
    sub = vol[:max(np_rows, 1)]
    rr = oracle.depth1d_pile_run(sub, dmin, dmax, D, s_hat)
    nn = onp.depth1d_pile_run(sub, np.float32(dmin), np.float32(dmax), D, s_hat) if np_rows else None
    for k_o, k_n in (("edge_confidence", "Ce"), ("edge_mask", "Ce_mask"), ("depth_idx", "idx"), ("score", "score"),
                     ("depth_raw", "depth_raw"), ("rbar", "rbar"), ("disp_confidence", "Cd"), ("depth", "depth")):
        assert nn is None or np.array_equal(getattr(rr, k_o), nn[k_n]), (name, k_o)
    out = {k: getattr(r, k) for k in OUT_KEYS}
    meta = dict(dmin=dmin, dmax=dmax, D=D, s_hat=s_hat)
    meta.update(extra or {})
    np.savez_compressed(os.path.join(HERE, name + ".npz"), meta=json.dumps(meta), **out)
    print("%-10s V,S,U,C=%s scanned=%d rejected=%d" % (name, vol.shape, int((r.depth_idx >= 0).sum()),
                                                       int(((r.edge_mask == 0) & (r.edge_confidence == 0)).sum())))
    return r


def main():
    anchors = {}
    # ---- c1: crop of the reference's 000.tif, 9 identical views --------------
    if os.path.exists(TIF):
        from PIL import Image
        img = np.array(Image.open(TIF), dtype=np.float32)
        assert img.shape == (960, 540), img.shape
        full_max = float(img.max())
        crop = np.ascontiguousarray(img[400:424])
        np.save(os.path.join(HERE, "c1_crop_000tif_rows400_424.npy"), crop)
        # SURVEY.md 8c anchor: edge mask of the whole image after the constructor's normalisation
        norm, scale = oracle.normalize_f32(img[:, None, :, None], -1.0)
        Ce, m = oracle.edge_confidence_pile(np.ascontiguousarray(norm), 0)
        anchors = dict(tif_shape=list(img.shape), tif_max=full_max, scale_used=scale,
                       mask_count=int((m > 0).sum()), shadow_cut_count=int((Ce == 0).sum()))
        print("c1 anchor:", anchors)
    else:
        crop = np.load(os.path.join(HERE, "c1_crop_000tif_rows400_424.npy"))
        anchors = json.load(open(os.path.join(HERE, "c1_anchor.json")))
        full_max = anchors["tif_max"]
    json.dump(anchors, open(os.path.join(HERE, "c1_anchor.json"), "w"), indent=1)
    norm_crop, _ = oracle.normalize_f32(crop, full_max)
    vol = np.ascontiguousarray(np.repeat(norm_crop[:, None, :, None], 9, axis=1))   # [24, 9, 540, 1]
    run_and_save("c1crop", vol, -2.0, 5.875, 64, extra=dict(tif_max=full_max, views=9), np_rows=1)
    # ---- c1 on the WHOLE frame (BASELINE.json configs[0]): the decoded data file travels as a fixture (data, 2 MB), and
    # the oracle's planes of the 9-view / 64-hypothesis run with it -----------------------------------------------------
    if os.path.exists(TIF):
        np.savez_compressed(os.path.join(HERE, "c1_000tif_960x540_f32.npz"), image=img)
    else:
        img = np.load(os.path.join(HERE, "c1_000tif_960x540_f32.npz"))["image"]
    norm_full, _ = oracle.normalize_f32(np.ascontiguousarray(img), full_max)
    vol_full = np.ascontiguousarray(np.repeat(norm_full[:, None, :, None], 9, axis=1))   # [960, 9, 540, 1]
    r = run_and_save("c1full", vol_full, -2.0, 5.875, 64, extra=dict(tif_max=full_max, views=9), np_rows=1)
    anchors["c1_scanned"] = int((r.depth_idx >= 0).sum())
    json.dump(anchors, open(os.path.join(HERE, "c1_anchor.json"), "w"), indent=1)

    # ---- full-SHAPE noise fields: BASELINE.json configs[2] / [4] at their view counts, hypothesis
    # counts and grids, a few scanlines each; the inputs are regenerated from their seeds (tests/test_oracle.py:
    # noise_case), only the oracle's planes are committed ------------------------------------------------------------
    from tests.test_oracle import NOISE_CASES, noise_volume
    for name, (V, S, U, C, D, dmin, dmax, seed) in NOISE_CASES.items():
        run_and_save(name, noise_volume(name), dmin, dmax, D, extra=dict(seed=seed, shape=[V, S, U, C]), np_rows=0)

    # ---- seeded synthetic cases ----------------------------------------------
    rng = np.random.default_rng(20261001)
    vs, _ = make_lightfield(96, 12, 33, 1, seed=20261002, dmin=-1.0, dmax=2.0, band=3)
    vs[6:] = rng.uniform(0.0, 1.0, size=vs[6:].shape).astype(np.float32)          # half structure, half noise
    np.save(os.path.join(HERE, "rand1_input.npy"), vs.astype(np.float32))
    run_and_save("rand1", vs, -1.0, 2.9375, 48)

    v3, _ = make_lightfield(64, 8, 17, 3, seed=20261003, dmin=-1.0, dmax=1.0, band=2)
    v3[4:] = rng.uniform(0.0, 1.0, size=v3[4:].shape).astype(np.float32)
    np.save(os.path.join(HERE, "rgb_input.npy"), v3.astype(np.float32))
    run_and_save("rgb", v3, -1.0, 1.0, 24)

    ve = rng.uniform(0.2, 1.0, size=(6, 9, 64, 1)).astype(np.float32)
    ve[1] = 0.01            # dark scanline: shadow cut
    ve[2] = 0.5             # flat scanline: C_e = 0
    ve[3, :, 10:20] = 0.0   # dark stripe
    np.save(os.path.join(HERE, "edge_input.npy"), ve)
    run_and_save("edge", ve, 1.0, 1.0, 8)   # dmin == dmax: every hypothesis ties, index 0 wins


if __name__ == "__main__":
    main()
