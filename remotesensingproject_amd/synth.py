"""Synthetic light fields for the bench and the parity tests.

A scene is a stack of horizontal bands, each with one true disparity delta.
View s of scanline v is the scanline's texture resampled at
``u + reach - (s_hat - s) * delta`` so that the EPI line through ``(s_hat, u)``
has slope ``delta`` -- the line the reference samples at
``I = u + (s_hat - s) * d`` (rslf_depth_computation_core.hpp:542-552).
"""
from __future__ import annotations

import numpy as np

# name -> (U, V, S, C, D, dmin, dmax, seed); BASELINE.json configs c2, c3/c4, c5
CONFIGS = {
    "c1": dict(U=540, V=960, S=9, C=1, D=64, dmin=-2.0, dmax=5.875, seed=None),
    "c2": dict(U=512, V=512, S=33, C=1, D=128, dmin=-1.0, dmax=2.96875, seed=20260001),
    "c3": dict(U=1920, V=1080, S=101, C=1, D=256, dmin=-2.0, dmax=5.96875, seed=20260003),
    "c5": dict(U=4096, V=2160, S=201, C=3, D=512, dmin=-2.0, dmax=5.984375, seed=20260005),
    # the shapes of the reference's own published runs (report/rs_report.tex:427-437), synthetic fields: context lines of
    # the fine-to-coarse / 2-D sweep rows, not BASELINE configs
    "skysat_lr": dict(U=960, V=540, S=100, C=1, D=120, dmin=-1.0, dmax=4.0, seed=20260099),     # SkysatLR18, report:430
    "mansion_lr": dict(U=1146, V=720, S=100, C=3, D=120, dmin=0.0, dmax=4.0, seed=20260099),    # MansionLR, report:406,427
    # MansionLR's frame with 151 views: a rung of the on-chip kernel's ladder other than c5's (k2_scan_chip<false, 84, 0, true>)
    "mansion_151": dict(U=1146, V=720, S=151, C=3, D=120, dmin=0.0, dmax=4.0, seed=20260099),
}


def band_disparities(V: int, dmin: float, dmax: float, band: int = 32) -> np.ndarray:
    """Per-row true disparity: integer values in [dmin, dmax], cycling per band."""
    ints = np.arange(int(np.ceil(dmin)), int(np.floor(dmax)) + 1)
    return ints[(np.arange(V) // band) % len(ints)].astype(np.float32)


def make_lightfield(U: int, V: int, S: int, C: int = 1, *, seed: int, deltas=None,
                    dmin: float = -2.0, dmax: float = 5.0, band: int = 32,
                    s_hat: int | None = None, lo: float = 0.2, hi: float = 1.0,
                    rows: slice | None = None) -> tuple[np.ndarray, np.ndarray]:
    """Return (vol [V',S,U,C] float32 in [lo,hi), delta [V'] float32).

    ``rows`` selects a block of scanlines of the V-row scene (used by the
    multi-GPU shards: every rank draws the same textures and keeps its rows).
    Fractional deltas are resampled with a float32 2-tap lerp.
    """
    if s_hat is None:
        s_hat = S // 2
    if deltas is None:
        deltas = band_disparities(V, dmin, dmax, band)
    deltas = np.asarray(deltas, np.float32)
    max_ds = max(s_hat, S - 1 - s_hat)
    reach = int(np.ceil(float(np.abs(deltas).max()) * max_ds)) + 1 if V else 1
    rng = np.random.default_rng(seed)
    width = U + 2 * reach
    sel = range(V)[rows] if rows is not None else range(V)
    vol = np.empty((len(sel), S, U, C), np.float32)
    out_i = 0
    first = sel[0] if len(sel) else 0
    for v in range(V):
        # textures are drawn row by row so any row block sees the same values
        T = rng.uniform(lo, hi, size=(C, width)).astype(np.float32)
        if v < first:
            continue
        if out_i >= len(sel):
            break
        dl = float(deltas[v])
        for s in range(S):
            x0 = reach - (s_hat - s) * dl
            if float(x0).is_integer():
                a = int(x0)
                vol[out_i, s] = T[:, a:a + U].T
            else:
                x = (np.arange(U, dtype=np.float32) + np.float32(x0)).astype(np.float32)
                i0 = np.floor(x).astype(np.int64)
                t = (x - i0.astype(np.float32)).astype(np.float32)
                vol[out_i, s] = ((np.float32(1) - t) * T[:, i0] + t * T[:, i0 + 1]).T
        out_i += 1
    return vol, deltas[list(sel)]


def make_config(name: str, rows: slice | None = None):
    """Volume + sweep arguments for a BASELINE.json synthetic config."""
    c = CONFIGS[name]
    if c["seed"] is None:
        raise ValueError("config %s is not synthetic" % name)
    vol, delta = make_lightfield(c["U"], c["V"], c["S"], c["C"], seed=c["seed"],
                                 dmin=c["dmin"], dmax=c["dmax"], rows=rows)
    return vol, delta, c
