"""Shared helpers for the parity tests."""
from __future__ import annotations

import numpy as np

# BASELINE.md §7 / north_star: argmax indices bit-exact, confidences within 1e-5
TOL = 1e-5


def assert_pile_parity(got: dict, ref, *, exact_float: bool = True, label: str = ""):
    """got: dict of numpy planes from the HIP path; ref: oracle.PileResult.

    Integer / mask planes must match bit for bit.  Float planes are asserted
    within TOL (the bar north_star states) and, when exact_float, additionally
    required to be bit-identical -- the HIP kernels perform the same IEEE
    operations in the same order as the oracle, so any difference is a bug.
    """
    def eq(name, a, b):
        assert a.shape == b.shape, (label, name, a.shape, b.shape)
        af, bf = a.reshape(-1), b.reshape(-1)
        differ = af != bf
        if af.dtype.kind == "f":   # a NaN in the same place on both sides is agreement (NaN inputs propagate into C_e)
            differ &= ~(np.isnan(af) & np.isnan(bf))
        bad = np.flatnonzero(differ)
        assert bad.size == 0, "%s %s: %d mismatches, first at %s: got %r want %r" % (
            label, name, bad.size, np.unravel_index(bad[0], a.shape), a.reshape(-1)[bad[0]], b.reshape(-1)[bad[0]])

    eq("edge_mask", got["edge_mask"], ref.edge_mask)
    eq("depth_idx", got["depth_idx"], ref.depth_idx)
    for name, a, b in (("edge_confidence", got["edge_confidence"], ref.edge_confidence),
                       ("score", got["score"], ref.score),
                       ("disp_confidence", got["disp_confidence"], ref.disp_confidence),
                       ("rbar", got["rbar"], ref.rbar),
                       ("depth_raw", got["depth_raw"], ref.depth_raw),
                       ("depth", got["depth"], ref.depth)):
        assert a.shape == b.shape, (label, name, a.shape, b.shape)
        with np.errstate(invalid="ignore"):
            d = np.abs(a.astype(np.float64) - b.astype(np.float64))
        d = np.where(np.isnan(a) & np.isnan(b), 0.0, d)
        d = np.where(np.isinf(a) & (a == b), 0.0, d)
        err = d.max() if a.size else 0.0
        assert err <= TOL, "%s %s: max abs err %g > %g" % (label, name, err, TOL)
        if exact_float and name != "disp_confidence":
            eq(name, a, b)
