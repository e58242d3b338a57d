"""Independent numpy restatement of the hot path, written matrix-pass by
matrix-pass the way the reference's OpenCV code runs (S x D temporaries per
pixel), to cross-check oracle/rslf_oracle.c, which restructures the loops.

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see rslf_oracle.h).  Small cases
only: it is a Python loop over pixels.

All arrays are float32 and every numpy ufunc call on them is one IEEE binary32
operation per element; sums over s use an explicit sequential loop (np.sum is
pairwise and would re-associate).

Reference lines followed are given per function; paths are relative to
/root/reference/RSLightFields/.
"""
from __future__ import annotations

import numpy as np

F = np.float32
SQRT3 = 1.73205080757  # types.cpp:84


def default_params() -> dict:
    """include/rslf_depth_computation_core.hpp:16-31, 74-99."""
    return dict(
        edge_score_threshold=F(0.02),
        raw_score_threshold=F(0.0),
        mean_shift_max_iter=F(10),
        edge_confidence_filter_size=9,
        median_filter_size=5,
        median_filter_epsilon=F(0.1),
        slope_factor=F(1.0),
        cut_shadows=True,
        shadow_level=F(0.05 * SQRT3),
        kernel_bandwidth=F(0.2),
        interpolation=0,             # core.hpp:76: Interpolation1DLinear; 1 / 2 = nearest (see rslf_oracle.h)
        edge_confidence_opening_type=2,   # cv::MORPH_ELLIPSE, core.hpp:28
        edge_confidence_opening_size=1,   # core.hpp:29: 1 = no opening
        use_disp_confidence_score=False,  # core.hpp:35 (_USE_DISP_CONFIDENCE_SCORE, commented out in the reference)
        disp_score_threshold=F(0.01),     # core.hpp:22
    )


def _norm(x: np.ndarray) -> np.ndarray:
    """norm<float> / norm<Vec3f> over the last axis (src/rslf_types.cpp:80-91)."""
    if x.shape[-1] == 1:
        return (np.abs(x[..., 0]).astype(np.float64) * SQRT3).astype(F)
    return np.sqrt((x.astype(np.float64) ** 2).sum(axis=-1)).astype(F)


def _reflect101(i: np.ndarray, n: int) -> np.ndarray:
    i = i.copy()
    if n == 1:
        return np.zeros_like(i)
    while True:
        lo, hi = i < 0, i >= n
        if not (lo.any() or hi.any()):
            return i
        i[lo] = -i[lo]
        i[hi] = 2 * n - 2 - i[hi]


def edge_confidence_row(row: np.ndarray, p: dict):
    """core.hpp:426-478. row [U,C] f32 -> (C_e [U], mask [U])."""
    U, C = row.shape
    fs = p["edge_confidence_filter_size"]
    ctr = (fs - 1) // 2
    Ce = np.zeros(U, F)
    u = np.arange(U)
    for j in range(fs):
        if j == ctr:
            continue
        q = _reflect101(u + j - ctr, U)
        for c in range(C):
            t = row[:, c] - row[q, c]          # filter2D, +1 centre / -1 at j
            Ce = Ce + t * t                    # core.cpp:6-23
    if p["cut_shadows"]:
        Ce[_norm(row) < p["shadow_level"]] = F(0)
    mask = np.where(Ce > p["edge_score_threshold"], 255, 0).astype(np.uint8)
    return Ce, mask


def structuring_element(shape: int, k: int) -> np.ndarray:
    """cv::getStructuringElement(shape, Size(k, k)), default anchor (OpenCV 3.x morph.cpp): [k,k] bool."""
    r = c = k // 2
    el = np.zeros((k, k), bool)
    for i in range(k):
        if shape == 0 or (shape == 1 and i == k // 2):
            el[i, :] = True
        elif shape == 1:
            el[i, k // 2] = True
        else:
            dy = i - r
            if abs(dy) <= r:
                inv_r2 = 1.0 / (r * r) if r else 0.0
                dx = int(np.rint(c * np.sqrt((r * r - dy * dy) * inv_r2)))     # cvRound
                el[i, max(c - dx, 0):min(c + dx + 1, k)] = True
    return el


def morph_open(mask: np.ndarray, shape: int, k: int) -> np.ndarray:
    """cv::morphologyEx(MORPH_OPEN): erosion then dilation over the structuring element, anchor (k/2, k/2);
    outside the image counts as 255 for the erosion and 0 for the dilation (morphologyDefaultBorderValue).
    Written with shifted planes, unlike the C oracle's per-pixel loops."""
    el = structuring_element(shape, k)
    a = k // 2
    V, U = mask.shape

    def one(src, fill, reduce):
        pad = np.full((V + 2 * k, U + 2 * k), fill, np.uint8)
        pad[k:k + V, k:k + U] = src
        acc = np.full((V, U), fill, np.uint8)
        for i in range(k):
            for j in range(k):
                if el[i, j]:
                    y0, x0 = k + i - a, k + j - a
                    acc = reduce(acc, pad[y0:y0 + V, x0:x0 + U])
        return acc

    return one(one(mask, 255, np.minimum), 0, np.maximum)


def _edge_confidence_plane(vol, s, p):
    """compute_1D_edge_confidence_pile (core.hpp:728-770): every row, then the optional opening."""
    V, S, U, C = vol.shape
    Ce = np.zeros((V, U), F); cm = np.zeros((V, U), np.uint8)
    for v in range(V):
        Ce[v], cm[v] = edge_confidence_row(vol[v, s], p)
    if int(p.get("edge_confidence_opening_size", 1)) > 1:      # core.hpp:759-768
        cm = morph_open(cm, int(p.get("edge_confidence_opening_type", 2)), int(p["edge_confidence_opening_size"]))
    return Ce, cm


def _kernel(delta: np.ndarray, inv_h2: F, k1: F) -> np.ndarray:
    """BandwidthKernel::evaluate_mat (src/rslf_kernels.cpp:16-26, 39-54).
    delta [S,D,C] -> K [S,D]; NaN -> 0."""
    with np.errstate(invalid="ignore"):
        if delta.shape[-1] == 1:
            q = (k1 * delta[..., 0]) * delta[..., 0]
        else:
            qc = (inv_h2 * delta) * delta
            q = (qc[..., 0] + qc[..., 2]) + qc[..., 1]   # OpenCV 3.x reduceC_ order
        o = F(1.0) - q
        return np.where(o > 0, o, F(0)).astype(F)


def scan_pixel(epi: np.ndarray, u: int, dmin: F, dmax: F, D: int, s_hat: int, p: dict):
    """core.hpp:527-625 for one pixel. epi [S,U,C]. Returns (Dv, score, rbar, K)."""
    S, U, C = epi.shape
    h = F(p["kernel_bandwidth"])
    inv_h2 = F(1.0 / np.float64(h * h))        # kernels.hpp:43
    k1 = F(3.0) * inv_h2                       # kernels.cpp:21

    Sv = (s_hat - np.arange(S)).astype(F)      # core.hpp:542
    d = np.arange(D).astype(F)
    Dv = F(dmin) + (d * (F(dmax) - F(dmin))) / F(D - 1)   # core.hpp:548
    I = Sv[:, None] * Dv[None, :]              # core.hpp:550 (gemm, K = 1)
    I = I * F(p["slope_factor"])               # core.hpp:551
    I = I + F(u)                               # core.hpp:552

    rows = np.arange(S)[:, None]
    mode = int(p.get("interpolation", 0))
    if mode == 0:
        i0 = np.floor(I).astype(np.int64)          # interp.hpp:179-181
        i1 = np.ceil(I).astype(np.int64)
        t = I - i0.astype(F)
        valid = ~((i0 < 0) | (i1 > U - 1))         # interp.hpp:182
        i0c, i1c = np.clip(i0, 0, U - 1), np.clip(i1, 0, U - 1)
        R = (F(1) - t)[..., None] * epi[rows, i0c] + t[..., None] * epi[rows, i1c]   # interp.hpp:184
    else:
        if mode == 2:                              # interp.hpp:118: the float matrix read through an int pointer
            r = np.ascontiguousarray(I, F).view(np.int32).astype(np.int64)
        else:                                      # interp.hpp:121: std::round, halves away from zero
            with np.errstate(invalid="ignore", over="ignore"):
                r = (np.sign(I) * np.floor(np.abs(I.astype(np.float64)) + 0.5)).astype(np.int64)
            r = np.where(np.abs(I) < 2.0e9, r, -1)
        valid = (r > -1) & (r < U)                 # interp.hpp:122
        R = epi[rows, np.clip(r, 0, U - 1)]        # interp.hpp:124
    R = np.where(valid[..., None], R, F(np.nan)).astype(F)   # interp.hpp:129 / :189
    card = np.zeros(D, F)
    for s in range(S):
        card = card + valid[s].astype(F)       # interp.hpp:185

    rbar = R[s_hat].copy()                     # core.hpp:577  [D,C]
    with np.errstate(invalid="ignore"):
        R0 = np.where(R > 0, R, F(0)).astype(F)  # core.hpp:580

    n_iter = 0
    while F(n_iter) < p["mean_shift_max_iter"]:   # core.hpp:584 (float bound)
        n_iter += 1
    K = None
    B = None
    for _ in range(n_iter):
        with np.errstate(invalid="ignore"):
            delta = R - rbar[None]             # core.hpp:591
        K = _kernel(delta, inv_h2, k1)         # core.hpp:595
        P = R0 * K[..., None]                  # core.hpp:599
        A = P[0].copy()                        # core.hpp:602: sequential rows
        B = K[0].copy()                        # core.hpp:603
        for s in range(1, S):
            A = A + P[s]
            B = B + K[s]
        with np.errstate(divide="ignore", invalid="ignore"):
            q = np.where(B[:, None] != 0, A / B[:, None], F(0)).astype(F)   # core.hpp:606 (3.x: /0 -> 0)
        rbar = np.where(q > 0, q, F(0)).astype(F)                            # core.hpp:609
    with np.errstate(divide="ignore", invalid="ignore"):
        score = np.where(card != 0, B / card, F(0)).astype(F)               # core.hpp:616-620
    score = np.where(score > 0, score, F(0)).astype(F)                      # core.hpp:622
    return Dv, score, rbar, K


def depth_epi(epi, dmin_u, dmax_u, D, s_hat, Ce_u, Ce_mask_u, p, mask_u=None):
    """core.hpp:480-661 for one EPI [S,U,C]."""
    S, U, C = epi.shape
    Ce = Ce_u.astype(F).copy()
    cm = Ce_mask_u.astype(np.uint8).copy()
    scan_mask = cm if mask_u is None else (cm & mask_u)      # core.hpp:510-513
    scan_mask = scan_mask.copy()
    Cd = np.zeros(U, F)
    depth = np.zeros(U, F)
    rbar_o = np.zeros((U, C), F)
    idx = np.full(U, -1, np.int32)
    score_o = np.zeros(U, F)
    for u in np.flatnonzero(scan_mask):                      # core.hpp:516, 527
        Dv, score, rbar, _ = scan_pixel(epi, int(u), dmin_u[u], dmax_u[u], D, s_hat, p)
        best = int(np.argmax(score))                         # first maximum (core.hpp:634)
        mx = np.float64(score[best])
        if mx > np.float64(p["raw_score_threshold"]):       # core.hpp:636
            depth[u] = Dv[best]
            mean = score.astype(np.float64).sum() / D
            Cd[u] = F(np.float64(Ce[u]) * abs(mx - mean))    # core.hpp:641
            rbar_o[u] = rbar[best]
            idx[u] = best
            score_o[u] = score[best]
        else:
            Ce[u] = 0                                        # core.hpp:655-656
            cm[u] = 0
    return dict(Ce=Ce, Ce_mask=cm, Cd=Cd, depth=depth, rbar=rbar_o, idx=idx, score=score_o)


def selective_median(src, vol, s_hat, mask, size, eps):
    """core.hpp:663-718. vol [V,S,U,C]."""
    V, S, U, C = vol.shape
    w = int((size - 1) / 2)                                  # core.hpp:686: C++ division truncates (size 0 -> 0)
    dst = np.zeros((V, U), F)
    ref = vol[:, s_hat]                                      # [V,U,C]
    for v in range(V):
        for u in range(U):
            if not mask[v, u]:
                continue
            k0, k1 = max(0, v - w), min(V, v + w + 1)
            l0, l1 = max(0, u - w), min(U, u + w + 1)
            diff = ref[v, u][None, None] - ref[k0:k1, l0:l1]
            ok = (mask[k0:k1, l0:l1] != 0) & (_norm(diff) < F(eps))
            vals = np.sort(src[k0:k1, l0:l1][ok])
            # nth_element at n/2; an empty candidate set (NaN centre radiance) is undefined in the reference
            # (core.hpp:713-714 reads buffer[0] of a cleared vector): defined as 0 here, in the C oracle and in K3
            dst[v, u] = vals[len(vals) // 2] if len(vals) else F(0)
    return dst


def depth1d_pile_run(vol, dmin, dmax, D, s_hat=-1, p=None):
    """Depth1DComputer_pile ctor + run() (dc.hpp:425-565) on a normalised volume."""
    p = p or default_params()
    V, S, U, C = vol.shape
    if s_hat < 0 or s_hat > S - 1:
        s_hat = int(np.floor((0.0 + S) / 2))
    out = dict(Ce=np.zeros((V, U), F), Ce_mask=np.zeros((V, U), np.uint8), Cd=np.zeros((V, U), F),
               depth_raw=np.zeros((V, U), F), rbar=np.zeros((V, U, C), F),
               idx=np.full((V, U), -1, np.int32), score=np.zeros((V, U), F))
    dmin_u = np.full(U, dmin, F)
    dmax_u = np.full(U, dmax, F)
    Ce_all, cm_all = _edge_confidence_plane(vol, s_hat, p)
    for v in range(V):
        r = depth_epi(vol[v], dmin_u, dmax_u, D, s_hat, Ce_all[v], cm_all[v], p)
        out["Ce"][v], out["Ce_mask"][v], out["Cd"][v] = r["Ce"], r["Ce_mask"], r["Cd"]
        out["depth_raw"][v], out["rbar"][v], out["idx"][v], out["score"][v] = r["depth"], r["rbar"], r["idx"], r["score"]
    out["depth"] = selective_median(out["depth_raw"], vol, s_hat, out["Ce_mask"],
                                    p["median_filter_size"], p["median_filter_epsilon"])
    return out


def depth2d_run(vol, dmin, dmax, D, p=None, propagation_epsilon=F(0.1)):
    """Depth2DComputer ctor + run() (dc.hpp:651-805): compute_2D_edge_confidence (core.hpp:901-931)
    and compute_2D_depth_epi (core.hpp:933-1133, default build: propagation gated by the edge mask)."""
    p = p or default_params()
    V, S, U, C = vol.shape
    Ce = np.zeros((S, V, U), F); cm = np.zeros((S, V, U), np.uint8)
    for s in range(S):
        Ce[s], cm[s] = _edge_confidence_plane(vol, s, p)
    Cd = np.zeros((S, V, U), F); depth = np.zeros((S, V, U), F); rbar = np.zeros((S, V, U, C), F)
    mask = cm.copy()                                            # core.hpp:958-965
    s_mid = int(np.floor(S / 2.0))
    order = [s_mid]
    for off in range(1, S - s_mid):                             # core.hpp:981-990
        order.append(s_mid + off)
        if s_mid - off > -1:
            order.append(s_mid - off)
    dmin_u = np.full(U, dmin, F); dmax_u = np.full(U, dmax, F)
    for s_hat in order:
        for v in range(V):                                      # core.hpp:1012-1028 (scan, per EPI)
            r = depth_epi(vol[v], dmin_u, dmax_u, D, s_hat, Ce[s_hat, v], cm[s_hat, v], p, mask_u=mask[s_hat, v])
            mask[s_hat, v] = cm[s_hat, v] & mask[s_hat, v]      # core.hpp:511 (AND in place, before rejections)
            scanned = mask[s_hat, v] > 0
            Ce[s_hat, v], cm[s_hat, v] = r["Ce"], r["Ce_mask"]
            sel = r["idx"] >= 0
            Cd[s_hat, v][sel] = r["Cd"][sel]
            depth[s_hat, v][sel] = r["depth"][sel]              # raw depths land in the stored plane
            rbar[s_hat, v][sel] = r["rbar"][sel]
        filtered = selective_median(depth[s_hat], vol, s_hat, cm[s_hat], p["median_filter_size"], p["median_filter_epsilon"])
        for v in range(V):                                      # core.hpp:1088-1129
            for u in range(U):
                if p.get("use_disp_confidence_score", False):       # core.hpp:1097-1098
                    if not Cd[s_hat, v, u] > F(p["disp_score_threshold"]):
                        continue
                elif not cm[s_hat, v, u]:                            # core.hpp:1102
                    continue
                cur = filtered[v, u]
                for s in range(S):
                    off = F(F(cur * F(s_hat - s)) * F(p["slope_factor"]))
                    ri = u + int(np.sign(off) * np.floor(np.abs(off) + F(0.5)))   # std::round: half away from zero
                    if -1 < ri < U and mask[s, v, ri]:
                        if _norm((vol[v, s, ri] - rbar[s_hat, v, u])[None])[0] < F(propagation_epsilon):
                            depth[s, v, ri] = cur
                            mask[s, v, ri] = 0
                            Cd[s, v, ri] = Cd[s_hat, v, u]
    return dict(Ce=Ce, Ce_mask=cm, Cd=Cd, depth=depth, rbar=rbar, scan_mask=mask)


# ---- "next" row: fine-to-coarse primitives (restated with array ops) -----------------------------

_GK = np.array([0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125], F)   # small_gaussian_tab[3]


def _reflect(i, n):
    """cv::BORDER_REFLECT index map."""
    i = np.array(i, copy=True)
    if n == 1:
        return np.zeros_like(i)
    while True:
        lo, hi = i < 0, i >= n
        if not (lo.any() or hi.any()):
            return i
        i[lo] = -i[lo] - 1
        i[hi] = 2 * n - 1 - i[hi]


def gaussian7_reflect(img):
    """cv::GaussianBlur(7x7, sigma 0, BORDER_REFLECT) on [R,W,C] float32 (fine_to_coarse_core.cpp:38)."""
    R, W, C = img.shape
    xs = np.arange(W)
    tmp = _GK[0] * img[:, _reflect(xs - 3, W)]
    for j in range(1, 7):
        tmp = tmp + _GK[j] * img[:, _reflect(xs + j - 3, W)]
    ys = np.arange(R)
    out = _GK[3] * tmp
    for j in range(1, 4):
        out = out + _GK[3 + j] * (tmp[_reflect(ys + j, R)] + tmp[_reflect(ys - j, R)])
    return out.astype(F)


def halve_area(img):
    """cv::resize(0.5, 0.5, INTER_LINEAR) = OpenCV's 2x2 area-fast path (fine_to_coarse_core.cpp:42)."""
    R, W, C = img.shape
    R2, W2 = int(np.rint(R * 0.5)), int(np.rint(W * 0.5))      # cvRound: ties to even
    out = np.zeros((R2, W2, C), F)
    Rf, Wf = R // 2, W // 2
    a = img[0:2 * Rf:2, 0:2 * Wf:2] + img[1:2 * Rf:2, 0:2 * Wf:2]
    b = img[0:2 * Rf:2, 1:2 * Wf:2] + img[1:2 * Rf:2, 1:2 * Wf:2]
    out[:Rf, :Wf] = (a + b) * F(0.25)
    for y in range(R2):
        for x in range(W2):
            if y < Rf and x < Wf:
                continue
            vals = [img[2 * y + sy, 2 * x + sx] for sy in range(2) for sx in range(2) if 2 * y + sy < R and 2 * x + sx < W]
            acc = np.zeros(C, F)
            for vv in vals:
                acc = acc + vv
            out[y, x] = acc / F(len(vals)) if vals else 0
    return out


def downsample_epis(raw):
    """rslf::downsample_EPIs: [V,S,U,C] -> [V2,S,U2,C]."""
    V, S, U, C = raw.shape
    outs = [halve_area(gaussian7_reflect(np.ascontiguousarray(raw[:, s]))) for s in range(S)]
    return np.ascontiguousarray(np.stack(outs, axis=1))


def downsample_epis_u8(levels):
    """rslf::downsample_EPIs on CV_8U Mats, independently of the C restatement: uchar levels in, uchar levels out
    (as float32).  GaussianBlur 7x7 (sigma 0) on 8U = the exact integer convolution with {8,28,56,72,56,28,8}/256 per
    axis, BORDER_REFLECT, rounded half up once; resize x0.5 = INTER_AREA's 2x2 mean, (sum + 2) >> 2, and at an odd
    border cvRound(sum / count)."""
    V, S, U, C = levels.shape
    k = np.array([8, 28, 56, 72, 56, 28, 8], np.int64)
    V2, U2 = int(np.rint(V * 0.5)), int(np.rint(U * 0.5))
    out = np.zeros((V2, S, U2, C), F)

    def refl(i, n):
        i = np.asarray(i)
        i = np.where(i < 0, -i - 1, i)
        return np.where(i >= n, 2 * n - 1 - i, i)

    for s in range(S):
        img = levels[:, s].astype(np.int64)                                     # [V, U, C]
        rows = sum(k[j] * img[:, refl(np.arange(U) + j - 3, U)] for j in range(7))
        full = sum(k[j] * rows[refl(np.arange(V) + j - 3, V)] for j in range(7))
        blur = (full + 32768) >> 16
        for y in range(V2):
            for x in range(U2):
                blk = blur[2 * y:2 * y + 2, 2 * x:2 * x + 2]                    # the pixels that exist
                if blk.shape[0] == 2 and blk.shape[1] == 2:
                    out[y, s, x] = (blk.sum(axis=(0, 1)) + 2) >> 2
                elif blk.size:
                    out[y, s, x] = np.rint(blk.sum(axis=(0, 1)).astype(F) / F(blk.shape[0] * blk.shape[1]))
    return out


def tighten_bounds(depth_up, mask_up, dmin_down, dmax_down):
    """rslf_fine_to_coarse.hpp:202-294."""
    S, Vu, Uu = depth_up.shape
    _, Vd, Ud = dmin_down.shape
    lo, hi = dmin_down.copy(), dmax_down.copy()
    for s in range(S):
        for v in range(Vd):
            for u in range(Ud):
                cands = []
                rows = [min(2 * v, Vu - 1)]
                if rows[0] + 1 < Vu:
                    rows.append(rows[0] + 1)
                uu = min(2 * u, Uu - 1)
                for r in rows:
                    left = [c for c in range(uu - 1, 0, -1) if mask_up[s, r, c] > 0]
                    right = [c for c in range(uu + 1, Uu) if mask_up[s, r, c] > 0]
                    if left and right:
                        cands += [depth_up[s, r, left[0]], depth_up[s, r, right[0]]]
                if len(cands) > 1:
                    lo[s, v, u], hi[s, v, u] = min(cands), max(cands)
    return lo, hi


def resize_linear(src, R2, W2):
    """cv::resize(..., INTER_LINEAR) of a float plane (upscaling, fine_to_coarse_core.cpp:104)."""
    R, W = src.shape
    sx_scale, sy_scale = 1.0 / (W2 / W), 1.0 / (R2 / R)
    fx = ((np.arange(W2) + 0.5) * sx_scale - 0.5).astype(F)
    sx = np.floor(fx).astype(np.int64)
    fx = (fx - sx.astype(F)).astype(F)
    neg = sx < 0
    fx[neg], sx[neg] = 0, 0
    edge = sx + 1 >= W
    xmax = int(np.flatnonzero(edge)[0]) if edge.any() else W2
    far = sx >= W - 1
    fx[far], sx[far] = 0, W - 1
    fy = ((np.arange(R2) + 0.5) * sy_scale - 0.5).astype(F)
    sy = np.floor(fy).astype(np.int64)
    fy = (fy - sy.astype(F)).astype(F)
    y0, y1 = np.clip(sy, 0, R - 1), np.clip(sy + 1, 0, R - 1)
    a0, a1 = (F(1) - fx), fx
    sx1 = np.minimum(sx + 1, W - 1)
    def hrow(rows):
        full = rows[:, sx] * a0 + rows[:, sx1] * a1
        flat = rows[:, sx] * F(1)
        return np.where(np.arange(W2)[None] < xmax, full, flat).astype(F)
    r0, r1 = hrow(src[y0]), hrow(src[y1])
    return (r0 * (F(1) - fy)[:, None] + r1 * fy[:, None]).astype(F)


def resize_nearest(src, R2, W2):
    R, W = src.shape
    sy = np.minimum(np.floor(np.arange(R2) * (1.0 / (R2 / R))).astype(np.int64), R - 1)
    sx = np.minimum(np.floor(np.arange(W2) * (1.0 / (W2 / W))).astype(np.int64), W - 1)
    return src[sy][:, sx]


def median3(src):
    R, W = src.shape
    p = np.pad(src, 1, mode="edge")
    stack = np.stack([p[dy:dy + R, dx:dx + W] for dy in range(3) for dx in range(3)], axis=0)
    return np.sort(stack, axis=0)[4].astype(F)


def fuse(disp_pyr, valid_pyr):
    """rslf::fuse_disp_maps for one view (fine_to_coarse_core.cpp:93-131)."""
    P = len(disp_pyr)
    md, mm = disp_pyr[P - 1].copy(), valid_pyr[P - 1].copy()
    for p in range(P - 1, 0, -1):
        R, W = disp_pyr[p - 1].shape
        up = resize_linear(md, R, W)
        upm = resize_nearest(mm, R, W)
        inval = valid_pyr[p - 1] == 0
        md = np.where(inval, F(0) + up, disp_pyr[p - 1]).astype(F)
        mm = valid_pyr[p - 1] | upm
    return median3(md), mm
