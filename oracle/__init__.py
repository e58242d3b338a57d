"""ctypes loader for the CPU oracle (oracle/rslf_oracle.c).

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see rslf_oracle.h): the reference
holds no golden vectors for this path and cannot be built without OpenCV 3.x.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
package; the product (remotesensingproject_amd/, include/) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "librslf_oracle.so")
_lib = None


class OracleParams(C.Structure):
    """Mirror of oracle_params (= rslf::Depth1DParameters, core.hpp:66-142)."""

    _fields_ = [
        ("edge_score_threshold", C.c_float),
        ("raw_score_threshold", C.c_float),
        ("mean_shift_max_iter", C.c_float),
        ("edge_confidence_filter_size", C.c_int),
        ("median_filter_size", C.c_int),
        ("median_filter_epsilon", C.c_float),
        ("slope_factor", C.c_float),
        ("cut_shadows", C.c_int),
        ("shadow_level", C.c_float),
        ("kernel_bandwidth", C.c_float),
        ("interpolation", C.c_int),     # 0 linear, 1 nearest (interp.hpp:80-92), 2 nearest as built (interp.hpp:118)
        ("edge_confidence_opening_type", C.c_int),   # cv::MORPH_RECT 0 / CROSS 1 / ELLIPSE 2
        ("edge_confidence_opening_size", C.c_int),   # 1 = off
        ("use_disp_confidence_score", C.c_int),      # _USE_DISP_CONFIDENCE_SCORE (core.hpp:35), 0 = default build
        ("disp_score_threshold", C.c_float),         # 0.01
    ]


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile). Returns the .so path."""
    src = os.path.join(_HERE, "rslf_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B"], check=True, capture_output=True)
    return _SO


_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.oracle_default_params.argtypes = [C.POINTER(OracleParams)]
        L.oracle_normalize_u8.argtypes = [_u8p, _f32p, C.c_size_t]
        L.oracle_normalize_f32.argtypes = [_f32p, _f32p, C.c_size_t, C.c_float]
        L.oracle_normalize_f32.restype = C.c_float
        L.oracle_edge_confidence_pile.argtypes = [
            _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _f32p, _u8p, C.POINTER(OracleParams)]
        L.oracle_depth_epi.argtypes = [
            _f32p, C.c_int, C.c_int, C.c_int, _f32p, _f32p, C.c_int, C.c_int,
            _f32p, _u8p, _f32p, _f32p, _f32p, C.POINTER(OracleParams),
            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_selective_median.argtypes = [
            _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _u8p, C.c_float]
        L.oracle_depth_epi_pile.argtypes = [
            _f32p, C.c_int, C.c_int, C.c_int, C.c_int, _f32p, _f32p, C.c_int, C.c_int,
            _f32p, _u8p, _f32p, _f32p, _f32p, C.POINTER(OracleParams),
            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_depth1d_pile_run.argtypes = [
            _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int,
            C.POINTER(OracleParams), _f32p, _u8p, _f32p, _f32p, _f32p, _i32p, _f32p, _f32p]
        L.oracle_depth2d_run.argtypes = [
            _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.POINTER(OracleParams), C.c_float,
            _f32p, _u8p, _f32p, _f32p, _f32p, _u8p]
        L.oracle_edge_confidence_2d.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, C.c_int, _f32p, _u8p, C.POINTER(OracleParams)]
        L.oracle_depth_epi_2d.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, C.c_int, _f32p, _f32p, C.c_int, _f32p, _u8p, _f32p,
                                          _f32p, _f32p, C.POINTER(OracleParams), C.c_float, _u8p]
        L.oracle_f2c_out_dims.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.oracle_downsample_epis.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, C.c_int, _f32p]
        L.oracle_downsample_epis_u8.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, C.c_int, _f32p]
        L.oracle_f2c_tighten_bounds.argtypes = [_f32p, _u8p, C.c_int, C.c_int, C.c_int, _f32p, _f32p, C.c_int, C.c_int]
        L.oracle_f2c_fuse.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                      C.c_int, _f32p, _u8p]
        L.oracle_num_threads.restype = C.c_int
        L.oracle_set_num_threads.argtypes = [C.c_int]
        L.oracle_set_assumptions.argtypes = [C.c_int]
        _lib = L
    return _lib


def default_params() -> OracleParams:
    p = OracleParams()
    lib().oracle_default_params(C.byref(p))
    return p


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


@dataclass
class PileResult:
    """Outputs of Depth1DComputer_pile::run() (dc.hpp:131-135) + parity extras."""

    edge_confidence: np.ndarray   # C_e   [V,U] f32
    edge_mask: np.ndarray         # mask  [V,U] u8 (0/255)
    disp_confidence: np.ndarray   # C_d   [V,U] f32
    depth: np.ndarray             # best depth after the selective median [V,U] f32
    rbar: np.ndarray              # [V,U,C] f32
    depth_idx: np.ndarray         # argmax index, -1 if none [V,U] i32
    score: np.ndarray             # score[d*] [V,U] f32
    depth_raw: np.ndarray         # best depth before the median [V,U] f32


def normalize_u8(x: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.uint8)
    out = np.empty(x.shape, np.float32)
    lib().oracle_normalize_u8(x.reshape(-1), out.reshape(-1), x.size)
    return out


def normalize_f32(x: np.ndarray, scale: float = -1.0):
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty(x.shape, np.float32)
    s = lib().oracle_normalize_f32(x.reshape(-1), out.reshape(-1), x.size, scale)
    return out, float(s)


def edge_confidence_pile(vol: np.ndarray, s: int, params: OracleParams | None = None):
    """vol [V,S,U,C] f32 -> (C_e [V,U], mask [V,U])."""
    vol = np.ascontiguousarray(vol, dtype=np.float32)
    V, S, U, Cc = vol.shape
    p = params or default_params()
    Ce = np.zeros((V, U), np.float32)
    mask = np.zeros((V, U), np.uint8)
    lib().oracle_edge_confidence_pile(vol.reshape(-1), V, S, U, Cc, s, Ce.reshape(-1), mask.reshape(-1), C.byref(p))
    return Ce, mask


def selective_median(src, vol, s_hat, mask, size=5, epsilon=np.float32(0.1)):
    vol = np.ascontiguousarray(vol, dtype=np.float32)
    V, S, U, Cc = vol.shape
    src = np.ascontiguousarray(src, dtype=np.float32)
    mask = np.ascontiguousarray(mask, dtype=np.uint8)
    dst = np.zeros((V, U), np.float32)
    lib().oracle_selective_median(src.reshape(-1), dst.reshape(-1), vol.reshape(-1), V, S, U, Cc, s_hat, size,
                                  mask.reshape(-1), float(epsilon))
    return dst


def depth_epi(epi, dmin_u, dmax_u, dim_d, s_hat, Ce_u, Ce_mask_u, params=None, mask_u=None, want_K=False):
    """One EPI [S,U,C] (core.hpp:480-661). Returns dict of updated rows."""
    epi = np.ascontiguousarray(epi, dtype=np.float32)
    S, U, Cc = epi.shape
    p = params or default_params()
    Ce = np.array(Ce_u, np.float32, copy=True)
    cm = np.array(Ce_mask_u, np.uint8, copy=True)
    Cd = np.zeros(U, np.float32)
    depth = np.zeros(U, np.float32)
    rbar = np.zeros((U, Cc), np.float32)
    idx = np.full(U, -1, np.int32)
    score = np.zeros(U, np.float32)
    K = np.zeros((S, U), np.float32) if want_K else None
    m = None if mask_u is None else np.array(mask_u, np.uint8, copy=True)
    lib().oracle_depth_epi(epi.reshape(-1), S, U, Cc,
                           np.ascontiguousarray(dmin_u, np.float32), np.ascontiguousarray(dmax_u, np.float32),
                           dim_d, s_hat, Ce, cm, Cd, depth, rbar.reshape(-1), C.byref(p),
                           _ptr(m), _ptr(idx), _ptr(score), _ptr(K))
    return dict(Ce=Ce, Ce_mask=cm, Cd=Cd, depth=depth, rbar=rbar, idx=idx, score=score, K=K, mask=m)


def depth_epi_pile(vol, dmin_vu, dmax_vu, dim_d, s_hat, Ce_vu, Ce_mask_vu, params=None, mask_vu=None) -> PileResult:
    """core.hpp:772-893 on caller-supplied C_e / mask planes (copied)."""
    vol = np.ascontiguousarray(vol, dtype=np.float32)
    V, S, U, Cc = vol.shape
    p = params or default_params()
    Ce = np.array(Ce_vu, np.float32, copy=True)
    cm = np.array(Ce_mask_vu, np.uint8, copy=True)
    Cd = np.zeros((V, U), np.float32)
    depth = np.zeros((V, U), np.float32)
    rbar = np.zeros((V, U, Cc), np.float32)
    idx = np.full((V, U), -1, np.int32)
    score = np.zeros((V, U), np.float32)
    raw = np.zeros((V, U), np.float32)
    m = None if mask_vu is None else np.array(mask_vu, np.uint8, copy=True)
    lib().oracle_depth_epi_pile(vol.reshape(-1), V, S, U, Cc,
                                np.ascontiguousarray(dmin_vu, np.float32).reshape(-1),
                                np.ascontiguousarray(dmax_vu, np.float32).reshape(-1),
                                dim_d, s_hat, Ce.reshape(-1), cm.reshape(-1), Cd.reshape(-1), depth.reshape(-1),
                                rbar.reshape(-1), C.byref(p), _ptr(m), _ptr(idx), _ptr(score), _ptr(raw))
    return PileResult(Ce, cm, Cd, depth, rbar, idx, score, raw)


def depth1d_pile_run(vol, dmin, dmax, dim_d, s_hat=-1, params=None) -> PileResult:
    """Depth1DComputer_pile ctor+run() (dc.hpp:425-565) on a normalised volume."""
    vol = np.ascontiguousarray(vol, dtype=np.float32)
    V, S, U, Cc = vol.shape
    p = params or default_params()
    Ce = np.zeros((V, U), np.float32)
    cm = np.zeros((V, U), np.uint8)
    Cd = np.zeros((V, U), np.float32)
    depth = np.zeros((V, U), np.float32)
    rbar = np.zeros((V, U, Cc), np.float32)
    idx = np.full((V, U), -1, np.int32)
    score = np.zeros((V, U), np.float32)
    raw = np.zeros((V, U), np.float32)
    lib().oracle_depth1d_pile_run(vol.reshape(-1), V, S, U, Cc, dmin, dmax, dim_d, s_hat, C.byref(p),
                                  Ce.reshape(-1), cm.reshape(-1), Cd.reshape(-1), depth.reshape(-1),
                                  rbar.reshape(-1), idx.reshape(-1), score.reshape(-1), raw.reshape(-1))
    return PileResult(Ce, cm, Cd, depth, rbar, idx, score, raw)


@dataclass
class SweepResult:
    """Outputs of Depth2DComputer::run() (dc.hpp:208-215), all [S,V,U(,C)]."""

    edge_confidence: np.ndarray
    edge_mask: np.ndarray
    disp_confidence: np.ndarray
    depth: np.ndarray
    rbar: np.ndarray
    scan_mask: np.ndarray     # the running masks after the last view (not kept by the reference)


def depth2d_run(vol, dmin, dmax, dim_d, params=None, propagation_epsilon=0.1) -> SweepResult:
    """Depth2DComputer ctor+run() (dc.hpp:651-805) on a normalised volume [V,S,U,C]."""
    vol = np.ascontiguousarray(vol, dtype=np.float32)
    V, S, U, Cc = vol.shape
    p = params or default_params()
    Ce = np.zeros((S, V, U), np.float32)
    cm = np.zeros((S, V, U), np.uint8)
    Cd = np.zeros((S, V, U), np.float32)
    depth = np.zeros((S, V, U), np.float32)
    rbar = np.zeros((S, V, U, Cc), np.float32)
    sm = np.zeros((S, V, U), np.uint8)
    lib().oracle_depth2d_run(vol.reshape(-1), V, S, U, Cc, dmin, dmax, dim_d, C.byref(p), np.float32(propagation_epsilon),
                             Ce.reshape(-1), cm.reshape(-1), Cd.reshape(-1), depth.reshape(-1), rbar.reshape(-1), sm.reshape(-1))
    return SweepResult(Ce, cm, Cd, depth, rbar, sm)


def depth2d_run_planes(vol, dmin_svu, dmax_svu, dim_d, params=None, propagation_epsilon=0.1) -> SweepResult:
    """compute_2D_edge_confidence + compute_2D_depth_epi with per-pixel hypothesis ranges [S,V,U]
    (what FineToCoarse hands to every level but the finest, rslf_fine_to_coarse.hpp:171-299)."""
    vol = np.ascontiguousarray(vol, dtype=np.float32)
    V, S, U, Cc = vol.shape
    p = params or default_params()
    Ce = np.zeros((S, V, U), np.float32); cm = np.zeros((S, V, U), np.uint8)
    Cd = np.zeros((S, V, U), np.float32); depth = np.zeros((S, V, U), np.float32)
    rbar = np.zeros((S, V, U, Cc), np.float32); sm = np.zeros((S, V, U), np.uint8)
    L = lib()
    L.oracle_edge_confidence_2d(vol.reshape(-1), V, S, U, Cc, Ce.reshape(-1), cm.reshape(-1), C.byref(p))
    L.oracle_depth_epi_2d(vol.reshape(-1), V, S, U, Cc, np.ascontiguousarray(dmin_svu, np.float32).reshape(-1),
                          np.ascontiguousarray(dmax_svu, np.float32).reshape(-1), dim_d, Ce.reshape(-1), cm.reshape(-1),
                          Cd.reshape(-1), depth.reshape(-1), rbar.reshape(-1), C.byref(p), np.float32(propagation_epsilon),
                          sm.reshape(-1))
    return SweepResult(Ce, cm, Cd, depth, rbar, sm)


def downsample_epis(raw_vsuc: np.ndarray) -> np.ndarray:
    """rslf::downsample_EPIs (fine_to_coarse_core.cpp:14-60) on [V,S,U,C] float32."""
    a = np.ascontiguousarray(raw_vsuc, np.float32)
    V, S, U, Cc = a.shape
    v2, u2 = C.c_int(), C.c_int()
    lib().oracle_f2c_out_dims(V, U, C.byref(v2), C.byref(u2))
    out = np.zeros((v2.value, S, u2.value, Cc), np.float32)
    lib().oracle_downsample_epis(a.reshape(-1), V, S, U, Cc, out.reshape(-1))
    return out


def downsample_epis_u8(levels_vsuc: np.ndarray) -> np.ndarray:
    """rslf::downsample_EPIs on CV_8U Mats: uchar levels (0..255) carried in float32, uchar arithmetic."""
    a = np.ascontiguousarray(levels_vsuc, np.float32)
    V, S, U, Cc = a.shape
    v2, u2 = C.c_int(), C.c_int()
    lib().oracle_f2c_out_dims(V, U, C.byref(v2), C.byref(u2))
    out = np.zeros((v2.value, S, u2.value, Cc), np.float32)
    lib().oracle_downsample_epis_u8(a.reshape(-1), V, S, U, Cc, out.reshape(-1))
    return out


def f2c_tighten_bounds(depth_up_svu, mask_up_svu, dmin_down_svu, dmax_down_svu):
    """rslf_fine_to_coarse.hpp:171-299; returns new (dmin, dmax) for the coarser level."""
    du = np.ascontiguousarray(depth_up_svu, np.float32); mu = np.ascontiguousarray(mask_up_svu, np.uint8)
    S, Vu, Uu = du.shape
    lo = np.array(dmin_down_svu, np.float32, copy=True); hi = np.array(dmax_down_svu, np.float32, copy=True)
    _, Vd, Ud = lo.shape
    lib().oracle_f2c_tighten_bounds(du.reshape(-1), mu.reshape(-1), S, Vu, Uu, lo.reshape(-1), hi.reshape(-1), Vd, Ud)
    return lo, hi


def f2c_fuse(disp_pyr, valid_pyr):
    """rslf::fuse_disp_maps (fine_to_coarse_core.cpp:69-135) for one view: lists of [V_p,U_p] planes, finest first."""
    P = len(disp_pyr)
    d = [np.ascontiguousarray(x, np.float32) for x in disp_pyr]
    m = [np.ascontiguousarray(x, np.uint8) for x in valid_pyr]
    dp = (C.c_void_p * P)(*[x.ctypes.data for x in d]); mp = (C.c_void_p * P)(*[x.ctypes.data for x in m])
    Vp = (C.c_int * P)(*[x.shape[0] for x in d]); Up = (C.c_int * P)(*[x.shape[1] for x in d])
    out = np.zeros(d[0].shape, np.float32); ov = np.zeros(d[0].shape, np.uint8)
    lib().oracle_f2c_fuse(dp, mp, Vp, Up, P, out.reshape(-1), ov.reshape(-1))
    return out, ov


def fine_to_coarse_run(raw_vsuc, dmin, dmax, dim_d, params=None, max_pyr_depth=-1, accept_all_last_scale=True,
                       min_spatial_dim=10, propagation_epsilon=0.1, is_u8=False):
    """rslf::FineToCoarse (rslf_fine_to_coarse.hpp:103-299 + get_results :302-324) on a RAW float32 volume
    [V,S,U,C]: every level normalises by its own max (epi_scale_factor = -1 in each Depth2DComputer), levels are
    built while min(V,U) > _MIN_SPATIAL_DIM, slope_factor = U_p / U_0.
    Returns dict(levels=[SweepResult...], dims=[(V_p,U_p)...], fused_map [S,V,U], fused_valid [S,V,U])."""
    import copy
    base = params or default_params()
    raw = np.ascontiguousarray(raw_vsuc, np.float32)
    U0 = raw.shape[2]
    vols, pars = [], []
    if max_pyr_depth < 1:
        max_pyr_depth = 1 << 30
    cur = raw
    V, U = cur.shape[0], cur.shape[2]
    count = 0
    while V > min_spatial_dim and U > min_spatial_dim and count < max_pyr_depth:   # f2c.hpp:130
        count += 1
        p = copy.copy(base) if not isinstance(base, OracleParams) else OracleParams.from_buffer_copy(base)
        p.slope_factor = np.float32((0.0 + U) / U0)                                # f2c.hpp:139
        if is_u8:   # uchar EPIs: every level scales by 1/255 (dc.hpp:696-699) and the pyramid is built in uchar arithmetic
            norm = normalize_u8(cur.astype(np.uint8))
        else:
            norm, _ = normalize_f32(cur, -1.0)                                     # Depth2DComputer ctor, dc.hpp:671-705
        vols.append(norm); pars.append(p)
        cur = downsample_epis_u8(cur) if is_u8 else downsample_epis(cur)           # f2c.hpp:145-147 (the RAW EPIs go down)
        V, U = cur.shape[0], cur.shape[2]
    S = raw.shape[1]
    thr = np.float32(base.edge_score_threshold)
    levels, valids = [], []
    dmin_p = dmax_p = None
    for lvl, (vol, p) in enumerate(zip(vols, pars)):
        Vp, Up = vol.shape[0], vol.shape[2]
        if lvl == 0:
            lo = np.full((S, Vp, Up), dmin, np.float32); hi = np.full((S, Vp, Up), dmax, np.float32)
        else:
            lo = np.full((S, Vp, Up), dmin, np.float32); hi = np.full((S, Vp, Up), dmax, np.float32)
            lo, hi = f2c_tighten_bounds(levels[-1].depth, valids[-1], lo, hi)      # f2c.hpp:176-294
        r = depth2d_run_planes(vol, lo, hi, dim_d, p, propagation_epsilon)
        last = lvl == len(vols) - 1
        if last and accept_all_last_scale:
            valid = np.where(r.edge_confidence > -1, 255, 0).astype(np.uint8)      # dc.hpp:911
        else:
            valid = np.where(r.edge_confidence > thr, 255, 0).astype(np.uint8)     # dc.hpp:906
        levels.append(r); valids.append(valid)
    fused = np.zeros((S,) + levels[0].depth.shape[1:], np.float32)
    fvalid = np.zeros(fused.shape, np.uint8)
    for s in range(S):                                                             # fine_to_coarse_core.cpp:84
        fused[s], fvalid[s] = f2c_fuse([lv.depth[s] for lv in levels], [vm[s] for vm in valids])
    return dict(levels=levels, valids=valids, dims=[(v.shape[0], v.shape[2]) for v in vols], fused_map=fused, fused_valid=fvalid,
                params=pars)


def sweep_pixels_scanned(reset: bool = True) -> int:
    """Pixels the sweeps (depth2d_run, depth2d_run_planes, fine_to_coarse_run) have scanned since the last reset."""
    L = lib()
    L.oracle_sweep_pixels_scanned.restype = C.c_longlong
    L.oracle_sweep_pixels_scanned.argtypes = [C.c_int]
    return int(L.oracle_sweep_pixels_scanned(1 if reset else 0))


def num_threads() -> int:
    return int(lib().oracle_num_threads())


def usable_cpus() -> int:
    """CPUs this process may actually use: the affinity mask capped by the cgroup quota
    (a GPU box gives a container a share of a much larger host)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


ASSUME_RGB_SUM_IN_ORDER, ASSUME_MUL_SCALE_LAST, ASSUME_DIV0_IEEE, ASSUME_MAX_NAN_TAIL = 1, 2, 4, 8
# the fine-to-coarse pyramid's readings (rslf_fine_to_coarse_core.cpp:22-41, :69-135)
ASSUME_GAUSS_ROW_SYMM, ASSUME_GAUSS_COL_ORDER, ASSUME_AREA_SCALAR, ASSUME_SIZE_FLOOR, ASSUME_RESIZE_FLOAT = 16, 32, 64, 128, 256


def set_assumptions(flags: int) -> None:
    """What-if switches of tools/blast_radius.py (0 = the assumptions of record; tests never change it)."""
    lib().oracle_set_assumptions(int(flags))


def set_num_threads(n: int) -> None:
    lib().oracle_set_num_threads(int(n))
