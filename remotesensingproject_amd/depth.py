"""Host-side mirror of the reference's interface for the 1-D pile path.

Names, argument meaning and defaults follow RSLightFields
(/root/reference/RSLightFields/include/):
  Depth1DParameters                 rslf_depth_computation_core.hpp:66-142
  compute_1D_edge_confidence_pile   rslf_depth_computation_core.hpp:279-287
  compute_1D_depth_epi_pile         rslf_depth_computation_core.hpp:293-310
  selective_median_filter           rslf_depth_computation_core.hpp:366-375
  Depth1DComputer_pile              rslf_depth_computation.hpp:93-143, :425-565
The reference's `Mat`s become torch CUDA tensors (device memory + streams are
all torch is used for); the arithmetic runs in librslf_hip.so through the C-ABI
of include/rslf_hip.h.  No CPU path exists here.
"""
from __future__ import annotations

import atexit
import ctypes as C
import sys
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import RslfParams, RslfStats, RslfVolumeDesc, check


class Interpolation1DLinear:
    """include/rslf_interpolation.hpp:59-67, :155-193 (RSLF_INTERP_LINEAR)."""
    mode = 0


class Interpolation1DNearestNeighbour:
    """include/rslf_interpolation.hpp:46-54.  `as_built=False`: the class as its scalar interpolate() states it
    (:80-92, RSLF_INTERP_NEAREST).  `as_built=True`: what interpolate_mat() executes in the reference, the float
    index matrix read through an int pointer (:118, RSLF_INTERP_NEAREST_AS_BUILT)."""

    def __init__(self, as_built: bool = False):
        self.mode = 2 if as_built else 1


@dataclass
class Depth1DParameters:
    """rslf::Depth1DParameters<T> with the reference's member names and defaults."""

    par_edge_score_threshold: float = 0.02
    par_line_score_threshold: float = 0.02
    par_disp_score_threshold: float = 0.01
    par_raw_score_threshold: float = 0.0
    par_mean_shift_max_iter: float = 10.0
    par_edge_confidence_filter_size: int = 9
    par_edge_confidence_opening_type: int = 2
    par_edge_confidence_opening_size: int = 1
    par_median_filter_size: int = 5
    par_median_filter_epsilon: float = 0.1
    par_propagation_epsilon: float = 0.1
    par_slope_factor: float = 1.0
    par_cut_shadows: bool = True
    par_shadow_level: float = 0.05 * 1.73205080757
    par_kernel_bandwidth: float = 0.2   # BandwidthKernel(_BANDWIDTH_KERNEL_PARAMETER), core.hpp:78
    # par_interpolation_class (core.hpp:76-77, :108): an Interpolation1DLinear / Interpolation1DNearestNeighbour
    # instance, or the RSLF_INTERP_* integer itself
    par_interpolation_class: object = 0
    # the reference's commented-out build switch _USE_DISP_CONFIDENCE_SCORE (core.hpp:35): propagation gated by
    # C_d > par_disp_score_threshold instead of the edge mask
    par_use_disp_confidence_score: bool = False

    @staticmethod
    def get_default() -> "Depth1DParameters":
        return Depth1DParameters()

    def to_c(self) -> RslfParams:
        p = RslfParams()
        for f, _ in RslfParams._fields_:
            if f == "interpolation":
                p.interpolation = int(getattr(self.par_interpolation_class, "mode", self.par_interpolation_class))
                continue
            v = getattr(self, "par_" + f)
            setattr(p, f, int(v) if isinstance(getattr(p, f), int) else float(v))
        return p


class Context:
    """rslf_ctx bound to one GPU; launches go to torch's current stream on it."""

    def __init__(self, device: int | torch.device | None = None):
        if device is None:
            device = torch.cuda.current_device()
        self.device = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        h = C.c_void_p()
        check(_lib.lib().rslf_ctx_create(self.device.index or 0, C.byref(h)), "rslf_ctx_create")
        self._h = h
        self.use_current_stream()

    def use_current_stream(self) -> None:
        s = torch.cuda.current_stream(self.device).cuda_stream
        check(_lib.lib().rslf_ctx_set_stream(self._h, C.c_void_p(s)), "rslf_ctx_set_stream")

    def synchronize(self) -> None:
        check(_lib.lib().rslf_ctx_synchronize(self._h), "rslf_ctx_synchronize")

    _DEBUG_DEFAULTS = dict(force_scan=0, force_groups=0, force_packed=-1, px=-1, stream_share=1, stream_groups=0, stream_lds_kib=80,
                           claim_skip=1, time_all=0, row_split=1)   # = the library's own defaults (rslf_internal.hpp, plan::kStreamLdsBytes)
    _FORCE_SCAN = {None: 0, "auto": 0, "generic": 1, "stream": 2}

    def set_debug(self, **hooks) -> None:
        """rslf_ctx_set_debug: test / tuning hooks of THIS context (kernel variant, launch shape).  force_scan also
        takes "generic" / "stream" / None."""
        for k, v in hooks.items():
            if k == "force_scan" and not isinstance(v, int):
                v = self._FORCE_SCAN[v or None]
            check(_lib.lib().rslf_ctx_set_debug(self._h, k.encode(), int(v)), "rslf_ctx_set_debug(%s)" % k)

    def reset_debug(self) -> None:
        self.set_debug(**self._DEBUG_DEFAULTS)

    def last_scan_kernel_ms(self) -> float:
        ms = C.c_float()
        check(_lib.lib().rslf_last_scan_kernel_ms(self._h, C.byref(ms)), "rslf_last_scan_kernel_ms")
        return float(ms.value)

    def scan_time_total_ms(self) -> tuple[float, int]:
        """With set_debug(time_all=1): (summed K2 milliseconds, scan launches) since the last call; resets the sum."""
        ms, n = C.c_float(), C.c_int()
        check(_lib.lib().rslf_scan_time_total_ms(self._h, C.byref(ms), C.byref(n)), "rslf_scan_time_total_ms")
        return float(ms.value), int(n.value)

    def close(self) -> None:
        # at interpreter shutdown the HIP runtime may already be gone: leave the handle to the OS
        if getattr(self, "_h", None) and not sys.is_finalizing():
            _lib.lib().rslf_ctx_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


_DEFAULT_CTX: dict[int, Context] = {}


@atexit.register
def _drop_default_contexts() -> None:
    # release while the HIP runtime is still alive (module globals die too late for that)
    for c in list(_DEFAULT_CTX.values()):
        c._h = None
    _DEFAULT_CTX.clear()


def default_context(device=None) -> Context:
    idx = torch.cuda.current_device() if device is None else torch.device(device).index or 0
    if idx not in _DEFAULT_CTX:
        _DEFAULT_CTX[idx] = Context(idx)
    return _DEFAULT_CTX[idx]


def _ptr(t: Optional[torch.Tensor]) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


class Volume:
    """The light-field slab in HBM, [V][S][pitch][C] float32 (rslf_volume)."""

    def __init__(self, ctx: Context, V: int, S: int, U: int, C_: int):
        self.ctx = ctx
        h = C.c_void_p()
        check(_lib.lib().rslf_volume_create(ctx._h, V, S, U, C_, C.byref(h)), "rslf_volume_create")
        self._h = h
        self.V, self.S, self.U, self.C = V, S, U, C_
        self.scale_used: float | None = None

    # -- constructors -----------------------------------------------------
    @staticmethod
    def from_epis(epis: Sequence[np.ndarray], epi_scale_factor: float = -1.0, ctx: Context | None = None) -> "Volume":
        """The reference's constructor input: a Vec<Mat> of V EPIs, each [S,U] or
        [S,U,3], uint8 or float32 (dc.hpp:425-477)."""
        ctx = ctx or default_context()
        e0 = np.asarray(epis[0])
        S, U = e0.shape[:2]
        C_ = 1 if e0.ndim == 2 else e0.shape[2]
        vol = Volume(ctx, len(epis), S, U, C_)
        is_u8 = e0.dtype == np.uint8
        arrs = [np.ascontiguousarray(e, dtype=np.uint8 if is_u8 else np.float32) for e in epis]
        for a in arrs:
            if a.shape != arrs[0].shape:
                raise ValueError("all EPIs must have the same shape")
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        L = _lib.lib()
        if is_u8:
            check(L.rslf_volume_upload_epis_u8(vol._h, ptrs, 0), "rslf_volume_upload_epis_u8")
            vol.scale_used = 255.0
        else:
            su = C.c_float()
            check(L.rslf_volume_upload_epis_f32(vol._h, ptrs, 0, float(epi_scale_factor), C.byref(su)),
                  "rslf_volume_upload_epis_f32")
            vol.scale_used = float(su.value)
        return vol

    @staticmethod
    def from_images(imgs: Sequence[np.ndarray], epi_scale_factor: float = -1.0, ctx: Context | None = None,
                    transpose: bool = False, rotate_180: bool = False) -> "Volume":
        """Image-major input, images each [V,U] or [V,U,3] -- what rslf::build_epis_from_imgs
        (rslf_io.cpp:194-227) consumes, with its `transpose` / `rotate_180` options: transposed, the EPI of a
        scanline has one row per image COLUMN and one column per image."""
        ctx = ctx or default_context()
        i0 = np.asarray(imgs[0])
        V, cols = i0.shape[:2]
        C_ = 1 if i0.ndim == 2 else i0.shape[2]
        S, U = (cols, len(imgs)) if transpose else (len(imgs), cols)
        vol = Volume(ctx, V, S, U, C_)
        is_u8 = i0.dtype == np.uint8
        arrs = [np.ascontiguousarray(e, dtype=np.uint8 if is_u8 else np.float32) for e in imgs]
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        L = _lib.lib()
        plain = not (transpose or rotate_180)
        if is_u8:
            if plain:
                check(L.rslf_volume_upload_images_u8(vol._h, ptrs, 0), "rslf_volume_upload_images_u8")
            else:
                check(L.rslf_volume_upload_images_xf_u8(vol._h, ptrs, 0, int(transpose), int(rotate_180)),
                      "rslf_volume_upload_images_xf_u8")
            vol.scale_used = 255.0
        else:
            su = C.c_float()
            if plain:
                check(L.rslf_volume_upload_images_f32(vol._h, ptrs, 0, float(epi_scale_factor), C.byref(su)),
                      "rslf_volume_upload_images_f32")
            else:
                check(L.rslf_volume_upload_images_xf_f32(vol._h, ptrs, 0, float(epi_scale_factor), C.byref(su), int(transpose),
                                                         int(rotate_180)), "rslf_volume_upload_images_xf_f32")
            vol.scale_used = float(su.value)
        return vol

    @staticmethod
    def from_dense(vsuc, epi_scale_factor: float = 1.0, ctx: Context | None = None) -> "Volume":
        """Dense [V,S,U] / [V,S,U,C] float32, numpy (uploaded) or a CUDA tensor
        (packed on the device)."""
        ctx = ctx or default_context()
        if isinstance(vsuc, np.ndarray):
            a = np.ascontiguousarray(vsuc, np.float32)
            if a.ndim == 4 and a.shape[3] == 1:
                a = a[..., 0]
            return Volume.from_epis(list(a), epi_scale_factor, ctx)
        t = vsuc
        if t.dim() == 3:
            t = t.unsqueeze(-1)
        if not t.is_cuda or t.dtype != torch.float32:
            raise ValueError("from_dense needs a float32 CUDA tensor or a numpy array")
        t = t.contiguous()
        V, S, U, C_ = t.shape
        vol = Volume(ctx, V, S, U, C_)
        ctx.use_current_stream()
        su = C.c_float()
        check(_lib.lib().rslf_volume_pack_device_f32(vol._h, _ptr(t), float(epi_scale_factor), C.byref(su)),
              "rslf_volume_pack_device_f32")
        vol.scale_used = float(su.value)
        return vol

    def describe(self) -> RslfVolumeDesc:
        d = RslfVolumeDesc()
        check(_lib.lib().rslf_volume_describe(self._h, C.byref(d)), "rslf_volume_describe")
        return d

    def close(self) -> None:
        if getattr(self, "_h", None) and not sys.is_finalizing():
            _lib.lib().rslf_volume_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


# ---- the reference's free functions -------------------------------------

def compute_1D_edge_confidence_pile(vol: Volume, a_s: int, a_edge_confidence_v_u: torch.Tensor,
                                    a_parameters: Depth1DParameters | None = None) -> torch.Tensor:
    """core.hpp:279-287.  Accumulates INTO a_edge_confidence_v_u ([V,U] f32
    CUDA, pass zeros) and returns the new mask ([V,U] u8) -- the reference
    allocates the mask itself (core.hpp:740)."""
    p = (a_parameters or Depth1DParameters()).to_c()
    mask = torch.empty((vol.V, vol.U), dtype=torch.uint8, device=a_edge_confidence_v_u.device)
    vol.ctx.use_current_stream()
    check(_lib.lib().rslf_edge_confidence_pile(vol.ctx._h, vol._h, a_s, C.byref(p), _ptr(a_edge_confidence_v_u), _ptr(mask)),
          "rslf_edge_confidence_pile")
    return mask


def compute_1D_depth_epi_pile(vol: Volume, a_dmin_v_u, a_dmax_v_u, a_dim_d: int, a_s_hat: int,
                              a_edge_confidence_v_u: torch.Tensor, a_edge_confidence_mask_v_u: torch.Tensor,
                              a_disp_confidence_v_u: torch.Tensor, a_best_depth_v_u: torch.Tensor,
                              a_rbar_v_u: torch.Tensor, a_parameters: Depth1DParameters | None = None,
                              a_mask_v_u: torch.Tensor | None = None, *, idx_v_u: torch.Tensor | None = None,
                              score_v_u: torch.Tensor | None = None, depth_raw_v_u: torch.Tensor | None = None,
                              want_stats: bool = False, a_K_r_m_rbar_v_s_u: torch.Tensor | None = None) -> RslfStats | None:
    """core.hpp:293-310.  a_dmin_v_u / a_dmax_v_u are [V,U] f32 CUDA tensors or
    Python floats (the constant planes of dc.hpp:486-487).  All planes are
    updated in place; a_best_depth_v_u ends as the selective median (core.hpp:892).
    a_K_r_m_rbar_v_s_u ([V,S,U] f32, the reference's optional last argument, core.hpp:309): receives
    K(r - rbar)[:, d*] for every pixel that got a disparity (core.hpp:647-651)."""
    p = (a_parameters or Depth1DParameters()).to_c()
    planes = isinstance(a_dmin_v_u, torch.Tensor)
    st = RslfStats() if want_stats else None
    vol.ctx.use_current_stream()
    if a_K_r_m_rbar_v_s_u is not None and idx_v_u is None:
        idx_v_u = torch.empty((vol.V, vol.U), dtype=torch.int32, device=a_best_depth_v_u.device)
    check(_lib.lib().rslf_depth_epi_pile(
        vol.ctx._h, vol._h, _ptr(a_dmin_v_u if planes else None), _ptr(a_dmax_v_u if planes else None),
        0.0 if planes else float(a_dmin_v_u), 0.0 if planes else float(a_dmax_v_u), a_dim_d, a_s_hat,
        _ptr(a_edge_confidence_v_u), _ptr(a_edge_confidence_mask_v_u), _ptr(a_disp_confidence_v_u),
        _ptr(a_best_depth_v_u), _ptr(a_rbar_v_u), C.byref(p), _ptr(a_mask_v_u), _ptr(idx_v_u), _ptr(score_v_u),
        _ptr(depth_raw_v_u), C.byref(st) if st is not None else None), "rslf_depth_epi_pile")
    if a_K_r_m_rbar_v_s_u is not None:
        check(_lib.lib().rslf_kernel_columns_pile(
            vol.ctx._h, vol._h, _ptr(a_dmin_v_u if planes else None), _ptr(a_dmax_v_u if planes else None),
            0.0 if planes else float(a_dmin_v_u), 0.0 if planes else float(a_dmax_v_u), a_dim_d, a_s_hat, C.byref(p),
            _ptr(idx_v_u), _ptr(a_K_r_m_rbar_v_s_u)), "rslf_kernel_columns_pile")
    return st


def selective_median_filter(a_src: torch.Tensor, vol: Volume, a_s_hat: int, a_size: int, a_mask_v_u: torch.Tensor,
                            a_epsilon: float) -> torch.Tensor:
    """core.hpp:366-375; returns a_dst."""
    dst = torch.empty_like(a_src)
    vol.ctx.use_current_stream()
    check(_lib.lib().rslf_selective_median(vol.ctx._h, vol._h, _ptr(a_src), _ptr(dst), a_s_hat, a_size,
                                           _ptr(a_mask_v_u), float(a_epsilon)), "rslf_selective_median")
    return dst


# ---- the reference's class -------------------------------------------------

class Depth1DComputer_pile:
    """rslf::Depth1DComputer_pile<T> (dc.hpp:93-143).

    `epis` is the reference's Vec<Mat> (a list of V arrays [S,U] or [S,U,3],
    uint8 or float32), a dense numpy / CUDA array [V,S,U(,C)], or a Volume that
    is already resident.  After run(), the result members hold CUDA tensors."""

    def __init__(self, epis, dmin: float, dmax: float, dim_d: int, s_hat: int = -1, epi_scale_factor: float = -1.0,
                 parameters: Depth1DParameters | None = None, ctx: Context | None = None):
        self.m_parameters = parameters or Depth1DParameters.get_default()
        if isinstance(epis, Volume):
            self.m_epis = epis
        elif isinstance(epis, (list, tuple)):
            self.m_epis = Volume.from_epis(epis, epi_scale_factor, ctx)
        elif isinstance(epis, np.ndarray):
            a = epis[..., 0] if (epis.ndim == 4 and epis.shape[3] == 1) else epis
            self.m_epis = Volume.from_epis(list(a), epi_scale_factor, ctx)
        else:
            self.m_epis = Volume.from_dense(epis, epi_scale_factor if epi_scale_factor > 0 else 1.0, ctx)
        vol = self.m_epis
        self.m_dim_d = int(dim_d)
        self.m_dmin, self.m_dmax = float(dmin), float(dmax)
        # dc.hpp:490-498
        self.m_s_hat = int(np.floor((0.0 + vol.S) / 2)) if (s_hat < 0 or s_hat > vol.S - 1) else int(s_hat)
        dev = vol.ctx.device
        V, U, C_ = vol.V, vol.U, vol.C
        self.m_edge_confidence_v_u = torch.empty((V, U), dtype=torch.float32, device=dev)
        self.m_edge_confidence_mask_v_u = torch.empty((V, U), dtype=torch.uint8, device=dev)
        self.m_disp_confidence_v_u = torch.empty((V, U), dtype=torch.float32, device=dev)
        self.m_best_depth_v_u = torch.empty((V, U), dtype=torch.float32, device=dev)
        self.m_rbar_v_u = torch.empty((V, U, C_), dtype=torch.float32, device=dev)
        # parity witnesses, not in the reference
        self.m_depth_idx_v_u = torch.empty((V, U), dtype=torch.int32, device=dev)
        self.m_score_v_u = torch.empty((V, U), dtype=torch.float32, device=dev)
        self.m_depth_raw_v_u = torch.empty((V, U), dtype=torch.float32, device=dev)
        self.stats: RslfStats | None = None

    def run(self, want_stats: bool = True) -> None:
        """dc.hpp:513-565."""
        vol = self.m_epis
        p = self.m_parameters.to_c()
        st = RslfStats() if want_stats else None
        vol.ctx.use_current_stream()
        check(_lib.lib().rslf_depth1d_pile_run(
            vol.ctx._h, vol._h, self.m_dmin, self.m_dmax, self.m_dim_d, self.m_s_hat, C.byref(p),
            _ptr(self.m_edge_confidence_v_u), _ptr(self.m_edge_confidence_mask_v_u), _ptr(self.m_disp_confidence_v_u),
            _ptr(self.m_best_depth_v_u), _ptr(self.m_rbar_v_u), _ptr(self.m_depth_idx_v_u), _ptr(self.m_score_v_u),
            _ptr(self.m_depth_raw_v_u), C.byref(st) if st is not None else None), "rslf_depth1d_pile_run")
        self.stats = st

    def get_s_hat(self) -> int:
        return self.m_s_hat

    def results(self) -> dict:
        """Host copies of every output plane."""
        torch.cuda.synchronize(self.m_epis.ctx.device)
        return dict(
            edge_confidence=self.m_edge_confidence_v_u.cpu().numpy(),
            edge_mask=self.m_edge_confidence_mask_v_u.cpu().numpy(),
            disp_confidence=self.m_disp_confidence_v_u.cpu().numpy(),
            depth=self.m_best_depth_v_u.cpu().numpy(),
            rbar=self.m_rbar_v_u.cpu().numpy(),
            depth_idx=self.m_depth_idx_v_u.cpu().numpy(),
            score=self.m_score_v_u.cpu().numpy(),
            depth_raw=self.m_depth_raw_v_u.cpu().numpy(),
        )


class MultiDevice:
    """rslf_multi: Depth1DComputer_pile's constructor + run() + result Mats in one call on HOST arrays, the scanlines cut
    into one block per device (and every block into chunks whose upload, kernels and download overlap).  Host planes
    come back as numpy arrays; no collective is involved (each device writes its rows of the caller's planes)."""

    def __init__(self, devices: Sequence[int] | None = None):
        devs = list(devices) if devices else []
        arr = (C.c_int * max(1, len(devs)))(*devs) if devs else None
        h = C.c_void_p()
        check(_lib.lib().rslf_multi_create(arr, len(devs), C.byref(h)), "rslf_multi_create")
        self._h = h

    def device_count(self) -> int:
        return int(_lib.lib().rslf_multi_device_count(self._h))

    def set_chunk_rows(self, rows: int) -> None:
        check(_lib.lib().rslf_multi_set_chunk_rows(self._h, int(rows)), "rslf_multi_set_chunk_rows")

    def peer_access(self) -> list:
        """n x n matrix: [i][k] = 1 if worker i reaches worker k's memory directly (same GPU, or peer access enabled over
        xGMI), 0 if copies between them stage through the host (rslf_multi_peer_access)."""
        n = self.device_count()
        return [[int(_lib.lib().rslf_multi_peer_access(self._h, i, k)) for k in range(n)] for i in range(n)]

    def depth1d_pile(self, epis: Sequence[np.ndarray], dmin: float, dmax: float, dim_d: int, s_hat: int = -1,
                     epi_scale_factor: float = -1.0, parameters: Depth1DParameters | None = None) -> dict:
        """epis: the reference's Vec<Mat> -- V arrays [S,U] or [S,U,3], all uint8 or all float32."""
        first = np.asarray(epis[0])
        dt = first.dtype
        if dt not in (np.uint8, np.float32):
            raise TypeError("EPIs must be uint8 or float32 (dc.hpp:149-154)")
        # keeps the buffers alive over the call; a thousand EPIs: no per-array conversions or ctypes objects where none are needed
        keep = [e if (type(e) is np.ndarray and e.dtype == dt and e.flags.c_contiguous) else np.ascontiguousarray(e, dtype=dt) for e in epis]
        V = len(keep)
        S, U = keep[0].shape[0], keep[0].shape[1]
        C_ = 1 if keep[0].ndim == 2 else keep[0].shape[2]
        if any(e.shape != keep[0].shape for e in keep):
            raise ValueError("every EPI must have the shape of the first, %s" % (keep[0].shape,))
        ptrs = (C.c_void_p * V)(*[e.__array_interface__["data"][0] for e in keep])
        out = dict(edge_confidence=np.empty((V, U), np.float32), edge_mask=np.empty((V, U), np.uint8),
                   disp_confidence=np.empty((V, U), np.float32), depth=np.empty((V, U), np.float32),
                   rbar=np.empty((V, U, C_), np.float32), depth_idx=np.empty((V, U), np.int32),
                   score=np.empty((V, U), np.float32), depth_raw=np.empty((V, U), np.float32))
        hp = [out[k].ctypes.data_as(C.c_void_p) for k in ("edge_confidence", "edge_mask", "disp_confidence", "depth", "rbar",
                                                         "depth_idx", "score", "depth_raw")]
        p = (parameters or Depth1DParameters()).to_c()
        st = RslfStats()
        L = _lib.lib()
        if dt == np.uint8:
            check(L.rslf_multi_depth1d_pile_u8(self._h, ptrs, 0, V, S, U, C_, float(dmin), float(dmax), int(dim_d), int(s_hat),
                                               C.byref(p), *hp, C.byref(st)), "rslf_multi_depth1d_pile_u8")
            self.scale_used = 255.0
        else:
            su = C.c_float()
            check(L.rslf_multi_depth1d_pile_f32(self._h, ptrs, 0, V, S, U, C_, float(epi_scale_factor), float(dmin), float(dmax),
                                                int(dim_d), int(s_hat), C.byref(p), *hp, C.byref(st), C.byref(su)),
                  "rslf_multi_depth1d_pile_f32")
            self.scale_used = float(su.value)
        self.stats = st
        return out

    def depth2d(self, epis: Sequence[np.ndarray], dmin: float, dmax: float, dim_d: int, epi_scale_factor: float = -1.0,
                parameters: Depth1DParameters | None = None) -> dict:
        """Depth2DComputer (constructor + run + getters) over this object's devices: the 2-D sweep cut into one block of
        scanlines per device, the neighbours' boundary rows exchanged by peer copy on every visit
        (rslf_multi_depth2d_run_f32 / _u8).  Host EPIs in, numpy planes [S, V, U] out."""
        first = np.asarray(epis[0])
        dt = first.dtype
        if dt not in (np.uint8, np.float32):
            raise TypeError("EPIs must be uint8 or float32 (dc.hpp:149-154)")
        keep = [e if (type(e) is np.ndarray and e.dtype == dt and e.flags.c_contiguous) else np.ascontiguousarray(e, dtype=dt) for e in epis]
        V = len(keep)
        S, U = keep[0].shape[0], keep[0].shape[1]
        C_ = 1 if keep[0].ndim == 2 else keep[0].shape[2]
        if any(e.shape != keep[0].shape for e in keep):
            raise ValueError("every EPI must have the shape of the first, %s" % (keep[0].shape,))
        ptrs = (C.c_void_p * V)(*[e.__array_interface__["data"][0] for e in keep])
        out = dict(edge_confidence=np.empty((S, V, U), np.float32), edge_mask=np.empty((S, V, U), np.uint8),
                   disp_confidence=np.empty((S, V, U), np.float32), depth=np.empty((S, V, U), np.float32),
                   rbar=np.empty((S, V, U, C_), np.float32), scan_mask=np.empty((S, V, U), np.uint8))
        hp = [out[k].ctypes.data_as(C.c_void_p) for k in ("edge_confidence", "edge_mask", "disp_confidence", "depth", "rbar", "scan_mask")]
        p = (parameters or Depth1DParameters()).to_c()
        st = RslfStats()
        L = _lib.lib()
        if dt == np.uint8:
            check(L.rslf_multi_depth2d_run_u8(self._h, ptrs, 0, V, S, U, C_, float(dmin), float(dmax), int(dim_d), C.byref(p), *hp,
                                              C.byref(st)), "rslf_multi_depth2d_run_u8")
            self.scale_used = 255.0
        else:
            su = C.c_float()
            check(L.rslf_multi_depth2d_run_f32(self._h, ptrs, 0, V, S, U, C_, float(epi_scale_factor), float(dmin), float(dmax),
                                               int(dim_d), C.byref(p), *hp, C.byref(st), C.byref(su)), "rslf_multi_depth2d_run_f32")
            self.scale_used = float(su.value)
        self.stats = st
        return out

    def fine_to_coarse(self, epis: Sequence[np.ndarray], d_min: float, d_max: float, dim_d: int, epi_scale_factor: float = -1.0,
                       parameters: Depth1DParameters | None = None, max_pyr_depth: int = -1, accept_all_last_scale: bool = True):
        """FineToCoarse (constructor + run + get_results) over this object's devices (rslf_multi_fine_to_coarse_run_host):
        every level's sweep sharded by scanline.  Returns (out_map [S,V,U] f32, out_validity [S,V,U] u8, levels)."""
        first = np.asarray(epis[0])
        dt = first.dtype
        if dt not in (np.uint8, np.float32):
            raise TypeError("EPIs must be uint8 or float32 (dc.hpp:149-154)")
        keep = [e if (type(e) is np.ndarray and e.dtype == dt and e.flags.c_contiguous) else np.ascontiguousarray(e, dtype=dt) for e in epis]
        V = len(keep)
        S, U = keep[0].shape[0], keep[0].shape[1]
        C_ = 1 if keep[0].ndim == 2 else keep[0].shape[2]
        ptrs = (C.c_void_p * V)(*[e.__array_interface__["data"][0] for e in keep])
        out_map = np.empty((S, V, U), np.float32)
        out_valid = np.empty((S, V, U), np.uint8)
        p = (parameters or Depth1DParameters()).to_c()
        st, nl = RslfStats(), C.c_int()
        check(_lib.lib().rslf_multi_fine_to_coarse_run_host(self._h, ptrs, 1 if dt == np.uint8 else 0, V, S, U, C_, 0, float(d_min), float(d_max),
                                                            int(dim_d), float(epi_scale_factor), C.byref(p), int(max_pyr_depth),
                                                            1 if accept_all_last_scale else 0, out_map.ctypes.data_as(C.c_void_p),
                                                            out_valid.ctypes.data_as(C.c_void_p), C.byref(nl), C.byref(st)),
              "rslf_multi_fine_to_coarse_run_host")
        self.stats = st
        return out_map, out_valid, int(nl.value)

    def depth1d_pile_device_out(self, epis: Sequence[np.ndarray], dmin: float, dmax: float, dim_d: int, out_device: int = 0,
                                s_hat: int = -1, epi_scale_factor: float = -1.0, parameters: Depth1DParameters | None = None) -> dict:
        """The same with the result planes left on `out_device` as CUDA tensors (float32 EPIs): every worker copies its
        rows there with a peer copy (rslf_multi_depth1d_pile_f32_dev)."""
        keep = [np.ascontiguousarray(e, dtype=np.float32) for e in epis]
        V = len(keep)
        S, U = keep[0].shape[0], keep[0].shape[1]
        C_ = 1 if keep[0].ndim == 2 else keep[0].shape[2]
        ptrs = (C.c_void_p * V)(*[e.ctypes.data for e in keep])
        dev = torch.device("cuda", out_device)
        mk = lambda shape, dt: torch.empty(shape, dtype=dt, device=dev)
        out = dict(edge_confidence=mk((V, U), torch.float32), edge_mask=mk((V, U), torch.uint8), disp_confidence=mk((V, U), torch.float32),
                   depth=mk((V, U), torch.float32), rbar=mk((V, U, C_), torch.float32), depth_idx=mk((V, U), torch.int32),
                   score=mk((V, U), torch.float32), depth_raw=mk((V, U), torch.float32))
        torch.cuda.synchronize(dev)
        hp = [_ptr(out[k]) for k in ("edge_confidence", "edge_mask", "disp_confidence", "depth", "rbar", "depth_idx", "score", "depth_raw")]
        p = (parameters or Depth1DParameters()).to_c()
        st, su = RslfStats(), C.c_float()
        check(_lib.lib().rslf_multi_depth1d_pile_f32_dev(self._h, ptrs, 0, V, S, U, C_, float(epi_scale_factor), float(dmin), float(dmax),
                                                         int(dim_d), int(s_hat), C.byref(p), int(out_device), *hp, C.byref(st), C.byref(su)),
              "rslf_multi_depth1d_pile_f32_dev")
        self.stats, self.scale_used = st, float(su.value)
        return out

    def close(self) -> None:
        if getattr(self, "_h", None) and not sys.is_finalizing():
            _lib.lib().rslf_multi_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


# ---- "next" row: the 2-D sweep -------------------------

def compute_2D_edge_confidence(vol: Volume, a_edge_confidence_s_v_u: torch.Tensor,
                               a_parameters: Depth1DParameters | None = None) -> torch.Tensor:
    """core.hpp:323-330.  Accumulates into a_edge_confidence_s_v_u ([S,V,U] f32, pass zeros);
    returns the masks [S,V,U] u8."""
    p = (a_parameters or Depth1DParameters()).to_c()
    mask = torch.empty((vol.S, vol.V, vol.U), dtype=torch.uint8, device=a_edge_confidence_s_v_u.device)
    vol.ctx.use_current_stream()
    check(_lib.lib().rslf_edge_confidence_2d(vol.ctx._h, vol._h, C.byref(p), _ptr(a_edge_confidence_s_v_u), _ptr(mask)),
          "rslf_edge_confidence_2d")
    return mask


def compute_2D_depth_epi(vol: Volume, a_dmin_s_v_u, a_dmax_s_v_u, a_dim_d: int, a_edge_confidence_s_v_u: torch.Tensor,
                         a_edge_confidence_mask_s_v_u: torch.Tensor, a_disp_confidence_s_v_u: torch.Tensor,
                         a_best_depth_s_v_u: torch.Tensor, a_rbar_s_v_u: torch.Tensor,
                         a_parameters: Depth1DParameters | None = None, *, scan_mask_s_v_u: torch.Tensor | None = None,
                         want_stats: bool = False) -> RslfStats | None:
    """core.hpp:336-351 (the line-confidence argument does not exist in the default build).
    a_dmin_s_v_u / a_dmax_s_v_u: [S,V,U] f32 CUDA tensors or Python floats."""
    p = (a_parameters or Depth1DParameters()).to_c()
    planes = isinstance(a_dmin_s_v_u, torch.Tensor)
    st = RslfStats() if want_stats else None
    vol.ctx.use_current_stream()
    check(_lib.lib().rslf_depth_epi_2d(
        vol.ctx._h, vol._h, _ptr(a_dmin_s_v_u if planes else None), _ptr(a_dmax_s_v_u if planes else None),
        0.0 if planes else float(a_dmin_s_v_u), 0.0 if planes else float(a_dmax_s_v_u), a_dim_d,
        _ptr(a_edge_confidence_s_v_u), _ptr(a_edge_confidence_mask_s_v_u), _ptr(a_disp_confidence_s_v_u),
        _ptr(a_best_depth_s_v_u), _ptr(a_rbar_s_v_u), C.byref(p), _ptr(scan_mask_s_v_u),
        C.byref(st) if st is not None else None), "rslf_depth_epi_2d")
    return st


class Depth2DComputer:
    """rslf::Depth2DComputer<T> (dc.hpp:166-225, :651-805): disparities for every view of the light field,
    visiting the views from the centre outwards and propagating along EPI lines."""

    def __init__(self, epis, dmin: float, dmax: float, dim_d: int, epi_scale_factor: float = -1.0,
                 parameters: Depth1DParameters | None = None, verbose: bool = False, ctx: Context | None = None):
        self.m_parameters = parameters or Depth1DParameters.get_default()
        if isinstance(epis, Volume):
            self.m_epis = epis
        elif isinstance(epis, (list, tuple)):
            self.m_epis = Volume.from_epis(epis, epi_scale_factor, ctx)
        elif isinstance(epis, np.ndarray):
            a = epis[..., 0] if (epis.ndim == 4 and epis.shape[3] == 1) else epis
            self.m_epis = Volume.from_epis(list(a), epi_scale_factor, ctx)
        else:
            self.m_epis = Volume.from_dense(epis, epi_scale_factor if epi_scale_factor > 0 else 1.0, ctx)
        vol = self.m_epis
        self.m_dim_d, self.m_dmin, self.m_dmax = int(dim_d), float(dmin), float(dmax)
        self.m_accept_all = False
        dev = vol.ctx.device
        S, V, U, C_ = vol.S, vol.V, vol.U, vol.C
        self.m_edge_confidence_s_v_u = torch.empty((S, V, U), dtype=torch.float32, device=dev)
        self.m_edge_confidence_mask_s_v_u = torch.empty((S, V, U), dtype=torch.uint8, device=dev)
        self.m_disp_confidence_s_v_u = torch.empty((S, V, U), dtype=torch.float32, device=dev)
        self.m_best_depth_s_v_u = torch.empty((S, V, U), dtype=torch.float32, device=dev)
        self.m_rbar_s_v_u = torch.empty((S, V, U, C_), dtype=torch.float32, device=dev)
        self.m_scan_mask_s_v_u = torch.empty((S, V, U), dtype=torch.uint8, device=dev)
        self.stats: RslfStats | None = None

    def run(self, want_stats: bool = True) -> None:
        """dc.hpp:748-805."""
        vol = self.m_epis
        p = self.m_parameters.to_c()
        st = RslfStats() if want_stats else None
        vol.ctx.use_current_stream()
        check(_lib.lib().rslf_depth2d_run(
            vol.ctx._h, vol._h, self.m_dmin, self.m_dmax, self.m_dim_d, C.byref(p), _ptr(self.m_edge_confidence_s_v_u),
            _ptr(self.m_edge_confidence_mask_s_v_u), _ptr(self.m_disp_confidence_s_v_u), _ptr(self.m_best_depth_s_v_u),
            _ptr(self.m_rbar_s_v_u), _ptr(self.m_scan_mask_s_v_u), C.byref(st) if st is not None else None),
            "rslf_depth2d_run")
        self.stats = st

    def get_depths_s_v_u(self) -> torch.Tensor:
        return self.m_best_depth_s_v_u

    def set_accept_all(self, b: bool) -> None:
        self.m_accept_all = bool(b)

    def get_valid_depths_mask_s_v_u(self) -> torch.Tensor:
        """dc.hpp:893-915, default build: C_e > edge threshold (or everything > -1 with accept_all)."""
        thr = -1.0 if self.m_accept_all else float(np.float32(self.m_parameters.par_edge_score_threshold))
        return (self.m_edge_confidence_s_v_u > thr).to(torch.uint8) * 255

    def results(self) -> dict:
        torch.cuda.synchronize(self.m_epis.ctx.device)
        return dict(edge_confidence=self.m_edge_confidence_s_v_u.cpu().numpy(),
                    edge_mask=self.m_edge_confidence_mask_s_v_u.cpu().numpy(),
                    disp_confidence=self.m_disp_confidence_s_v_u.cpu().numpy(),
                    depth=self.m_best_depth_s_v_u.cpu().numpy(), rbar=self.m_rbar_s_v_u.cpu().numpy(),
                    scan_mask=self.m_scan_mask_s_v_u.cpu().numpy())


# ---- "next" row: the single-EPI class -------------------

def compute_1D_depth_epi(vol: Volume, a_dmin_v_u, a_dmax_v_u, a_dim_d: int, a_s_hat: int,
                         a_edge_confidence_v_u: torch.Tensor, a_edge_confidence_mask_v_u: torch.Tensor,
                         a_disp_confidence_v_u: torch.Tensor, a_best_depth_v_u: torch.Tensor, a_rbar_v_u: torch.Tensor,
                         a_parameters: Depth1DParameters | None = None, a_mask_v_u: torch.Tensor | None = None, *,
                         idx_v_u: torch.Tensor | None = None, score_v_u: torch.Tensor | None = None,
                         want_stats: bool = False) -> RslfStats | None:
    """core.hpp:251-267 for every EPI of the volume: the scan alone, no selective median."""
    p = (a_parameters or Depth1DParameters()).to_c()
    planes = isinstance(a_dmin_v_u, torch.Tensor)
    st = RslfStats() if want_stats else None
    vol.ctx.use_current_stream()
    check(_lib.lib().rslf_depth_epi_scan(
        vol.ctx._h, vol._h, _ptr(a_dmin_v_u if planes else None), _ptr(a_dmax_v_u if planes else None),
        0.0 if planes else float(a_dmin_v_u), 0.0 if planes else float(a_dmax_v_u), a_dim_d, a_s_hat,
        _ptr(a_edge_confidence_v_u), _ptr(a_edge_confidence_mask_v_u), _ptr(a_disp_confidence_v_u),
        _ptr(a_best_depth_v_u), _ptr(a_rbar_v_u), C.byref(p), _ptr(a_mask_v_u), _ptr(idx_v_u), _ptr(score_v_u),
        C.byref(st) if st is not None else None), "rslf_depth_epi_scan")
    return st


class Depth1DComputer:
    """rslf::Depth1DComputer<T> (dc.hpp:26-70, :256-371): ONE EPI ([S,U] or [S,U,3], uint8 or float32), edge
    confidence + scan, no median.  Results are [U] CUDA tensors."""

    def __init__(self, epi, dmin: float, dmax: float, dim_d: int, s_hat: int = -1, epi_scale_factor: float = -1.0,
                 parameters: Depth1DParameters | None = None, ctx: Context | None = None):
        self.m_parameters = parameters or Depth1DParameters.get_default()
        self.m_epi = Volume.from_epis([np.asarray(epi)], epi_scale_factor, ctx)
        vol = self.m_epi
        self.m_dim_d, self.m_dmin, self.m_dmax = int(dim_d), float(dmin), float(dmax)
        self.m_s_hat = int(np.floor((0.0 + vol.S) / 2)) if (s_hat < 0 or s_hat > vol.S - 1) else int(s_hat)   # dc.hpp:303-311
        dev = vol.ctx.device
        U, C_ = vol.U, vol.C
        self.m_edge_confidence_u = torch.empty((1, U), dtype=torch.float32, device=dev)
        self.m_edge_confidence_mask_u = torch.empty((1, U), dtype=torch.uint8, device=dev)
        self.m_disp_confidence_u = torch.empty((1, U), dtype=torch.float32, device=dev)
        self.m_best_depth_u = torch.empty((1, U), dtype=torch.float32, device=dev)
        self.m_rbar_u = torch.empty((1, U, C_), dtype=torch.float32, device=dev)
        self.m_depth_idx_u = torch.empty((1, U), dtype=torch.int32, device=dev)
        self.m_score_u = torch.empty((1, U), dtype=torch.float32, device=dev)
        self.stats: RslfStats | None = None

    def run(self) -> None:
        """dc.hpp:325-371."""
        vol = self.m_epi
        p = self.m_parameters.to_c()
        st = RslfStats()
        vol.ctx.use_current_stream()
        check(_lib.lib().rslf_depth1d_run(
            vol.ctx._h, vol._h, self.m_dmin, self.m_dmax, self.m_dim_d, self.m_s_hat, C.byref(p),
            _ptr(self.m_edge_confidence_u), _ptr(self.m_edge_confidence_mask_u), _ptr(self.m_disp_confidence_u),
            _ptr(self.m_best_depth_u), _ptr(self.m_rbar_u), _ptr(self.m_depth_idx_u), _ptr(self.m_score_u), C.byref(st)),
            "rslf_depth1d_run")
        self.stats = st

    def results(self) -> dict:
        torch.cuda.synchronize(self.m_epi.ctx.device)
        return dict(edge_confidence=self.m_edge_confidence_u[0].cpu().numpy(), edge_mask=self.m_edge_confidence_mask_u[0].cpu().numpy(),
                    disp_confidence=self.m_disp_confidence_u[0].cpu().numpy(), depth=self.m_best_depth_u[0].cpu().numpy(),
                    rbar=self.m_rbar_u[0].cpu().numpy(), depth_idx=self.m_depth_idx_u[0].cpu().numpy(),
                    score=self.m_score_u[0].cpu().numpy())


# ---- "next" row: fine-to-coarse ---------------------------

_MIN_SPATIAL_DIM = 10   # rslf_fine_to_coarse.hpp:8


def downsample_EPIs(raw_vsuc: torch.Tensor, ctx: Context | None = None, is_u8: bool = False) -> torch.Tensor:
    """rslf::downsample_EPIs (src/rslf_fine_to_coarse_core.cpp:14-60) on a dense RAW float32 CUDA volume
    [V,S,U,C] -> [V2,S,U2,C].  is_u8: the values are uchar levels (a CV_8U light field) and the blur and the halving
    run in uchar arithmetic, as the reference's Mats of the input's own type do."""
    ctx = ctx or default_context(raw_vsuc.device)
    t = raw_vsuc.contiguous()
    V, S, U, C_ = t.shape
    v2, u2 = C.c_int(), C.c_int()
    check(_lib.lib().rslf_f2c_level_dims(V, U, C.byref(v2), C.byref(u2)), "rslf_f2c_level_dims")
    out = torch.empty((v2.value, S, u2.value, C_), dtype=torch.float32, device=t.device)
    ctx.use_current_stream()
    if is_u8:
        check(_lib.lib().rslf_downsample_epis_u8(ctx._h, _ptr(t), V, S, U, C_, _ptr(out)), "rslf_downsample_epis_u8")
    else:
        check(_lib.lib().rslf_downsample_epis_f32(ctx._h, _ptr(t), V, S, U, C_, _ptr(out)), "rslf_downsample_epis_f32")
    return out


class FineToCoarse:
    """rslf::FineToCoarse<T> (include/rslf_fine_to_coarse.hpp:26-81, :103-324): a pyramid of Depth2DComputers,
    each level halving (v, u) -- never s --, with slope_factor = U_p / U_0, per-pixel hypothesis ranges
    tightened from the finer level, and a coarse-to-fine fusion of the disparity maps.

    `epis`: the reference's Vec<Mat> (list of V arrays [S,U] / [S,U,3]) or a dense array [V,S,U(,C)],
    float32 or uint8.  A uint8 light field keeps uchar arithmetic through the pyramid, as the reference's CV_8U Mats do."""

    def __init__(self, epis, d_min: float, d_max: float, dim_d: int, epi_scale_factor: float = -1.0,
                 parameters: Depth1DParameters | None = None, max_pyr_depth: int = -1, accept_all_last_scale: bool = True,
                 ctx: Context | None = None):
        import copy
        self.m_parameters = parameters or Depth1DParameters.get_default()
        ctx = ctx or default_context()
        dev = ctx.device
        a = np.stack([np.asarray(e) for e in epis]) if isinstance(epis, (list, tuple)) else np.asarray(epis)
        if a.ndim == 3:
            a = a[..., None]
        self._is_u8 = a.dtype == np.uint8
        raw = torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
        start_dim_u = raw.shape[2]
        if max_pyr_depth < 1:
            max_pyr_depth = 1 << 30
        self.m_computers: list[Depth2DComputer] = []
        self.m_parameter_instances: list[Depth1DParameters] = []
        dim_v, dim_u, counter = raw.shape[0], raw.shape[2], 0
        L = _lib.lib()
        while dim_v > _MIN_SPATIAL_DIM and dim_u > _MIN_SPATIAL_DIM and counter < max_pyr_depth:   # f2c.hpp:130
            counter += 1
            new_parameters = copy.copy(self.m_parameters)
            new_parameters.par_slope_factor = float(np.float32((0.0 + dim_u) / start_dim_u))       # f2c.hpp:139
            # Depth2DComputer's constructor normalises ITS input: uchar by 1/255, float by the level's own max
            # unless a scale factor was given (dc.hpp:671-705)
            if self._is_u8:
                scale = 255.0
            elif epi_scale_factor < 0:
                mx = C.c_float()
                ctx.use_current_stream()
                check(L.rslf_device_max_f32(ctx._h, _ptr(raw), raw.numel(), C.byref(mx)), "rslf_device_max_f32")
                scale = float(mx.value)
            else:
                scale = float(epi_scale_factor)
            vol = Volume.from_dense(raw, scale, ctx)
            self.m_computers.append(Depth2DComputer(vol, d_min, d_max, dim_d, parameters=new_parameters))
            self.m_parameter_instances.append(new_parameters)
            raw = downsample_EPIs(raw, ctx, self._is_u8)                                           # f2c.hpp:145-147
            dim_v, dim_u = raw.shape[0], raw.shape[2]
        if not self.m_computers:
            raise ValueError("light field smaller than _MIN_SPATIAL_DIM: no pyramid level")
        if accept_all_last_scale:
            self.m_computers[-1].set_accept_all(True)                                              # f2c.hpp:157-158
        self._dmin, self._dmax = float(d_min), float(d_max)

    def run(self) -> None:
        """f2c.hpp:171-299."""
        L = _lib.lib()
        for p, comp in enumerate(self.m_computers):
            vol = comp.m_epis
            S, V, U = vol.S, vol.V, vol.U
            if p == 0:
                comp.run(want_stats=True)
            else:
                up = self.m_computers[p - 1]
                dev = vol.ctx.device
                dmin = torch.full((S, V, U), self._dmin, dtype=torch.float32, device=dev)
                dmax = torch.full((S, V, U), self._dmax, dtype=torch.float32, device=dev)
                mask_up = up.get_valid_depths_mask_s_v_u()
                vol.ctx.use_current_stream()
                check(L.rslf_f2c_tighten_bounds(vol.ctx._h, _ptr(up.m_best_depth_s_v_u), _ptr(mask_up), S, up.m_epis.V, up.m_epis.U,
                                                _ptr(dmin), _ptr(dmax), V, U), "rslf_f2c_tighten_bounds")
                comp.m_dmin_s_v_u, comp.m_dmax_s_v_u = dmin, dmax
                # Depth2DComputer::run with per-pixel ranges (edit_dmin / edit_dmax, dc.hpp:201-203)
                for t in (comp.m_edge_confidence_s_v_u, comp.m_disp_confidence_s_v_u, comp.m_best_depth_s_v_u, comp.m_rbar_s_v_u):
                    t.zero_()
                comp.m_edge_confidence_mask_s_v_u = compute_2D_edge_confidence(vol, comp.m_edge_confidence_s_v_u, comp.m_parameters)
                comp.stats = compute_2D_depth_epi(vol, dmin, dmax, comp.m_dim_d, comp.m_edge_confidence_s_v_u,
                                                  comp.m_edge_confidence_mask_s_v_u, comp.m_disp_confidence_s_v_u,
                                                  comp.m_best_depth_s_v_u, comp.m_rbar_s_v_u, comp.m_parameters,
                                                  scan_mask_s_v_u=comp.m_scan_mask_s_v_u, want_stats=True)

    def get_results(self):
        """f2c.hpp:302-324 -> (out_map_s_v_u [S,V,U] f32, out_validity_s_v_u [S,V,U] u8) at the finest scale."""
        comps = self.m_computers
        P, S = len(comps), comps[0].m_epis.S
        disp = [c.m_best_depth_s_v_u.contiguous() for c in comps]
        valid = [c.get_valid_depths_mask_s_v_u().contiguous() for c in comps]
        dp = (C.c_void_p * P)(*[t.data_ptr() for t in disp])
        vp = (C.c_void_p * P)(*[t.data_ptr() for t in valid])
        Vp = (C.c_int * P)(*[c.m_epis.V for c in comps])
        Up = (C.c_int * P)(*[c.m_epis.U for c in comps])
        dev = comps[0].m_epis.ctx.device
        out_map = torch.empty((S, comps[0].m_epis.V, comps[0].m_epis.U), dtype=torch.float32, device=dev)
        out_valid = torch.empty((S, comps[0].m_epis.V, comps[0].m_epis.U), dtype=torch.uint8, device=dev)
        ctx = comps[0].m_epis.ctx
        ctx.use_current_stream()
        check(_lib.lib().rslf_f2c_fuse(ctx._h, dp, vp, Vp, Up, P, S, _ptr(out_map), _ptr(out_valid)), "rslf_f2c_fuse")
        return out_map, out_valid
