"""ctypes binding of include/rslf_hip.h (librslf_hip.so).

There is no fallback: if the library is missing or fails to load, or no gfx950
device is visible when a context is created, this raises.
"""
from __future__ import annotations

import ctypes as C
import os

from . import build as _build

_LIB = None

ABI_VERSION = 6

# every symbol include/rslf_hip.h declares
SYMBOLS = [
    "rslf_abi_version", "rslf_status_string", "rslf_last_error", "rslf_device_count", "rslf_default_params",
    "rslf_ctx_create", "rslf_ctx_destroy", "rslf_ctx_set_stream", "rslf_ctx_synchronize", "rslf_ctx_set_debug", "rslf_debug_inject",
    "rslf_multi_create", "rslf_multi_destroy", "rslf_multi_device_count", "rslf_multi_set_chunk_rows", "rslf_multi_peer_access",
    "rslf_multi_depth1d_pile_f32", "rslf_multi_depth1d_pile_u8", "rslf_multi_depth1d_pile_f32_dev",
    "rslf_multi_depth2d_run_f32", "rslf_multi_depth2d_run_u8", "rslf_multi_fine_to_coarse_run_host",
    "rslf_volume_create", "rslf_volume_destroy", "rslf_volume_describe",
    "rslf_volume_upload_epis_f32", "rslf_volume_upload_epis_u8",
    "rslf_volume_upload_images_f32", "rslf_volume_upload_images_u8", "rslf_volume_pack_device_f32",
    "rslf_edge_confidence_pile", "rslf_depth_epi_pile", "rslf_selective_median",
    "rslf_depth1d_pile_run", "rslf_depth1d_pile_run_host", "rslf_last_scan_kernel_ms", "rslf_scan_time_total_ms",
    "rslf_edge_confidence_2d", "rslf_depth_epi_2d", "rslf_depth2d_run",
    "rslf_sweep_begin", "rslf_sweep_visit_scan", "rslf_sweep_visit_finish", "rslf_sweep_end",
    "rslf_depth_epi_scan", "rslf_depth1d_run",
    "rslf_f2c_level_dims", "rslf_downsample_epis_f32", "rslf_downsample_epis_u8", "rslf_device_max_f32", "rslf_f2c_tighten_bounds", "rslf_f2c_fuse",
    "rslf_depth2d_run_host", "rslf_fine_to_coarse_run_host", "rslf_kernel_columns_pile", "rslf_volume_upload_images_xf_f32", "rslf_volume_upload_images_xf_u8",
]


class RslfParams(C.Structure):
    """rslf_params == rslf::Depth1DParameters<T> (core.hpp:66-142)."""

    _fields_ = [
        ("edge_score_threshold", C.c_float),
        ("line_score_threshold", C.c_float),
        ("disp_score_threshold", C.c_float),
        ("raw_score_threshold", C.c_float),
        ("mean_shift_max_iter", C.c_float),
        ("edge_confidence_filter_size", C.c_int),
        ("edge_confidence_opening_type", C.c_int),
        ("edge_confidence_opening_size", C.c_int),
        ("median_filter_size", C.c_int),
        ("median_filter_epsilon", C.c_float),
        ("propagation_epsilon", C.c_float),
        ("slope_factor", C.c_float),
        ("cut_shadows", C.c_int),
        ("shadow_level", C.c_float),
        ("kernel_bandwidth", C.c_float),
        ("interpolation", C.c_int),
        ("use_disp_confidence_score", C.c_int),
    ]


class RslfVolumeDesc(C.Structure):
    _fields_ = [
        ("V", C.c_int), ("S", C.c_int), ("U", C.c_int), ("C", C.c_int),
        ("pitch", C.c_int),
        ("d_base", C.c_void_p),
        ("bytes", C.c_size_t),
        ("min_value", C.c_float),
        ("max_value", C.c_float),
    ]


class RslfStats(C.Structure):
    _fields_ = [
        ("pixels_scanned", C.c_int64),
        ("units", C.c_int64),
        ("scan_kernel", C.c_int),
        ("s_pad", C.c_int),
    ]


class RslfError(RuntimeError):
    def __init__(self, status: int, where: str):
        L = lib()
        msg = L.rslf_last_error().decode(errors="replace")
        what = L.rslf_status_string(status).decode()
        super().__init__("%s: %s (%d): %s" % (where, what, status, msg))
        self.status = status


def library_path() -> str:
    # RSLF_LIBRARY: load another build of the same ABI (A/B timing of kernel variants)
    return os.environ.get("RSLF_LIBRARY") or _build.SO


def lib():
    """Load librslf_hip.so (building it first if hipcc is at hand and it is stale)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # torch first: it brings its own copy of the HIP runtime (same soname as /opt/rocm's), and a process must not end up
    # with the library bound to one copy and torch to the other -- whichever loads first serves both, and with ours
    # first torch's streams and tensors and the library's context no longer see the same devices
    import torch  # noqa: F401
    path = library_path()
    if not os.path.exists(path):
        try:
            _build.build()
        except Exception as e:  # noqa: BLE001
            raise RuntimeError(
                "librslf_hip.so is missing and could not be built (%s). The HIP library IS the product: "
                "there is no CPU fallback. Run `python -m remotesensingproject_amd.build`." % e) from e
    L = C.CDLL(path)
    vp, ci, cf = C.c_void_p, C.c_int, C.c_float
    L.rslf_abi_version.restype = ci
    L.rslf_status_string.restype = C.c_char_p
    L.rslf_status_string.argtypes = [ci]
    L.rslf_last_error.restype = C.c_char_p
    L.rslf_device_count.restype = ci
    L.rslf_default_params.restype = None
    L.rslf_default_params.argtypes = [C.POINTER(RslfParams)]
    L.rslf_ctx_create.argtypes = [ci, C.POINTER(vp)]
    L.rslf_ctx_destroy.argtypes = [vp]
    L.rslf_ctx_set_stream.argtypes = [vp, vp]
    L.rslf_ctx_set_debug.argtypes = [vp, C.c_char_p, C.c_int]
    L.rslf_ctx_synchronize.argtypes = [vp]
    L.rslf_volume_create.argtypes = [vp, ci, ci, ci, ci, C.POINTER(vp)]
    L.rslf_volume_destroy.argtypes = [vp]
    L.rslf_volume_describe.argtypes = [vp, C.POINTER(RslfVolumeDesc)]
    L.rslf_volume_upload_epis_f32.argtypes = [vp, C.POINTER(vp), C.c_size_t, cf, C.POINTER(cf)]
    L.rslf_volume_upload_epis_u8.argtypes = [vp, C.POINTER(vp), C.c_size_t]
    L.rslf_volume_upload_images_f32.argtypes = [vp, C.POINTER(vp), C.c_size_t, cf, C.POINTER(cf)]
    L.rslf_volume_upload_images_u8.argtypes = [vp, C.POINTER(vp), C.c_size_t]
    L.rslf_volume_upload_images_xf_f32.argtypes = [vp, C.POINTER(vp), C.c_size_t, cf, C.POINTER(cf), ci, ci]
    L.rslf_volume_upload_images_xf_u8.argtypes = [vp, C.POINTER(vp), C.c_size_t, ci, ci]
    L.rslf_volume_pack_device_f32.argtypes = [vp, vp, cf, C.POINTER(cf)]
    L.rslf_edge_confidence_pile.argtypes = [vp, vp, ci, C.POINTER(RslfParams), vp, vp]
    L.rslf_depth_epi_pile.argtypes = [vp, vp, vp, vp, cf, cf, ci, ci, vp, vp, vp, vp, vp, C.POINTER(RslfParams),
                                      vp, vp, vp, vp, C.POINTER(RslfStats)]
    L.rslf_selective_median.argtypes = [vp, vp, vp, vp, ci, ci, vp, cf]
    L.rslf_depth_epi_scan.argtypes = [vp, vp, vp, vp, cf, cf, ci, ci, vp, vp, vp, vp, vp, C.POINTER(RslfParams),
                                      vp, vp, vp, C.POINTER(RslfStats)]
    L.rslf_depth1d_run.argtypes = [vp, vp, cf, cf, ci, ci, C.POINTER(RslfParams), vp, vp, vp, vp, vp, vp, vp,
                                   C.POINTER(RslfStats)]
    L.rslf_depth1d_pile_run.argtypes = [vp, vp, cf, cf, ci, ci, C.POINTER(RslfParams), vp, vp, vp, vp, vp, vp, vp, vp,
                                        C.POINTER(RslfStats)]
    L.rslf_depth1d_pile_run_host.argtypes = L.rslf_depth1d_pile_run.argtypes
    L.rslf_last_scan_kernel_ms.argtypes = [vp, C.POINTER(cf)]
    L.rslf_scan_time_total_ms.argtypes = [vp, C.POINTER(cf), C.POINTER(ci)]
    L.rslf_sweep_begin.argtypes = [vp, vp, vp, vp, ci, ci, ci]
    L.rslf_sweep_visit_scan.argtypes = [vp, vp, vp, vp, cf, cf, ci, ci, vp, vp, vp, vp, vp, C.POINTER(RslfParams)]
    L.rslf_sweep_visit_finish.argtypes = [vp, vp, ci, vp, vp, vp, vp, C.POINTER(RslfParams)]
    L.rslf_sweep_end.argtypes = [vp, ci, ci, C.POINTER(RslfStats)]
    L.rslf_multi_create.argtypes = [C.POINTER(ci), ci, C.POINTER(vp)]
    L.rslf_multi_destroy.argtypes = [vp]
    L.rslf_multi_device_count.argtypes = [vp]
    L.rslf_multi_set_chunk_rows.argtypes = [vp, ci]
    L.rslf_multi_peer_access.argtypes = [vp, ci, ci]
    L.rslf_debug_inject.argtypes = [C.c_char_p, ci]
    L.rslf_multi_depth1d_pile_f32.argtypes = [vp, C.POINTER(vp), C.c_size_t, ci, ci, ci, ci, cf, cf, cf, ci, ci, C.POINTER(RslfParams),
                                              vp, vp, vp, vp, vp, vp, vp, vp, C.POINTER(RslfStats), C.POINTER(cf)]
    L.rslf_multi_depth1d_pile_f32_dev.argtypes = [vp, C.POINTER(vp), C.c_size_t, ci, ci, ci, ci, cf, cf, cf, ci, ci, C.POINTER(RslfParams), ci,
                                                  vp, vp, vp, vp, vp, vp, vp, vp, C.POINTER(RslfStats), C.POINTER(cf)]
    L.rslf_multi_depth1d_pile_u8.argtypes = [vp, C.POINTER(vp), C.c_size_t, ci, ci, ci, ci, cf, cf, ci, ci, C.POINTER(RslfParams),
                                             vp, vp, vp, vp, vp, vp, vp, vp, C.POINTER(RslfStats)]
    L.rslf_multi_depth2d_run_f32.argtypes = [vp, C.POINTER(vp), C.c_size_t, ci, ci, ci, ci, cf, cf, cf, ci, C.POINTER(RslfParams),
                                             vp, vp, vp, vp, vp, vp, C.POINTER(RslfStats), C.POINTER(cf)]
    L.rslf_multi_depth2d_run_u8.argtypes = [vp, C.POINTER(vp), C.c_size_t, ci, ci, ci, ci, cf, cf, ci, C.POINTER(RslfParams),
                                            vp, vp, vp, vp, vp, vp, C.POINTER(RslfStats)]
    L.rslf_multi_fine_to_coarse_run_host.argtypes = [vp, C.POINTER(vp), ci, ci, ci, ci, ci, C.c_size_t, cf, cf, ci, cf, C.POINTER(RslfParams),
                                                     ci, ci, vp, vp, C.POINTER(ci), C.POINTER(RslfStats)]
    L.rslf_kernel_columns_pile.argtypes = [vp, vp, vp, vp, cf, cf, ci, ci, C.POINTER(RslfParams), vp, vp]
    L.rslf_depth2d_run_host.argtypes = [vp, vp, cf, cf, ci, C.POINTER(RslfParams), vp, vp, vp, vp, vp, C.POINTER(RslfStats)]
    L.rslf_fine_to_coarse_run_host.argtypes = [vp, C.POINTER(vp), ci, ci, ci, ci, ci, C.c_size_t, cf, cf, ci, cf, C.POINTER(RslfParams),
                                               ci, ci, vp, vp, C.POINTER(ci), C.POINTER(RslfStats)]
    L.rslf_f2c_level_dims.argtypes = [ci, ci, C.POINTER(ci), C.POINTER(ci)]
    L.rslf_downsample_epis_f32.argtypes = [vp, vp, ci, ci, ci, ci, vp]
    L.rslf_downsample_epis_u8.argtypes = [vp, vp, ci, ci, ci, ci, vp]
    L.rslf_device_max_f32.argtypes = [vp, vp, C.c_size_t, C.POINTER(cf)]
    L.rslf_f2c_tighten_bounds.argtypes = [vp, vp, vp, ci, ci, ci, vp, vp, ci, ci]
    L.rslf_f2c_fuse.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(ci), C.POINTER(ci), ci, ci, vp, vp]
    L.rslf_edge_confidence_2d.argtypes = [vp, vp, C.POINTER(RslfParams), vp, vp]
    L.rslf_depth_epi_2d.argtypes = [vp, vp, vp, vp, cf, cf, ci, vp, vp, vp, vp, vp, C.POINTER(RslfParams), vp, C.POINTER(RslfStats)]
    L.rslf_depth2d_run.argtypes = [vp, vp, cf, cf, ci, C.POINTER(RslfParams), vp, vp, vp, vp, vp, vp, C.POINTER(RslfStats)]
    for name in SYMBOLS:
        f = getattr(L, name)   # AttributeError here = the library does not export the ABI
        if f.restype is C.c_int and name not in ("rslf_abi_version", "rslf_device_count"):
            pass
    if L.rslf_abi_version() != ABI_VERSION:
        raise RuntimeError("librslf_hip.so has ABI %d, binding expects %d" % (L.rslf_abi_version(), ABI_VERSION))
    _LIB = L
    return L


def check(status: int, where: str) -> None:
    if status != 0:
        raise RslfError(status, where)


def default_params() -> RslfParams:
    p = RslfParams()
    lib().rslf_default_params(C.byref(p))
    return p
