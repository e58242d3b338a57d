/*
 * rslf_hip.h -- C-ABI of the MI355X-native EPI depth scan.
 *
 * This is the drop-in boundary for ONE path of RSLightFields
 * (14chanwa/remotesensingProject): everything below
 * rslf::Depth1DComputer_pile<T>::run() -- the scanline-parallel edge
 * confidence, the per-pixel x per-hypothesis EPI-slope scan with its mean
 * shift, and the selective median -- runs as hand-written HIP kernels for
 * gfx950.  The reference has no FFI of its own (it is one C++11/OpenCV/OpenMP
 * library); the seam is the template free-function signatures cited per entry
 * point below (paths relative to /root/reference/RSLightFields/).  The C++11
 * host class that keeps the reference's constructor/run() shape is
 * include/rslf_hip.hpp; the binding a maintainer would add on the reference
 * side is shown in INTEGRATION.md.
 *
 * Conventions
 *  - plain C, no C++/torch/OpenCV types; every function returns an rslf_status
 *    (0 = ok, <0 = error) and never throws across the boundary: every entry
 *    point is a function-try-block (std::bad_alloc -> RSLF_ERR_ALLOC, anything
 *    else -> RSLF_ERR_INTERNAL), and worker threads are joined on every path.
 *    (The reference returns void and has no error convention: SURVEY.md 8b.)
 *  - pointers named d_* are DEVICE pointers, h_* are HOST pointers; all
 *    buffers are caller-owned; planes are dense row-major [V][U].
 *  - work is enqueued on the context's stream (rslf_ctx_set_stream, a
 *    hipStream_t passed as void*); *_host entry points synchronise before
 *    returning, device entry points do not.
 *  - a context is not re-entrant; use one per host thread / stream.
 *  - there is NO CPU fallback behind this ABI.
 */
#ifndef RSLF_HIP_H
#define RSLF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RSLF_ABI_VERSION 6

typedef enum rslf_status {
    RSLF_OK = 0,
    RSLF_ERR_INVALID_ARG = -1,   /* null pointer, bad dimension, dim_d < 2, ... */
    RSLF_ERR_UNSUPPORTED = -2,   /* channel count other than 1 or 3, filter sizes the kernels do not cover */
    RSLF_ERR_HIP = -3,           /* a HIP runtime call failed; see rslf_last_error() */
    RSLF_ERR_NO_DEVICE = -4,     /* no gfx950 device visible */
    RSLF_ERR_ALLOC = -5,         /* device or host memory could not be had (hipMalloc, std::bad_alloc) */
    RSLF_ERR_INTERNAL = -6       /* a C++ exception was stopped at the boundary; its text is in rslf_last_error() */
} rslf_status;

/* Mirrors rslf::Depth1DParameters<T> -- include/rslf_depth_computation_core.hpp:66-142
 * (defaults :16-31, :74-99).  Same member meaning, par_ prefix dropped.  The two
 * polymorphic plug-ins (:108-109) become plain members: `interpolation` selects
 * between the reference's two interpolation classes (include/rslf_interpolation.hpp),
 * and the kernel is BandwidthKernel(h) (src/rslf_kernels.cpp:16-54), h = kernel_bandwidth. */
typedef struct rslf_params {
    float edge_score_threshold;         /* 0.02 */
    float line_score_threshold;         /* 0.02  (unused by this path, kept for 1:1 layout) */
    float disp_score_threshold;         /* 0.01  (only with use_disp_confidence_score) */
    float raw_score_threshold;          /* 0 */
    float mean_shift_max_iter;          /* 10; a float in the reference (:115) */
    int   edge_confidence_filter_size;  /* 9 */
    int   edge_confidence_opening_type; /* cv::MORPH_RECT 0 / MORPH_CROSS 1 / MORPH_ELLIPSE 2 (default) */
    int   edge_confidence_opening_size; /* 1 = no morphological opening (default); k in 2..31: cv::morphologyEx(MORPH_OPEN)
                                           with getStructuringElement(type, Size(k, k)) on the edge mask (core.hpp:759-768) */
    int   median_filter_size;           /* 5; any size >= 0: window half-width (size - 1) / 2, core.hpp:686 */
    float median_filter_epsilon;        /* 0.1 */
    float propagation_epsilon;          /* 0.1 (unused by this path) */
    float slope_factor;                 /* 1.0 */
    int   cut_shadows;                  /* 1 */
    float shadow_level;                 /* 0.05 * 1.73205080757 */
    float kernel_bandwidth;             /* 0.2 (_BANDWIDTH_KERNEL_PARAMETER, :26) */
    int   interpolation;                /* par_interpolation_class (:76-77, :108): RSLF_INTERP_*, default LINEAR */
    int   use_disp_confidence_score;    /* the reference's commented-out build switch _USE_DISP_CONFIDENCE_SCORE (:35): the 2-D
                                           sweep's propagation is gated by C_d > disp_score_threshold (:1097-1098) instead of the
                                           edge mask (:1102).  0 = the default build */
} rslf_params;

/* par_interpolation_class.
 * LINEAR            Interpolation1DLinear (include/rslf_interpolation.hpp:155-193), the reference default.
 * NEAREST           Interpolation1DNearestNeighbour as its scalar interpolate() states it (:80-92):
 *                   index (int)std::round(x), NaN outside [0, U-1].
 * NEAREST_AS_BUILT  what its interpolate_mat() -- the method the scan calls (core.hpp:561) -- executes:
 *                   the float index matrix is read through `indices.ptr<int>` (:118), so the "index" is
 *                   the BIT PATTERN of x, in range only for x = +0.  Kept so that a caller who selects
 *                   the reference's commented-out line (core.hpp:77) gets the reference's result bit for bit.
 * The two nearest modes run on the generic scan kernel only. */
#define RSLF_INTERP_LINEAR            0
#define RSLF_INTERP_NEAREST           1
#define RSLF_INTERP_NEAREST_AS_BUILT  2

typedef struct rslf_ctx rslf_ctx;       /* device, stream, scratch */
typedef struct rslf_volume rslf_volume; /* light-field slab resident in HBM */

typedef struct rslf_volume_desc {
    int V, S, U, C;       /* scanlines, views, columns, channels */
    int pitch;            /* floats per (v,s,c) row, multiple of 64, > U (zero padded) */
    void* d_base;         /* device address of element (v=0,s=0,c=0,u=0) */
    size_t bytes;         /* allocation size */
    float min_value;      /* over the normalised volume */
    float max_value;
} rslf_volume_desc;

/* Per-call statistics (nullable wherever taken). */
typedef struct rslf_stats {
    int64_t pixels_scanned;   /* pixels whose scan mask was set on entry (core.hpp:515-527) */
    int64_t units;            /* pixels_scanned * dim_d  = (pixel, hypothesis) pairs */
    int     scan_kernel;      /* which K2 variant ran: see RSLF_SCAN_* */
    int     s_pad;            /* register slots per lane of the register variant, else 0 */
} rslf_stats;

#define RSLF_SCAN_GENERIC   0  /* any S, C in {1,3}, any sign: re-gathers every mean-shift pass */
#define RSLF_SCAN_STREAM    2  /* volume >= 0, any S: a resident prefix of the samples (registers + LDS), the rest re-gathered every pass */
#define RSLF_SCAN_REG       1  /* volume >= 0 and C=1, S<=192 or C=3, S<=48: every sample held in registers */
#define RSLF_SCAN_CHIP      3  /* volume >= 0, C=3, S in [123, 220], dense launch with one hypothesis grid: one wave per SIMD, all but
                                  three samples of a unit on chip (VGPRs + AGPRs + LDS, a ladder of instantiations 8 views apart;
                                  201 views = BASELINE.json's c5 is the top rung), packed-fp32 passes */
#define RSLF_SCAN_REG_PX    4  /* RSLF_SCAN_REG's shapes on a packed (sparse) launch: a wave owns one pixel, its lanes the hypotheses */
#define RSLF_SCAN_STREAM_PX 5  /* RSLF_SCAN_STREAM's shapes on a packed (sparse) launch, the same way */

int         rslf_abi_version(void);
const char* rslf_status_string(int status);
const char* rslf_last_error(void);       /* thread-local text of the last failure */
int         rslf_device_count(void);
/* rslf::Depth1DParameters<T>::Depth1DParameters() -- core.hpp:74-99 */
void        rslf_default_params(rslf_params* p);

/* ---- context ---------------------------------------------------------- */
int rslf_ctx_create(int device, rslf_ctx** out);
int rslf_ctx_destroy(rslf_ctx* ctx);
int rslf_ctx_set_stream(rslf_ctx* ctx, void* hip_stream);   /* NULL = default stream */
int rslf_ctx_synchronize(rslf_ctx* ctx);
/* Test and tuning hooks of ONE context (nothing process-global, no environment variable is read):
 *   "force_scan"     0 automatic | 1 generic scan kernel | 2 streaming scan kernel (never the on-chip one)
 *   "force_groups"   0 automatic | 1..64 hypothesis groups per tile
 *   "force_packed"   -1 automatic | 0 row tiles | 1 one packed pixel list
 *   "px"             -1 automatic | 0 never | 1 whenever it can run: packed launches of a register or streaming kernel put a pixel's
 *                    HYPOTHESES in the lanes of a wave (k2_scan_reg_px, k2_scan_stream_px) instead of 64 pixels
 *   "row_split"      1 (default) packed launches of stream-class volumes scan the rows that hold >= 64 pixels as row tiles of
 *                    the packed list and leave the pixel-per-wave launch the rest | 0 the pixel-per-wave launch takes all
 *   "claim_skip"     1 (default) the 2-D sweep's claims skip views with nothing left to paint within reach | 0 off
 *   "stream_share"   63-pixel tiles sharing taps between lanes in the streaming kernel: 1 (default) where the samples gathered
 *                    again on every pass are at least a quarter of the views | 0 never | 2 always
 *   "stream_groups"  0 automatic | hypothesis groups per tile of the streaming kernel's dense launches
 *   "stream_lds_kib" dynamic LDS of one streaming workgroup, KiB (default 80)
 *   "time_all"       0 (default) | 1: every scan launch is bracketed by its own pair of HIP events, summed and reset by
 *                    rslf_scan_time_total_ms (the K2 time of a whole sweep or pyramid; each event costs ~5.6 us in the queue)
 * Results do not depend on these (the parity tests drive every combination; "claim_skip" on / off is compared plane by
 * plane in tests/test_gpu_sweep2d.py); speed does.  One qualification: the disparity confidence C_d = C_e * |max - mean of
 * the scores| takes the mean through a double sum whose ORDER differs between launch shapes (per wave in the row kernels,
 * a butterfly over lanes in k2_scan_reg_px / k2_scan_stream_px); sums of <= 4096 floats in [0, 1] are exact in a double
 * unless a score is below 2^-21, so C_d of two launch shapes is equal to within 1e-5 (the tolerance the reference's float
 * output is held to), not guaranteed bitwise.  Every other plane -- masks, arg-max indices, disparities, scores, r-bar,
 * C_e -- is bit-identical whatever the hooks say. */
int rslf_ctx_set_debug(rslf_ctx* ctx, const char* key, int value);
/* Fault injection for the tests of the error paths (process-wide, off unless armed): the next `count` visits of `site`
 * fail as if the runtime had -- "worker": a device worker of rslf_multi_* throws std::runtime_error; "thread_create":
 * std::thread cannot be started (the work then runs on the calling thread); "alloc": std::bad_alloc in a worker.
 * The entry point reports a status; nothing is left running, nothing leaks.  count = 0 disarms. */
int rslf_debug_inject(const char* site, int count);

/* ---- volume: replaces Depth1DComputer_pile's constructor --------------- */
/* include/rslf_depth_computation.hpp:425-477: the constructor copies the
 * caller's Vec<Mat> of V EPIs (each S x U, C channels interleaved) into float
 * Mats scaled to [0,1].  Here the copy lands in one HBM slab laid out
 * [V][S][pitch][C]: one zero-padded row of `pitch` pixels per (EPI, view), channels
 * interleaved like the reference's Mat rows, so the two lerp taps of a sample, all
 * channels, are 2*C consecutive floats. */
int rslf_volume_create(rslf_ctx* ctx, int V, int S, int U, int C, rslf_volume** out);
int rslf_volume_destroy(rslf_volume* vol);   /* waits for the device; contexts and volumes may be destroyed in either order */
int rslf_volume_describe(const rslf_volume* vol, rslf_volume_desc* out);

/* Host EPIs as the reference holds them: h_epis[v] -> S rows of U*C values,
 * row_stride_bytes apart (cv::Mat::step).  u8: x * float(1/255) (dc.hpp:470).
 * f32: x * float(1/double(scale)); scale < 0 => the max over all EPIs
 * (dc.hpp:442-460, :474).  scale_used is nullable. */
int rslf_volume_upload_epis_f32(rslf_volume* vol, const float* const* h_epis, size_t row_stride_bytes,
                                float epi_scale_factor, float* scale_used);
int rslf_volume_upload_epis_u8(rslf_volume* vol, const uint8_t* const* h_epis, size_t row_stride_bytes);
/* Host image stack as read from disk, h_imgs[s] -> V rows of U*C values: the
 * input of rslf::build_epis_from_imgs (src/rslf_io.cpp:194-227); the
 * [s][v][u] -> [v][s][u] transposition happens on the device. */
int rslf_volume_upload_images_f32(rslf_volume* vol, const float* const* h_imgs, size_t row_stride_bytes,
                                  float epi_scale_factor, float* scale_used);
int rslf_volume_upload_images_u8(rslf_volume* vol, const uint8_t* const* h_imgs, size_t row_stride_bytes);
/* The same with the two per-EPI options of rslf::build_epis_from_imgs (src/rslf_io.cpp:194-227, arguments `transpose`,
 * `rotate_180`): the EPI of scanline v is E[i][x] = h_imgs[i](v, x); transpose stores E^T -- the volume then has
 * S = image columns and U = number of images, h_imgs holds vol->U images of V rows x vol->S columns -- and
 * rotate_180 turns the (transposed) EPI by 180 degrees, i.e. reverses both of its axes. */
int rslf_volume_upload_images_xf_f32(rslf_volume* vol, const float* const* h_imgs, size_t row_stride_bytes,
                                     float epi_scale_factor, float* scale_used, int transpose, int rotate_180);
int rslf_volume_upload_images_xf_u8(rslf_volume* vol, const uint8_t* const* h_imgs, size_t row_stride_bytes,
                                    int transpose, int rotate_180);
/* Device-resident dense [V][S][U][C] float32 (already on this GPU). */
int rslf_volume_pack_device_f32(rslf_volume* vol, const float* d_vsuc, float epi_scale_factor, float* scale_used);

/* ---- the hot path, device pointers ------------------------------------ */
/* rslf::compute_1D_edge_confidence_pile -- core.hpp:279-287, impl :728-770
 * (per row :426-478).  d_Ce_vu [V][U] f32 is IN/OUT: the reference accumulates
 * into the caller's plane (src/rslf_depth_computation_core.cpp:10), pass zeros.
 * d_Ce_mask_vu [V][U] u8 out (0/255). */
int rslf_edge_confidence_pile(rslf_ctx* ctx, const rslf_volume* vol, int s, const rslf_params* p,
                              float* d_Ce_vu, uint8_t* d_Ce_mask_vu);

/* rslf::compute_1D_depth_epi_pile -- core.hpp:293-310, impl :772-893
 * (per EPI :480-661, selective median :663-718).
 *   d_dmin_vu/d_dmax_vu  [V][U] f32 per-pixel hypothesis range; both NULL =>
 *                        the scalars dmin/dmax (what dc.hpp:486-487 fills in)
 *   d_Ce_vu, d_Ce_mask_vu  in/out: a pixel whose best score is not above
 *                        raw_score_threshold gets C_e = 0, mask = 0 (:653-657)
 *   d_Cd_vu, d_depth_vu, d_rbar_vu  in/out, written only at scanned pixels;
 *                        d_rbar_vu is [V][U][C]; d_depth_vu is then REPLACED by
 *                        the selective median of itself (:881-892), 0 where the
 *                        edge mask is 0
 *   d_mask_vu            nullable in/out scan mask, AND-ed with the edge mask
 *                        in place (:510-511); NULL => edge mask alone (:513)
 *   d_idx_vu             nullable out, int32 argmax index d*, -1 if none
 *                        (not in the reference; the bit-exactness witness)
 *   d_score_vu           nullable out, score[d*]
 *   d_depth_raw_vu       nullable out, d_depth_vu before the median */
int rslf_depth_epi_pile(rslf_ctx* ctx, const rslf_volume* vol,
                        const float* d_dmin_vu, const float* d_dmax_vu, float dmin, float dmax,
                        int dim_d, int s_hat,
                        float* d_Ce_vu, uint8_t* d_Ce_mask_vu, float* d_Cd_vu, float* d_depth_vu,
                        float* d_rbar_vu, const rslf_params* p, uint8_t* d_mask_vu,
                        int32_t* d_idx_vu, float* d_score_vu, float* d_depth_raw_vu,
                        rslf_stats* stats);

/* rslf::compute_1D_depth_epi -- core.hpp:251-267, impl :480-661 -- for every EPI of the volume and
 * nothing else (no selective median): the first half of rslf_depth_epi_pile, same arguments, and all
 * the single-EPI class rslf::Depth1DComputer<T> needs. */
int rslf_depth_epi_scan(rslf_ctx* ctx, const rslf_volume* vol,
                        const float* d_dmin_vu, const float* d_dmax_vu, float dmin, float dmax,
                        int dim_d, int s_hat,
                        float* d_Ce_vu, uint8_t* d_Ce_mask_vu, float* d_Cd_vu, float* d_depth_vu,
                        float* d_rbar_vu, const rslf_params* p, uint8_t* d_mask_vu,
                        int32_t* d_idx_vu, float* d_score_vu, rslf_stats* stats);

/* rslf::Depth1DComputer<T>::run() -- include/rslf_depth_computation.hpp:325-371 with the
 * constructor's zero-filled outputs (:313-322): edge confidence and scan of each EPI on its own, no
 * median.  The reference's class holds one EPI; a volume of V EPIs is V independent instances. */
int rslf_depth1d_run(rslf_ctx* ctx, const rslf_volume* vol, float dmin, float dmax, int dim_d, int s_hat,
                     const rslf_params* p,
                     float* d_Ce_vu, uint8_t* d_Ce_mask_vu, float* d_Cd_vu, float* d_depth_vu,
                     float* d_rbar_vu, int32_t* d_idx_vu, float* d_score_vu, rslf_stats* stats);

/* The optional last argument of compute_1D_depth_epi / _pile (a_K_r_m_rbar_(v_)s_u, core.hpp:266, :309; written at
 * :647-651): for every pixel that received a disparity, K(r - rbar)[:, d*] of its winning hypothesis, i.e. the
 * kernel values of the last mean-shift pass over the S views.  d_idx_vu is the arg-max plane a scan produced
 * (rslf_depth_epi_scan / rslf_depth_epi_pile with the same volume, ranges, dim_d, s_hat and parameters);
 * d_K_vsu is [V][S][U] (the reference's Vec<Mat> of V mats S x U).  Pixels with idx < 0 are left untouched, as
 * the reference leaves them.  Costs 1/dim_d of a scan. */
int rslf_kernel_columns_pile(rslf_ctx* ctx, const rslf_volume* vol, const float* d_dmin_vu, const float* d_dmax_vu,
                             float dmin, float dmax, int dim_d, int s_hat, const rslf_params* p,
                             const int32_t* d_idx_vu, float* d_K_vsu);

/* rslf::selective_median_filter -- core.hpp:366-375, impl :663-718.
 * d_dst_vu must not alias d_src_vu; it is fully written (0 where mask is 0).
 * `size` = a_size: the window is width = (size - 1) / 2 pixels either side (:686), so an even size is the next smaller
 * odd window and 0 the 1 x 1 window; any size >= 0 runs (the report documents 11, report/rs_report.tex:388). */
int rslf_selective_median(rslf_ctx* ctx, const rslf_volume* vol, const float* d_src_vu, float* d_dst_vu,
                          int s_hat, int size, const uint8_t* d_mask_vu, float epsilon);

/* rslf::Depth1DComputer_pile<T>::run() -- include/rslf_depth_computation.hpp:513-565
 * with the output allocation of the constructor (:486-510): zero-fills the
 * outputs (the reference leaves C_e and C_d uninitialised, :501-504), s_hat < 0
 * or > S-1 => floor(S/2) (:490-498), then the two calls above, no scan mask. */
int rslf_depth1d_pile_run(rslf_ctx* ctx, const rslf_volume* vol, float dmin, float dmax, int dim_d, int s_hat,
                          const rslf_params* p,
                          float* d_Ce_vu, uint8_t* d_Ce_mask_vu, float* d_Cd_vu, float* d_depth_vu,
                          float* d_rbar_vu, int32_t* d_idx_vu, float* d_score_vu, float* d_depth_raw_vu,
                          rslf_stats* stats);

/* ---- the hot path, host pointers (cv::Mat::data in / out) -------------- */
/* Same as rslf_depth1d_pile_run with host result planes: runs on the device,
 * copies back, synchronises.  Any output may be NULL. */
int rslf_depth1d_pile_run_host(rslf_ctx* ctx, const rslf_volume* vol, float dmin, float dmax, int dim_d, int s_hat,
                               const rslf_params* p,
                               float* h_Ce_vu, uint8_t* h_Ce_mask_vu, float* h_Cd_vu, float* h_depth_vu,
                               float* h_rbar_vu, int32_t* h_idx_vu, float* h_score_vu, float* h_depth_raw_vu,
                               rslf_stats* stats);

/* ---- the hot path from host buffers, pipelined, over one or several devices --------------------------------- */
/* Depth1DComputer_pile<T>'s constructor + run() + result Mats (include/rslf_depth_computation.hpp:425-565) in ONE
 * call on host buffers: h_epis[v] is EPI v (S rows of U pixels, C channels interleaved, rows row_stride_bytes apart,
 * 0 = dense), the h_* planes are the caller's [V][U] result planes (any may be NULL).
 *
 * The V EPIs are cut into one contiguous block of scanlines per device of the rslf_multi (K1 and K2 are
 * independent per scanline, core.hpp:743-757 / :799-854; the median reads +-2 rows, core.hpp:686 -- those rows are
 * recomputed, not exchanged), every block into chunks, and per device the upload of chunk k+1, the kernels of chunk k
 * and the download of chunk k-1 overlap.  Each device's rows land directly at their offset in the caller's planes:
 * there is no collective.  The float input's default scale (epi_scale_factor < 0: the maximum over ALL EPIs,
 * dc.hpp:442-460) is taken once over the whole input, never per block.  Results are bit-identical to
 * rslf_volume_upload_epis_* + rslf_depth1d_pile_run_host on one device.
 *
 * devices may name the same GPU more than once (two workers on one GPU); NULL / 0 = device 0 alone. */
typedef struct rslf_multi rslf_multi;
int rslf_multi_create(const int* devices, int n_devices, rslf_multi** out);
int rslf_multi_destroy(rslf_multi* m);
int rslf_multi_device_count(const rslf_multi* m);
/* 1 if worker `from` can map worker `to`'s memory (hipDeviceCanAccessPeer) and the access has been enabled -- copies
 * between them then go device to device over xGMI; 0 if the copies stage through the host (still correct); 1 for two
 * workers on one GPU; <0 on a bad index.  rslf_multi_create enables peer access for every pair that allows it. */
int rslf_multi_peer_access(const rslf_multi* m, int from, int to);
int rslf_multi_set_chunk_rows(rslf_multi* m, int rows);   /* scanlines per chunk; 0 = automatic (about 8 chunks per device) */
int rslf_multi_depth1d_pile_f32(rslf_multi* m, const float* const* h_epis, size_t row_stride_bytes, int V, int S, int U, int C,
                                float epi_scale_factor, float dmin, float dmax, int dim_d, int s_hat, const rslf_params* p,
                                float* h_Ce_vu, uint8_t* h_Ce_mask_vu, float* h_Cd_vu, float* h_depth_vu, float* h_rbar_vu,
                                int32_t* h_idx_vu, float* h_score_vu, float* h_depth_raw_vu, rslf_stats* stats, float* scale_used);
/* Device-out form: the result planes live on device `out_device` (any device, one of the rslf_multi's or not); every
 * worker copies its rows to their place there with hipMemcpyPeerAsync -- point-to-point over xGMI, no collective and no
 * staging on the host.  Returns when the planes are complete. */
int rslf_multi_depth1d_pile_f32_dev(rslf_multi* m, const float* const* h_epis, size_t row_stride_bytes, int V, int S, int U, int C,
                                    float epi_scale_factor, float dmin, float dmax, int dim_d, int s_hat, const rslf_params* p,
                                    int out_device, float* d_Ce_vu, uint8_t* d_Ce_mask_vu, float* d_Cd_vu, float* d_depth_vu,
                                    float* d_rbar_vu, int32_t* d_idx_vu, float* d_score_vu, float* d_depth_raw_vu, rslf_stats* stats,
                                    float* scale_used);
int rslf_multi_depth1d_pile_u8(rslf_multi* m, const uint8_t* const* h_epis, size_t row_stride_bytes, int V, int S, int U, int C,
                               float dmin, float dmax, int dim_d, int s_hat, const rslf_params* p,
                               float* h_Ce_vu, uint8_t* h_Ce_mask_vu, float* h_Cd_vu, float* h_depth_vu, float* h_rbar_vu,
                               int32_t* h_idx_vu, float* h_score_vu, float* h_depth_raw_vu, rslf_stats* stats);

/* ---- "next" row: the 2-D sweep over all views ------ */
/* Planes here are [S][V][U] (the reference's Vec<Mat> indexed by s, dc.hpp:208-215),
 * d_rbar_svu is [S][V][U][C]. */

/* rslf::compute_2D_edge_confidence -- core.hpp:323-330, impl :901-931: the pile edge
 * confidence for every view.  d_Ce_svu in/out (accumulates, pass zeros). */
int rslf_edge_confidence_2d(rslf_ctx* ctx, const rslf_volume* vol, const rslf_params* p,
                            float* d_Ce_svu, uint8_t* d_Ce_mask_svu);

/* rslf::compute_2D_depth_epi -- core.hpp:336-351, impl :933-1133 (default build: neither
 * _USE_DISP_CONFIDENCE_SCORE nor _USE_LINE_CONFIDENCE_SCORE, so propagation is gated by the edge
 * mask, :1099-1103).  Views are visited s_hat = floor(S/2), s_hat+1, s_hat-1, ... (:981-990); each
 * visit runs rslf_depth_epi_pile on the running mask (:1012-1028) and then propagates the
 * median-filtered disparities along their EPI lines into every view (:1088-1129).
 *   d_dmin_svu/d_dmax_svu  [S][V][U], both NULL => scalars dmin/dmax (dc.hpp:718-722)
 *   d_scan_mask_svu        nullable out: the running masks after the last visit
 *   stats->units           sums the units of all visits */
int rslf_depth_epi_2d(rslf_ctx* ctx, const rslf_volume* vol, const float* d_dmin_svu, const float* d_dmax_svu,
                      float dmin, float dmax, int dim_d,
                      float* d_Ce_svu, uint8_t* d_Ce_mask_svu, float* d_Cd_svu, float* d_depth_svu,
                      float* d_rbar_svu, const rslf_params* p, uint8_t* d_scan_mask_svu, rslf_stats* stats);

/* The sweep one visit at a time -- for callers that shard the scanlines of a sweep over devices (DESIGN.md
 * "Multi-GPU").  Scan and propagation are local to a scanline (core.hpp:1012-1028, :1088-1129) but the selective median
 * of a visit reads +-(size-1)/2 scanlines of the visited view's raw disparities and edge mask (core.hpp:686), which a
 * neighbouring device has just written: between rslf_sweep_visit_scan and rslf_sweep_visit_finish the caller brings
 * those rows of d_depth_svu[s_hat] and d_Ce_mask_svu[s_hat] up to date (a 2-row exchange with each neighbour).
 *   begin:  scanlines [v_lo, v_hi) of the volume are this device's own; the others are halo rows -- their running mask is
 *           cleared, so nothing is ever scanned or painted on them here.  (0, V) = an unsharded sweep.
 *   visits: in the reference's order, centre view outwards (core.hpp:981-990).
 *   end:    restores the context (call it with ok = 0 after a failed step); stats count the own scanlines only.
 * rslf_depth_epi_2d is exactly begin + (scan, finish) per view + end on (0, V). */
int rslf_sweep_begin(rslf_ctx* ctx, const rslf_volume* vol, const uint8_t* d_Ce_mask_svu, uint8_t* d_scan_mask_svu, int dim_d,
                     int v_lo, int v_hi);
int rslf_sweep_visit_scan(rslf_ctx* ctx, const rslf_volume* vol, const float* d_dmin_svu, const float* d_dmax_svu, float dmin,
                          float dmax, int dim_d, int s_hat, float* d_Ce_svu, uint8_t* d_Ce_mask_svu, float* d_Cd_svu,
                          float* d_depth_svu, float* d_rbar_svu, const rslf_params* p);
int rslf_sweep_visit_finish(rslf_ctx* ctx, const rslf_volume* vol, int s_hat, uint8_t* d_Ce_mask_svu, float* d_Cd_svu,
                            float* d_depth_svu, float* d_rbar_svu, const rslf_params* p);
int rslf_sweep_end(rslf_ctx* ctx, int ok, int dim_d, rslf_stats* stats);

/* rslf::Depth2DComputer<T>::run() -- include/rslf_depth_computation.hpp:748-805 with the
 * constructor's output allocation (:718-750): zero-fills the outputs, then the two calls above. */
int rslf_depth2d_run(rslf_ctx* ctx, const rslf_volume* vol, float dmin, float dmax, int dim_d, const rslf_params* p,
                     float* d_Ce_svu, uint8_t* d_Ce_mask_svu, float* d_Cd_svu, float* d_depth_svu,
                     float* d_rbar_svu, uint8_t* d_scan_mask_svu, rslf_stats* stats);

/* ---- "next" row: fine-to-coarse ---------------------- */
/* rslf::FineToCoarse<T> (include/rslf_fine_to_coarse.hpp:26-81) is a host-side loop over pyramid
 * levels, each a Depth2DComputer (rslf_depth2d_run / rslf_depth_epi_2d above).  These are the pieces
 * between the levels; the loop itself lives in the host wrapper (depth.py: FineToCoarse).  Raw
 * (un-normalised) volumes here are dense float32 [V][S][U][C] on the device; uchar light fields keep uchar
 * arithmetic through the pyramid (rslf_downsample_epis_u8), as the reference's CV_8U Mats do. */

/* Level dimensions of cv::resize(0.5, 0.5): cvRound(V/2), cvRound(U/2) (ties to even). */
int rslf_f2c_level_dims(int V, int U, int* V2, int* U2);
/* rslf::downsample_EPIs -- src/rslf_fine_to_coarse_core.cpp:14-60: per view, cv::GaussianBlur(7x7,
 * sigma 0, BORDER_REFLECT) then cv::resize(0.5, 0.5, INTER_LINEAR).  d_out_vsuc is [V2][S][U2][C]. */
int rslf_downsample_epis_f32(rslf_ctx* ctx, const float* d_in_vsuc, int V, int S, int U, int C, float* d_out_vsuc);
/* The same for CV_8U light fields, in uchar arithmetic as the reference runs it (its Mats keep the input's type):
 * d_in / d_out hold uchar levels 0..255 as float32.  GaussianBlur on 8U = exact integer convolution (the taps are
 * exact in 8 fractional bits) rounded half up once -- the result of both 8U paths of OpenCV 3.4 (<= 3.4.0 8-bit
 * fixed-point filter, >= 3.4.1 ufixedpoint16); resize = INTER_AREA's (sum + 2) >> 2. */
int rslf_downsample_epis_u8(rslf_ctx* ctx, const float* d_in_vsuc, int V, int S, int U, int C, float* d_out_vsuc);
/* max over a device buffer: the per-level epi_scale_factor of Depth2DComputer's constructor
 * (include/rslf_depth_computation.hpp:671-690).  Synchronises. */
int rslf_device_max_f32(rslf_ctx* ctx, const float* d_values, size_t n, float* h_max);   /* returns the value: waits */
/* The three helpers below enqueue on the context's stream and return, like every device entry point. */
/* FineToCoarse::run(), bound tightening -- include/rslf_fine_to_coarse.hpp:171-299: d_dmin/d_dmax of
 * the coarser level ([S][V_down][U_down], in/out) from the finer level's disparities and validity. */
int rslf_f2c_tighten_bounds(rslf_ctx* ctx, const float* d_depth_up_svu, const uint8_t* d_valid_up_svu, int S, int V_up,
                            int U_up, float* d_dmin_down_svu, float* d_dmax_down_svu, int V_down, int U_down);
/* rslf::fuse_disp_maps -- src/rslf_fine_to_coarse_core.cpp:69-135, all views at once.
 * d_disp[p] / d_valid[p]: level p planes [S][Vp[p]][Up[p]] (p = 0 finest); outputs [S][Vp[0]][Up[0]]. */
int rslf_f2c_fuse(rslf_ctx* ctx, const float* const* d_disp, const uint8_t* const* d_valid, const int* Vp, const int* Up,
                  int P, int S, float* d_out_map_svu, uint8_t* d_out_valid_svu);

/* ---- the rows around the path, host pointers --------------------------- */
/* rslf::Depth2DComputer<T>::run() with host result planes ([S][V][U], rbar [S][V][U][C]; any may be NULL). */
int rslf_depth2d_run_host(rslf_ctx* ctx, const rslf_volume* vol, float dmin, float dmax, int dim_d, const rslf_params* p,
                          float* h_Ce_svu, uint8_t* h_Ce_mask_svu, float* h_Cd_svu, float* h_depth_svu,
                          float* h_rbar_svu, rslf_stats* stats);

/* rslf::FineToCoarse<T>: constructor + run() + get_results() -- include/rslf_fine_to_coarse.hpp:103-324 --
 * from the reference's Vec<Mat> (h_epis[v] -> S rows of U*C values, row_stride_bytes apart; is_u8
 * selects uchar input).  The whole pyramid loop runs inside the library.
 *   h_out_map_svu / h_out_valid_svu  [S][V][U] fused disparity map and validity at the finest scale
 *   n_levels (nullable)              pyramid depth that was built
 *   stats->units                     sums every visit of every level */
int rslf_fine_to_coarse_run_host(rslf_ctx* ctx, const void* const* h_epis, int is_u8, int V, int S, int U, int C,
                                 size_t row_stride_bytes, float d_min, float d_max, int dim_d, float epi_scale_factor,
                                 const rslf_params* p, int max_pyr_depth, int accept_all_last_scale,
                                 float* h_out_map_svu, uint8_t* h_out_valid_svu, int* n_levels, rslf_stats* stats);

/* Depth2DComputer<T>'s constructor + run() + getters (rslf_depth_computation.hpp:651-805) over the context's devices,
 * host EPIs in, host [S][V][U] planes out.  The 2-D sweep is cut into one block of scanlines per device; every visit
 * exchanges the neighbours' boundary rows (median reach) of the visited view's raw disparities and edge mask by peer
 * copy between the scan and the median -- the one exchange step of the path.  One host thread queues the work of all
 * devices; events keep the order.  Planes may be NULL (not wanted).  Bit-identical to rslf_depth2d_run on one volume. */
int rslf_multi_depth2d_run_f32(rslf_multi* m, const float* const* h_epis, size_t row_stride_bytes, int V, int S, int U, int C,
                               float epi_scale_factor, float dmin, float dmax, int dim_d, const rslf_params* p,
                               float* h_Ce_svu, uint8_t* h_Ce_mask_svu, float* h_Cd_svu, float* h_depth_svu,
                               float* h_rbar_svu, uint8_t* h_scan_mask_svu, rslf_stats* stats, float* scale_used);
int rslf_multi_depth2d_run_u8(rslf_multi* m, const uint8_t* const* h_epis, size_t row_stride_bytes, int V, int S, int U, int C,
                              float dmin, float dmax, int dim_d, const rslf_params* p, float* h_Ce_svu,
                              uint8_t* h_Ce_mask_svu, float* h_Cd_svu, float* h_depth_svu, float* h_rbar_svu,
                              uint8_t* h_scan_mask_svu, rslf_stats* stats);

/* FineToCoarse<T>'s constructor + run() + get_results() (rslf_fine_to_coarse.hpp:103-324) over the context's devices:
 * every level's 2-D sweep runs sharded as in rslf_multi_depth2d_run_* (with the level's tightened per-pixel ranges); the
 * pyramid, the bound tightening and the fusion run on the first device, which also holds every level's planes: the
 * other devices fetch their rows from it and return their results to it by peer copies.
 * Arguments and results as rslf_fine_to_coarse_run_host; bit-identical to it. */
int rslf_multi_fine_to_coarse_run_host(rslf_multi* m, const void* const* h_epis, int is_u8, int V, int S, int U, int C,
                                       size_t row_stride_bytes, float d_min, float d_max, int dim_d, float epi_scale_factor,
                                       const rslf_params* p, int max_pyr_depth, int accept_all_last_scale,
                                       float* h_out_map_svu, uint8_t* h_out_valid_svu, int* n_levels, rslf_stats* stats);

/* ---- measurement ------------------------------------------------------ */
/* Duration in milliseconds of the last scan-kernel launch (K2) of this
 * context, from HIP events recorded on the context's stream around that
 * launch.  Blocks until the launch has finished.  Within a 2-D sweep only
 * the first (dense, centre-view) visit is timed: an event is a packet of
 * its own in the queue, and two per sparse visit cost a tenth of the visit. */
int rslf_last_scan_kernel_ms(rslf_ctx* ctx, float* ms);
/* With rslf_ctx_set_debug(ctx, "time_all", 1): the SUM of the durations of every scan launch sequence queued on this
 * context since the last call (or since the hook was set) and their number; blocks until they have finished, then starts
 * a new sum.  For the roofline of the rows around the path: a 2-D sweep is one dense and S - 1 sparse scan launches, a
 * fine-to-coarse run that per pyramid level (core.hpp:993-1028). */
int rslf_scan_time_total_ms(rslf_ctx* ctx, float* ms, int* launches);

#ifdef __cplusplus
}
#endif
#endif /* RSLF_HIP_H */
