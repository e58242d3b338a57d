// C-ABI of the gfx950 EPI depth scan (include/rslf_hip.h) -- host side of
// librslf_hip.so: contexts, the HBM slab, kernel selection and launches.
// No OpenCV, no torch, no CPU compute path.
#include "../../include/rslf_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <deque>
#include <new>
#include <string>
#include <vector>

#include "k1_edge.hpp"
#include "k2_scan.hpp"
#include "k3_median.hpp"
#include "k4_propagate.hpp"
#include "k5_f2c.hpp"

using namespace rslf;

// ---- errors ---------------------------------------------------------------

static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                               \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            return fail(RSLF_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// ---- objects --------------------------------------------------------------

// Streaming scan kernel: two workgroups per CU (two waves per SIMD) share the CU's 160 KiB of LDS.  The kernel has no
// static LDS (the merge of the four waves' results uses the head of their dynamic regions, EpilogueBlock), so each
// workgroup gets 80 KiB of dynamic LDS.
static constexpr size_t kStreamLdsBytes = (size_t)80 << 10;
constexpr int kSweepGroups = 32;    // workgroups sharing a packed tile's hypotheses on a sweep's sparse visits
// Dense launches of the streaming kernel: this many workgroups share one tile's hypotheses.  The workgroups an
// XCD runs together then sit on two or three tiles instead of a whole scanline, and what they gather from
// stays inside the XCD's 4 MiB L2 (k2_scan.hpp)
constexpr int kStreamGroups = 16;
static constexpr size_t kPartialBudget = (size_t)256 << 20;   // bytes of (tile, group, lane) records per grouped scan launch

struct rslf_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    // scratch, grown on demand (never inside a timed launch sequence after the first call)
    int* list = nullptr;
    int* count = nullptr;
    float* depth_tmp = nullptr;
    size_t plane_cap = 0;   // pixels list/depth_tmp can hold
    int count_cap = 0;
    unsigned long long* total = nullptr;   // device counter
    float* partial = nullptr;              // pack min/max partials
    size_t partial_cap = 0;
    float* minmax = nullptr;               // device [2]
    void* staging = nullptr;
    size_t staging_cap = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool ev_valid = false;
    int last_spad = 0;   // register-scan slot count of the last K2 launch, 0 = none
    int last_kernel = 0; // RSLF_SCAN_* of the last K2 launch
    int num_cus = 0;           // compute units of the device (how many workgroups a launch needs to fill it)
    bool keep_total = false;   // the 2-D sweep sums the scanned pixels of all its visits
    int scan_groups = 1;       // hypothesis groups per tile for the next scan launches (the 2-D sweep raises it)
    bool scan_packed = false;  // next scan launches use one packed pixel list (sparse visits of the 2-D sweep)
    // test / tuning hooks (rslf_ctx_set_debug), per context: 0 / -1 = automatic
    int force_scan = 0;        // 1 generic kernel, 2 streaming kernel
    int force_groups = 0;      // hypothesis groups per tile
    int force_packed = -1;     // 0 / 1
    int stream_groups = 0;     // streaming kernel, dense launches: hypothesis groups per tile (0 = kStreamGroups)
    bool stream_share = true;  // streaming kernel: 63-pixel row tiles whose tail shares taps between neighbouring lanes
    size_t stream_lds_bytes = kStreamLdsBytes;   // dynamic LDS of one streaming workgroup
    bool stream_attr_set = false;
    Partial* scan_partial = nullptr;   // [tile][group][64] records of grouped scan launches
    size_t partial_rec_cap = 0;
    int* scan_ticket = nullptr;        // [tile] of the same launches: which group merges the tile (zero between launches)
    size_t ticket_cap = 0;
    bool packed_n_clean = false;       // the packed list's length is already 0 (the sweep's apply pass resets it)
    int precompacted = 0;              // the next scan's pixel lists and total are already in place: 1 = per-row lists (K1 +
                                       // compaction in one launch), 2 = the packed list (a sweep's apply pass made it)
    int sweep_expect = -1;             // the view the sweep visits next (core.hpp:981-990), -1 once all are done
    bool sweep_open = false;           // between rslf_sweep_begin and rslf_sweep_end
    bool sweep_first = true;           // the next visit is the sweep's first (dense) one
    uint8_t* sweep_mask_run = nullptr; // the running masks [S][V][U] of the open sweep
    // 2-D sweep scratch
    int* winner = nullptr;        // [S][V][U]
    uint8_t* dirty = nullptr;     // [S][V][ceil(U/256)]: segments of the winner rows that hold a claim (all 0 between visits)
    size_t dirty_cap = 0;
    uint8_t* sweep_mask = nullptr;
    float* filtered = nullptr;    // [V][U] median of the visited view, the propagation's source
    size_t sweep_cap = 0;         // entries winner / sweep_mask can hold (S*V*U)
    size_t sweep_plane_cap = 0;   // floats `filtered` can hold (V*U)
    // grow-only scratch of the once-per-level helpers (pyramid, tightening, fusion): reused across calls, so
    // these helpers neither allocate nor free -- and so never force a device-wide synchronisation
    static constexpr int kHelperSlots = 4;
    void* helper[kHelperSlots] = {nullptr, nullptr, nullptr, nullptr};
    size_t helper_cap[kHelperSlots] = {0, 0, 0, 0};
};

struct rslf_volume {
    rslf_ctx* ctx = nullptr;
    int device = 0;   // kept here too: a volume may be destroyed after its context
    int V = 0, S = 0, U = 0, C = 0, pitch = 0;
    float* base = nullptr;
    size_t bytes = 0;
    float min_value = 0.0f, max_value = 0.0f;
    bool filled = false;
};

static VolView view_of(const rslf_volume* vol)
{
    VolView w;
    w.base = vol->base;
    w.V = vol->V;
    w.S = vol->S;
    w.U = vol->U;
    w.C = vol->C;
    w.pitch = vol->pitch;
    w.stride_s = (long long)vol->pitch * vol->C;
    w.stride_v = (long long)vol->S * w.stride_s;
    return w;
}

static int ensure_plane_scratch(rslf_ctx* ctx, int V, int U)
{
    const size_t n = (size_t)V * U;
    if (n > ctx->plane_cap) {
        if (ctx->list)
            HIP_TRY(hipFree(ctx->list));
        if (ctx->depth_tmp)
            HIP_TRY(hipFree(ctx->depth_tmp));
        ctx->list = nullptr;
        ctx->depth_tmp = nullptr;
        ctx->plane_cap = 0;
        HIP_TRY(hipMalloc(&ctx->list, n * sizeof(int)));
        HIP_TRY(hipMalloc(&ctx->depth_tmp, n * sizeof(float)));
        ctx->plane_cap = n;
    }
    if (V > ctx->count_cap) {
        if (ctx->count)
            HIP_TRY(hipFree(ctx->count));
        ctx->count = nullptr;
        ctx->count_cap = 0;
        HIP_TRY(hipMalloc(&ctx->count, (size_t)V * sizeof(int)));
        ctx->count_cap = V;
    }
    return RSLF_OK;
}

// Records and tickets of grouped scan launches (k2_scan.hpp): grow-only, so a context allocates them once.
static int ensure_group_scratch(rslf_ctx* ctx, size_t recs, size_t tiles)
{
    if (recs > ctx->partial_rec_cap) {
        HIP_TRY(hipFree(ctx->scan_partial));
        ctx->scan_partial = nullptr;
        ctx->partial_rec_cap = 0;
        HIP_TRY(hipMalloc(&ctx->scan_partial, recs * sizeof(Partial)));
        ctx->partial_rec_cap = recs;
    }
    if (tiles > ctx->ticket_cap) {
        HIP_TRY(hipFree(ctx->scan_ticket));
        ctx->scan_ticket = nullptr;
        ctx->ticket_cap = 0;
        HIP_TRY(hipMalloc(&ctx->scan_ticket, tiles * sizeof(int)));
        HIP_TRY(hipMemsetAsync(ctx->scan_ticket, 0, tiles * sizeof(int), ctx->stream));   // the kernels leave it at zero
        ctx->ticket_cap = tiles;
    }
    return RSLF_OK;
}

// ---- misc -----------------------------------------------------------------

extern "C" int rslf_abi_version(void) { return RSLF_ABI_VERSION; }

extern "C" const char* rslf_status_string(int status)
{
    switch (status) {
    case RSLF_OK: return "ok";
    case RSLF_ERR_INVALID_ARG: return "invalid argument";
    case RSLF_ERR_UNSUPPORTED: return "unsupported configuration";
    case RSLF_ERR_HIP: return "HIP runtime error";
    case RSLF_ERR_NO_DEVICE: return "no gfx950 device";
    case RSLF_ERR_ALLOC: return "allocation failed";
    default: return "unknown status";
    }
}

extern "C" const char* rslf_last_error(void) { return g_err; }

extern "C" int rslf_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}

extern "C" void rslf_default_params(rslf_params* p)
{
    if (!p)
        return;
    // include/rslf_depth_computation_core.hpp:16-31, :74-99
    p->edge_score_threshold = (float)0.02;
    p->line_score_threshold = (float)0.02;
    p->disp_score_threshold = (float)0.01;
    p->raw_score_threshold = (float)0;
    p->mean_shift_max_iter = (float)10;
    p->edge_confidence_filter_size = 9;
    p->edge_confidence_opening_type = 2;
    p->edge_confidence_opening_size = 1;
    p->median_filter_size = 5;
    p->median_filter_epsilon = (float)0.1;
    p->propagation_epsilon = (float)0.1;
    p->slope_factor = (float)1.0;
    p->cut_shadows = 1;
    p->shadow_level = (float)(0.05 * 1.73205080757);
    p->kernel_bandwidth = (float)0.2;
    p->interpolation = RSLF_INTERP_LINEAR;   // core.hpp:76
    p->use_disp_confidence_score = 0;        // core.hpp:35: commented out in the reference
}

static ScanConsts make_scan_consts(const rslf_params* p)
{
    ScanConsts k;
    k.slope = p->slope_factor;
    const float h = p->kernel_bandwidth;
    const float hh = h * h;
    k.inv_h2 = (float)(1.0 / (double)hh);   // include/rslf_kernels.hpp:43
    k.k1 = 3.0f * k.inv_h2;                 // src/rslf_kernels.cpp:21
    k.raw_thr = p->raw_score_threshold;
    int n = 0;
    while ((float)n < p->mean_shift_max_iter && n < (1 << 20))   // core.hpp:584, float bound
        n++;
    k.n_iter = n;
    k.interp = p->interpolation;
    return k;
}

static int check_params(const rslf_params* p)
{
    if (!p)
        return fail(RSLF_ERR_INVALID_ARG, "params is NULL");
    if (p->edge_confidence_opening_size > 31)
        return fail(RSLF_ERR_UNSUPPORTED, "edge_confidence_opening_size=%d: structuring elements up to 31 x 31", p->edge_confidence_opening_size);
    if (p->edge_confidence_opening_size > 1 && (p->edge_confidence_opening_type < 0 || p->edge_confidence_opening_type > 2))
        return fail(RSLF_ERR_INVALID_ARG, "edge_confidence_opening_type=%d is not cv::MORPH_RECT (0), MORPH_CROSS (1) or MORPH_ELLIPSE (2)",
                    p->edge_confidence_opening_type);
    if (p->edge_confidence_filter_size < 1 || (p->edge_confidence_filter_size & 1) == 0)
        return fail(RSLF_ERR_INVALID_ARG, "edge_confidence_filter_size must be odd and >= 1");
    if (p->median_filter_size < 1 || (p->median_filter_size & 1) == 0 || p->median_filter_size > kMedianMaxSize)
        return fail(RSLF_ERR_UNSUPPORTED, "median_filter_size must be odd and <= %d", kMedianMaxSize);
    if (!(p->kernel_bandwidth > 0.0f))
        return fail(RSLF_ERR_INVALID_ARG, "kernel_bandwidth must be > 0");
    if (!(p->mean_shift_max_iter > 0.0f))
        return fail(RSLF_ERR_INVALID_ARG, "mean_shift_max_iter must be > 0");
    if (p->interpolation < RSLF_INTERP_LINEAR || p->interpolation > RSLF_INTERP_NEAREST_AS_BUILT)
        return fail(RSLF_ERR_INVALID_ARG, "interpolation=%d is not one of RSLF_INTERP_*", p->interpolation);
    return RSLF_OK;
}

// ---- context --------------------------------------------------------------

extern "C" int rslf_ctx_create(int device, rslf_ctx** out)
{
    if (!out)
        return fail(RSLF_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(RSLF_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= n)
        return fail(RSLF_ERR_INVALID_ARG, "device %d out of range (0..%d)", device, n - 1);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(RSLF_ERR_NO_DEVICE, "device %d is %s; this library holds gfx950 code objects only", device, prop.gcnArchName);
    HIP_TRY(hipSetDevice(device));
    rslf_ctx* ctx = new (std::nothrow) rslf_ctx();
    if (!ctx)
        return fail(RSLF_ERR_ALLOC, "out of host memory");
    ctx->device = device;
    (void)hipDeviceGetAttribute(&ctx->num_cus, hipDeviceAttributeMultiprocessorCount, device);
    hipError_t e = hipMalloc(&ctx->total, 2 * sizeof(unsigned long long));   // [0] scanned pixels, [1] packed-list length (int)
    if (e == hipSuccess)
        e = hipMalloc(&ctx->minmax, 2 * sizeof(float));
    if (e == hipSuccess)
        e = hipEventCreate(&ctx->ev0);
    if (e == hipSuccess)
        e = hipEventCreate(&ctx->ev1);
    if (e != hipSuccess) {
        delete ctx;
        return fail(RSLF_ERR_HIP, "context setup failed: %s", hipGetErrorString(e));
    }
    *out = ctx;
    return RSLF_OK;
}

extern "C" int rslf_ctx_destroy(rslf_ctx* ctx)
{
    if (!ctx)
        return RSLF_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(ctx->list);
    (void)hipFree(ctx->count);
    (void)hipFree(ctx->depth_tmp);
    (void)hipFree(ctx->total);
    (void)hipFree(ctx->partial);
    (void)hipFree(ctx->minmax);
    (void)hipFree(ctx->staging);
    (void)hipFree(ctx->scan_partial);
    (void)hipFree(ctx->scan_ticket);
    for (int i = 0; i < rslf_ctx::kHelperSlots; i++)
        (void)hipFree(ctx->helper[i]);
    (void)hipFree(ctx->winner);
    (void)hipFree(ctx->dirty);
    (void)hipFree(ctx->sweep_mask);
    (void)hipFree(ctx->filtered);
    if (ctx->ev0)
        (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1)
        (void)hipEventDestroy(ctx->ev1);
    delete ctx;
    return RSLF_OK;
}

extern "C" int rslf_ctx_set_stream(rslf_ctx* ctx, void* hip_stream)
{
    if (!ctx)
        return fail(RSLF_ERR_INVALID_ARG, "ctx is NULL");
    ctx->stream = (hipStream_t)hip_stream;
    return RSLF_OK;
}

extern "C" int rslf_ctx_set_debug(rslf_ctx* ctx, const char* key, int value)
{
    if (!ctx || !key)
        return fail(RSLF_ERR_INVALID_ARG, "ctx/key is NULL");
    if (strcmp(key, "force_scan") == 0 && value >= 0 && value <= 2)
        ctx->force_scan = value;
    else if (strcmp(key, "force_groups") == 0 && value >= 0 && value <= 64)
        ctx->force_groups = value;
    else if (strcmp(key, "force_packed") == 0 && value >= -1 && value <= 1)
        ctx->force_packed = value;
    else if (strcmp(key, "stream_share") == 0 && (value == 0 || value == 1))
        ctx->stream_share = value != 0;
    else if (strcmp(key, "stream_groups") == 0 && value >= 0 && value <= 64)
        ctx->stream_groups = value;
    else if (strcmp(key, "stream_lds_kib") == 0 && value >= 16 && value <= 152) {
        ctx->stream_lds_bytes = (size_t)value << 10;
        ctx->stream_attr_set = false;
    } else
        return fail(RSLF_ERR_INVALID_ARG, "rslf_ctx_set_debug: unknown key or value out of range: %s = %d", key, value);
    return RSLF_OK;
}

extern "C" int rslf_ctx_synchronize(rslf_ctx* ctx)
{
    if (!ctx)
        return fail(RSLF_ERR_INVALID_ARG, "ctx is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return RSLF_OK;
}

// ---- volume ---------------------------------------------------------------

extern "C" int rslf_volume_create(rslf_ctx* ctx, int V, int S, int U, int C, rslf_volume** out)
{
    if (!ctx || !out)
        return fail(RSLF_ERR_INVALID_ARG, "ctx/out is NULL");
    *out = nullptr;
    if (V < 1 || S < 1 || U < 1)
        return fail(RSLF_ERR_INVALID_ARG, "bad dimensions V=%d S=%d U=%d", V, S, U);
    if (C != 1 && C != 3)
        return fail(RSLF_ERR_UNSUPPORTED, "C=%d: the reference instantiates float and cv::Vec3f only (dc.hpp:149-154)", C);
    if (V > 65535 || S > 65535)
        return fail(RSLF_ERR_UNSUPPORTED, "V=%d / S=%d: the per-scanline kernels index scanlines and views with grid.y / grid.z "
                                          "(at most 65535)", V, S);
    if ((long long)S * C * (((long long)U + 1 + 63) / 64 * 64) > ((long long)1 << 29))
        return fail(RSLF_ERR_UNSUPPORTED, "one EPI (S*pitch*C floats) must stay below 2 GiB: the scan addresses it with 32-bit byte offsets");
    HIP_TRY(hipSetDevice(ctx->device));
    rslf_volume* vol = new (std::nothrow) rslf_volume();
    if (!vol)
        return fail(RSLF_ERR_ALLOC, "out of host memory");
    vol->ctx = ctx;
    vol->device = ctx->device;
    vol->V = V;
    vol->S = S;
    vol->U = U;
    vol->C = C;
    vol->pitch = ((U + 1 + 63) / 64) * 64;   // pixels per row, > U: the second lerp tap of u = U-1 lands on zeros
    vol->bytes = (size_t)V * S * C * vol->pitch * sizeof(float);
    hipError_t e = hipMalloc(&vol->base, vol->bytes);
    if (e != hipSuccess) {
        delete vol;
        return fail(RSLF_ERR_ALLOC, "hipMalloc(%zu) for the volume failed: %s", vol->bytes, hipGetErrorString(e));
    }
    *out = vol;
    return RSLF_OK;
}

extern "C" int rslf_volume_destroy(rslf_volume* vol)
{
    if (!vol)
        return RSLF_OK;
    // hipFree waits for the device's outstanding work, so the slab outlives every launch that reads it; the
    // context is not touched (it may already be gone -- contexts and volumes can be destroyed in either order)
    (void)hipSetDevice(vol->device);
    (void)hipFree(vol->base);
    delete vol;
    return RSLF_OK;
}

extern "C" int rslf_volume_describe(const rslf_volume* vol, rslf_volume_desc* out)
{
    if (!vol || !out)
        return fail(RSLF_ERR_INVALID_ARG, "vol/out is NULL");
    out->V = vol->V;
    out->S = vol->S;
    out->U = vol->U;
    out->C = vol->C;
    out->pitch = vol->pitch;
    out->d_base = vol->base;
    out->bytes = vol->bytes;
    out->min_value = vol->min_value;
    out->max_value = vol->max_value;
    return RSLF_OK;
}

static int ensure_partial(rslf_ctx* ctx, size_t rows)
{
    if (rows > ctx->partial_cap) {
        if (ctx->partial)
            HIP_TRY(hipFree(ctx->partial));
        ctx->partial = nullptr;
        ctx->partial_cap = 0;
        HIP_TRY(hipMalloc(&ctx->partial, rows * 2 * sizeof(float)));
        ctx->partial_cap = rows;
    }
    return RSLF_OK;
}

static int ensure_staging(rslf_ctx* ctx, size_t bytes)
{
    if (bytes > ctx->staging_cap) {
        if (ctx->staging)
            HIP_TRY(hipFree(ctx->staging));
        ctx->staging = nullptr;
        ctx->staging_cap = 0;
        HIP_TRY(hipMalloc(&ctx->staging, bytes));
        ctx->staging_cap = bytes;
    }
    return RSLF_OK;
}

// Helper scratch slot `slot`, at least `bytes` large.  Growing it frees the old buffer, which waits for the
// device; after the first call of a given size nothing is allocated any more.
static int helper_scratch(rslf_ctx* ctx, int slot, size_t bytes, void** out)
{
    if (bytes > ctx->helper_cap[slot]) {
        if (ctx->helper[slot])
            HIP_TRY(hipFree(ctx->helper[slot]));
        ctx->helper[slot] = nullptr;
        ctx->helper_cap[slot] = 0;
        HIP_TRY(hipMalloc(&ctx->helper[slot], bytes));
        ctx->helper_cap[slot] = bytes;
    }
    *out = ctx->helper[slot];
    return RSLF_OK;
}

__global__ void k_init_minmax(float* minmax)
{
    minmax[0] = INFINITY;
    minmax[1] = -INFINITY;
}

static int minmax_begin(rslf_ctx* ctx)
{
    hipLaunchKernelGGL(k_init_minmax, dim3(1), dim3(1), 0, ctx->stream, ctx->minmax);
    HIP_TRY(hipGetLastError());
    return RSLF_OK;
}

static int minmax_end(rslf_volume* vol)
{
    rslf_ctx* ctx = vol->ctx;
    float mm[2];
    HIP_TRY(hipMemcpyAsync(mm, ctx->minmax, sizeof(mm), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    vol->min_value = mm[0];
    vol->max_value = mm[1];
    vol->filled = true;
    return RSLF_OK;
}

// Pack rows [V0, V0+Vn) from a device buffer holding just those rows.
template <typename SrcT>
static int pack_chunk(rslf_volume* vol, const SrcT* d_src, int V0, int Vn, bool image_major, float scale)
{
    rslf_ctx* ctx = vol->ctx;
    const size_t rows = (size_t)Vn * vol->S;
    int rc = ensure_partial(ctx, rows);
    if (rc)
        return rc;
    if (image_major)
        hipLaunchKernelGGL((k0_pack<SrcT, true>), dim3((unsigned)rows), dim3(256), 0, ctx->stream, d_src, vol->base, V0, Vn, Vn,
                           vol->S, vol->U, vol->C, vol->pitch, scale, ctx->partial);
    else
        hipLaunchKernelGGL((k0_pack<SrcT, false>), dim3((unsigned)rows), dim3(256), 0, ctx->stream, d_src, vol->base, V0, Vn, Vn,
                           vol->S, vol->U, vol->C, vol->pitch, scale, ctx->partial);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(k0_minmax_final, dim3(1), dim3(256), 0, ctx->stream, ctx->partial, (int)rows, ctx->minmax);
    HIP_TRY(hipGetLastError());
    return RSLF_OK;
}

// Host upload in scanline chunks through a bounded device staging buffer.
template <typename SrcT>
static int upload_host(rslf_volume* vol, const SrcT* const* h_ptrs, size_t row_stride_bytes, bool image_major, float scale)
{
    rslf_ctx* ctx = vol->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t row_bytes = (size_t)vol->U * vol->C * sizeof(SrcT);
    if (row_stride_bytes == 0)
        row_stride_bytes = row_bytes;
    if (row_stride_bytes < row_bytes)
        return fail(RSLF_ERR_INVALID_ARG, "row_stride_bytes %zu < row size %zu", row_stride_bytes, row_bytes);
    const size_t epi_bytes = row_bytes * vol->S;
    const size_t budget = (size_t)256 << 20;
    int chunk = (int)std::max<size_t>(1, budget / epi_bytes);
    chunk = std::min(chunk, vol->V);
    int rc = ensure_staging(ctx, (size_t)chunk * epi_bytes);
    if (rc)
        return rc;
    rc = minmax_begin(ctx);
    if (rc)
        return rc;
    for (int v0 = 0; v0 < vol->V; v0 += chunk) {
        const int vn = std::min(chunk, vol->V - v0);
        if (!image_major) {
            // h_ptrs[v] -> S rows; staging [vn][S][U*C]
            for (int i = 0; i < vn; i++)
                if (!h_ptrs[v0 + i])
                    return fail(RSLF_ERR_INVALID_ARG, "h_epis[%d] is NULL", v0 + i);
            // Dense rows (the usual cv::Mat): an EPI is one run of bytes, and EPIs that follow one another in host memory
            // (a stacked array) are one run together -- plain 1-D copies, which move pageable memory at the link's rate
            // (57 GB/s measured, tools/probe_h2d.py) where the 2-D form with its 8 KB rows reached about 10.
            for (int i = 0; i < vn;) {
                if (row_stride_bytes != row_bytes) {
                    HIP_TRY(hipMemcpy2DAsync((char*)ctx->staging + (size_t)i * epi_bytes, row_bytes, h_ptrs[v0 + i], row_stride_bytes,
                                             row_bytes, vol->S, hipMemcpyHostToDevice, ctx->stream));
                    i++;
                    continue;
                }
                int n = 1;
                while (i + n < vn && (const char*)h_ptrs[v0 + i + n] == (const char*)h_ptrs[v0 + i] + (size_t)n * epi_bytes)
                    n++;
                HIP_TRY(hipMemcpyAsync((char*)ctx->staging + (size_t)i * epi_bytes, h_ptrs[v0 + i], (size_t)n * epi_bytes,
                                       hipMemcpyHostToDevice, ctx->stream));
                i += n;
            }
        } else {
            // h_ptrs[s] -> V rows; staging [S][vn][U*C]
            for (int s = 0; s < vol->S; s++) {
                if (!h_ptrs[s])
                    return fail(RSLF_ERR_INVALID_ARG, "h_imgs[%d] is NULL", s);
                if (row_stride_bytes == row_bytes)   // dense rows: one run of bytes (see above)
                    HIP_TRY(hipMemcpyAsync((char*)ctx->staging + (size_t)s * vn * row_bytes, (const char*)h_ptrs[s] + (size_t)v0 * row_bytes,
                                           (size_t)vn * row_bytes, hipMemcpyHostToDevice, ctx->stream));
                else
                    HIP_TRY(hipMemcpy2DAsync((char*)ctx->staging + (size_t)s * vn * row_bytes, row_bytes,
                                             (const char*)h_ptrs[s] + (size_t)v0 * row_stride_bytes, row_stride_bytes, row_bytes, vn,
                                             hipMemcpyHostToDevice, ctx->stream));
            }
        }
        rc = pack_chunk<SrcT>(vol, (const SrcT*)ctx->staging, v0, vn, image_major, scale);
        if (rc)
            return rc;
        // the staging buffer is reused by the next chunk
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    return minmax_end(vol);
}

// Image-major upload with build_epis_from_imgs' transpose / rotate_180 options: n_imgs images of V rows x cols.
template <typename SrcT>
static int upload_images_xf(rslf_volume* vol, const SrcT* const* h_imgs, size_t row_stride_bytes, float scale, int transpose,
                            int rotate_180)
{
    rslf_ctx* ctx = vol->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    const int n_imgs = transpose ? vol->U : vol->S;   // the slab holds E^T when transposed
    const int cols = transpose ? vol->S : vol->U;
    const size_t row_bytes = (size_t)cols * vol->C * sizeof(SrcT);
    if (row_stride_bytes == 0)
        row_stride_bytes = row_bytes;
    if (row_stride_bytes < row_bytes)
        return fail(RSLF_ERR_INVALID_ARG, "row_stride_bytes %zu < row size %zu", row_stride_bytes, row_bytes);
    const size_t epi_bytes = row_bytes * n_imgs;
    int chunk = (int)std::max<size_t>(1, ((size_t)256 << 20) / epi_bytes);
    chunk = std::min(chunk, vol->V);
    int rc = ensure_staging(ctx, (size_t)chunk * epi_bytes);
    if (rc)
        return rc;
    rc = minmax_begin(ctx);
    if (rc)
        return rc;
    for (int v0 = 0; v0 < vol->V; v0 += chunk) {
        const int vn = std::min(chunk, vol->V - v0);
        for (int i = 0; i < n_imgs; i++) {   // staging [n_imgs][vn][cols*C]
            if (!h_imgs[i])
                return fail(RSLF_ERR_INVALID_ARG, "h_imgs[%d] is NULL", i);
            if (row_stride_bytes == row_bytes)
                HIP_TRY(hipMemcpyAsync((char*)ctx->staging + (size_t)i * vn * row_bytes, (const char*)h_imgs[i] + (size_t)v0 * row_bytes,
                                       (size_t)vn * row_bytes, hipMemcpyHostToDevice, ctx->stream));
            else
                HIP_TRY(hipMemcpy2DAsync((char*)ctx->staging + (size_t)i * vn * row_bytes, row_bytes,
                                         (const char*)h_imgs[i] + (size_t)v0 * row_stride_bytes, row_stride_bytes, row_bytes, vn,
                                         hipMemcpyHostToDevice, ctx->stream));
        }
        const size_t rows = (size_t)vn * vol->S;
        rc = ensure_partial(ctx, rows);
        if (rc)
            return rc;
        hipLaunchKernelGGL((k0_pack_images_xf<SrcT>), dim3((unsigned)rows), dim3(256), 0, ctx->stream, (const SrcT*)ctx->staging,
                           vol->base, v0, vn, n_imgs, cols, vol->S, vol->U, vol->C, vol->pitch, scale, transpose, rotate_180,
                           ctx->partial);
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(k0_minmax_final, dim3(1), dim3(256), 0, ctx->stream, ctx->partial, (int)rows, ctx->minmax);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(ctx->stream));   // the staging buffer is reused by the next chunk
    }
    return minmax_end(vol);
}

// dc.hpp:442-460: epi_scale_factor = max over every value of every EPI
static float host_max_f32(const float* const* h_ptrs, int n_ptrs, int rows, size_t row_stride_bytes, size_t row_elems, float start)
{
    float m = start;
    for (int i = 0; i < n_ptrs; i++) {
        for (int r = 0; r < rows; r++) {
            const float* p = (const float*)((const char*)h_ptrs[i] + (size_t)r * row_stride_bytes);
            for (size_t k = 0; k < row_elems; k++)
                if (p[k] > m)
                    m = p[k];
        }
    }
    return m;
}

static float scale_of(float epi_scale_factor)
{
    return (float)(1.0 / (double)epi_scale_factor);   // dc.hpp:474 through cvtScale's float scale
}

extern "C" int rslf_volume_upload_epis_f32(rslf_volume* vol, const float* const* h_epis, size_t row_stride_bytes,
                                           float epi_scale_factor, float* scale_used)
{
    if (!vol || !h_epis)
        return fail(RSLF_ERR_INVALID_ARG, "vol/h_epis is NULL");
    const size_t row_elems = (size_t)vol->U * vol->C;
    const size_t stride = row_stride_bytes ? row_stride_bytes : row_elems * sizeof(float);
    if (epi_scale_factor < 0)
        epi_scale_factor = host_max_f32(h_epis, vol->V, vol->S, stride, row_elems, epi_scale_factor);
    if (scale_used)
        *scale_used = epi_scale_factor;
    return upload_host<float>(vol, h_epis, stride, false, scale_of(epi_scale_factor));
}

extern "C" int rslf_volume_upload_epis_u8(rslf_volume* vol, const uint8_t* const* h_epis, size_t row_stride_bytes)
{
    if (!vol || !h_epis)
        return fail(RSLF_ERR_INVALID_ARG, "vol/h_epis is NULL");
    return upload_host<uint8_t>(vol, h_epis, row_stride_bytes, false, (float)(1.0 / 255.0));   // dc.hpp:470
}

extern "C" int rslf_volume_upload_images_f32(rslf_volume* vol, const float* const* h_imgs, size_t row_stride_bytes,
                                             float epi_scale_factor, float* scale_used)
{
    if (!vol || !h_imgs)
        return fail(RSLF_ERR_INVALID_ARG, "vol/h_imgs is NULL");
    const size_t row_elems = (size_t)vol->U * vol->C;
    const size_t stride = row_stride_bytes ? row_stride_bytes : row_elems * sizeof(float);
    if (epi_scale_factor < 0)
        epi_scale_factor = host_max_f32(h_imgs, vol->S, vol->V, stride, row_elems, epi_scale_factor);
    if (scale_used)
        *scale_used = epi_scale_factor;
    return upload_host<float>(vol, h_imgs, stride, true, scale_of(epi_scale_factor));
}

extern "C" int rslf_volume_upload_images_u8(rslf_volume* vol, const uint8_t* const* h_imgs, size_t row_stride_bytes)
{
    if (!vol || !h_imgs)
        return fail(RSLF_ERR_INVALID_ARG, "vol/h_imgs is NULL");
    return upload_host<uint8_t>(vol, h_imgs, row_stride_bytes, true, (float)(1.0 / 255.0));
}

extern "C" int rslf_volume_upload_images_xf_f32(rslf_volume* vol, const float* const* h_imgs, size_t row_stride_bytes,
                                                float epi_scale_factor, float* scale_used, int transpose, int rotate_180)
{
    if (!vol || !h_imgs)
        return fail(RSLF_ERR_INVALID_ARG, "vol/h_imgs is NULL");
    const int n_imgs = transpose ? vol->U : vol->S;
    const size_t row_elems = (size_t)(transpose ? vol->S : vol->U) * vol->C;
    const size_t stride = row_stride_bytes ? row_stride_bytes : row_elems * sizeof(float);
    if (epi_scale_factor < 0)
        epi_scale_factor = host_max_f32(h_imgs, n_imgs, vol->V, stride, row_elems, epi_scale_factor);
    if (scale_used)
        *scale_used = epi_scale_factor;
    return upload_images_xf<float>(vol, h_imgs, stride, scale_of(epi_scale_factor), transpose != 0, rotate_180 != 0);
}

extern "C" int rslf_volume_upload_images_xf_u8(rslf_volume* vol, const uint8_t* const* h_imgs, size_t row_stride_bytes, int transpose,
                                               int rotate_180)
{
    if (!vol || !h_imgs)
        return fail(RSLF_ERR_INVALID_ARG, "vol/h_imgs is NULL");
    return upload_images_xf<uint8_t>(vol, h_imgs, row_stride_bytes, (float)(1.0 / 255.0), transpose != 0, rotate_180 != 0);
}

extern "C" int rslf_volume_pack_device_f32(rslf_volume* vol, const float* d_vsuc, float epi_scale_factor, float* scale_used)
{
    if (!vol || !d_vsuc)
        return fail(RSLF_ERR_INVALID_ARG, "vol/d_vsuc is NULL");
    if (epi_scale_factor < 0)
        return fail(RSLF_ERR_INVALID_ARG, "pack_device needs an explicit epi_scale_factor (> 0); 1.0 keeps the values");
    rslf_ctx* ctx = vol->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    if (scale_used)
        *scale_used = epi_scale_factor;
    int rc = minmax_begin(ctx);
    if (rc)
        return rc;
    rc = pack_chunk<float>(vol, d_vsuc, 0, vol->V, false, scale_of(epi_scale_factor));
    if (rc)
        return rc;
    return minmax_end(vol);
}

// cv::getStructuringElement(shape, Size(k, k)) with the default anchor, as OpenCV 3.x builds it (imgproc/src/morph.cpp):
// RECT every column; CROSS the anchor row entirely, elsewhere the anchor column; ELLIPSE the columns
// [c - dx, c + dx + 1), dx = cvRound(c * sqrt((r*r - dy*dy) / (r*r))), r = c = k/2, dy = i - r.
static MorphElement structuring_element(int shape, int k)
{
    MorphElement el;
    el.k = k;
    const int r = k / 2, c = k / 2;
    const double inv_r2 = r ? 1.0 / ((double)r * r) : 0.0;
    for (int i = 0; i < 31; i++)
        el.rows[i] = 0;
    for (int i = 0; i < k; i++) {
        int j1 = 0, j2 = 0;
        if (shape == 0 || (shape == 1 && i == k / 2)) {
            j2 = k;
        } else if (shape == 1) {
            j1 = k / 2;
            j2 = j1 + 1;
        } else {
            const int dy = i - r;
            if (std::abs(dy) <= r) {
                const int dx = (int)std::lrint(c * std::sqrt((r * r - dy * dy) * inv_r2));
                j1 = std::max(c - dx, 0);
                j2 = std::min(c + dx + 1, k);
            }
        }
        for (int j = j1; j < j2; j++)
            el.rows[i] |= 1u << j;
    }
    return el;
}

// ---- hot path -------------------------------------------------------------

extern "C" int rslf_edge_confidence_pile(rslf_ctx* ctx, const rslf_volume* vol, int s, const rslf_params* p,
                                         float* d_Ce_vu, uint8_t* d_Ce_mask_vu)
{
    if (!ctx || !vol || !d_Ce_vu || !d_Ce_mask_vu)
        return fail(RSLF_ERR_INVALID_ARG, "NULL argument");
    int rc = check_params(p);
    if (rc)
        return rc;
    if (s < 0 || s >= vol->S)
        return fail(RSLF_ERR_INVALID_ARG, "s=%d outside [0,%d)", s, vol->S);
    if (!vol->filled)
        return fail(RSLF_ERR_INVALID_ARG, "volume has not been filled");
    HIP_TRY(hipSetDevice(ctx->device));
    EdgeConsts ec;
    ec.filter_size = p->edge_confidence_filter_size;
    ec.cut_shadows = p->cut_shadows;
    ec.shadow_level = p->shadow_level;
    ec.edge_thr = p->edge_score_threshold;
    const dim3 grid((vol->U + 255) / 256, vol->V);
    if (vol->C == 1)
        hipLaunchKernelGGL(k1_edge_confidence<1>, grid, dim3(256), 0, ctx->stream, view_of(vol), s, ec, d_Ce_vu, d_Ce_mask_vu);
    else
        hipLaunchKernelGGL(k1_edge_confidence<3>, grid, dim3(256), 0, ctx->stream, view_of(vol), s, ec, d_Ce_vu, d_Ce_mask_vu);
    HIP_TRY(hipGetLastError());
    if (p->edge_confidence_opening_size > 1) {   // core.hpp:759-768
        rc = ensure_plane_scratch(ctx, vol->V, vol->U);
        if (rc)
            return rc;
        const MorphElement el = structuring_element(p->edge_confidence_opening_type, p->edge_confidence_opening_size);
        uint8_t* tmp = reinterpret_cast<uint8_t*>(ctx->depth_tmp);   // V*U floats: room for a byte plane
        hipLaunchKernelGGL(k1_morph_pass, grid, dim3(256), 0, ctx->stream, d_Ce_mask_vu, tmp, vol->V, vol->U, el, 0);   // erode
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(k1_morph_pass, grid, dim3(256), 0, ctx->stream, tmp, d_Ce_mask_vu, vol->V, vol->U, el, 1);   // dilate
        HIP_TRY(hipGetLastError());
    }
    return RSLF_OK;
}

extern "C" int rslf_selective_median(rslf_ctx* ctx, const rslf_volume* vol, const float* d_src_vu, float* d_dst_vu,
                                     int s_hat, int size, const uint8_t* d_mask_vu, float epsilon)
{
    if (!ctx || !vol || !d_src_vu || !d_dst_vu || !d_mask_vu)
        return fail(RSLF_ERR_INVALID_ARG, "NULL argument");
    if (d_src_vu == d_dst_vu)
        return fail(RSLF_ERR_INVALID_ARG, "selective median cannot run in place");
    if (size < 1 || (size & 1) == 0 || size > kMedianMaxSize)
        return fail(RSLF_ERR_UNSUPPORTED, "median size must be odd and <= %d", kMedianMaxSize);
    if (s_hat < 0 || s_hat >= vol->S)
        return fail(RSLF_ERR_INVALID_ARG, "s_hat=%d outside [0,%d)", s_hat, vol->S);
    HIP_TRY(hipSetDevice(ctx->device));
    const dim3 grid((vol->U + 255) / 256, vol->V);
    const size_t lds = size == 5 ? 0 : (size_t)size * size * 256 * sizeof(float);   // one candidate slot per window pixel and thread (5 x 5 sorts in registers)
    if (vol->C == 1)
        hipLaunchKernelGGL(k3_selective_median<1>, grid, dim3(256), lds, ctx->stream, view_of(vol), d_src_vu, d_dst_vu, d_mask_vu,
                           s_hat, size, epsilon);
    else
        hipLaunchKernelGGL(k3_selective_median<3>, grid, dim3(256), lds, ctx->stream, view_of(vol), d_src_vu, d_dst_vu, d_mask_vu,
                           s_hat, size, epsilon);
    HIP_TRY(hipGetLastError());
    return RSLF_OK;
}

// Register-variant slot counts compiled into this library (multiples of 8), per channel count: those that run at
// two or more waves per SIMD (C*SPAD + working registers <= 256).  Beyond them -- C=1 above 192 views, RGB above 48
// -- the streaming kernel with its resident prefix is faster than a one-wave register variant (55-61 vs 44-47
// TFLOP/s, profiles/r01_k2_variants.md), so none is built.
#ifndef RSLF_SPAD_LIST_1CH
#define RSLF_SPAD_LIST_1CH(X) X(8) X(16) X(24) X(32) X(40) X(48) X(56) X(64) X(72) X(80) X(88) X(96) X(104) X(112) X(120) X(128) \
    X(144) X(160) X(176) X(192)
#endif
#ifndef RSLF_SPAD_LIST_3CH
#define RSLF_SPAD_LIST_3CH(X) X(8) X(16) X(24) X(32) X(40) X(48)
#endif

static int launch_scan_reg(int spad, int C, const ScanArgs& a, dim3 grid, hipStream_t stream)
{
#define RSLF_CASE(N)                                                                                        \
    case N:                                                                                                 \
        if (a.packed)                                                                                       \
            hipLaunchKernelGGL((k2_scan_reg_packed<N, RSLF_C>), grid, dim3(64 * kScanWaves), 0, stream, a); \
        else                                                                                                \
            hipLaunchKernelGGL((k2_scan_reg<N, RSLF_C>), grid, dim3(64 * kScanWaves), 0, stream, a);        \
        return RSLF_OK;
    if (C == 1) {
        switch (spad) {
#define RSLF_C 1
            RSLF_SPAD_LIST_1CH(RSLF_CASE)
#undef RSLF_C
        default:
            break;
        }
    } else if (C == 3) {
        switch (spad) {
#define RSLF_C 3
            RSLF_SPAD_LIST_3CH(RSLF_CASE)
#undef RSLF_C
        default:
            break;
        }
    }
#undef RSLF_CASE
    return fail(RSLF_ERR_UNSUPPORTED, "no register scan kernel with %d slots x %d channels", spad, C);
}

// Smallest compiled slot count >= S (0 = none: the generic kernel runs).
static int pick_spad(int S, int C)
{
    int best = 0;
#define RSLF_PICK(N) \
    if (N >= S && best == 0) best = N;
    if (C == 1) {
        RSLF_SPAD_LIST_1CH(RSLF_PICK)
    } else if (C == 3) {
        RSLF_SPAD_LIST_3CH(RSLF_PICK)
    }
#undef RSLF_PICK
    return best;
}

static void fill_stats(rslf_ctx* ctx, unsigned long long tot, int dim_d, rslf_stats* stats)
{
    stats->pixels_scanned = (int64_t)tot;
    stats->units = (int64_t)tot * dim_d;
    stats->scan_kernel = ctx->last_kernel;
    stats->s_pad = ctx->last_spad;
}

extern "C" int rslf_depth_epi_scan(rslf_ctx* ctx, const rslf_volume* vol, const float* d_dmin_vu, const float* d_dmax_vu,
                                   float dmin, float dmax, int dim_d, int s_hat, float* d_Ce_vu, uint8_t* d_Ce_mask_vu,
                                   float* d_Cd_vu, float* d_depth_vu, float* d_rbar_vu, const rslf_params* p,
                                   uint8_t* d_mask_vu, int32_t* d_idx_vu, float* d_score_vu, rslf_stats* stats)
{
    if (!ctx || !vol || !d_Ce_vu || !d_Ce_mask_vu || !d_Cd_vu || !d_depth_vu || !d_rbar_vu)
        return fail(RSLF_ERR_INVALID_ARG, "NULL argument");
    int rc = check_params(p);
    if (rc)
        return rc;
    if ((d_dmin_vu == nullptr) != (d_dmax_vu == nullptr))
        return fail(RSLF_ERR_INVALID_ARG, "d_dmin_vu and d_dmax_vu must both be given or both be NULL");
    if (dim_d < 2)
        return fail(RSLF_ERR_INVALID_ARG, "dim_d=%d: the hypothesis grid divides by dim_d-1 (core.hpp:548)", dim_d);
    if (s_hat < 0 || s_hat >= vol->S)
        return fail(RSLF_ERR_INVALID_ARG, "s_hat=%d outside [0,%d)", s_hat, vol->S);
    if (!vol->filled)
        return fail(RSLF_ERR_INVALID_ARG, "volume has not been filled");
    HIP_TRY(hipSetDevice(ctx->device));
    rc = ensure_plane_scratch(ctx, vol->V, vol->U);
    if (rc)
        return rc;

    const size_t n = (size_t)vol->V * vol->U;
    hipStream_t st = ctx->stream;
    if (d_idx_vu)
        HIP_TRY(hipMemsetAsync(d_idx_vu, 0xFF, n * sizeof(int32_t), st));   // -1
    if (d_score_vu)
        HIP_TRY(hipMemsetAsync(d_score_vu, 0, n * sizeof(float), st));
    const int precompacted = ctx->precompacted;   // 1: rslf_depth1d_pile_run's K1 left row lists and total; 2: packed list (sweep)
    ctx->precompacted = 0;
    if (!ctx->keep_total && !precompacted)
        HIP_TRY(hipMemsetAsync(ctx->total, 0, sizeof(unsigned long long), st));

    // Which kernel?  Register variant: S within the compiled slot counts, and radiances in [0, 1e6] so that
    // max(R,0) == R and the 1e30 sentinel dwarfs them.  Streaming variant: same precondition, any S whose
    // offset table fits the LDS.  Otherwise the generic kernel.
    int spad = 0;
    if (vol->min_value >= 0.0f && vol->max_value <= 1.0e6f)
        spad = pick_spad(vol->S, vol->C);
    bool stream_ok = vol->min_value >= 0.0f && vol->max_value <= 1.0e6f &&
                     (size_t)kScanWaves * vol->S * sizeof(float) <= (size_t)48 << 10;
    if (p->interpolation != RSLF_INTERP_LINEAR) {      // nearest-neighbour sampling: generic kernel only
        spad = 0;
        stream_ok = false;
    } else if (ctx->force_scan == 1) {                 // parity tests exercise every variant on small cases
        spad = 0;
        stream_ok = false;
    } else if (ctx->force_scan == 2) {
        spad = 0;
    }
    const bool use_stream = !spad && stream_ok;

    // hypothesis groups per tile and packed tiles: 1 / off unless the caller expects a sparse launch ...
    int groups = std::max(1, ctx->scan_groups);
    bool packed = ctx->scan_packed;
    // ... or the streaming kernel runs a dense launch: its workgroups then share tiles so that what an XCD's
    // workgroups gather from at any one time fits its L2 (kStreamGroups)
    if (use_stream && groups == 1 && !packed)
        groups = ctx->stream_groups > 0 ? ctx->stream_groups : kStreamGroups;
    // A dense launch of a register kernel whose grid is only a few rounds of workgroups pays for its last, partly empty
    // round: 139 scanlines of c3 (one GPU's share of eight) are 5.4 rounds and took 9.3 ms where 8.5 would do.  Sharing
    // each tile's hypotheses among 2-8 workgroups makes the rounds shorter and more numerous (measured on that shard:
    // 9.25 / 8.73 / 8.58 / 8.46 ms with 1 / 2 / 4 / 8 groups; a quarter of the field 17.2 -> 16.6 ms with 4).
    if (spad && groups == 1 && !packed && ctx->num_cus > 0) {
        const long long tiles = (long long)vol->V * ((vol->U + 63) / 64);
        const long long resident = (long long)ctx->num_cus * scan_reg_waves(spad, vol->C);
        // ... as long as a wave keeps at least eight hypotheses and ~256 (hypothesis, view) pairs: below that its fixed
        // costs per tile (offset table, merge) outweigh the shorter rounds (c1, 64 hypotheses over 9 views: +25 % with
        // eight groups, +3 % with two)
        while (groups < 8 && tiles * groups < 40 * resident && dim_d >= 8 * kScanWaves * 2 * groups &&
               (long long)(dim_d / (kScanWaves * 2 * groups)) * vol->S >= 256)
            groups *= 2;
    }
    if (ctx->force_groups > 0)   // parity tests: sparse-launch shapes on the pile path too (rslf_ctx_set_debug)
        groups = std::min(64, ctx->force_groups);
    if (ctx->force_packed >= 0)
        packed = ctx->force_packed != 0;
    if (n > (size_t)INT32_MAX || precompacted == 1)
        packed = false;   // entry counts are ints; precompacted: the row lists are what K1 wrote
    if (precompacted == 2)
        packed = true;    // the previous visit's apply pass left the packed list and its length
    // enough hypotheses to share out?  (One per wave instead of two on the sweep's sparse launches measured slower.)
    while (groups > 1 && dim_d < 2 * kScanWaves * groups)
        groups /= 2;

    int* packed_n = reinterpret_cast<int*>(ctx->total + 1);
    if (precompacted) {
        // nothing to compact
    } else if (packed) {
        if (!ctx->packed_n_clean)
            HIP_TRY(hipMemsetAsync(packed_n, 0, sizeof(int), st));
        hipLaunchKernelGGL(k_compact_mask_packed, dim3(vol->V), dim3(256), 0, st, d_Ce_mask_vu, d_mask_vu, vol->U, ctx->list,
                           ctx->count, ctx->total, packed_n);
    } else {
        hipLaunchKernelGGL(k_compact_mask, dim3(vol->V), dim3(256), 0, st, d_Ce_mask_vu, d_mask_vu, vol->U, ctx->list,
                           ctx->count, ctx->total);
    }
    HIP_TRY(hipGetLastError());

    ScanArgs a;
    a.vol = view_of(vol);
    a.list = ctx->list;
    a.count = ctx->count;
    a.dmin_vu = d_dmin_vu;
    a.dmax_vu = d_dmax_vu;
    a.dmin = dmin;
    a.dmax = dmax;
    a.dim_d = dim_d;
    a.s_hat = s_hat;
    a.k = make_scan_consts(p);
    a.Ce = d_Ce_vu;
    a.Ce_mask = d_Ce_mask_vu;
    a.Cd = d_Cd_vu;
    a.depth = d_depth_vu;
    a.rbar = d_rbar_vu;
    a.idx = d_idx_vu;
    a.score = d_score_vu;
    // the streaming kernel's row tiles leave lane 63 to its neighbour's right tap (k2_scan.hpp, DENSE)
    a.tile_w = (use_stream && !packed && ctx->stream_share) ? 63 : 64;
    // 63-entry tiles: a row's last tile takes up to 64 entries (scan_tile)
    a.tiles_per_row = a.tile_w == 63 ? std::max(1, (vol->U + 61) / 63) : (vol->U + a.tile_w - 1) / a.tile_w;
    {
        // largest position the scan can form, and one ulp of it below 1
        int e = 0;
        (void)frexpf((float)vol->U + 2.0f, &e);            // U + 2 < 2^e
        a.stream_frac_max = 1.0f - ldexpf(1.0f, std::max(e - 24, -24));
    }
    a.packed = packed ? 1 : 0;
    a.packed_n = packed_n;
    a.packed_adapt = 0;
    a.stream_park = 0;
    a.stream_wave_floats = 0;
    a.partial = nullptr;
    a.ticket = nullptr;
    a.v0 = 0;

    // Grouped launches leave one 32-byte record per (tile, group, lane) for the tile's last group to merge.  Row-tile
    // launches bound them by kPartialBudget and go by blocks of scanlines.  Packed launches of the register / generic
    // kernels settle their group count on the device (packed_groups, k2_scan.hpp): more than one group only while
    // tiles x groups <= kPackedItemTarget, so that many records (4 MiB) serve any list length.  The streaming kernel keeps
    // its groups whatever the length (they are its L2 locality) and halves them here until the worst case fits the budget.
    int rows_per_launch = vol->V;
    a.packed_adapt = (packed && !use_stream) ? 1 : 0;
    if (groups > 1 && !packed) {
        const size_t per_row = (size_t)a.tiles_per_row * groups * 64 * sizeof(Partial);
        rows_per_launch = (int)std::min<size_t>((size_t)vol->V, std::max<size_t>(1, kPartialBudget / per_row));
    }
    if (packed && use_stream)
        while (groups > 1 && ((n + 63) / 64) * groups * 64 * sizeof(Partial) > kPartialBudget)
            groups /= 2;
    a.groups = groups;
    if (groups > 1) {
        const size_t tiles_all = (n + 63) / 64;
        const size_t tiles_max = !packed ? (size_t)rows_per_launch * a.tiles_per_row
                                         : a.packed_adapt ? std::min<size_t>(tiles_all, kPackedItemTarget / 2) : tiles_all;
        const size_t recs = (packed && a.packed_adapt) ? std::min<size_t>(tiles_all * groups, kPackedItemTarget) * 64 : tiles_max * groups * 64;
        rc = ensure_group_scratch(ctx, recs, tiles_max);
        if (rc)
            return rc;
        a.partial = ctx->scan_partial;
        a.ticket = ctx->scan_ticket;
    }

    size_t lds = 0;
    if (use_stream) {
        // too many samples for the register file (k2_scan_stream).  LDS per wave: the S view offsets, the parked
        // samples and the staging slots of the re-gathered tail; as many batches of parked samples as the
        // workgroup's LDS share leaves room for (and never past the end of the views)
        const int batch = vol->C == 1 ? 8 : 4;
        const int nres = stream_resident_for(vol->S, vol->C);
        const size_t s4 = ((size_t)vol->S + 3) & ~(size_t)3;
        int park = 0;
        if (nres > 0) {
            size_t room = ctx->stream_lds_bytes / kScanWaves / sizeof(float);   // floats per wave
            room -= std::min(room, s4);
            park = (int)(room / ((size_t)vol->C * 64));
            park = std::min(park, vol->S - nres);
            park -= park % batch;
        }
        a.stream_park = park;
        a.stream_wave_floats = (int)(s4 + (size_t)park * vol->C * 64);
        a.stream_wave_floats = std::max(a.stream_wave_floats, 2 * (64 + (3 + vol->C) * 32));   // room for the wave's EpilogueBlock
        lds = (size_t)kScanWaves * a.stream_wave_floats * sizeof(float);
        if (!ctx->stream_attr_set) {   // more than the 64 KiB a kernel gets without asking
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k2_scan_stream<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)ctx->stream_lds_bytes));
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k2_scan_stream<3>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)ctx->stream_lds_bytes));
            ctx->stream_attr_set = true;
        }
    }

    HIP_TRY(hipGetLastError());   // anything an earlier enqueue left behind is not this launch's fault
    ctx->last_spad = spad;
    ctx->last_kernel = spad ? RSLF_SCAN_REG : (stream_ok ? RSLF_SCAN_STREAM : RSLF_SCAN_GENERIC);
    // The events that time K2 are marker packets of their own: ~5.6 us each before the next kernel starts (measured,
    // tools/probe_gaps.py) -- nothing beside a 66 ms scan, a tenth of a sweep's sparse visit.  A sweep times its first
    // (dense) visit only.
    const bool timed = !ctx->sweep_open || ctx->sweep_first;
    if (timed)
        HIP_TRY(hipEventRecord(ctx->ev0, st));
    for (int v0 = 0; v0 < vol->V; v0 += rows_per_launch) {
        const int rows = std::min(rows_per_launch, vol->V - v0);
        // row tiles: ceil(U/64) per scanline; packed tiles: at most ceil(V*U/64), the device knows how many
        const long long tiles = packed ? (long long)((n + 63) / 64) : (long long)rows * a.tiles_per_row;
        if (tiles * groups > (long long)1 << 30)
            return fail(RSLF_ERR_UNSUPPORTED, "%lld tiles x %d groups exceeds the grid limit", tiles, groups);
        a.v0 = v0;
        a.logical_blocks = (int)(tiles * groups);   // `groups` workgroups per tile, their waves split the hypotheses
        a.per_xcd = (a.logical_blocks + 7) / 8;
        // packed: a fixed grid strides over the items (k2_scan.hpp); ~4 workgroups per CU cover any occupancy
        const dim3 grid(packed ? (unsigned)std::min<long long>(tiles * groups, 1024) : (unsigned)(a.per_xcd * 8));
        if (spad) {
            rc = launch_scan_reg(spad, vol->C, a, grid, st);
            if (rc)
                return rc;
        } else if (use_stream) {
            if (vol->C == 1)
                hipLaunchKernelGGL(k2_scan_stream<1>, grid, dim3(64 * kScanWaves), lds, st, a);
            else
                hipLaunchKernelGGL(k2_scan_stream<3>, grid, dim3(64 * kScanWaves), lds, st, a);
        } else if (vol->C == 1) {
            hipLaunchKernelGGL(k2_scan_generic<1>, grid, dim3(64 * kScanWaves), 0, st, a);
        } else {
            hipLaunchKernelGGL(k2_scan_generic<3>, grid, dim3(64 * kScanWaves), 0, st, a);
        }
        HIP_TRY(hipGetLastError());   // grouped launches merge their records themselves (scan_epilogue): no combine launch
        if (packed)
            break;   // one launch covers the packed list
    }
    if (timed) {
        HIP_TRY(hipEventRecord(ctx->ev1, st));
        ctx->ev_valid = true;
    }

    if (stats) {
        unsigned long long tot = 0;
        HIP_TRY(hipMemcpyAsync(&tot, ctx->total, sizeof(tot), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        fill_stats(ctx, tot, dim_d, stats);
    }
    return RSLF_OK;
}

extern "C" int rslf_kernel_columns_pile(rslf_ctx* ctx, const rslf_volume* vol, const float* d_dmin_vu, const float* d_dmax_vu,
                                        float dmin, float dmax, int dim_d, int s_hat, const rslf_params* p,
                                        const int32_t* d_idx_vu, float* d_K_vsu)
{
    if (!ctx || !vol || !d_idx_vu || !d_K_vsu)
        return fail(RSLF_ERR_INVALID_ARG, "NULL argument");
    int rc = check_params(p);
    if (rc)
        return rc;
    if ((d_dmin_vu == nullptr) != (d_dmax_vu == nullptr))
        return fail(RSLF_ERR_INVALID_ARG, "d_dmin_vu and d_dmax_vu must both be given or both be NULL");
    if (dim_d < 2)
        return fail(RSLF_ERR_INVALID_ARG, "dim_d=%d: the hypothesis grid divides by dim_d-1 (core.hpp:548)", dim_d);
    if (s_hat < 0 || s_hat >= vol->S)
        return fail(RSLF_ERR_INVALID_ARG, "s_hat=%d outside [0,%d)", s_hat, vol->S);
    if (!vol->filled)
        return fail(RSLF_ERR_INVALID_ARG, "volume has not been filled");
    HIP_TRY(hipSetDevice(ctx->device));
    ScanArgs a = {};
    a.vol = view_of(vol);
    a.dmin_vu = d_dmin_vu;
    a.dmax_vu = d_dmax_vu;
    a.dmin = dmin;
    a.dmax = dmax;
    a.dim_d = dim_d;
    a.s_hat = s_hat;
    a.k = make_scan_consts(p);
    a.groups = 1;
    const dim3 grid((vol->U + 255) / 256, vol->V);
    if (vol->C == 1)
        hipLaunchKernelGGL(k2_kernel_column<1>, grid, dim3(256), 0, ctx->stream, a, d_idx_vu, d_K_vsu);
    else
        hipLaunchKernelGGL(k2_kernel_column<3>, grid, dim3(256), 0, ctx->stream, a, d_idx_vu, d_K_vsu);
    HIP_TRY(hipGetLastError());
    return RSLF_OK;
}

extern "C" int rslf_depth_epi_pile(rslf_ctx* ctx, const rslf_volume* vol, const float* d_dmin_vu, const float* d_dmax_vu,
                                   float dmin, float dmax, int dim_d, int s_hat, float* d_Ce_vu, uint8_t* d_Ce_mask_vu,
                                   float* d_Cd_vu, float* d_depth_vu, float* d_rbar_vu, const rslf_params* p,
                                   uint8_t* d_mask_vu, int32_t* d_idx_vu, float* d_score_vu, float* d_depth_raw_vu,
                                   rslf_stats* stats)
{
    if (!ctx || !vol || !d_depth_vu)
        return fail(RSLF_ERR_INVALID_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = ensure_plane_scratch(ctx, vol->V, vol->U);
    if (rc)
        return rc;
    const size_t n = (size_t)vol->V * vol->U;
    hipStream_t st = ctx->stream;
    // core.hpp:799-854: the scan of every EPI writes the RAW disparities -- into the caller's raw plane if one is
    // wanted (over the zeros best_depth starts from, dc.hpp:507), else into scratch, where no background is needed:
    // the median reads the raw plane at mask pixels only, and every mask pixel has been written by the scan ...
    // With a caller's scan mask, mask pixels that are NOT scanned now keep the disparity the plane came in with
    // (a_best_depth_v_u is in/out, core.hpp:305), and the median reads them: the raw plane then starts as a copy.
    float* raw = d_depth_raw_vu ? d_depth_raw_vu : ctx->depth_tmp;
    if (d_mask_vu)
        HIP_TRY(hipMemcpyAsync(raw, d_depth_vu, n * sizeof(float), hipMemcpyDeviceToDevice, st));
    else if (d_depth_raw_vu)
        HIP_TRY(hipMemsetAsync(d_depth_raw_vu, 0, n * sizeof(float), st));
    rc = rslf_depth_epi_scan(ctx, vol, d_dmin_vu, d_dmax_vu, dmin, dmax, dim_d, s_hat, d_Ce_vu, d_Ce_mask_vu, d_Cd_vu, raw,
                             d_rbar_vu, p, d_mask_vu, d_idx_vu, d_score_vu, nullptr);
    if (rc)
        return rc;
    // ... then core.hpp:881-892: median over the EDGE mask, result replaces best_depth -- written straight into the
    // caller's plane (every pixel: 0 where the mask is 0, core.hpp:678-679), so no plane is copied
    rc = rslf_selective_median(ctx, vol, raw, d_depth_vu, s_hat, p->median_filter_size, d_Ce_mask_vu, p->median_filter_epsilon);
    if (rc)
        return rc;

    if (stats) {
        unsigned long long tot = 0;
        HIP_TRY(hipMemcpyAsync(&tot, ctx->total, sizeof(tot), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        fill_stats(ctx, tot, dim_d, stats);
    }
    return RSLF_OK;
}

static int resolve_s_hat(int s_hat, int S)
{
    if (s_hat < 0 || s_hat > S - 1)
        return (int)std::floor((0.0 + S) / 2);   // dc.hpp:490-494
    return s_hat;
}

extern "C" int rslf_depth1d_pile_run(rslf_ctx* ctx, const rslf_volume* vol, float dmin, float dmax, int dim_d, int s_hat,
                                     const rslf_params* p, float* d_Ce_vu, uint8_t* d_Ce_mask_vu, float* d_Cd_vu,
                                     float* d_depth_vu, float* d_rbar_vu, int32_t* d_idx_vu, float* d_score_vu,
                                     float* d_depth_raw_vu, rslf_stats* stats)
{
    if (!ctx || !vol || !d_Ce_vu || !d_Ce_mask_vu || !d_Cd_vu || !d_depth_vu || !d_rbar_vu)
        return fail(RSLF_ERR_INVALID_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(ctx->device));
    s_hat = resolve_s_hat(s_hat, vol->S);
    const size_t n = (size_t)vol->V * vol->U;
    hipStream_t st = ctx->stream;
    // dc.hpp:501-510 (C_e and C_d are uninitialised there; zero is the intended start)
    HIP_TRY(hipMemsetAsync(d_Ce_vu, 0, n * sizeof(float), st));
    HIP_TRY(hipMemsetAsync(d_Cd_vu, 0, n * sizeof(float), st));
    HIP_TRY(hipMemsetAsync(d_depth_vu, 0, n * sizeof(float), st));
    HIP_TRY(hipMemsetAsync(d_rbar_vu, 0, n * vol->C * sizeof(float), st));
    int rc = check_params(p);
    if (rc)
        return rc;
    // dc.hpp:538 + the findNonZero of dc.hpp:547's callee (core.hpp:513-516) in ONE launch when nothing sits between
    // them: no opening of the mask (core.hpp:759-768) and row tiles.  A pile step is then three launches -- edge
    // confidence + compaction, scan, selective median -- and no plane is copied.
    const bool fuse = p->edge_confidence_opening_size <= 1 && ctx->force_packed != 1 && !ctx->scan_packed && vol->filled &&
                      (size_t)vol->V * vol->U <= (size_t)INT32_MAX;
    if (fuse) {
        rc = ensure_plane_scratch(ctx, vol->V, vol->U);
        if (rc)
            return rc;
        EdgeConsts ec;
        ec.filter_size = p->edge_confidence_filter_size;
        ec.cut_shadows = p->cut_shadows;
        ec.shadow_level = p->shadow_level;
        ec.edge_thr = p->edge_score_threshold;
        if (!ctx->keep_total)
            HIP_TRY(hipMemsetAsync(ctx->total, 0, sizeof(unsigned long long), st));
        if (vol->C == 1)
            hipLaunchKernelGGL(k1_edge_confidence_compact<1>, dim3(vol->V), dim3(256), 0, st, view_of(vol), s_hat, ec, d_Ce_vu,
                               d_Ce_mask_vu, ctx->list, ctx->count, ctx->total);
        else
            hipLaunchKernelGGL(k1_edge_confidence_compact<3>, dim3(vol->V), dim3(256), 0, st, view_of(vol), s_hat, ec, d_Ce_vu,
                               d_Ce_mask_vu, ctx->list, ctx->count, ctx->total);
        HIP_TRY(hipGetLastError());
        ctx->precompacted = 1;
    } else {
        rc = rslf_edge_confidence_pile(ctx, vol, s_hat, p, d_Ce_vu, d_Ce_mask_vu);   // dc.hpp:538
        if (rc)
            return rc;
    }
    rc = rslf_depth_epi_pile(ctx, vol, nullptr, nullptr, dmin, dmax, dim_d, s_hat, d_Ce_vu, d_Ce_mask_vu, d_Cd_vu,   // dc.hpp:547
                             d_depth_vu, d_rbar_vu, p, nullptr, d_idx_vu, d_score_vu, d_depth_raw_vu, stats);
    ctx->precompacted = 0;
    return rc;
}

extern "C" int rslf_depth1d_run(rslf_ctx* ctx, const rslf_volume* vol, float dmin, float dmax, int dim_d, int s_hat,
                                const rslf_params* p, float* d_Ce_vu, uint8_t* d_Ce_mask_vu, float* d_Cd_vu, float* d_depth_vu,
                                float* d_rbar_vu, int32_t* d_idx_vu, float* d_score_vu, rslf_stats* stats)
{
    if (!ctx || !vol || !d_Ce_vu || !d_Ce_mask_vu || !d_Cd_vu || !d_depth_vu || !d_rbar_vu)
        return fail(RSLF_ERR_INVALID_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(ctx->device));
    s_hat = resolve_s_hat(s_hat, vol->S);   // dc.hpp:303-311
    const size_t n = (size_t)vol->V * vol->U;
    hipStream_t st = ctx->stream;
    // dc.hpp:313-322: zero-initialised outputs
    HIP_TRY(hipMemsetAsync(d_Ce_vu, 0, n * sizeof(float), st));
    HIP_TRY(hipMemsetAsync(d_Cd_vu, 0, n * sizeof(float), st));
    HIP_TRY(hipMemsetAsync(d_depth_vu, 0, n * sizeof(float), st));
    HIP_TRY(hipMemsetAsync(d_rbar_vu, 0, n * vol->C * sizeof(float), st));
    int rc = rslf_edge_confidence_pile(ctx, vol, s_hat, p, d_Ce_vu, d_Ce_mask_vu);   // dc.hpp:347
    if (rc)
        return rc;
    return rslf_depth_epi_scan(ctx, vol, nullptr, nullptr, dmin, dmax, dim_d, s_hat, d_Ce_vu, d_Ce_mask_vu, d_Cd_vu,   // dc.hpp:356
                               d_depth_vu, d_rbar_vu, p, nullptr, d_idx_vu, d_score_vu, stats);
}

extern "C" int rslf_depth1d_pile_run_host(rslf_ctx* ctx, const rslf_volume* vol, float dmin, float dmax, int dim_d, int s_hat,
                                          const rslf_params* p, float* h_Ce_vu, uint8_t* h_Ce_mask_vu, float* h_Cd_vu,
                                          float* h_depth_vu, float* h_rbar_vu, int32_t* h_idx_vu, float* h_score_vu,
                                          float* h_depth_raw_vu, rslf_stats* stats)
{
    if (!ctx || !vol)
        return fail(RSLF_ERR_INVALID_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t n = (size_t)vol->V * vol->U;
    // one device block: Ce, Cd, depth, raw, score | rbar (n*C) | idx | mask
    const size_t f_planes = 5 + (size_t)vol->C;
    const size_t bytes = n * (f_planes * sizeof(float) + sizeof(int32_t) + 1);
    char* blk = nullptr;
    hipError_t e = hipMalloc(&blk, bytes);
    if (e != hipSuccess)
        return fail(RSLF_ERR_ALLOC, "hipMalloc(%zu) for result planes failed: %s", bytes, hipGetErrorString(e));
    float* d_Ce = (float*)blk;
    float* d_Cd = d_Ce + n;
    float* d_depth = d_Cd + n;
    float* d_raw = d_depth + n;
    float* d_score = d_raw + n;
    float* d_rbar = d_score + n;
    int32_t* d_idx = (int32_t*)(d_rbar + n * vol->C);
    uint8_t* d_mask = (uint8_t*)(d_idx + n);
    int rc = rslf_depth1d_pile_run(ctx, vol, dmin, dmax, dim_d, s_hat, p, d_Ce, d_mask, d_Cd, d_depth, d_rbar, d_idx, d_score,
                                   d_raw, nullptr);
    hipStream_t st = ctx->stream;
    auto pull = [&](void* h, const void* d, size_t b) -> hipError_t {
        return h ? hipMemcpyAsync(h, d, b, hipMemcpyDeviceToHost, st) : hipSuccess;
    };
    if (rc == RSLF_OK) {
        hipError_t ce = pull(h_Ce_vu, d_Ce, n * 4);
        if (ce == hipSuccess) ce = pull(h_Ce_mask_vu, d_mask, n);
        if (ce == hipSuccess) ce = pull(h_Cd_vu, d_Cd, n * 4);
        if (ce == hipSuccess) ce = pull(h_depth_vu, d_depth, n * 4);
        if (ce == hipSuccess) ce = pull(h_rbar_vu, d_rbar, n * 4 * vol->C);
        if (ce == hipSuccess) ce = pull(h_idx_vu, d_idx, n * 4);
        if (ce == hipSuccess) ce = pull(h_score_vu, d_score, n * 4);
        if (ce == hipSuccess) ce = pull(h_depth_raw_vu, d_raw, n * 4);
        if (ce == hipSuccess) ce = hipStreamSynchronize(st);
        if (ce != hipSuccess)
            rc = fail(RSLF_ERR_HIP, "result download failed: %s", hipGetErrorString(ce));
    } else {
        (void)hipStreamSynchronize(st);
    }
    if (rc == RSLF_OK && stats) {
        unsigned long long tot = 0;
        if (hipMemcpy(&tot, ctx->total, sizeof(tot), hipMemcpyDeviceToHost) == hipSuccess) {
            stats->pixels_scanned = (int64_t)tot;
            stats->units = (int64_t)tot * dim_d;
            stats->scan_kernel = ctx->last_kernel;
            stats->s_pad = ctx->last_spad;
        }
    }
    (void)hipFree(blk);
    return rc;
}

// ---- "next" row: the 2-D sweep ----------------------------------------------

extern "C" int rslf_edge_confidence_2d(rslf_ctx* ctx, const rslf_volume* vol, const rslf_params* p, float* d_Ce_svu,
                                       uint8_t* d_Ce_mask_svu)
{
    if (!ctx || !vol || !d_Ce_svu || !d_Ce_mask_svu)
        return fail(RSLF_ERR_INVALID_ARG, "NULL argument");
    const size_t n = (size_t)vol->V * vol->U;
    if (p && p->edge_confidence_opening_size <= 1 && vol->S <= 65535 && vol->V <= 65535) {   // every view in one launch
        int rc = check_params(p);
        if (rc)
            return rc;
        if (!vol->filled)
            return fail(RSLF_ERR_INVALID_ARG, "volume has not been filled");
        HIP_TRY(hipSetDevice(ctx->device));
        EdgeConsts ec;
        ec.filter_size = p->edge_confidence_filter_size;
        ec.cut_shadows = p->cut_shadows;
        ec.shadow_level = p->shadow_level;
        ec.edge_thr = p->edge_score_threshold;
        const dim3 grid((vol->U + 255) / 256, vol->V, vol->S);
        if (vol->C == 1)
            hipLaunchKernelGGL(k1_edge_confidence_views<1>, grid, dim3(256), 0, ctx->stream, view_of(vol), ec, d_Ce_svu, d_Ce_mask_svu);
        else
            hipLaunchKernelGGL(k1_edge_confidence_views<3>, grid, dim3(256), 0, ctx->stream, view_of(vol), ec, d_Ce_svu, d_Ce_mask_svu);
        HIP_TRY(hipGetLastError());
        return RSLF_OK;
    }
    for (int s = 0; s < vol->S; s++) {   // core.hpp:918-934
        int rc = rslf_edge_confidence_pile(ctx, vol, s, p, d_Ce_svu + (size_t)s * n, d_Ce_mask_svu + (size_t)s * n);
        if (rc)
            return rc;
    }
    return RSLF_OK;
}

// Two capacities: winner / running mask hold S*V*U entries, the median plane V*U.  (A single S*V*U
// capacity once let a later volume with fewer views but larger planes overrun the plane: found by
// tools/fuzz_sweep.py.)
static int ensure_sweep_scratch(rslf_ctx* ctx, const rslf_volume* vol)
{
    const size_t n = (size_t)vol->S * vol->V * vol->U;
    if (n > ctx->sweep_cap) {
        (void)hipFree(ctx->winner);
        (void)hipFree(ctx->sweep_mask);
        ctx->winner = nullptr;
        ctx->sweep_mask = nullptr;
        ctx->sweep_cap = 0;
        HIP_TRY(hipMalloc(&ctx->winner, n * sizeof(int)));
        HIP_TRY(hipMalloc(&ctx->sweep_mask, n));
        ctx->sweep_cap = n;
        // every claim pass is undone by its apply pass, so one fill lasts
        HIP_TRY(hipMemsetAsync(ctx->winner, 0x7F, n * sizeof(int), ctx->stream));
    }
    const size_t flags = (size_t)vol->S * vol->V * ((vol->U + 255) / 256);
    if (flags > ctx->dirty_cap) {
        (void)hipFree(ctx->dirty);
        ctx->dirty = nullptr;
        ctx->dirty_cap = 0;
        HIP_TRY(hipMalloc(&ctx->dirty, flags));
        ctx->dirty_cap = flags;
        HIP_TRY(hipMemsetAsync(ctx->dirty, 0, flags, ctx->stream));   // every apply pass leaves them at 0 again
    }
    const size_t plane = (size_t)vol->V * vol->U;
    if (plane > ctx->sweep_plane_cap) {
        (void)hipFree(ctx->filtered);
        ctx->filtered = nullptr;
        ctx->sweep_plane_cap = 0;
        HIP_TRY(hipMalloc(&ctx->filtered, plane * sizeof(float)));
        ctx->sweep_plane_cap = plane;
    }
    return RSLF_OK;
}

// The sweep one visit at a time (rslf_sweep_*), and rslf_depth_epi_2d on top of it.  The launch shape of the visits
// (hypothesis groups, packed tiles, running total) is context state the scan reads: rslf_sweep_end restores it, and after
// an error the winners are refilled on the next sweep (a claim pass whose apply never ran leaves them set).
// core.hpp:981-990: the centre view, then outwards, alternating (an even view count never reaches view 0)
static std::vector<int> sweep_order(int S)
{
    std::vector<int> order;
    const int s_mid = (int)std::floor(S / 2.0);
    order.push_back(s_mid);
    for (int off = 1; off < S - s_mid; off++) {
        order.push_back(s_mid + off);
        if (s_mid - off > -1)
            order.push_back(s_mid - off);
    }
    return order;
}

static int sweep_view_after(int S, int s_hat)
{
    const std::vector<int> order = sweep_order(S);
    for (size_t i = 0; i + 1 < order.size(); i++)
        if (order[i] == s_hat)
            return order[i + 1];
    return -1;
}

static void sweep_close(rslf_ctx* ctx, bool ok)
{
    ctx->keep_total = false;
    ctx->scan_groups = 1;
    ctx->scan_packed = false;
    ctx->packed_n_clean = false;
    ctx->precompacted = 0;
    ctx->sweep_expect = -1;
    if (!ok) {
        ctx->sweep_cap = 0;   // claims without their apply pass may be left behind: fresh winners and flags next time
        ctx->dirty_cap = 0;
    }
    ctx->sweep_open = false;
}

extern "C" int rslf_sweep_begin(rslf_ctx* ctx, const rslf_volume* vol, const uint8_t* d_Ce_mask_svu, uint8_t* d_scan_mask_svu,
                                int dim_d, int v_lo, int v_hi)
{
    if (!ctx || !vol || !d_Ce_mask_svu)
        return fail(RSLF_ERR_INVALID_ARG, "NULL argument");
    if (v_lo < 0 || v_hi > vol->V || v_lo >= v_hi)
        return fail(RSLF_ERR_INVALID_ARG, "active scanlines [%d, %d) outside the volume's %d", v_lo, v_hi, vol->V);
    HIP_TRY(hipSetDevice(ctx->device));
    if (ctx->sweep_open)
        sweep_close(ctx, false);   // a sweep left open by a caller's error path
    int rc = ensure_sweep_scratch(ctx, vol);
    if (rc)
        return rc;
    const int S = vol->S, V = vol->V, U = vol->U;
    const size_t n = (size_t)V * U;
    hipStream_t st = ctx->stream;
    uint8_t* mask_svu = d_scan_mask_svu ? d_scan_mask_svu : ctx->sweep_mask;
    // core.hpp:958-965: running masks start as clones of the edge masks ...
    HIP_TRY(hipMemcpyAsync(mask_svu, d_Ce_mask_svu, (size_t)S * n, hipMemcpyDeviceToDevice, st));
    // ... except on halo scanlines (a sharded sweep): never scanned, never painted here -- their owner does both
    if (v_lo > 0)
        HIP_TRY(hipMemset2DAsync(mask_svu, n, 0, (size_t)v_lo * U, S, st));
    if (v_hi < V)
        HIP_TRY(hipMemset2DAsync(mask_svu + (size_t)v_hi * U, n, 0, (size_t)(V - v_hi) * U, S, st));
    HIP_TRY(hipMemsetAsync(ctx->total, 0, sizeof(unsigned long long), st));
    {   // the sparse visits' records, sized before the first visit (no allocation in the middle of the sequence)
        int g = kSweepGroups;
        while (g > 1 && dim_d < 2 * kScanWaves * g)
            g /= 2;
        // (the same choice of kernel as rslf_depth_epi_scan makes for linear interpolation without debug hooks; should it
        // differ, that call sizes the records itself)
        const bool in_range = vol->min_value >= 0.0f && vol->max_value <= 1.0e6f;
        const bool stream = in_range && pick_spad(S, vol->C) == 0 && (size_t)kScanWaves * S * sizeof(float) <= (size_t)48 << 10;
        const size_t tiles_all = (n + 63) / 64;
        if (stream)
            while (g > 1 && tiles_all * g * 64 * sizeof(Partial) > kPartialBudget)
                g /= 2;
        if (g > 1 && n <= (size_t)INT32_MAX) {
            rc = stream ? ensure_group_scratch(ctx, tiles_all * g * 64, tiles_all)
                        : ensure_group_scratch(ctx, std::min<size_t>(tiles_all * g, kPackedItemTarget) * 64,
                                               std::min<size_t>(tiles_all, kPackedItemTarget / 2));
            if (rc)
                return rc;
        }
    }
    ctx->keep_total = true;
    ctx->sweep_open = true;
    ctx->sweep_first = true;
    ctx->sweep_mask_run = mask_svu;
    ctx->sweep_expect = sweep_order(S)[0];
    ctx->precompacted = 0;
    return RSLF_OK;
}

extern "C" int rslf_sweep_visit_scan(rslf_ctx* ctx, const rslf_volume* vol, const float* d_dmin_svu, const float* d_dmax_svu,
                                     float dmin, float dmax, int dim_d, int s_hat, float* d_Ce_svu, uint8_t* d_Ce_mask_svu,
                                     float* d_Cd_svu, float* d_depth_svu, float* d_rbar_svu, const rslf_params* p)
{
    if (!ctx || !vol || !d_Ce_svu || !d_Ce_mask_svu || !d_Cd_svu || !d_depth_svu || !d_rbar_svu)
        return fail(RSLF_ERR_INVALID_ARG, "NULL argument");
    if (!ctx->sweep_open)
        return fail(RSLF_ERR_INVALID_ARG, "rslf_sweep_visit_scan without rslf_sweep_begin");
    if ((d_dmin_svu == nullptr) != (d_dmax_svu == nullptr))
        return fail(RSLF_ERR_INVALID_ARG, "d_dmin_svu and d_dmax_svu must both be given or both be NULL");
    if (s_hat < 0 || s_hat >= vol->S)
        return fail(RSLF_ERR_INVALID_ARG, "s_hat=%d outside [0,%d)", s_hat, vol->S);
    if (s_hat != ctx->sweep_expect)   // the previous visit has already listed this view's pixels (k4_propagate_apply)
        return fail(RSLF_ERR_INVALID_ARG, "the sweep visits view %d next (core.hpp:981-990), not %d", ctx->sweep_expect, s_hat);
    const size_t n = (size_t)vol->V * vol->U;
    // After the centre view, propagation has explained most pixels: a visit scans a few per scanline.
    // Pack them into one list and share each tile's hypotheses out over up to kSweepGroups workgroups (k2_scan.hpp).
    ctx->scan_groups = ctx->sweep_first ? 1 : kSweepGroups;
    ctx->scan_packed = !ctx->sweep_first;
    // core.hpp:1012-1028: the pile call is the scan of every EPI followed by the selective median.  In the
    // reference the stored plane keeps the RAW depths and only the local header is rebound to the median
    // (core.hpp:892), which the propagation then paints from: so the scan writes the view's depth plane and the median
    // goes to ctx->filtered (rslf_sweep_visit_finish) -- no plane copies.
    return rslf_depth_epi_scan(ctx, vol, d_dmin_svu ? d_dmin_svu + (size_t)s_hat * n : nullptr,
                               d_dmax_svu ? d_dmax_svu + (size_t)s_hat * n : nullptr, dmin, dmax, dim_d, s_hat,
                               d_Ce_svu + (size_t)s_hat * n, d_Ce_mask_svu + (size_t)s_hat * n, d_Cd_svu + (size_t)s_hat * n,
                               d_depth_svu + (size_t)s_hat * n, d_rbar_svu + (size_t)s_hat * n * vol->C, p,
                               ctx->sweep_mask_run + (size_t)s_hat * n, nullptr, nullptr, nullptr);
}

extern "C" int rslf_sweep_visit_finish(rslf_ctx* ctx, const rslf_volume* vol, int s_hat, uint8_t* d_Ce_mask_svu, float* d_Cd_svu,
                                       float* d_depth_svu, float* d_rbar_svu, const rslf_params* p)
{
    if (!ctx || !vol || !d_Ce_mask_svu || !d_Cd_svu || !d_depth_svu || !d_rbar_svu || !p)
        return fail(RSLF_ERR_INVALID_ARG, "NULL argument");
    if (!ctx->sweep_open)
        return fail(RSLF_ERR_INVALID_ARG, "rslf_sweep_visit_finish without rslf_sweep_begin");
    if (s_hat != ctx->sweep_expect)
        return fail(RSLF_ERR_INVALID_ARG, "rslf_sweep_visit_finish(%d): the open visit is view %d", s_hat, ctx->sweep_expect);
    if (p->median_filter_size < 1 || (p->median_filter_size & 1) == 0 || p->median_filter_size > kMedianMaxSize)
        return fail(RSLF_ERR_UNSUPPORTED, "median size must be odd and <= %d", kMedianMaxSize);
    HIP_TRY(hipSetDevice(ctx->device));
    const int S = vol->S, V = vol->V, U = vol->U, C = vol->C;
    const size_t n = (size_t)V * U;
    hipStream_t st = ctx->stream;
    uint8_t* mask_svu = ctx->sweep_mask_run;
    const dim3 grid_vu((U + 255) / 256, V);
    if ((long long)S * V > (1ll << 31) - 1 || U > 65536)
        return fail(RSLF_ERR_UNSUPPORTED, "%d views x %d scanlines x %d columns: too large for one apply launch", S, V, U);
    // the 5 x 5 window sorts in registers (selective_median_pixel_5x5); other sizes keep their candidates in LDS
    const size_t median_lds = p->median_filter_size == 5 ? 0 : (size_t)p->median_filter_size * p->median_filter_size * 256 * sizeof(float);
    int* packed_n = reinterpret_cast<int*>(ctx->total + 1);
    float* depth = d_depth_svu + (size_t)s_hat * n;
    float* Cd = d_Cd_svu + (size_t)s_hat * n;
    float* rbar = d_rbar_svu + (size_t)s_hat * n * C;
    uint8_t* cem = d_Ce_mask_svu + (size_t)s_hat * n;
    // core.hpp:881-892 (selective median over the edge mask) and :1088-1129 (propagation) -- the median and the
    // claims of a pixel in one launch (k34_median_claim), then the apply pass, which also lists the pixels the NEXT
    // visit scans; a visit is three launches: scan (its groups merge their records themselves), median + claims,
    // apply + compaction
    int s_next = sweep_view_after(S, s_hat);
    const int s_after = s_next;
    if (ctx->force_packed == 0 || n > (size_t)INT32_MAX)
        s_next = -1;   // that scan will not take a packed list: it compacts for itself
    if (C == 1)
        hipLaunchKernelGGL(k34_median_claim<1>, grid_vu, dim3(256), median_lds, st, view_of(vol), s_hat, depth, ctx->filtered, cem,
                           p->median_filter_size, p->median_filter_epsilon, rbar, mask_svu, ctx->winner, ctx->dirty, p->slope_factor,
                           p->propagation_epsilon, p->use_disp_confidence_score ? Cd : nullptr, p->disp_score_threshold, packed_n);
    else
        hipLaunchKernelGGL(k34_median_claim<3>, grid_vu, dim3(256), median_lds, st, view_of(vol), s_hat, depth, ctx->filtered, cem,
                           p->median_filter_size, p->median_filter_epsilon, rbar, mask_svu, ctx->winner, ctx->dirty, p->slope_factor,
                           p->propagation_epsilon, p->use_disp_confidence_score ? Cd : nullptr, p->disp_score_threshold, packed_n);
    HIP_TRY(hipGetLastError());
    const unsigned apply_blocks = (unsigned)((s_next >= 0 ? V : 0) + ((long long)S * V + kApplyRowsPerBlock - 1) / kApplyRowsPerBlock);
    hipLaunchKernelGGL(k4_propagate_apply, dim3(apply_blocks), dim3(256), 0, st, S, V, U, s_hat, ctx->filtered, Cd, d_depth_svu,
                       d_Cd_svu, mask_svu, ctx->winner, ctx->dirty, s_next, s_next >= 0 ? d_Ce_mask_svu + (size_t)s_next * n : nullptr, ctx->list,
                       ctx->count, ctx->total, packed_n);
    HIP_TRY(hipGetLastError());
    ctx->packed_n_clean = s_next < 0;       // k34_median_claim zeroed the packed list's length; a listing apply pass set it again
    ctx->precompacted = s_next >= 0 ? 2 : 0;
    ctx->sweep_expect = s_after;
    ctx->sweep_first = false;
    return RSLF_OK;
}

extern "C" int rslf_sweep_end(rslf_ctx* ctx, int ok, int dim_d, rslf_stats* stats)
{
    if (!ctx)
        return fail(RSLF_ERR_INVALID_ARG, "ctx is NULL");
    const bool was_open = ctx->sweep_open;
    sweep_close(ctx, ok != 0 && was_open);
    if (ok && was_open && stats) {
        unsigned long long tot = 0;
        HIP_TRY(hipSetDevice(ctx->device));
        HIP_TRY(hipMemcpyAsync(&tot, ctx->total, sizeof(tot), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        stats->pixels_scanned = (int64_t)tot;
        stats->units = (int64_t)tot * dim_d;
        stats->scan_kernel = ctx->last_kernel;
        stats->s_pad = ctx->last_spad;
    }
    return RSLF_OK;
}

extern "C" int rslf_depth_epi_2d(rslf_ctx* ctx, const rslf_volume* vol, const float* d_dmin_svu, const float* d_dmax_svu,
                                 float dmin, float dmax, int dim_d, float* d_Ce_svu, uint8_t* d_Ce_mask_svu, float* d_Cd_svu,
                                 float* d_depth_svu, float* d_rbar_svu, const rslf_params* p, uint8_t* d_scan_mask_svu,
                                 rslf_stats* stats)
{
    if (!ctx || !vol || !d_Ce_svu || !d_Ce_mask_svu || !d_Cd_svu || !d_depth_svu || !d_rbar_svu)
        return fail(RSLF_ERR_INVALID_ARG, "NULL argument");
    int rc = check_params(p);
    if (rc)
        return rc;
    rc = rslf_sweep_begin(ctx, vol, d_Ce_mask_svu, d_scan_mask_svu, dim_d, 0, vol->V);
    if (rc)
        return rc;
    for (int s_hat : sweep_order(vol->S)) {   // core.hpp:981-990
        rc = rslf_sweep_visit_scan(ctx, vol, d_dmin_svu, d_dmax_svu, dmin, dmax, dim_d, s_hat, d_Ce_svu, d_Ce_mask_svu, d_Cd_svu,
                                   d_depth_svu, d_rbar_svu, p);
        if (!rc)
            rc = rslf_sweep_visit_finish(ctx, vol, s_hat, d_Ce_mask_svu, d_Cd_svu, d_depth_svu, d_rbar_svu, p);
        if (rc) {
            const std::string msg = g_err;
            (void)rslf_sweep_end(ctx, 0, dim_d, nullptr);
            return fail(rc, "%s", msg.c_str());
        }
    }
    return rslf_sweep_end(ctx, 1, dim_d, stats);
}

extern "C" int rslf_depth2d_run(rslf_ctx* ctx, const rslf_volume* vol, float dmin, float dmax, int dim_d, const rslf_params* p,
                                float* d_Ce_svu, uint8_t* d_Ce_mask_svu, float* d_Cd_svu, float* d_depth_svu, float* d_rbar_svu,
                                uint8_t* d_scan_mask_svu, rslf_stats* stats)
{
    if (!ctx || !vol || !d_Ce_svu || !d_Ce_mask_svu || !d_Cd_svu || !d_depth_svu || !d_rbar_svu)
        return fail(RSLF_ERR_INVALID_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t n = (size_t)vol->S * vol->V * vol->U;
    hipStream_t st = ctx->stream;
    // dc.hpp:733-750 (C_e and C_d are uninitialised there; zero is the intended start)
    HIP_TRY(hipMemsetAsync(d_Ce_svu, 0, n * sizeof(float), st));
    HIP_TRY(hipMemsetAsync(d_Cd_svu, 0, n * sizeof(float), st));
    HIP_TRY(hipMemsetAsync(d_depth_svu, 0, n * sizeof(float), st));
    HIP_TRY(hipMemsetAsync(d_rbar_svu, 0, n * vol->C * sizeof(float), st));
    int rc = rslf_edge_confidence_2d(ctx, vol, p, d_Ce_svu, d_Ce_mask_svu);   // dc.hpp:772
    if (rc)
        return rc;
    return rslf_depth_epi_2d(ctx, vol, nullptr, nullptr, dmin, dmax, dim_d, d_Ce_svu, d_Ce_mask_svu, d_Cd_svu, d_depth_svu,   // dc.hpp:780
                             d_rbar_svu, p, d_scan_mask_svu, stats);
}

// ---- "next" row: fine-to-coarse ------------------------------------------------

extern "C" int rslf_f2c_level_dims(int V, int U, int* V2, int* U2)
{
    if (!V2 || !U2 || V < 1 || U < 1)
        return fail(RSLF_ERR_INVALID_ARG, "bad arguments");
    *V2 = (int)std::lrint(V * 0.5);   // cvRound: ties to even
    *U2 = (int)std::lrint(U * 0.5);
    return RSLF_OK;
}

namespace {
struct DevBuf {   // scoped device scratch for the once-per-level helpers
    void* p = nullptr;
    ~DevBuf() { (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
    void release()
    {
        (void)hipFree(p);
        p = nullptr;
    }
};
}  // namespace

extern "C" int rslf_downsample_epis_f32(rslf_ctx* ctx, const float* d_in_vsuc, int V, int S, int U, int C, float* d_out_vsuc)
{
    if (!ctx || !d_in_vsuc || !d_out_vsuc || V < 1 || S < 1 || U < 1 || (C != 1 && C != 3))
        return fail(RSLF_ERR_INVALID_ARG, "bad arguments");
    HIP_TRY(hipSetDevice(ctx->device));
    int V2, U2;
    rslf_f2c_level_dims(V, U, &V2, &U2);
    if (V2 < 1 || U2 < 1)
        return fail(RSLF_ERR_INVALID_ARG, "level too small to halve");
    void* tmp_p = nullptr;
    int rc = helper_scratch(ctx, 0, (size_t)V * S * U * C * sizeof(float), &tmp_p);
    if (rc)
        return rc;
    hipStream_t st = ctx->stream;
    const long long row_blocks = (long long)V * S * ((U * C + 255) / 256);
    if (row_blocks > (1ll << 31) - 1)
        return fail(RSLF_ERR_UNSUPPORTED, "volume too large for one downsampling launch");
    hipLaunchKernelGGL(k5_gauss_rows, dim3((unsigned)row_blocks), dim3(256), 0, st, d_in_vsuc, (float*)tmp_p, (long long)V * S, U, C);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(k5_gauss_cols_halve, dim3((U2 * C + 255) / 256, S, V2), dim3(256), 0, st, (const float*)tmp_p, d_out_vsuc,
                       V, S, U, C, V2, U2);
    HIP_TRY(hipGetLastError());
    return RSLF_OK;   // enqueued on the context's stream like every device entry point
}

extern "C" int rslf_downsample_epis_u8(rslf_ctx* ctx, const float* d_in_vsuc, int V, int S, int U, int C, float* d_out_vsuc)
{
    if (!ctx || !d_in_vsuc || !d_out_vsuc || V < 1 || S < 1 || U < 1 || (C != 1 && C != 3))
        return fail(RSLF_ERR_INVALID_ARG, "bad arguments");
    HIP_TRY(hipSetDevice(ctx->device));
    int V2, U2;
    rslf_f2c_level_dims(V, U, &V2, &U2);
    if (V2 < 1 || U2 < 1)
        return fail(RSLF_ERR_INVALID_ARG, "level too small to halve");
    void* tmp_p = nullptr;
    int rc = helper_scratch(ctx, 0, (size_t)V * S * U * C * sizeof(int), &tmp_p);
    if (rc)
        return rc;
    hipStream_t st = ctx->stream;
    const long long row_blocks = (long long)V * S * ((U * C + 255) / 256);
    if (row_blocks > (1ll << 31) - 1)
        return fail(RSLF_ERR_UNSUPPORTED, "volume too large for one downsampling launch");
    hipLaunchKernelGGL(k5_gauss_rows_u8, dim3((unsigned)row_blocks), dim3(256), 0, st, d_in_vsuc, (int*)tmp_p, (long long)V * S, U, C);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(k5_gauss_cols_halve_u8, dim3((U2 * C + 255) / 256, S, V2), dim3(256), 0, st, (const int*)tmp_p, d_out_vsuc,
                       V, S, U, C, V2, U2);
    HIP_TRY(hipGetLastError());
    return RSLF_OK;
}

extern "C" int rslf_device_max_f32(rslf_ctx* ctx, const float* d_values, size_t n, float* h_max)
{
    if (!ctx || !d_values || !h_max || n == 0)
        return fail(RSLF_ERR_INVALID_ARG, "bad arguments");
    HIP_TRY(hipSetDevice(ctx->device));
    const int blocks = (int)std::min<size_t>((n + 255) / 256, 2048);
    void* part_p = nullptr;
    int rc = helper_scratch(ctx, 1, (size_t)2048 * sizeof(float), &part_p);
    if (rc)
        return rc;
    hipLaunchKernelGGL(k5_max_partial, dim3(blocks), dim3(256), 0, ctx->stream, d_values, (long long)n, (float*)part_p);
    HIP_TRY(hipGetLastError());
    std::vector<float> h(blocks);
    HIP_TRY(hipMemcpyAsync(h.data(), part_p, (size_t)blocks * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    float m = h[0];
    for (int i = 1; i < blocks; i++)
        m = std::max(m, h[i]);
    *h_max = m;
    return RSLF_OK;
}

extern "C" int rslf_f2c_tighten_bounds(rslf_ctx* ctx, const float* d_depth_up_svu, const uint8_t* d_valid_up_svu, int S, int V_up,
                                       int U_up, float* d_dmin_down_svu, float* d_dmax_down_svu, int V_down, int U_down)
{
    if (!ctx || !d_depth_up_svu || !d_valid_up_svu || !d_dmin_down_svu || !d_dmax_down_svu || S < 1 || V_up < 1 || U_up < 1 ||
        V_down < 1 || U_down < 1)
        return fail(RSLF_ERR_INVALID_ARG, "bad arguments");
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t n_up = (size_t)S * V_up * U_up;
    void *left_p = nullptr, *right_p = nullptr;
    int rc = helper_scratch(ctx, 2, n_up * sizeof(int), &left_p);
    if (!rc)
        rc = helper_scratch(ctx, 3, n_up * sizeof(int), &right_p);
    if (rc)
        return rc;
    hipStream_t st = ctx->stream;
    const long long rows = (long long)S * V_up;
    hipLaunchKernelGGL(k5_nearest_valid, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, d_valid_up_svu, rows, U_up,
                       (int*)left_p, (int*)right_p);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(k5_tighten, dim3((U_down + 255) / 256, V_down, S), dim3(256), 0, st, d_depth_up_svu, (const int*)left_p,
                       (const int*)right_p, S, V_up, U_up, d_dmin_down_svu, d_dmax_down_svu, V_down, U_down);
    HIP_TRY(hipGetLastError());
    return RSLF_OK;   // enqueued, not awaited
}

extern "C" int rslf_f2c_fuse(rslf_ctx* ctx, const float* const* d_disp, const uint8_t* const* d_valid, const int* Vp, const int* Up,
                             int P, int S, float* d_out_map_svu, uint8_t* d_out_valid_svu)
{
    if (!ctx || !d_disp || !d_valid || !Vp || !Up || P < 1 || S < 1 || !d_out_map_svu || !d_out_valid_svu)
        return fail(RSLF_ERR_INVALID_ARG, "bad arguments");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const size_t n0 = (size_t)S * Vp[0] * Up[0];
    // two ping-pong buffers at the finest size hold the running map / mask of every step
    void *mapA = nullptr, *mapB = nullptr, *mskA = nullptr, *mskB = nullptr;
    int rc = helper_scratch(ctx, 0, n0 * sizeof(float), &mapA);
    if (!rc)
        rc = helper_scratch(ctx, 2, n0 * sizeof(float), &mapB);
    if (!rc)
        rc = helper_scratch(ctx, 1, std::max<size_t>(n0, 2048 * sizeof(float)), &mskA);
    if (!rc)
        rc = helper_scratch(ctx, 3, n0, &mskB);
    if (rc)
        return rc;
    float* map_down = (float*)mapA;
    float* map_next = (float*)mapB;
    uint8_t* msk_down = (uint8_t*)mskA;
    uint8_t* msk_next = (uint8_t*)mskB;
    const size_t nl = (size_t)S * Vp[P - 1] * Up[P - 1];
    HIP_TRY(hipMemcpyAsync(map_down, d_disp[P - 1], nl * sizeof(float), hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(msk_down, d_valid[P - 1], nl, hipMemcpyDeviceToDevice, st));
    for (int p = P - 1; p > 0; p--) {   // fine_to_coarse_core.cpp:98-123
        const int R = Vp[p], W = Up[p], R2 = Vp[p - 1], W2 = Up[p - 1];
        hipLaunchKernelGGL(k5_fuse_step, dim3((W2 + 255) / 256, R2, S), dim3(256), 0, st, map_down, msk_down, R, W, d_disp[p - 1],
                           d_valid[p - 1], map_next, msk_next, R2, W2);
        HIP_TRY(hipGetLastError());
        std::swap(map_down, map_next);
        std::swap(msk_down, msk_next);
    }
    hipLaunchKernelGGL(k5_median3, dim3((Up[0] + 255) / 256, Vp[0], S), dim3(256), 0, st, map_down, d_out_map_svu, Vp[0], Up[0]);   // :127
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(d_out_valid_svu, msk_down, n0, hipMemcpyDeviceToDevice, st));
    return RSLF_OK;   // enqueued, not awaited
}

// ---- host-pointer forms of the rows around the path -----------------------------

extern "C" int rslf_depth2d_run_host(rslf_ctx* ctx, const rslf_volume* vol, float dmin, float dmax, int dim_d, const rslf_params* p,
                                     float* h_Ce_svu, uint8_t* h_Ce_mask_svu, float* h_Cd_svu, float* h_depth_svu,
                                     float* h_rbar_svu, rslf_stats* stats)
{
    if (!ctx || !vol)
        return fail(RSLF_ERR_INVALID_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t n = (size_t)vol->S * vol->V * vol->U;
    DevBuf Ce, Cd, depth, rbar, mask;
    HIP_TRY(Ce.alloc(n * 4));
    HIP_TRY(Cd.alloc(n * 4));
    HIP_TRY(depth.alloc(n * 4));
    HIP_TRY(rbar.alloc(n * 4 * vol->C));
    HIP_TRY(mask.alloc(n));
    int rc = rslf_depth2d_run(ctx, vol, dmin, dmax, dim_d, p, (float*)Ce.p, (uint8_t*)mask.p, (float*)Cd.p, (float*)depth.p,
                              (float*)rbar.p, nullptr, stats);
    if (rc)
        return rc;
    hipStream_t st = ctx->stream;
    if (h_Ce_svu) HIP_TRY(hipMemcpyAsync(h_Ce_svu, Ce.p, n * 4, hipMemcpyDeviceToHost, st));
    if (h_Ce_mask_svu) HIP_TRY(hipMemcpyAsync(h_Ce_mask_svu, mask.p, n, hipMemcpyDeviceToHost, st));
    if (h_Cd_svu) HIP_TRY(hipMemcpyAsync(h_Cd_svu, Cd.p, n * 4, hipMemcpyDeviceToHost, st));
    if (h_depth_svu) HIP_TRY(hipMemcpyAsync(h_depth_svu, depth.p, n * 4, hipMemcpyDeviceToHost, st));
    if (h_rbar_svu) HIP_TRY(hipMemcpyAsync(h_rbar_svu, rbar.p, n * 4 * vol->C, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return RSLF_OK;
}

namespace {
// One pyramid level of rslf::FineToCoarse: its Depth2DComputer's volume and result planes.
struct F2cLevel {
    rslf_volume* vol = nullptr;
    int V = 0, U = 0;
    DevBuf Ce, Cd, depth, rbar, mask, valid, dmin, dmax;
    rslf_params params;
    ~F2cLevel() { rslf_volume_destroy(vol); }
};

__global__ __launch_bounds__(256) void k_u8_to_f32(const uint8_t* __restrict__ in, float* __restrict__ out, long long n)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = (float)in[i];
}
__global__ __launch_bounds__(256) void k_fill_f32(float* __restrict__ out, long long n, float value)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = value;
}
// get_valid_depths_mask_s_v_u (dc.hpp:893-915, default build): C_e > thr, or everything with accept_all
__global__ __launch_bounds__(256) void k_valid_mask(const float* __restrict__ Ce, uint8_t* __restrict__ out, long long n, float thr)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = (Ce[i] > thr) ? 255 : 0;
}
inline unsigned stream_blocks(size_t n) { return (unsigned)std::min<size_t>((n + 255) / 256, 8192); }
}  // namespace

extern "C" int rslf_fine_to_coarse_run_host(rslf_ctx* ctx, const void* const* h_epis, int is_u8, int V, int S, int U, int C,
                                            size_t row_stride_bytes, float d_min, float d_max, int dim_d, float epi_scale_factor,
                                            const rslf_params* p, int max_pyr_depth, int accept_all_last_scale,
                                            float* h_out_map_svu, uint8_t* h_out_valid_svu, int* n_levels, rslf_stats* stats)
{
    if (!ctx || !h_epis || !h_out_map_svu || !h_out_valid_svu || V < 1 || S < 1 || U < 1 || (C != 1 && C != 3))
        return fail(RSLF_ERR_INVALID_ARG, "bad arguments");
    int rc = check_params(p);
    if (rc)
        return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const size_t elem = is_u8 ? 1 : 4;
    const size_t row_bytes = (size_t)U * C * elem;
    if (row_stride_bytes == 0)
        row_stride_bytes = row_bytes;

    // the raw (un-normalised) finest level as a dense float volume [V][S][U][C]
    DevBuf raw;
    HIP_TRY(raw.alloc((size_t)V * S * U * C * sizeof(float)));
    {
        DevBuf stage;
        void* dst = raw.p;
        if (is_u8) {
            HIP_TRY(stage.alloc((size_t)V * S * row_bytes));
            dst = stage.p;
        }
        for (int v = 0; v < V; v++) {
            if (!h_epis[v])
                return fail(RSLF_ERR_INVALID_ARG, "h_epis[%d] is NULL", v);
            if (row_stride_bytes == row_bytes)   // dense rows: one run of bytes per EPI (upload_host)
                HIP_TRY(hipMemcpyAsync((char*)dst + (size_t)v * S * row_bytes, h_epis[v], (size_t)S * row_bytes, hipMemcpyHostToDevice, st));
            else
                HIP_TRY(hipMemcpy2DAsync((char*)dst + (size_t)v * S * row_bytes, row_bytes, h_epis[v], row_stride_bytes, row_bytes, S,
                                         hipMemcpyHostToDevice, st));
        }
        if (is_u8) {
            const size_t n = (size_t)V * S * U * C;
            hipLaunchKernelGGL(k_u8_to_f32, dim3(stream_blocks(n)), dim3(256), 0, st, (const uint8_t*)stage.p, (float*)raw.p,
                               (long long)n);
            HIP_TRY(hipGetLastError());
        }
        HIP_TRY(hipStreamSynchronize(st));
    }

    // constructor: rslf_fine_to_coarse.hpp:103-159
    if (max_pyr_depth < 1)
        max_pyr_depth = 1 << 30;
    std::vector<F2cLevel*> levels;
    struct Cleanup {
        std::vector<F2cLevel*>& l;
        ~Cleanup() { for (F2cLevel* x : l) delete x; }
    } cleanup{levels};
    const int start_dim_u = U;
    int dim_v = V, dim_u = U, counter = 0;
    DevBuf cur;           // raw volume of the level being built (level 0 borrows `raw`)
    float* cur_p = (float*)raw.p;
    while (dim_v > 10 && dim_u > 10 && counter < max_pyr_depth) {   // _MIN_SPATIAL_DIM, f2c.hpp:8, :130
        counter++;
        F2cLevel* lv = new (std::nothrow) F2cLevel();
        if (!lv)
            return fail(RSLF_ERR_ALLOC, "out of host memory");
        levels.push_back(lv);
        lv->V = dim_v;
        lv->U = dim_u;
        lv->params = *p;
        lv->params.slope_factor = (float)((0.0 + dim_u) / start_dim_u);   // f2c.hpp:139
        float scale = 255.0f;                                              // dc.hpp:696-699 (uchar)
        if (!is_u8) {
            scale = epi_scale_factor;
            if (scale < 0) {                                               // dc.hpp:671-690: this level's own max
                rc = rslf_device_max_f32(ctx, cur_p, (size_t)dim_v * S * dim_u * C, &scale);
                if (rc)
                    return rc;
            }
        }
        rc = rslf_volume_create(ctx, dim_v, S, dim_u, C, &lv->vol);
        if (rc)
            return rc;
        rc = rslf_volume_pack_device_f32(lv->vol, cur_p, scale, nullptr);
        if (rc)
            return rc;
        int v2, u2;
        rslf_f2c_level_dims(dim_v, dim_u, &v2, &u2);
        if (v2 < 1 || u2 < 1)
            break;
        DevBuf next;                                                       // f2c.hpp:145-147: the RAW EPIs go down
        HIP_TRY(next.alloc((size_t)v2 * S * u2 * C * sizeof(float)));
        // uchar EPIs go down in uchar arithmetic, as the reference's CV_8U Mats do (fine_to_coarse_core.cpp:22-41)
        rc = is_u8 ? rslf_downsample_epis_u8(ctx, cur_p, dim_v, S, dim_u, C, (float*)next.p)
                   : rslf_downsample_epis_f32(ctx, cur_p, dim_v, S, dim_u, C, (float*)next.p);
        if (rc)
            return rc;
        std::swap(cur.p, next.p);   // `next` now frees the previous level's raw copy
        cur_p = (float*)cur.p;
        dim_v = v2;
        dim_u = u2;
    }
    if (levels.empty())
        return fail(RSLF_ERR_INVALID_ARG, "light field %dx%d is not larger than _MIN_SPATIAL_DIM: no pyramid level", V, U);
    const int P = (int)levels.size();

    // run(): rslf_fine_to_coarse.hpp:171-299
    int64_t pixels = 0;
    rslf_stats st1;
    for (int l = 0; l < P; l++) {
        F2cLevel& lv = *levels[l];
        const size_t n = (size_t)S * lv.V * lv.U;
        HIP_TRY(lv.Ce.alloc(n * 4));
        HIP_TRY(lv.Cd.alloc(n * 4));
        HIP_TRY(lv.depth.alloc(n * 4));
        HIP_TRY(lv.rbar.alloc(n * 4 * C));
        HIP_TRY(lv.mask.alloc(n));
        HIP_TRY(lv.valid.alloc(n));
        if (l == 0) {
            rc = rslf_depth2d_run(ctx, lv.vol, d_min, d_max, dim_d, &lv.params, (float*)lv.Ce.p, (uint8_t*)lv.mask.p, (float*)lv.Cd.p,
                                  (float*)lv.depth.p, (float*)lv.rbar.p, nullptr, &st1);
        } else {
            F2cLevel& up = *levels[l - 1];
            HIP_TRY(lv.dmin.alloc(n * 4));
            HIP_TRY(lv.dmax.alloc(n * 4));
            hipLaunchKernelGGL(k_fill_f32, dim3(stream_blocks(n)), dim3(256), 0, st, (float*)lv.dmin.p, (long long)n, d_min);
            hipLaunchKernelGGL(k_fill_f32, dim3(stream_blocks(n)), dim3(256), 0, st, (float*)lv.dmax.p, (long long)n, d_max);
            HIP_TRY(hipGetLastError());
            rc = rslf_f2c_tighten_bounds(ctx, (const float*)up.depth.p, (const uint8_t*)up.valid.p, S, up.V, up.U, (float*)lv.dmin.p,
                                         (float*)lv.dmax.p, lv.V, lv.U);
            if (rc)
                return rc;
            HIP_TRY(hipMemsetAsync(lv.Ce.p, 0, n * 4, st));
            HIP_TRY(hipMemsetAsync(lv.Cd.p, 0, n * 4, st));
            HIP_TRY(hipMemsetAsync(lv.depth.p, 0, n * 4, st));
            HIP_TRY(hipMemsetAsync(lv.rbar.p, 0, n * 4 * C, st));
            rc = rslf_edge_confidence_2d(ctx, lv.vol, &lv.params, (float*)lv.Ce.p, (uint8_t*)lv.mask.p);
            if (rc)
                return rc;
            rc = rslf_depth_epi_2d(ctx, lv.vol, (const float*)lv.dmin.p, (const float*)lv.dmax.p, d_min, d_max, dim_d,
                                   (float*)lv.Ce.p, (uint8_t*)lv.mask.p, (float*)lv.Cd.p, (float*)lv.depth.p, (float*)lv.rbar.p,
                                   &lv.params, nullptr, &st1);
        }
        if (rc)
            return rc;
        pixels += st1.pixels_scanned;
        // get_valid_depths_mask_s_v_u: the last level accepts everything when asked to (f2c.hpp:157-158)
        const bool all = accept_all_last_scale && l == P - 1;
        hipLaunchKernelGGL(k_valid_mask, dim3(stream_blocks(n)), dim3(256), 0, st, (const float*)lv.Ce.p, (uint8_t*)lv.valid.p,
                           (long long)n, all ? -1.0f : p->edge_score_threshold);
        HIP_TRY(hipGetLastError());
    }

    // get_results(): rslf_fine_to_coarse.hpp:302-324
    std::vector<const float*> dp(P);
    std::vector<const uint8_t*> vp(P);
    std::vector<int> Vp(P), Up(P);
    for (int l = 0; l < P; l++) {
        dp[l] = (const float*)levels[l]->depth.p;
        vp[l] = (const uint8_t*)levels[l]->valid.p;
        Vp[l] = levels[l]->V;
        Up[l] = levels[l]->U;
    }
    const size_t n0 = (size_t)S * V * U;
    DevBuf omap, ovalid;
    HIP_TRY(omap.alloc(n0 * 4));
    HIP_TRY(ovalid.alloc(n0));
    rc = rslf_f2c_fuse(ctx, dp.data(), vp.data(), Vp.data(), Up.data(), P, S, (float*)omap.p, (uint8_t*)ovalid.p);
    if (rc)
        return rc;
    HIP_TRY(hipMemcpyAsync(h_out_map_svu, omap.p, n0 * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(h_out_valid_svu, ovalid.p, n0, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (n_levels)
        *n_levels = P;
    if (stats) {
        *stats = st1;
        stats->pixels_scanned = pixels;
        stats->units = pixels * dim_d;
    }
    return RSLF_OK;
}

extern "C" int rslf_last_scan_kernel_ms(rslf_ctx* ctx, float* ms)
{
    if (!ctx || !ms)
        return fail(RSLF_ERR_INVALID_ARG, "NULL argument");
    if (!ctx->ev_valid)
        return fail(RSLF_ERR_INVALID_ARG, "no scan kernel has been launched on this context");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipEventSynchronize(ctx->ev1));
    HIP_TRY(hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
    return RSLF_OK;
}

// ---- host pointers in, host planes out: pipelined, over one or several devices ------------------------------
//
// Depth1DComputer_pile's constructor + run() + getters (dc.hpp:425-565) in ONE call on host buffers (cv::Mat::data in,
// cv::Mat::data out).  Scanlines are independent up to the median's halo (DESIGN.md "Multi-GPU"), so the V EPIs are cut
// into blocks, one per device, and every block into chunks; a chunk is computed with `halo` recomputed rows either
// side and only its own rows are copied out, straight to their place in the caller's planes -- no collective.  Per
// device one host thread keeps three things in flight: the kernels of chunk k, the upload of chunk k+1 and the download
// of chunk k-1 (two volumes and two sets of result planes; copies to and from pageable host memory hold the host
// thread, never the GPU).  The result is bit-identical to the one-volume run.

#include <mutex>
#include <string>
#include <thread>

struct rslf_multi {
    struct Dev {
        rslf_ctx* ctx = nullptr;
        hipStream_t s_up = nullptr, s_comp = nullptr, s_down = nullptr;
        hipEvent_t done[2] = {nullptr, nullptr};
        rslf_volume* vol[2] = {nullptr, nullptr};
        int vol_rows[2] = {0, 0}, vol_S = 0, vol_U = 0, vol_C = 0;
        char* planes[2] = {nullptr, nullptr};
        size_t planes_cap = 0;
        char* pin[2] = {nullptr, nullptr};   // pinned host staging for EPIs scattered over the heap (Vec<Mat>)
        size_t pin_cap = 0;
        char* arena = nullptr;               // the sweep forms' planes, kept from call to call (and from level to level)
        size_t arena_cap = 0;
    };
    std::vector<Dev> devs;
    int chunk_rows = 0;   // 0 = automatic
};

static void multi_free_dev(rslf_multi::Dev& d)
{
    if (!d.ctx)
        return;
    (void)hipSetDevice(d.ctx->device);
    for (int i = 0; i < 2; i++) {
        if (d.vol[i])
            (void)rslf_volume_destroy(d.vol[i]);
        (void)hipFree(d.planes[i]);
        (void)hipHostFree(d.pin[i]);
        if (i == 0)
            (void)hipFree(d.arena);
        if (d.done[i])
            (void)hipEventDestroy(d.done[i]);
    }
    if (d.s_up)
        (void)hipStreamDestroy(d.s_up);
    if (d.s_comp)
        (void)hipStreamDestroy(d.s_comp);
    if (d.s_down)
        (void)hipStreamDestroy(d.s_down);
    (void)rslf_ctx_destroy(d.ctx);
    d = rslf_multi::Dev();
}

extern "C" int rslf_multi_destroy(rslf_multi* m)
{
    if (!m)
        return RSLF_OK;
    for (auto& d : m->devs)
        multi_free_dev(d);
    delete m;
    return RSLF_OK;
}

extern "C" int rslf_multi_create(const int* devices, int n_devices, rslf_multi** out)
{
    if (!out)
        return fail(RSLF_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (n_devices < 0 || n_devices > 64 || (n_devices > 0 && !devices))
        return fail(RSLF_ERR_INVALID_ARG, "bad device list");
    rslf_multi* m = new (std::nothrow) rslf_multi();
    if (!m)
        return fail(RSLF_ERR_ALLOC, "out of host memory");
    const int n = n_devices > 0 ? n_devices : 1;
    m->devs.resize(n);
    for (int i = 0; i < n; i++) {
        rslf_multi::Dev& d = m->devs[i];
        int rc = rslf_ctx_create(n_devices > 0 ? devices[i] : 0, &d.ctx);   // a device may appear more than once
        hipError_t e = hipSuccess;
        if (rc == RSLF_OK) {
            // the upload stream outranks the compute stream: its blit and pack kernels then take the slots the scan's
            // workgroups free as they finish, instead of waiting behind the whole scan of the previous chunk
            int prio_lo = 0, prio_hi = 0;
            (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
            e = hipStreamCreateWithPriority(&d.s_up, hipStreamNonBlocking, prio_hi);
            if (e == hipSuccess) e = hipStreamCreateWithPriority(&d.s_comp, hipStreamNonBlocking, prio_lo);
            if (e == hipSuccess) e = hipStreamCreateWithFlags(&d.s_down, hipStreamNonBlocking);
            for (int k = 0; k < 2 && e == hipSuccess; k++)
                e = hipEventCreateWithFlags(&d.done[k], hipEventDisableTiming);
            if (e != hipSuccess)
                rc = fail(RSLF_ERR_HIP, "stream / event creation failed: %s", hipGetErrorString(e));
        }
        if (rc != RSLF_OK) {
            const std::string msg = g_err;
            (void)rslf_multi_destroy(m);
            return fail(rc, "%s", msg.c_str());
        }
    }
    *out = m;
    return RSLF_OK;
}

extern "C" int rslf_multi_device_count(const rslf_multi* m) { return m ? (int)m->devs.size() : 0; }

extern "C" int rslf_multi_set_chunk_rows(rslf_multi* m, int rows)
{
    if (!m || rows < 0)
        return fail(RSLF_ERR_INVALID_ARG, "bad argument");
    m->chunk_rows = rows;
    return RSLF_OK;
}

namespace {

struct MultiJob {
    const void* const* h_epis;
    bool is_u8;
    size_t row_stride_bytes;
    int V, S, U, C;
    float scale_arg;      // f32: the divisor (already resolved, > 0 or as given); u8: unused
    float dmin, dmax;
    int dim_d, s_hat;
    const rslf_params* p;
    float* h_Ce;
    uint8_t* h_mask;
    float* h_Cd;
    float* h_depth;
    float* h_rbar;
    int32_t* h_idx;
    float* h_score;
    float* h_raw;
    int halo;
    int out_device;   // -1: the result planes are host memory; >= 0: they live on this device (peer copies)
};

struct Chunk {
    int a, b;     // owned rows [a, b) of the whole field
    int lo, hi;   // computed rows: owned + halo, clipped
};

struct PlanePtrs {
    float *Ce, *Cd, *depth, *raw, *score, *rbar;
    int32_t* idx;
    uint8_t* mask;
};

PlanePtrs carve(char* blk, size_t n, int C)
{
    PlanePtrs q;
    q.Ce = (float*)blk;
    q.Cd = q.Ce + n;
    q.depth = q.Cd + n;
    q.raw = q.depth + n;
    q.score = q.raw + n;
    q.rbar = q.score + n;
    q.idx = (int32_t*)(q.rbar + n * C);
    q.mask = (uint8_t*)(q.idx + n);
    return q;
}

// One device's share: rows [r0, r1) of the field, chunk by chunk.  Returns an rslf status; `err` receives the message.
int multi_worker(rslf_multi::Dev& d, const MultiJob& j, int r0, int r1, int chunk_rows, long long* scanned, int* kernel,
                 int* spad, std::string* err)
{
#define MW_FAIL(rc_)                  \
    do {                              \
        *err = g_err;                 \
        return (rc_);                 \
    } while (0)
#define MW_HIP(expr)                                                                                        \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess) {                                                                             \
            (void)fail(RSLF_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            MW_FAIL(RSLF_ERR_HIP);                                                                          \
        }                                                                                                   \
    } while (0)
    *scanned = 0;
    if (r1 <= r0)
        return RSLF_OK;
    rslf_ctx* ctx = d.ctx;
    MW_HIP(hipSetDevice(ctx->device));
    // Do the EPIs follow one another in host memory (a stacked array) or are they scattered over the heap (a Vec<Mat>)?
    const size_t in_row_bytes = (size_t)j.U * j.C * (j.is_u8 ? 1 : sizeof(float));
    const size_t in_epi_bytes = in_row_bytes * j.S;
    bool scattered = (j.row_stride_bytes ? j.row_stride_bytes : in_row_bytes) != in_row_bytes;
    for (int i = r0 + 1; i < r1 && !scattered; i++)
        scattered = j.h_epis[i] && j.h_epis[i - 1] && (const char*)j.h_epis[i] != (const char*)j.h_epis[i - 1] + in_epi_bytes;
    // Chunks.  A given size: uniform.  Automatic: a short first chunk so that the kernels start early (its upload is the
    // one copy nothing hides), then two large ones (stacked input; scattered input, which is gathered into pinned memory
    // first, takes a middling second chunk and pieces of about V/3.5) -- a chunk's scan is a grid of its own, and a
    // grid of 5.4 rounds of workgroups pays for 6 (eight equal chunks of a 1080-row field ran 16 % longer than one
    // launch over the whole field; measured with rocprofv3 on the host-in / host-out path).
    std::vector<Chunk> chunks;
    {
        std::vector<int> sizes;
        const int rows = r1 - r0;
        if (chunk_rows > 0) {
            for (int a = 0; a < rows; a += chunk_rows)
                sizes.push_back(std::min(chunk_rows, rows - a));
        } else {
            int left = rows;
            const int first = std::min(left, std::max(32, (rows + 15) / 16));
            sizes.push_back(first);
            left -= first;
            if (left > 0 && scattered) {   // gathered through pinned memory first: smaller steps keep the kernels fed
                const int second = std::min(left, std::max(32, (rows + 7) / 8));
                sizes.push_back(second);
                left -= second;
            }
            if (left > 0) {
                // the chunk before has to cover the next one's upload with its scan
                const int n = scattered ? std::max(1, (int)std::lround(left / (rows / 3.5))) : (left > 2 * first ? 2 : 1);
                for (int i = 0; i < n; i++) {
                    const int sz = (left + (n - i) - 1) / (n - i);
                    sizes.push_back(sz);
                    left -= sz;
                }
            }
        }
        int a = r0;
        for (int sz : sizes) {
            Chunk c;
            c.a = a;
            c.b = a + sz;
            c.lo = std::max(0, c.a - j.halo);
            c.hi = std::min(j.V, c.b + j.halo);
            chunks.push_back(c);
            a += sz;
        }
    }
    int max_rows = 0;
    for (const Chunk& c : chunks)
        max_rows = std::max(max_rows, c.hi - c.lo);
    // two volumes and two sets of result planes, kept from call to call while the shape allows
    for (int k = 0; k < 2; k++) {
        if (d.vol[k] && (d.vol_rows[k] < max_rows || d.vol_S != j.S || d.vol_U != j.U || d.vol_C != j.C)) {
            (void)rslf_volume_destroy(d.vol[k]);
            d.vol[k] = nullptr;
        }
    }
    d.vol_S = j.S, d.vol_U = j.U, d.vol_C = j.C;
    for (int k = 0; k < 2; k++) {
        if (!d.vol[k]) {
            int rc = rslf_volume_create(ctx, max_rows, j.S, j.U, j.C, &d.vol[k]);
            if (rc)
                MW_FAIL(rc);
            d.vol_rows[k] = max_rows;
        }
    }
    const size_t n_max = (size_t)max_rows * j.U;
    // result planes of one chunk + a copy of the scan's per-scanline pixel counts (the context's own array is
    // rewritten by the next chunk's kernels, which are already queued when this chunk is collected)
    const size_t plane_bytes = (n_max * ((5 + (size_t)j.C) * sizeof(float) + sizeof(int32_t) + 1) + 15) & ~(size_t)15;
    const size_t bytes = plane_bytes + (size_t)max_rows * sizeof(int);
    if (bytes > d.planes_cap) {
        for (int k = 0; k < 2; k++) {
            (void)hipFree(d.planes[k]);
            d.planes[k] = nullptr;
        }
        d.planes_cap = 0;
        for (int k = 0; k < 2; k++)
            MW_HIP(hipMalloc(&d.planes[k], bytes));
        d.planes_cap = bytes;
    }
    std::vector<int> counts((size_t)max_rows);
    // pinned staging for scattered EPIs
    const int pin_threads = std::max(1, std::min(8, (int)std::thread::hardware_concurrency() / 2));
    {
        const size_t need = (size_t)max_rows * in_epi_bytes;
        if (scattered && need > d.pin_cap) {
            for (int k = 0; k < 2; k++) {
                (void)hipHostFree(d.pin[k]);
                d.pin[k] = nullptr;
            }
            d.pin_cap = 0;
            for (int k = 0; k < 2; k++)
                MW_HIP(hipHostMalloc((void**)&d.pin[k], need, hipHostMallocDefault));
            d.pin_cap = need;
        }
    }

    // a volume object of the chunk's height over the (larger or equal) allocation: rows beyond are simply unused
    auto upload = [&](int k) -> int {
        const Chunk& c = chunks[k];
        rslf_volume* vol = d.vol[k & 1];
        vol->V = c.hi - c.lo;
        vol->bytes = (size_t)vol->V * vol->S * vol->C * vol->pitch * sizeof(float);
        ctx->stream = d.s_up;
        // EPIs that follow one another in host memory go up as they are: one pageable copy per run, at the link's rate.
        // EPIs scattered over the heap (a Vec<Mat>) would be one pageable copy each, and the runtime stages those through
        // its own bounce buffer on the calling thread at ~10 GB/s -- slower than the kernels consume them.  They are
        // gathered into a pinned buffer by a few host threads first (dense rows; ~25 GB/s per thread) and go up from there.
        const size_t esz = j.is_u8 ? 1 : sizeof(float);
        const size_t row_bytes = (size_t)j.U * j.C * esz;
        const size_t stride = j.row_stride_bytes ? j.row_stride_bytes : row_bytes;
        const size_t epi_bytes = row_bytes * j.S;
        const int rows = c.hi - c.lo;
        int runs = 1;
        for (int i = 1; i < rows; i++)
            if (stride != row_bytes || (const char*)j.h_epis[c.lo + i] != (const char*)j.h_epis[c.lo + i - 1] + epi_bytes)
                runs++;
        const void* const* src = j.h_epis + c.lo;
        std::vector<const void*> staged;
        size_t src_stride = j.row_stride_bytes;
        if (runs > 8 && d.pin[k & 1]) {
            char* pin = d.pin[k & 1];
            const int nt = std::max(1, std::min(rows, pin_threads));
            std::vector<std::thread> th;
            bool null_epi = false;
            for (int i = 0; i < rows; i++)
                null_epi |= j.h_epis[c.lo + i] == nullptr;
            if (null_epi)
                return fail(RSLF_ERR_INVALID_ARG, "an EPI pointer is NULL");
            for (int t = 0; t < nt; t++)
                th.emplace_back([&, t] {
                    const int i0 = (int)((long long)rows * t / nt), i1 = (int)((long long)rows * (t + 1) / nt);
                    for (int i = i0; i < i1; i++) {
                        const char* e = (const char*)j.h_epis[c.lo + i];
                        char* o = pin + (size_t)i * epi_bytes;
                        if (stride == row_bytes)
                            memcpy(o, e, epi_bytes);
                        else
                            for (int r = 0; r < j.S; r++)
                                memcpy(o + (size_t)r * row_bytes, e + (size_t)r * stride, row_bytes);
                    }
                });
            for (auto& t : th)
                t.join();
            staged.resize((size_t)rows);
            for (int i = 0; i < rows; i++)
                staged[(size_t)i] = pin + (size_t)i * epi_bytes;
            src = staged.data();
            src_stride = row_bytes;
        }
        int rc;
        if (j.is_u8)
            rc = upload_host<uint8_t>(vol, (const uint8_t* const*)src, src_stride, false, (float)(1.0 / 255.0));
        else
            rc = upload_host<float>(vol, (const float* const*)src, src_stride, false, scale_of(j.scale_arg));
        return rc;   // upload_host ends with a synchronisation of its stream (minmax_end): the pinned buffer is free again
    };
    auto compute = [&](int k) -> int {
        const Chunk& c = chunks[k];
        const size_t n = (size_t)(c.hi - c.lo) * j.U;
        const PlanePtrs q = carve(d.planes[k & 1], n, j.C);
        ctx->stream = d.s_comp;
        int rc = rslf_depth1d_pile_run(ctx, d.vol[k & 1], j.dmin, j.dmax, j.dim_d, j.s_hat, j.p, q.Ce, q.mask, q.Cd, q.depth, q.rbar,
                                       q.idx, q.score, q.raw, nullptr);
        if (rc)
            return rc;
        *kernel = ctx->last_kernel;
        *spad = ctx->last_spad;
        hipError_t e = hipMemcpyAsync(d.planes[k & 1] + plane_bytes, ctx->count, (size_t)(c.hi - c.lo) * sizeof(int),
                                      hipMemcpyDeviceToDevice, d.s_comp);
        if (e == hipSuccess)
            e = hipEventRecord(d.done[k & 1], d.s_comp);
        return e == hipSuccess ? RSLF_OK : fail(RSLF_ERR_HIP, "queueing the chunk's completion failed: %s", hipGetErrorString(e));
    };
    auto download = [&](int k) -> int {
        const Chunk& c = chunks[k];
        const int rows = c.hi - c.lo;
        const size_t n = (size_t)rows * j.U;
        const PlanePtrs q = carve(d.planes[k & 1], n, j.C);
        const size_t off = (size_t)(c.a - c.lo) * j.U, cnt = (size_t)(c.b - c.a) * j.U, dst = (size_t)c.a * j.U;
        hipError_t e = hipStreamWaitEvent(d.s_down, d.done[k & 1], 0);
        auto pull = [&](void* h, const void* dv, size_t esz, size_t mult) {
            if (e != hipSuccess || !h)
                return;
            if (j.out_device < 0)
                e = hipMemcpyAsync((char*)h + dst * esz * mult, (const char*)dv + off * esz * mult, cnt * esz * mult, hipMemcpyDeviceToHost,
                                   d.s_down);
            else   // device-out: each worker's rows go straight to their place in the planes on the output device (xGMI peer copy)
                e = hipMemcpyPeerAsync((char*)h + dst * esz * mult, j.out_device, (const char*)dv + off * esz * mult, ctx->device,
                                       cnt * esz * mult, d.s_down);
        };
        pull(j.h_Ce, q.Ce, 4, 1);
        pull(j.h_mask, q.mask, 1, 1);
        pull(j.h_Cd, q.Cd, 4, 1);
        pull(j.h_depth, q.depth, 4, 1);
        pull(j.h_rbar, q.rbar, 4, (size_t)j.C);
        pull(j.h_idx, q.idx, 4, 1);
        pull(j.h_score, q.score, 4, 1);
        pull(j.h_raw, q.raw, 4, 1);
        // pixels scanned on the owned rows: the per-scanline counts of the scan's pixel lists
        if (e == hipSuccess)
            e = hipMemcpyAsync(counts.data(), d.planes[k & 1] + plane_bytes, (size_t)rows * sizeof(int), hipMemcpyDeviceToHost, d.s_down);
        if (e == hipSuccess)
            e = hipStreamSynchronize(d.s_down);
        if (e != hipSuccess)
            return fail(RSLF_ERR_HIP, "result download failed: %s", hipGetErrorString(e));
        for (int r = c.a - c.lo; r < c.b - c.lo; r++)
            *scanned += counts[(size_t)r];
        return RSLF_OK;
    };

    hipStream_t saved = ctx->stream;
    int rc = upload(0);
    if (rc == RSLF_OK)
        rc = compute(0);
    for (int k = 0; rc == RSLF_OK && k < (int)chunks.size(); k++) {
        // chunk k's kernels are queued: feed the next chunk, then collect this one
        if (k + 1 < (int)chunks.size()) {
            rc = upload(k + 1);            // volume (k+1)&1 was last read by chunk k-1, whose download has completed
            if (rc == RSLF_OK)
                rc = compute(k + 1);       // planes (k+1)&1 likewise; queued behind chunk k on the compute stream
        }
        if (rc == RSLF_OK)
            rc = download(k);              // waits for chunk k's kernels; chunk k+1 runs meanwhile
    }
    if (rc != RSLF_OK) {
        *err = g_err;
        (void)hipDeviceSynchronize();
    }
    ctx->stream = saved;
    for (int k = 0; k < 2; k++) {   // restore the volumes' full height for the next call's reuse test
        d.vol[k]->V = d.vol_rows[k];
        d.vol[k]->bytes = (size_t)d.vol[k]->V * d.vol[k]->S * d.vol[k]->C * d.vol[k]->pitch * sizeof(float);
    }
    return rc;
#undef MW_FAIL
#undef MW_HIP
}

float host_max_f32_parallel(const float* const* h_epis, int V, int S, size_t stride, size_t row_elems, float start)
{
    const int nt = std::max(1, std::min<int>(8, std::min<int>((int)std::thread::hardware_concurrency(), V / 8)));
    std::vector<float> part((size_t)nt, start);
    std::vector<std::thread> th;
    for (int t = 0; t < nt; t++)
        th.emplace_back([&, t] {
            const int v0 = (int)((long long)V * t / nt), v1 = (int)((long long)V * (t + 1) / nt);
            part[(size_t)t] = host_max_f32(h_epis + v0, v1 - v0, S, stride, row_elems, start);
        });
    float m = start;
    for (int t = 0; t < nt; t++) {
        th[(size_t)t].join();
        m = std::max(m, part[(size_t)t]);
    }
    return m;
}

int multi_run(rslf_multi* m, MultiJob j, rslf_stats* stats)
{
    if (!m || !j.h_epis)
        return fail(RSLF_ERR_INVALID_ARG, "NULL argument");
    if (j.V < 1 || j.S < 1 || j.U < 1 || (j.C != 1 && j.C != 3))
        return fail(RSLF_ERR_INVALID_ARG, "bad dimensions V=%d S=%d U=%d C=%d", j.V, j.S, j.U, j.C);
    int rc = check_params(j.p);
    if (rc)
        return rc;
    for (int v = 0; v < j.V; v++)
        if (!j.h_epis[v])
            return fail(RSLF_ERR_INVALID_ARG, "h_epis[%d] is NULL", v);
    j.s_hat = resolve_s_hat(j.s_hat, j.S);
    // rows either side of a chunk that must be recomputed for the chunk's own rows to come out exact: the median
    // reads +-(size-1)/2 rows (core.hpp:686), and through the optional opening +-2*(k/2) rows more (core.hpp:759-768)
    j.halo = (j.p->median_filter_size - 1) / 2 + (j.p->edge_confidence_opening_size > 1 ? 2 * (j.p->edge_confidence_opening_size / 2) : 0);
    const int nd = (int)m->devs.size();
    std::vector<long long> scanned((size_t)nd, 0);
    std::vector<int> rcs((size_t)nd, RSLF_OK), kern((size_t)nd, 0), spads((size_t)nd, 0);
    std::vector<std::string> errs((size_t)nd);
    std::vector<std::thread> th;
    for (int i = 0; i < nd; i++) {
        const int r0 = (int)((long long)j.V * i / nd), r1 = (int)((long long)j.V * (i + 1) / nd);
        // chunks: enough of them to overlap the copies with the kernels, large enough to keep the halo's share small
        int chunk = m->chunk_rows > 0 ? std::max(1, std::min(m->chunk_rows, std::max(1, r1 - r0))) : 0;   // 0: the graded plan
        th.emplace_back([&, i, r0, r1, chunk] {
            rcs[(size_t)i] = multi_worker(m->devs[(size_t)i], j, r0, r1, chunk, &scanned[(size_t)i], &kern[(size_t)i], &spads[(size_t)i],
                                          &errs[(size_t)i]);
        });
    }
    for (auto& t : th)
        t.join();
    for (int i = 0; i < nd; i++)
        if (rcs[(size_t)i] != RSLF_OK)
            return fail(rcs[(size_t)i], "device %d: %s", m->devs[(size_t)i].ctx->device, errs[(size_t)i].c_str());
    if (stats) {
        long long tot = 0;
        for (long long s : scanned)
            tot += s;
        stats->pixels_scanned = tot;
        stats->units = tot * j.dim_d;
        stats->scan_kernel = kern[0];
        stats->s_pad = spads[0];
    }
    return RSLF_OK;
}

}  // namespace

static int multi_pile_f32(rslf_multi* m, const float* const* h_epis, size_t row_stride_bytes, int V, int S, int U, int C,
                          float epi_scale_factor, float dmin, float dmax, int dim_d, int s_hat, const rslf_params* p, int out_device,
                          float* h_Ce_vu, uint8_t* h_Ce_mask_vu, float* h_Cd_vu, float* h_depth_vu, float* h_rbar_vu, int32_t* h_idx_vu,
                          float* h_score_vu, float* h_depth_raw_vu, rslf_stats* stats, float* scale_used)
{
    if (!m || !h_epis || V < 1)
        return fail(RSLF_ERR_INVALID_ARG, "NULL argument");
    const size_t row_elems = (size_t)U * C;
    const size_t stride = row_stride_bytes ? row_stride_bytes : row_elems * sizeof(float);
    for (int v = 0; v < V; v++)
        if (!h_epis[v])
            return fail(RSLF_ERR_INVALID_ARG, "h_epis[%d] is NULL", v);
    // dc.hpp:442-460: the default scale is the maximum over ALL EPIs -- taken once here, never per block
    if (epi_scale_factor < 0)
        epi_scale_factor = host_max_f32_parallel(h_epis, V, S, stride, row_elems, epi_scale_factor);
    if (scale_used)
        *scale_used = epi_scale_factor;
    MultiJob j = {(const void* const*)h_epis, false, stride, V, S, U, C, epi_scale_factor, dmin, dmax, dim_d, s_hat, p,
                  h_Ce_vu, h_Ce_mask_vu, h_Cd_vu, h_depth_vu, h_rbar_vu, h_idx_vu, h_score_vu, h_depth_raw_vu, 0, out_device};
    return multi_run(m, j, stats);
}

extern "C" int rslf_multi_depth1d_pile_f32(rslf_multi* m, const float* const* h_epis, size_t row_stride_bytes, int V, int S, int U,
                                           int C, float epi_scale_factor, float dmin, float dmax, int dim_d, int s_hat,
                                           const rslf_params* p, float* h_Ce_vu, uint8_t* h_Ce_mask_vu, float* h_Cd_vu,
                                           float* h_depth_vu, float* h_rbar_vu, int32_t* h_idx_vu, float* h_score_vu,
                                           float* h_depth_raw_vu, rslf_stats* stats, float* scale_used)
{
    return multi_pile_f32(m, h_epis, row_stride_bytes, V, S, U, C, epi_scale_factor, dmin, dmax, dim_d, s_hat, p, -1, h_Ce_vu,
                          h_Ce_mask_vu, h_Cd_vu, h_depth_vu, h_rbar_vu, h_idx_vu, h_score_vu, h_depth_raw_vu, stats, scale_used);
}

extern "C" int rslf_multi_depth1d_pile_f32_dev(rslf_multi* m, const float* const* h_epis, size_t row_stride_bytes, int V, int S, int U,
                                               int C, float epi_scale_factor, float dmin, float dmax, int dim_d, int s_hat,
                                               const rslf_params* p, int out_device, float* d_Ce_vu, uint8_t* d_Ce_mask_vu,
                                               float* d_Cd_vu, float* d_depth_vu, float* d_rbar_vu, int32_t* d_idx_vu,
                                               float* d_score_vu, float* d_depth_raw_vu, rslf_stats* stats, float* scale_used)
{
    if (out_device < 0)
        return fail(RSLF_ERR_INVALID_ARG, "out_device %d", out_device);
    return multi_pile_f32(m, h_epis, row_stride_bytes, V, S, U, C, epi_scale_factor, dmin, dmax, dim_d, s_hat, p, out_device, d_Ce_vu,
                          d_Ce_mask_vu, d_Cd_vu, d_depth_vu, d_rbar_vu, d_idx_vu, d_score_vu, d_depth_raw_vu, stats, scale_used);
}

extern "C" int rslf_multi_depth1d_pile_u8(rslf_multi* m, const uint8_t* const* h_epis, size_t row_stride_bytes, int V, int S, int U,
                                          int C, float dmin, float dmax, int dim_d, int s_hat, const rslf_params* p, float* h_Ce_vu,
                                          uint8_t* h_Ce_mask_vu, float* h_Cd_vu, float* h_depth_vu, float* h_rbar_vu, int32_t* h_idx_vu,
                                          float* h_score_vu, float* h_depth_raw_vu, rslf_stats* stats)
{
    MultiJob j = {(const void* const*)h_epis, true, row_stride_bytes, V, S, U, C, 255.0f, dmin, dmax, dim_d, s_hat, p,
                  h_Ce_vu, h_Ce_mask_vu, h_Cd_vu, h_depth_vu, h_rbar_vu, h_idx_vu, h_score_vu, h_depth_raw_vu, 0, -1};
    return multi_run(m, j, stats);
}

// ---- Depth2DComputer::run over several devices (dc.hpp:748-805) -----------------------------------------------------
// The 2-D sweep sharded by scanline behind the C-ABI: every device holds a block of scanlines (+ the median's halo) of the
// volume and of the [S][rows][U] planes; a visit is scan on every device, then the neighbours' boundary rows of the visited
// view's raw disparities and edge mask by peer copy (the one real exchange step of the path, as sharding.ShardedDepth2D
// does it over RCCL), then median + propagation on every device.  ONE host thread drives all devices: every call only
// queues work, the order between devices is kept by events -- a device's finish waits for its neighbours to have
// fetched its boundary rows, because the apply pass rewrites them (core.hpp:1119-1121).
namespace {

struct Sweep2DDev {
    rslf_volume* vol = nullptr;
    float *Ce = nullptr, *Cd = nullptr, *depth = nullptr, *rbar = nullptr;
    float *dmin = nullptr, *dmax = nullptr;   // per-pixel hypothesis ranges over the held rows (a fine-to-coarse level), or NULL
    uint8_t *cem = nullptr, *scan_mask = nullptr;
    int lo = 0, hi = 0, a = 0, b = 0;   // rows held [lo, hi), rows owned [a, b)
    hipEvent_t ev_scan = nullptr, ev_fetch = nullptr;
    bool begun = false;
    hipStream_t saved_stream = nullptr;
    bool stream_swapped = false;
};

void sweep2d_free(rslf_multi* m, std::vector<Sweep2DDev>& ds)
{
    for (size_t i = 0; i < ds.size(); i++) {
        Sweep2DDev& d = ds[i];
        rslf_ctx* ctx = m->devs[i].ctx;
        (void)hipSetDevice(ctx->device);
        if (d.begun)
            (void)rslf_sweep_end(ctx, 0, 2, nullptr);
        (void)hipStreamSynchronize(ctx->stream);
        if (d.stream_swapped)
            ctx->stream = d.saved_stream;
        if (d.vol)
            (void)rslf_volume_destroy(d.vol);
        // (the planes live in the device's arena, which stays)
        if (d.ev_scan)
            (void)hipEventDestroy(d.ev_scan);
        if (d.ev_fetch)
            (void)hipEventDestroy(d.ev_fetch);
    }
}

// The fine-to-coarse form of a sweep: nothing passes through host memory.  The level's RAW volume, its per-pixel ranges
// and the two planes the next steps need live on the FIRST device; every device takes the rows it holds from there and
// leaves its own rows of the results there, by peer copies (plain device copies where it is the first device itself).
struct FirstDevicePlanes {
    const float* raw_vsuc = nullptr;   // [V][S][U][C] raw values of the level (replaces the host EPIs)
    const float* dmin_svu = nullptr;   // [S][V][U] ranges, or NULL for the scalar range
    const float* dmax_svu = nullptr;
    float* Ce_svu = nullptr;           // [S][V][U] results
    float* depth_svu = nullptr;
};

int multi_depth2d(rslf_multi* m, const void* const* h_epis, bool is_u8, size_t row_stride_bytes, int V, int S, int U, int C, float scale_arg,
                  float dmin, float dmax, int dim_d, const rslf_params* p, float* h_Ce_svu, uint8_t* h_Ce_mask_svu, float* h_Cd_svu,
                  float* h_depth_svu, float* h_rbar_svu, uint8_t* h_scan_mask_svu, rslf_stats* stats,
                  const FirstDevicePlanes* first = nullptr)
{
    if (!m || (!h_epis && !(first && first->raw_vsuc)))
        return fail(RSLF_ERR_INVALID_ARG, "NULL argument");
    if (first && ((first->dmin_svu == nullptr) != (first->dmax_svu == nullptr)))
        return fail(RSLF_ERR_INVALID_ARG, "dmin_svu and dmax_svu must both be given or both be NULL");
    const bool ranges = first && first->dmin_svu;
    const int dev0 = m->devs[0].ctx->device;
    if (V < 1 || S < 1 || U < 1 || (C != 1 && C != 3))
        return fail(RSLF_ERR_INVALID_ARG, "bad dimensions V=%d S=%d U=%d C=%d", V, S, U, C);
    int rc = check_params(p);
    if (rc)
        return rc;
    for (int v = 0; h_epis && v < V; v++)
        if (!h_epis[v])
            return fail(RSLF_ERR_INVALID_ARG, "h_epis[%d] is NULL", v);
    const int h_med = (p->median_filter_size - 1) / 2;
    const int halo = h_med + (p->edge_confidence_opening_size > 1 ? 2 * (p->edge_confidence_opening_size / 2) : 0);
    int nd = (int)m->devs.size();
    while (nd > 1 && V / nd < std::max(1, halo))   // a block must be able to fill its neighbours' halo rows
        nd--;
    std::vector<Sweep2DDev> ds((size_t)nd);
#define S2_TRY(expr)                                  \
    do {                                              \
        int rc_ = (expr);                             \
        if (rc_ != RSLF_OK) {                         \
            const std::string msg_ = g_err;           \
            sweep2d_free(m, ds);                      \
            return fail(rc_, "%s", msg_.c_str());     \
        }                                             \
    } while (0)
#define S2_HIP(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            sweep2d_free(m, ds);                                                                       \
            return fail(RSLF_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
        }                                                                                              \
    } while (0)
    // set-up: rows, volume, planes, edge confidence, sweep state
    for (int i = 0; i < nd; i++) {
        Sweep2DDev& d = ds[(size_t)i];
        rslf_ctx* ctx = m->devs[(size_t)i].ctx;
        S2_HIP(hipSetDevice(ctx->device));
        d.saved_stream = ctx->stream;
        d.stream_swapped = true;
        ctx->stream = m->devs[(size_t)i].s_comp;
        d.a = (int)((long long)V * i / nd);
        d.b = (int)((long long)V * (i + 1) / nd);
        d.lo = std::max(0, d.a - halo);
        d.hi = std::min(V, d.b + halo);
        const int rows = d.hi - d.lo;
        const size_t n = (size_t)S * rows * U;
        S2_HIP(hipEventCreateWithFlags(&d.ev_scan, hipEventDisableTiming));
        S2_HIP(hipEventCreateWithFlags(&d.ev_fetch, hipEventDisableTiming));
        S2_TRY(rslf_volume_create(ctx, rows, S, U, C, &d.vol));
        // a copy between this device and the first one (either direction), queued on this device's stream
        auto copy01 = [&](void* dst, int dst_dev, const void* src, int src_dev, size_t bytes) -> hipError_t {
            return dst_dev == src_dev ? hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream)
                                      : hipMemcpyPeerAsync(dst, dst_dev, src, src_dev, bytes, ctx->stream);
        };
        if (first && first->raw_vsuc) {   // the held rows of the level's raw volume: from the first device
            const size_t row_floats = (size_t)S * U * C;
            const float* src = first->raw_vsuc + (size_t)d.lo * row_floats;
            if (ctx->device != dev0) {
                S2_TRY(ensure_staging(ctx, (size_t)rows * row_floats * sizeof(float)));
                S2_HIP(copy01(ctx->staging, ctx->device, src, dev0, (size_t)rows * row_floats * sizeof(float)));
                src = (const float*)ctx->staging;
            }
            S2_TRY(rslf_volume_pack_device_f32(d.vol, src, scale_arg, nullptr));
        } else if (is_u8) {
            S2_TRY(upload_host<uint8_t>(d.vol, (const uint8_t* const*)h_epis + d.lo, row_stride_bytes, false, (float)(1.0 / 255.0)));
        } else {
            S2_TRY(upload_host<float>(d.vol, (const float* const*)h_epis + d.lo, row_stride_bytes, false, scale_of(scale_arg)));
        }
        {   // planes: one allocation per device, grown when a larger field comes (allocation calls synchronise the device)
            rslf_multi::Dev& md = m->devs[(size_t)i];
            const size_t nf = (n + 63) & ~(size_t)63;   // floats per plane, 256-byte aligned
            const size_t need = nf * sizeof(float) * (3 + (size_t)C + (ranges ? 2 : 0)) + 2 * nf;
            if (need > md.arena_cap) {
                (void)hipFree(md.arena);
                md.arena = nullptr;
                md.arena_cap = 0;
                S2_HIP(hipMalloc(&md.arena, need));
                md.arena_cap = need;
            }
            float* f = reinterpret_cast<float*>(md.arena);
            d.Ce = f, f += nf;
            d.Cd = f, f += nf;
            d.depth = f, f += nf;
            d.rbar = f, f += nf * C;
            if (ranges) {
                d.dmin = f, f += nf;
                d.dmax = f, f += nf;
            }
            d.cem = reinterpret_cast<uint8_t*>(f);
            d.scan_mask = d.cem + nf;
        }
        hipStream_t st = ctx->stream;
        if (ranges) {   // the held rows of every view's range planes: one run of bytes per view
            const size_t w = (size_t)rows * U;
            for (int sv = 0; sv < S; sv++) {
                S2_HIP(copy01(d.dmin + (size_t)sv * w, ctx->device, first->dmin_svu + ((size_t)sv * V + d.lo) * U, dev0, w * sizeof(float)));
                S2_HIP(copy01(d.dmax + (size_t)sv * w, ctx->device, first->dmax_svu + ((size_t)sv * V + d.lo) * U, dev0, w * sizeof(float)));
            }
        }
        S2_HIP(hipMemsetAsync(d.Ce, 0, n * sizeof(float), st));   // dc.hpp:733-750
        S2_HIP(hipMemsetAsync(d.Cd, 0, n * sizeof(float), st));
        S2_HIP(hipMemsetAsync(d.depth, 0, n * sizeof(float), st));
        S2_HIP(hipMemsetAsync(d.rbar, 0, n * C * sizeof(float), st));
        S2_TRY(rslf_edge_confidence_2d(ctx, d.vol, p, d.Ce, d.cem));                               // dc.hpp:772
        S2_TRY(rslf_sweep_begin(ctx, d.vol, d.cem, d.scan_mask, dim_d, d.a - d.lo, d.b - d.lo));  // dc.hpp:780
        d.begun = true;
    }
    // rows of plane `base` ([S][rows][U] elements of `esz` bytes) of view s_hat, local rows [r, r + h)
    auto rows_of = [&](const Sweep2DDev& d, void* base, size_t esz, int s_hat, int r) -> char* {
        return (char*)base + (((size_t)s_hat * (d.hi - d.lo) + r) * U) * esz;
    };
    for (int s_hat : sweep_order(S)) {   // core.hpp:981-990
        for (int i = 0; i < nd; i++) {
            Sweep2DDev& d = ds[(size_t)i];
            rslf_ctx* ctx = m->devs[(size_t)i].ctx;
            S2_HIP(hipSetDevice(ctx->device));
            S2_TRY(rslf_sweep_visit_scan(ctx, d.vol, d.dmin, d.dmax, dmin, dmax, dim_d, s_hat, d.Ce, d.cem, d.Cd, d.depth, d.rbar, p));
            S2_HIP(hipEventRecord(d.ev_scan, ctx->stream));
        }
        for (int i = 0; i < nd; i++) {   // fetch the neighbours' boundary rows into this device's halo rows
            Sweep2DDev& d = ds[(size_t)i];
            rslf_ctx* ctx = m->devs[(size_t)i].ctx;
            S2_HIP(hipSetDevice(ctx->device));
            for (int side = 0; side < 2 && h_med > 0; side++) {
                const int k = side == 0 ? i - 1 : i + 1;
                if (k < 0 || k >= nd)
                    continue;
                const Sweep2DDev& o = ds[(size_t)k];
                rslf_ctx* octx = m->devs[(size_t)k].ctx;
                S2_HIP(hipStreamWaitEvent(ctx->stream, o.ev_scan, 0));
                // side 0: the h rows above my block = the last h own rows of device i-1; side 1: the first h own rows of i+1
                const int dst_r = side == 0 ? (d.a - d.lo) - h_med : (d.b - d.lo);
                const int src_r = side == 0 ? (o.b - o.lo) - h_med : (o.a - o.lo);
                for (int pl = 0; pl < 2; pl++) {
                    const size_t esz = pl == 0 ? sizeof(float) : 1;
                    char* dst = rows_of(d, pl == 0 ? (void*)d.depth : (void*)d.cem, esz, s_hat, dst_r);
                    const char* src = rows_of(o, pl == 0 ? (void*)o.depth : (void*)o.cem, esz, s_hat, src_r);
                    const size_t bytes = (size_t)h_med * U * esz;
                    if (octx->device == ctx->device)
                        S2_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
                    else
                        S2_HIP(hipMemcpyPeerAsync(dst, ctx->device, src, octx->device, bytes, ctx->stream));
                }
            }
            S2_HIP(hipEventRecord(d.ev_fetch, ctx->stream));
        }
        for (int i = 0; i < nd; i++) {   // median + propagation, once the neighbours have my raw boundary rows
            Sweep2DDev& d = ds[(size_t)i];
            rslf_ctx* ctx = m->devs[(size_t)i].ctx;
            S2_HIP(hipSetDevice(ctx->device));
            if (i > 0)
                S2_HIP(hipStreamWaitEvent(ctx->stream, ds[(size_t)i - 1].ev_fetch, 0));
            if (i + 1 < nd)
                S2_HIP(hipStreamWaitEvent(ctx->stream, ds[(size_t)i + 1].ev_fetch, 0));
            S2_TRY(rslf_sweep_visit_finish(ctx, d.vol, s_hat, d.cem, d.Cd, d.depth, d.rbar, p));
        }
    }
    // collect: every device's own rows of every view land at their place in the caller's [S][V][U] planes
    long long scanned = 0;
    for (int i = 0; i < nd; i++) {
        Sweep2DDev& d = ds[(size_t)i];
        rslf_ctx* ctx = m->devs[(size_t)i].ctx;
        S2_HIP(hipSetDevice(ctx->device));
        rslf_stats st_i;
        memset(&st_i, 0, sizeof(st_i));
        d.begun = false;
        S2_TRY(rslf_sweep_end(ctx, 1, dim_d, &st_i));
        scanned += st_i.pixels_scanned;
        if (stats && i == 0) {
            stats->scan_kernel = st_i.scan_kernel;
            stats->s_pad = st_i.s_pad;
        }
        const int rows = d.hi - d.lo, own = d.b - d.a;
        auto pull = [&](void* h, const void* dv, size_t esz) -> hipError_t {
            if (!h)
                return hipSuccess;
            return hipMemcpy2DAsync((char*)h + (size_t)d.a * U * esz, (size_t)V * U * esz, (const char*)dv + (size_t)(d.a - d.lo) * U * esz,
                                    (size_t)rows * U * esz, (size_t)own * U * esz, S, hipMemcpyDeviceToHost, ctx->stream);
        };
        if (first && first->Ce_svu) {   // this device's own rows of the two planes the next steps read: to the first device
            const size_t w = (size_t)own * U;
            for (int sv = 0; sv < S; sv++) {
                const size_t src = ((size_t)sv * rows + (d.a - d.lo)) * U, dst = ((size_t)sv * V + d.a) * U;
                hipError_t e = ctx->device == dev0
                                   ? hipMemcpyAsync(first->Ce_svu + dst, d.Ce + src, w * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream)
                                   : hipMemcpyPeerAsync(first->Ce_svu + dst, dev0, d.Ce + src, ctx->device, w * sizeof(float), ctx->stream);
                S2_HIP(e);
                e = ctx->device == dev0
                        ? hipMemcpyAsync(first->depth_svu + dst, d.depth + src, w * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream)
                        : hipMemcpyPeerAsync(first->depth_svu + dst, dev0, d.depth + src, ctx->device, w * sizeof(float), ctx->stream);
                S2_HIP(e);
            }
        }
        S2_HIP(pull(h_Ce_svu, d.Ce, sizeof(float)));
        S2_HIP(pull(h_Ce_mask_svu, d.cem, 1));
        S2_HIP(pull(h_Cd_svu, d.Cd, sizeof(float)));
        S2_HIP(pull(h_depth_svu, d.depth, sizeof(float)));
        S2_HIP(pull(h_rbar_svu, d.rbar, sizeof(float) * C));
        S2_HIP(pull(h_scan_mask_svu, d.scan_mask, 1));
    }
    for (int i = 0; i < nd; i++) {
        S2_HIP(hipSetDevice(m->devs[(size_t)i].ctx->device));
        S2_HIP(hipStreamSynchronize(m->devs[(size_t)i].ctx->stream));
    }
    sweep2d_free(m, ds);
    if (stats) {
        stats->pixels_scanned = scanned;
        stats->units = scanned * dim_d;
    }
    return RSLF_OK;
#undef S2_TRY
#undef S2_HIP
}

}  // namespace

// FineToCoarse<T> (rslf_fine_to_coarse.hpp:103-324) over the context's devices.  Where the time goes -- every level's 2-D
// sweep -- runs sharded (multi_depth2d, with the level's tightened per-pixel ranges); the pyramid, the bound tightening
// and the fusion, cheap whole-image passes with non-local footprints, run on the first device, where every level's raw
// volume, ranges, disparities and confidences stay: the devices take their rows from there and leave their results there
// by peer copies (FirstDevicePlanes).  The host sees the EPIs going up once and the fused map coming down.
extern "C" int rslf_multi_fine_to_coarse_run_host(rslf_multi* m, const void* const* h_epis, int is_u8, int V, int S, int U, int C,
                                                  size_t row_stride_bytes, float d_min, float d_max, int dim_d, float epi_scale_factor,
                                                  const rslf_params* p, int max_pyr_depth, int accept_all_last_scale,
                                                  float* h_out_map_svu, uint8_t* h_out_valid_svu, int* n_levels, rslf_stats* stats)
{
    if (!m || !h_epis || !h_out_map_svu || !h_out_valid_svu || V < 1 || S < 1 || U < 1 || (C != 1 && C != 3))
        return fail(RSLF_ERR_INVALID_ARG, "bad arguments");
    int rc = check_params(p);
    if (rc)
        return rc;
    rslf_ctx* ctx = m->devs[0].ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const size_t elem = is_u8 ? 1 : 4;
    const size_t row_bytes = (size_t)U * C * elem;
    if (row_stride_bytes == 0)
        row_stride_bytes = row_bytes;

    struct Level {
        int V = 0, U = 0;
        float scale = 1.0f;
        rslf_params params;
        DevBuf raw;                 // [V][S][U][C] raw values, on the first device
        DevBuf Ce, depth, valid;    // [S][V][U], on the first device
    };
    std::deque<Level> levels;       // (a deque: the levels own device buffers and must not move)
    // constructor (f2c.hpp:103-159): the pyramid on the first device, every level's RAW volume kept there
    {
        levels.emplace_back();
        Level& l0 = levels.back();
        HIP_TRY(l0.raw.alloc((size_t)V * S * U * C * sizeof(float)));
        DevBuf stage;
        void* dst = l0.raw.p;
        if (is_u8) {
            HIP_TRY(stage.alloc((size_t)V * S * row_bytes));
            dst = stage.p;
        }
        for (int v = 0; v < V; v++) {
            if (!h_epis[v])
                return fail(RSLF_ERR_INVALID_ARG, "h_epis[%d] is NULL", v);
            if (row_stride_bytes == row_bytes)
                HIP_TRY(hipMemcpyAsync((char*)dst + (size_t)v * S * row_bytes, h_epis[v], (size_t)S * row_bytes, hipMemcpyHostToDevice, st));
            else
                HIP_TRY(hipMemcpy2DAsync((char*)dst + (size_t)v * S * row_bytes, row_bytes, h_epis[v], row_stride_bytes, row_bytes, S,
                                         hipMemcpyHostToDevice, st));
        }
        if (is_u8) {
            const size_t n = (size_t)V * S * U * C;
            hipLaunchKernelGGL(k_u8_to_f32, dim3(stream_blocks(n)), dim3(256), 0, st, (const uint8_t*)stage.p, (float*)l0.raw.p, (long long)n);
            HIP_TRY(hipGetLastError());
        }
        HIP_TRY(hipStreamSynchronize(st));
    }
    if (max_pyr_depth < 1)
        max_pyr_depth = 1 << 30;
    {
        const int start_dim_u = U;
        int dim_v = V, dim_u = U, counter = 0;
        bool have = true;   // levels.back() holds the raw volume of the level being described
        while (dim_v > 10 && dim_u > 10 && counter < max_pyr_depth) {   // _MIN_SPATIAL_DIM, f2c.hpp:8, :130
            counter++;
            Level& lv = levels.back();
            lv.V = dim_v;
            lv.U = dim_u;
            lv.params = *p;
            lv.params.slope_factor = (float)((0.0 + dim_u) / start_dim_u);   // f2c.hpp:139
            lv.scale = 255.0f;                                                // dc.hpp:696-699 (uchar)
            if (!is_u8) {
                lv.scale = epi_scale_factor;
                if (lv.scale < 0) {                                           // dc.hpp:671-690: this level's own max
                    rc = rslf_device_max_f32(ctx, (const float*)lv.raw.p, (size_t)dim_v * S * dim_u * C, &lv.scale);
                    if (rc)
                        return rc;
                }
            }
            have = false;
            int v2, u2;
            rslf_f2c_level_dims(dim_v, dim_u, &v2, &u2);
            if (v2 < 1 || u2 < 1)
                break;
            if (!(v2 > 10 && u2 > 10 && counter < max_pyr_depth))
                break;                                                        // no further level: nothing to build
            levels.emplace_back();
            Level& nx = levels.back();
            Level& cur = levels[levels.size() - 2];
            HIP_TRY(nx.raw.alloc((size_t)v2 * S * u2 * C * sizeof(float)));   // f2c.hpp:145-147: the RAW EPIs go down
            rc = is_u8 ? rslf_downsample_epis_u8(ctx, (const float*)cur.raw.p, dim_v, S, dim_u, C, (float*)nx.raw.p)
                       : rslf_downsample_epis_f32(ctx, (const float*)cur.raw.p, dim_v, S, dim_u, C, (float*)nx.raw.p);
            if (rc)
                return rc;
            have = true;
            dim_v = v2;
            dim_u = u2;
        }
        if (have)            // the loop never described the last buffer (the field itself is below _MIN_SPATIAL_DIM)
            levels.pop_back();
    }
    if (levels.empty())
        return fail(RSLF_ERR_INVALID_ARG, "light field %dx%d is not larger than _MIN_SPATIAL_DIM: no pyramid level", V, U);
    const int P = (int)levels.size();

    // run(): f2c.hpp:171-299 -- the sweeps over all devices, the tightening on the first
    int64_t pixels = 0;
    rslf_stats st1;
    memset(&st1, 0, sizeof(st1));
    for (int l = 0; l < P; l++) {
        Level& lv = levels[(size_t)l];
        const size_t n = (size_t)S * lv.V * lv.U;
        HIP_TRY(hipSetDevice(ctx->device));
        HIP_TRY(lv.Ce.alloc(n * 4));
        HIP_TRY(lv.depth.alloc(n * 4));
        HIP_TRY(lv.valid.alloc(n));
        DevBuf d_lo, d_hi;
        FirstDevicePlanes first;
        first.raw_vsuc = (const float*)lv.raw.p;
        first.Ce_svu = (float*)lv.Ce.p;
        first.depth_svu = (float*)lv.depth.p;
        if (l > 0) {
            Level& up = levels[(size_t)l - 1];
            HIP_TRY(d_lo.alloc(n * 4));
            HIP_TRY(d_hi.alloc(n * 4));
            hipLaunchKernelGGL(k_fill_f32, dim3(stream_blocks(n)), dim3(256), 0, st, (float*)d_lo.p, (long long)n, d_min);
            hipLaunchKernelGGL(k_fill_f32, dim3(stream_blocks(n)), dim3(256), 0, st, (float*)d_hi.p, (long long)n, d_max);
            HIP_TRY(hipGetLastError());
            rc = rslf_f2c_tighten_bounds(ctx, (const float*)up.depth.p, (const uint8_t*)up.valid.p, S, up.V, up.U, (float*)d_lo.p,
                                         (float*)d_hi.p, lv.V, lv.U);
            if (rc)
                return rc;
            first.dmin_svu = (const float*)d_lo.p;
            first.dmax_svu = (const float*)d_hi.p;
        }
        HIP_TRY(hipStreamSynchronize(st));   // what the other devices' streams are about to read is complete
        rc = multi_depth2d(m, nullptr, false, 0, lv.V, S, lv.U, C, lv.scale, d_min, d_max, dim_d, &lv.params, nullptr, nullptr, nullptr,
                           nullptr, nullptr, nullptr, &st1, &first);
        if (rc)
            return rc;
        pixels += st1.pixels_scanned;
        // get_valid_depths_mask_s_v_u (dc.hpp:893-915): C_e > threshold; the last level accepts everything when asked to
        HIP_TRY(hipSetDevice(ctx->device));
        const bool all = accept_all_last_scale && l == P - 1;
        hipLaunchKernelGGL(k_valid_mask, dim3(stream_blocks(n)), dim3(256), 0, st, (const float*)lv.Ce.p, (uint8_t*)lv.valid.p, (long long)n,
                           all ? -1.0f : p->edge_score_threshold);
        HIP_TRY(hipGetLastError());
        lv.raw.release();   // the level's raw volume has been taken by every device
    }

    // get_results(): f2c.hpp:302-324 on the first device
    std::vector<const float*> dp((size_t)P);
    std::vector<const uint8_t*> vp((size_t)P);
    std::vector<int> Vp((size_t)P), Up((size_t)P);
    for (int l = 0; l < P; l++) {
        dp[(size_t)l] = (const float*)levels[(size_t)l].depth.p;
        vp[(size_t)l] = (const uint8_t*)levels[(size_t)l].valid.p;
        Vp[(size_t)l] = levels[(size_t)l].V;
        Up[(size_t)l] = levels[(size_t)l].U;
    }
    const size_t n0 = (size_t)S * V * U;
    DevBuf omap, ovalid;
    HIP_TRY(omap.alloc(n0 * 4));
    HIP_TRY(ovalid.alloc(n0));
    rc = rslf_f2c_fuse(ctx, dp.data(), vp.data(), Vp.data(), Up.data(), P, S, (float*)omap.p, (uint8_t*)ovalid.p);
    if (rc)
        return rc;
    HIP_TRY(hipMemcpyAsync(h_out_map_svu, omap.p, n0 * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(h_out_valid_svu, ovalid.p, n0, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (n_levels)
        *n_levels = P;
    if (stats) {
        *stats = st1;
        stats->pixels_scanned = pixels;
        stats->units = pixels * dim_d;
    }
    return RSLF_OK;
}

extern "C" int rslf_multi_depth2d_run_f32(rslf_multi* m, const float* const* h_epis, size_t row_stride_bytes, int V, int S, int U, int C,
                                          float epi_scale_factor, float dmin, float dmax, int dim_d, const rslf_params* p,
                                          float* h_Ce_svu, uint8_t* h_Ce_mask_svu, float* h_Cd_svu, float* h_depth_svu,
                                          float* h_rbar_svu, uint8_t* h_scan_mask_svu, rslf_stats* stats, float* scale_used)
{
    if (!m || !h_epis || V < 1 || S < 1 || U < 1)
        return fail(RSLF_ERR_INVALID_ARG, "NULL argument");
    const size_t row_elems = (size_t)U * C;
    const size_t stride = row_stride_bytes ? row_stride_bytes : row_elems * sizeof(float);
    for (int v = 0; v < V; v++)
        if (!h_epis[v])
            return fail(RSLF_ERR_INVALID_ARG, "h_epis[%d] is NULL", v);
    if (epi_scale_factor < 0)   // dc.hpp:671-705: the maximum over ALL EPIs, taken once
        epi_scale_factor = host_max_f32_parallel(h_epis, V, S, stride, row_elems, epi_scale_factor);
    if (scale_used)
        *scale_used = epi_scale_factor;
    return multi_depth2d(m, (const void* const*)h_epis, false, stride, V, S, U, C, epi_scale_factor, dmin, dmax, dim_d, p, h_Ce_svu,
                         h_Ce_mask_svu, h_Cd_svu, h_depth_svu, h_rbar_svu, h_scan_mask_svu, stats);
}

extern "C" int rslf_multi_depth2d_run_u8(rslf_multi* m, const uint8_t* const* h_epis, size_t row_stride_bytes, int V, int S, int U, int C,
                                         float dmin, float dmax, int dim_d, const rslf_params* p, float* h_Ce_svu,
                                         uint8_t* h_Ce_mask_svu, float* h_Cd_svu, float* h_depth_svu, float* h_rbar_svu,
                                         uint8_t* h_scan_mask_svu, rslf_stats* stats)
{
    return multi_depth2d(m, (const void* const*)h_epis, true, row_stride_bytes, V, S, U, C, 255.0f, dmin, dmax, dim_d, p, h_Ce_svu,
                         h_Ce_mask_svu, h_Cd_svu, h_depth_svu, h_rbar_svu, h_scan_mask_svu, stats);
}
