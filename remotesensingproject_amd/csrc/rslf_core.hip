// librslf_hip.so, unit 1 of 9: errors, contexts, volumes, host upload / device pack (K0).
// C-ABI: include/rslf_hip.h.  No OpenCV, no torch, no CPU compute path.
#include "rslf_internal.hpp"

#include <algorithm>
#include <cmath>

#include "k0_pack.hpp"

using namespace rslf;

// ---- errors ---------------------------------------------------------------

namespace rslf {

char* last_error_buffer()
{
    static thread_local char g_err[512] = "";
    return g_err;
}

static std::atomic<int> g_inject[kInjectSites];

bool inject_hit(InjectSite site)
{
    std::atomic<int>& c = g_inject[site];
    int v = c.load(std::memory_order_relaxed);
    while (v > 0)
        if (c.compare_exchange_weak(v, v - 1, std::memory_order_relaxed))
            return true;
    return false;
}

}  // namespace rslf

extern "C" int rslf_debug_inject(const char* site, int count) RSLF_API_TRY
{
    if (!site || count < 0)
        return fail(RSLF_ERR_INVALID_ARG, "rslf_debug_inject: bad argument");
    static const char* const names[kInjectSites] = {"worker", "thread_create", "alloc"};
    for (int i = 0; i < kInjectSites; i++)
        if (strcmp(site, names[i]) == 0) {
            g_inject[i].store(count, std::memory_order_relaxed);
            return RSLF_OK;
        }
    return fail(RSLF_ERR_INVALID_ARG, "rslf_debug_inject: unknown site %s", site);
}
RSLF_API_CATCH

int rslf::ensure_plane_scratch(rslf_ctx* ctx, int V, int U)
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
        HIP_TRY(hipMalloc(&ctx->count, (size_t)2 * V * sizeof(int)));   // [count_cap] entries per row, then [count_cap] row bases of packed lists
        ctx->count_cap = V;
    }
    return RSLF_OK;
}

// Records and tickets of grouped scan launches (k2_scan.hpp): grow-only, so a context allocates them once.
int rslf::ensure_group_scratch(rslf_ctx* ctx, size_t recs, size_t tiles)
{
    if (recs > ctx->partial_rec_cap) {
        HIP_TRY(hipFree(ctx->scan_partial));
        ctx->scan_partial = nullptr;
        ctx->partial_rec_cap = 0;
        HIP_TRY(hipMalloc(&ctx->scan_partial, recs * plan::kPartialRecordBytes));
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

extern "C" int rslf_abi_version(void) RSLF_API_TRY
{
    return RSLF_ABI_VERSION;
}
RSLF_API_CATCH

extern "C" const char* rslf_status_string(int status)
{
    switch (status) {
    case RSLF_OK: return "ok";
    case RSLF_ERR_INVALID_ARG: return "invalid argument";
    case RSLF_ERR_UNSUPPORTED: return "unsupported configuration";
    case RSLF_ERR_HIP: return "HIP runtime error";
    case RSLF_ERR_NO_DEVICE: return "no gfx950 device";
    case RSLF_ERR_ALLOC: return "allocation failed";
    case RSLF_ERR_INTERNAL: return "internal error (an exception was stopped at the C boundary)";
    default: return "unknown status";
    }
}

extern "C" const char* rslf_last_error(void) { return last_error_buffer(); }

extern "C" int rslf_device_count(void) RSLF_API_TRY
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}
RSLF_API_CATCH

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

ScanConsts rslf::make_scan_consts(const rslf_params* p)
{
    ScanConsts k;
    k.slope = p->slope_factor;
    const float h = p->kernel_bandwidth;
    const float hh = h * h;
    k.inv_h2 = (float)(1.0 / (double)hh);   // include/rslf_kernels.hpp:43
    k.k1 = 3.0f * k.inv_h2;                 // src/rslf_kernels.cpp:21
    k.raw_thr = p->raw_score_threshold;
    k.n_iter = plan::mean_shift_passes(p->mean_shift_max_iter);   // core.hpp:584, float bound
    k.interp = p->interpolation;
    return k;
}

int rslf::check_params(const rslf_params* p)
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
    // any window the reference runs: width = (size - 1) / 2 (core.hpp:686), so even sizes and 0 are fine; a negative size
    // has no window at all (the reference then reads an empty vector, core.hpp:713-714)
    if (p->median_filter_size < 0 || p->median_filter_size > plan::kMedianMaxSize)
        return fail(RSLF_ERR_INVALID_ARG, "median_filter_size=%d: must be in [0, %d]", p->median_filter_size, plan::kMedianMaxSize);
    if (!(p->kernel_bandwidth > 0.0f))
        return fail(RSLF_ERR_INVALID_ARG, "kernel_bandwidth must be > 0");
    if (!(p->mean_shift_max_iter > 0.0f))
        return fail(RSLF_ERR_INVALID_ARG, "mean_shift_max_iter must be > 0");
    if (p->interpolation < RSLF_INTERP_LINEAR || p->interpolation > RSLF_INTERP_NEAREST_AS_BUILT)
        return fail(RSLF_ERR_INVALID_ARG, "interpolation=%d is not one of RSLF_INTERP_*", p->interpolation);
    return RSLF_OK;
}

// ---- context --------------------------------------------------------------

extern "C" int rslf_ctx_create(int device, rslf_ctx** out) RSLF_API_TRY
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
        (void)rslf_ctx_destroy(ctx);   // frees whatever of total / minmax / the events was made
        return fail(RSLF_ERR_HIP, "context setup failed: %s", hipGetErrorString(e));
    }
    *out = ctx;
    return RSLF_OK;
}
RSLF_API_CATCH

extern "C" int rslf_ctx_destroy(rslf_ctx* ctx) RSLF_API_TRY
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
    (void)hipFree(ctx->remain);
    (void)hipFree(ctx->sweep_mask);
    (void)hipFree(ctx->filtered);
    for (hipEvent_t e : ctx->ev_pool)
        (void)hipEventDestroy(e);
    if (ctx->ev0)
        (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1)
        (void)hipEventDestroy(ctx->ev1);
    delete ctx;
    return RSLF_OK;
}
RSLF_API_CATCH

extern "C" int rslf_ctx_set_stream(rslf_ctx* ctx, void* hip_stream) RSLF_API_TRY
{
    if (!ctx)
        return fail(RSLF_ERR_INVALID_ARG, "ctx is NULL");
    ctx->stream = (hipStream_t)hip_stream;
    return RSLF_OK;
}
RSLF_API_CATCH

extern "C" int rslf_ctx_set_debug(rslf_ctx* ctx, const char* key, int value) RSLF_API_TRY
{
    if (!ctx || !key)
        return fail(RSLF_ERR_INVALID_ARG, "ctx/key is NULL");
    if (strcmp(key, "force_scan") == 0 && value >= 0 && value <= 2)
        ctx->force_scan = value;
    else if (strcmp(key, "force_groups") == 0 && value >= 0 && value <= 64)
        ctx->force_groups = value;
    else if (strcmp(key, "force_packed") == 0 && value >= -1 && value <= 1)
        ctx->force_packed = value;
    else if (strcmp(key, "px") == 0 && value >= -1 && value <= 1)
        ctx->px_mode = value;
    else if (strcmp(key, "stream_share") == 0 && value >= 0 && value <= 2)
        ctx->stream_share = value;
    else if (strcmp(key, "row_split") == 0 && value >= 0 && value <= 65536)
        ctx->row_split = value;
    else if (strcmp(key, "claim_skip") == 0 && (value == 0 || value == 1))
        ctx->claim_skip = value;
    else if (strcmp(key, "time_all") == 0 && (value == 0 || value == 1)) {
        ctx->time_all = value;
        ctx->ev_used = 0;
    } else if (strcmp(key, "stream_groups") == 0 && value >= 0 && value <= 64)
        ctx->stream_groups = value;
    else if (strcmp(key, "stream_lds_kib") == 0 && value >= 16 && value <= 152) {
        ctx->stream_lds_bytes = (size_t)value << 10;
    } else
        return fail(RSLF_ERR_INVALID_ARG, "rslf_ctx_set_debug: unknown key or value out of range: %s = %d", key, value);
    return RSLF_OK;
}
RSLF_API_CATCH

extern "C" int rslf_ctx_synchronize(rslf_ctx* ctx) RSLF_API_TRY
{
    if (!ctx)
        return fail(RSLF_ERR_INVALID_ARG, "ctx is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return RSLF_OK;
}
RSLF_API_CATCH

// ---- volume ---------------------------------------------------------------

extern "C" int rslf_volume_create(rslf_ctx* ctx, int V, int S, int U, int C, rslf_volume** out) RSLF_API_TRY
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
RSLF_API_CATCH

extern "C" int rslf_volume_destroy(rslf_volume* vol) RSLF_API_TRY
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
RSLF_API_CATCH

extern "C" int rslf_volume_describe(const rslf_volume* vol, rslf_volume_desc* out) RSLF_API_TRY
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
RSLF_API_CATCH

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

int rslf::ensure_staging(rslf_ctx* ctx, size_t bytes)
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
int rslf::helper_scratch(rslf_ctx* ctx, int slot, size_t bytes, void** out)
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
int rslf::upload_host(rslf_volume* vol, const SrcT* const* h_ptrs, size_t row_stride_bytes, bool image_major, float scale)
{
    rslf_ctx* ctx = vol->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t row_bytes = (size_t)vol->U * vol->C * sizeof(SrcT);
    if (row_stride_bytes == 0)
        row_stride_bytes = row_bytes;
    if (row_stride_bytes < row_bytes)
        return fail(RSLF_ERR_INVALID_ARG, "row_stride_bytes %zu < row size %zu", row_stride_bytes, row_bytes);
    const size_t epi_bytes = row_bytes * vol->S;
    const int chunk = plan::staging_chunk_rows(epi_bytes, vol->V);
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
    const int chunk = plan::staging_chunk_rows(epi_bytes, vol->V);
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
float rslf::host_max_f32(const float* const* h_ptrs, int n_ptrs, int rows, size_t row_stride_bytes, size_t row_elems, float start)
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

extern "C" int rslf_volume_upload_epis_f32(rslf_volume* vol, const float* const* h_epis, size_t row_stride_bytes,
                                           float epi_scale_factor, float* scale_used) RSLF_API_TRY
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
RSLF_API_CATCH

extern "C" int rslf_volume_upload_epis_u8(rslf_volume* vol, const uint8_t* const* h_epis, size_t row_stride_bytes) RSLF_API_TRY
{
    if (!vol || !h_epis)
        return fail(RSLF_ERR_INVALID_ARG, "vol/h_epis is NULL");
    return upload_host<uint8_t>(vol, h_epis, row_stride_bytes, false, (float)(1.0 / 255.0));   // dc.hpp:470
}
RSLF_API_CATCH

extern "C" int rslf_volume_upload_images_f32(rslf_volume* vol, const float* const* h_imgs, size_t row_stride_bytes,
                                             float epi_scale_factor, float* scale_used) RSLF_API_TRY
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
RSLF_API_CATCH

extern "C" int rslf_volume_upload_images_u8(rslf_volume* vol, const uint8_t* const* h_imgs, size_t row_stride_bytes) RSLF_API_TRY
{
    if (!vol || !h_imgs)
        return fail(RSLF_ERR_INVALID_ARG, "vol/h_imgs is NULL");
    return upload_host<uint8_t>(vol, h_imgs, row_stride_bytes, true, (float)(1.0 / 255.0));
}
RSLF_API_CATCH

extern "C" int rslf_volume_upload_images_xf_f32(rslf_volume* vol, const float* const* h_imgs, size_t row_stride_bytes,
                                                float epi_scale_factor, float* scale_used, int transpose, int rotate_180) RSLF_API_TRY
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
RSLF_API_CATCH

extern "C" int rslf_volume_upload_images_xf_u8(rslf_volume* vol, const uint8_t* const* h_imgs, size_t row_stride_bytes, int transpose,
                                               int rotate_180) RSLF_API_TRY
{
    if (!vol || !h_imgs)
        return fail(RSLF_ERR_INVALID_ARG, "vol/h_imgs is NULL");
    return upload_images_xf<uint8_t>(vol, h_imgs, row_stride_bytes, (float)(1.0 / 255.0), transpose != 0, rotate_180 != 0);
}
RSLF_API_CATCH

extern "C" int rslf_volume_pack_device_f32(rslf_volume* vol, const float* d_vsuc, float epi_scale_factor, float* scale_used) RSLF_API_TRY
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
RSLF_API_CATCH

template int rslf::upload_host<float>(rslf_volume*, const float* const*, size_t, bool, float);
template int rslf::upload_host<uint8_t>(rslf_volume*, const uint8_t* const*, size_t, bool, float);

// dc.hpp:442-460 over all EPIs, by up to eight host threads
float rslf::host_max_f32_parallel(const float* const* h_epis, int V, int S, size_t stride, size_t row_elems, float start)
{
    const int nt = std::max(1, std::min<int>(8, std::min<int>((int)std::thread::hardware_concurrency(), V / 8)));
    std::vector<float> part((size_t)nt, start);
    {
        JoinGuard pool;
        for (int t = 0; t < nt; t++)
            pool.run([&, t] {
                int v0, v1;
                plan::split_range(V, t, nt, &v0, &v1);
                part[(size_t)t] = host_max_f32(h_epis + v0, v1 - v0, S, stride, row_elems, start);
            });
    }
    float m = start;
    for (int t = 0; t < nt; t++)
        m = std::max(m, part[(size_t)t]);
    return m;
}
