// librslf_hip.so, unit 7 of 9: fine-to-coarse (rslf_fine_to_coarse.hpp:103-324, rslf_fine_to_coarse_core.cpp:14-135) --
// the pyramid (Gaussian blur + halving), the bound tightening, the fusion (K5), the host-pointer form of the 2-D sweep and
// the native level loop.  C-ABI: include/rslf_hip.h.
#include "rslf_internal.hpp"

#include <algorithm>
#include <cmath>
#include <memory>

#include "k5_f2c.hpp"

using namespace rslf;

// ---- "next" row: fine-to-coarse ------------------------------------------------

extern "C" int rslf_f2c_level_dims(int V, int U, int* V2, int* U2) RSLF_API_TRY
{
    if (!V2 || !U2 || V < 1 || U < 1)
        return fail(RSLF_ERR_INVALID_ARG, "bad arguments");
    plan::f2c_level_dims(V, U, V2, U2);   // cvRound: ties to even
    return RSLF_OK;
}
RSLF_API_CATCH

extern "C" int rslf_downsample_epis_f32(rslf_ctx* ctx, const float* d_in_vsuc, int V, int S, int U, int C, float* d_out_vsuc) RSLF_API_TRY
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
RSLF_API_CATCH

extern "C" int rslf_downsample_epis_u8(rslf_ctx* ctx, const float* d_in_vsuc, int V, int S, int U, int C, float* d_out_vsuc) RSLF_API_TRY
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
RSLF_API_CATCH

extern "C" int rslf_device_max_f32(rslf_ctx* ctx, const float* d_values, size_t n, float* h_max) RSLF_API_TRY
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
RSLF_API_CATCH

extern "C" int rslf_f2c_tighten_bounds(rslf_ctx* ctx, const float* d_depth_up_svu, const uint8_t* d_valid_up_svu, int S, int V_up,
                                       int U_up, float* d_dmin_down_svu, float* d_dmax_down_svu, int V_down, int U_down) RSLF_API_TRY
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
RSLF_API_CATCH

extern "C" int rslf_f2c_fuse(rslf_ctx* ctx, const float* const* d_disp, const uint8_t* const* d_valid, const int* Vp, const int* Up,
                             int P, int S, float* d_out_map_svu, uint8_t* d_out_valid_svu) RSLF_API_TRY
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
RSLF_API_CATCH

// ---- host-pointer forms of the rows around the path -----------------------------

extern "C" int rslf_depth2d_run_host(rslf_ctx* ctx, const rslf_volume* vol, float dmin, float dmax, int dim_d, const rslf_params* p,
                                     float* h_Ce_svu, uint8_t* h_Ce_mask_svu, float* h_Cd_svu, float* h_depth_svu,
                                     float* h_rbar_svu, rslf_stats* stats) RSLF_API_TRY
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
RSLF_API_CATCH

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

int rslf::f2c_u8_to_f32(hipStream_t st, const uint8_t* in, float* out, size_t n)
{
    hipLaunchKernelGGL(k_u8_to_f32, dim3(stream_blocks(n)), dim3(256), 0, st, in, out, (long long)n);
    HIP_TRY(hipGetLastError());
    return RSLF_OK;
}
int rslf::f2c_fill_f32(hipStream_t st, float* out, size_t n, float value)
{
    hipLaunchKernelGGL(k_fill_f32, dim3(stream_blocks(n)), dim3(256), 0, st, out, (long long)n, value);
    HIP_TRY(hipGetLastError());
    return RSLF_OK;
}
int rslf::f2c_valid_mask(hipStream_t st, const float* Ce, uint8_t* out, size_t n, float thr)
{
    hipLaunchKernelGGL(k_valid_mask, dim3(stream_blocks(n)), dim3(256), 0, st, Ce, out, (long long)n, thr);
    HIP_TRY(hipGetLastError());
    return RSLF_OK;
}

extern "C" int rslf_fine_to_coarse_run_host(rslf_ctx* ctx, const void* const* h_epis, int is_u8, int V, int S, int U, int C,
                                            size_t row_stride_bytes, float d_min, float d_max, int dim_d, float epi_scale_factor,
                                            const rslf_params* p, int max_pyr_depth, int accept_all_last_scale,
                                            float* h_out_map_svu, uint8_t* h_out_valid_svu, int* n_levels, rslf_stats* stats) RSLF_API_TRY
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

    // constructor: rslf_fine_to_coarse.hpp:103-159 -- the level sizes are plan::f2c_pyramid
    const std::vector<plan::LevelDims> dims = plan::f2c_pyramid(V, U, max_pyr_depth);
    std::vector<std::unique_ptr<F2cLevel>> levels;
    DevBuf cur;           // raw volume of the level being built (level 0 borrows `raw`)
    float* cur_p = (float*)raw.p;
    for (size_t l = 0; l < dims.size(); l++) {
        const int dim_v = dims[l].V, dim_u = dims[l].U;
        levels.emplace_back(new F2cLevel());
        F2cLevel* lv = levels.back().get();
        lv->V = dim_v;
        lv->U = dim_u;
        lv->params = *p;
        lv->params.slope_factor = (float)((0.0 + dim_u) / U);             // f2c.hpp:139
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
        if (l + 1 == dims.size())
            break;
        const int v2 = dims[l + 1].V, u2 = dims[l + 1].U;
        DevBuf next;                                                       // f2c.hpp:145-147: the RAW EPIs go down
        HIP_TRY(next.alloc((size_t)v2 * S * u2 * C * sizeof(float)));
        // uchar EPIs go down in uchar arithmetic, as the reference's CV_8U Mats do (fine_to_coarse_core.cpp:22-41)
        rc = is_u8 ? rslf_downsample_epis_u8(ctx, cur_p, dim_v, S, dim_u, C, (float*)next.p)
                   : rslf_downsample_epis_f32(ctx, cur_p, dim_v, S, dim_u, C, (float*)next.p);
        if (rc)
            return rc;
        std::swap(cur.p, next.p);   // `next` now frees the previous level's raw copy
        cur_p = (float*)cur.p;
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
RSLF_API_CATCH
