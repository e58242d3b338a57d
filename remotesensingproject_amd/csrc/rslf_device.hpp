// Shared device-side definitions for the gfx950 EPI depth-scan kernels.
//
// Arithmetic contract (DESIGN.md "Numerics"): every float operation is one
// IEEE binary32 op in the reference's order; the translation unit is built
// with -ffp-contract=off and without fast-math, fp32 division is the
// correctly rounded default, fp32 denormals are preserved (hipcc default).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rslf {

constexpr int kWave = 64;              // gfx950 wavefront
constexpr float kSentinel = 1.0e30f;   // out-of-range sample marker of the register scan (see k2_scan.hpp)

// The light-field slab in HBM: [V][S][C][pitch] float32, zero padded rows.
struct VolView {
    const float* base;
    int V, S, U, C;
    int pitch;            // floats per row
    long long stride_s;   // floats between views   = C * pitch
    long long stride_v;   // floats between EPIs    = S * C * pitch
    __device__ __forceinline__ const float* row(int v, int s, int c) const
    {
        return base + (long long)v * stride_v + (long long)s * stride_s + (long long)c * pitch;
    }
};

// Scalars derived once on the host from rslf_params (rslf_abi.hip: make_consts).
struct ScanConsts {
    float slope;          // par_slope_factor
    float inv_h2;         // float(1.0 / double(h*h))       kernels.hpp:43
    float k1;             // 3.0f * inv_h2                  kernels.cpp:21
    float raw_thr;        // par_raw_score_threshold
    int   n_iter;         // #{i >= 0 : float(i) < par_mean_shift_max_iter}   core.hpp:584
};

struct EdgeConsts {
    int   filter_size;    // par_edge_confidence_filter_size
    int   cut_shadows;
    float shadow_level;
    float edge_thr;
};

// norm<float> / norm<cv::Vec3f>  (src/rslf_types.cpp:80-91)
__device__ __forceinline__ float norm1(float x)
{
    return (float)((double)fabsf(x) * 1.73205080757);
}
__device__ __forceinline__ float norm3(float x, float y, float z)
{
    double s = (double)x * (double)x;
    s += (double)y * (double)y;
    s += (double)z * (double)z;
    return (float)sqrt(s);
}

// K = max(1 - q, 0), NaN -> 0  (src/rslf_kernels.cpp:23-25 / :51-53) as ONE instruction:
// v_sub_f32 with the clamp output modifier clamps to [0, 1] and (DX10_CLAMP, the hipcc kernel
// default) sends NaN to 0.  q = (k*delta)*delta is >= 0 or NaN, so 1 - q <= 1 and the upper
// clamp never acts: the result equals cv::max(1 - q, 0) bit for bit.  v_max_f32 issues at
// about half the rate of v_sub_f32 on gfx950 (tools/ubench_valu.hip), so this is worth ~20 %
// of the mean-shift loop.
__device__ __forceinline__ float kernel_weight(float q)
{
    float k;
    asm("v_sub_f32_e64 %0, 1.0, %1 clamp" : "=v"(k) : "v"(q));
    return k;
}

// cv::BORDER_REFLECT_101
__device__ __forceinline__ int reflect101(int p, int len)
{
    if (len == 1)
        return 0;
    while (p < 0 || p >= len)
        p = (p < 0) ? -p : 2 * len - 2 - p;
    return p;
}

// Dispatch is round-robin over the 8 XCDs (block b runs on XCD b % 8, each
// with its own 4 MiB L2).  Remap so that every XCD walks one contiguous range
// of logical blocks: consecutive scanlines -- which read the same EPI rows --
// then share an L2.  `per_xcd` = ceil(logical_blocks / 8); the launch uses
// 8 * per_xcd blocks and surplus ones return.  Placement only affects speed.
__device__ __forceinline__ int xcd_logical_block(int b, int per_xcd)
{
    return (b & 7) * per_xcd + (b >> 3);
}

}  // namespace rslf
