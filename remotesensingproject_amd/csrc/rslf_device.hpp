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
    int   interp;         // RSLF_INTERP_*: the generic kernel alone handles the nearest modes
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
// of the mean-shift loop.  Written as asm on purpose: hipcc does fold
// __builtin_amdgcn_fmed3f(1 - q, 0, 1) into the same instruction, but then schedules each
// sample's 7-instruction chain back to back through two registers (+10 % cycles, profiles/).
__device__ __forceinline__ float kernel_weight(float q)
{
    float k;
    asm("v_sub_f32_e64 %0, 1.0, %1 clamp" : "=v"(k) : "v"(q));
    return k;
}

// interp.hpp:179-181 for a position x >= 0: i0 = (int)floor(x) and t = x - floor(x), one instruction each
// instead of floor + convert + subtract.  v_fract_f32 is x - floor(x) exactly for x >= 0 (for x < 0 close
// to an integer it stays below 1 where the subtraction rounds to 1 -- callers only use t where x is valid,
// i.e. non-negative); v_cvt_flr_i32_f32 converts with round toward -inf.
__device__ __forceinline__ float lerp_weight(float x)
{
    return __builtin_amdgcn_fractf(x);
}
__device__ __forceinline__ int floor_to_int(float x)
{
    int i;
    asm("v_cvt_flr_i32_f32_e32 %0, %1" : "=v"(i) : "v"(x));
    return i;
}

// Four samples of one mean-shift pass (1 channel), hand-scheduled: 28 VALU instructions issued
// stage by stage so that no instruction reads the result of the one before it (hipcc, left to
// itself, allocates two temporaries and emits the 7-instruction chain of each sample back to
// back, which stalls the issue: +10 % cycles measured).  The two running sums take
// the samples in ascending order, one IEEE add each -- the reference's sequential cv::reduce.
//   delta = R - rbar ; t = k1*delta ; q = t*delta ; K = clamp(1 - q) ; P = R*K ; A += P ; B += K
__device__ __forceinline__ void mean_shift_group4(float r0, float r1, float r2, float r3, float rbar, float k1,
                                                  float& A, float& B)
{
    float t0, t1, t2, t3, u0, u1, u2, u3;
    asm("v_sub_f32 %2, %10, %14\n\t"
        "v_sub_f32 %3, %11, %14\n\t"
        "v_sub_f32 %4, %12, %14\n\t"
        "v_sub_f32 %5, %13, %14\n\t"
        "v_mul_f32 %6, %15, %2\n\t"
        "v_mul_f32 %7, %15, %3\n\t"
        "v_mul_f32 %8, %15, %4\n\t"
        "v_mul_f32 %9, %15, %5\n\t"
        "v_mul_f32 %2, %2, %6\n\t"
        "v_mul_f32 %3, %3, %7\n\t"
        "v_mul_f32 %4, %4, %8\n\t"
        "v_mul_f32 %5, %5, %9\n\t"
        "v_sub_f32_e64 %2, 1.0, %2 clamp\n\t"
        "v_sub_f32_e64 %3, 1.0, %3 clamp\n\t"
        "v_sub_f32_e64 %4, 1.0, %4 clamp\n\t"
        "v_sub_f32_e64 %5, 1.0, %5 clamp\n\t"
        "v_mul_f32 %6, %10, %2\n\t"
        "v_mul_f32 %7, %11, %3\n\t"
        "v_mul_f32 %8, %12, %4\n\t"
        "v_mul_f32 %9, %13, %5\n\t"
        "v_add_f32 %0, %0, %6\n\t"
        "v_add_f32 %1, %1, %2\n\t"
        "v_add_f32 %0, %0, %7\n\t"
        "v_add_f32 %1, %1, %3\n\t"
        "v_add_f32 %0, %0, %8\n\t"
        "v_add_f32 %1, %1, %4\n\t"
        "v_add_f32 %0, %0, %9\n\t"
        "v_add_f32 %1, %1, %5"
        : "+v"(A), "+v"(B), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(u0), "=&v"(u1), "=&v"(u2), "=&v"(u3)
        : "v"(r0), "v"(r1), "v"(r2), "v"(r3), "v"(rbar), "s"(k1));
}

// One sample of the same pass (the tail of a view count that is not a multiple of four).
__device__ __forceinline__ void mean_shift_group1(float r0, float rbar, float k1, float& A, float& B)
{
    float t0, u0;
    asm("v_sub_f32 %2, %4, %5\n\t"
        "v_mul_f32 %3, %6, %2\n\t"
        "v_mul_f32 %2, %2, %3\n\t"
        "v_sub_f32_e64 %2, 1.0, %2 clamp\n\t"
        "v_mul_f32 %3, %4, %2\n\t"
        "v_add_f32 %0, %0, %3\n\t"
        "v_add_f32 %1, %1, %2"
        : "+v"(A), "+v"(B), "=&v"(t0), "=&v"(u0)
        : "v"(r0), "v"(rbar), "s"(k1));
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
