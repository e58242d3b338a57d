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

// The light-field slab in HBM: [V][S][pitch][C] float32 -- one row of `pitch` pixels per (EPI v, view s),
// channels interleaved as in the reference's cv::Mat rows, pixels U..pitch-1 zero.  Both lerp taps of a
// sample, all channels, are 2*C consecutive floats; for C = 1 the layout is a plain pitched row.
struct VolView {
    const float* base;
    int V, S, U, C;
    int pitch;            // pixels per row (multiple of 64, > U)
    long long stride_s;   // floats between views   = pitch * C
    long long stride_v;   // floats between EPIs    = S * pitch * C
    __device__ __forceinline__ const float* row(int v, int s) const
    {
        return base + (long long)v * stride_v + (long long)s * stride_s;
    }
};

// Scalars derived once on the host from rslf_params (rslf_core.hip: make_scan_consts).
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

// Two floats in an aligned register pair: operands of v_pk_add_f32 / v_pk_mul_f32.  Each half is the same
// IEEE operation as the scalar instruction (tools/ubench_pk.hip checks it bit for bit, sentinel and denormal
// products included).  At one wave per SIMD a wave issues one VALU instruction every ~5 clocks whatever it
// is, so there -- and only there -- a packed instruction does two samples' work for one issue slot.
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f2 kernel_weight_pk(f2 q)   // max(1 - q, 0) on both halves
{
    f2 k;
    const f2 one = {1.0f, 1.0f};
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1] clamp" : "=v"(k) : "v"(one), "v"(q));
    return k;
}

// Eight samples (four pairs) of one 1-channel mean-shift pass in packed instructions, issued stage by stage
// (a dependent packed instruction needs its producer several slots back).  Leaves P = R*K and K per pair;
// the caller adds them to the sums one sample at a time.
__device__ __forceinline__ void mean_shift_pk_octet(const f2 (&r)[4], f2 rb, f2 kq, f2 (&P)[4], f2 (&K)[4])
{
    const f2 one = {1.0f, 1.0f};
    f2 t0, t1, t2, t3;
    asm("v_pk_add_f32 %[k0], %[r0], %[rb] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %[k1], %[r1], %[rb] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %[k2], %[r2], %[rb] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %[k3], %[r3], %[rb] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_mul_f32 %[t0], %[kq], %[k0]\n\t"
        "v_pk_mul_f32 %[t1], %[kq], %[k1]\n\t"
        "v_pk_mul_f32 %[t2], %[kq], %[k2]\n\t"
        "v_pk_mul_f32 %[t3], %[kq], %[k3]\n\t"
        "v_pk_mul_f32 %[k0], %[k0], %[t0]\n\t"
        "v_pk_mul_f32 %[k1], %[k1], %[t1]\n\t"
        "v_pk_mul_f32 %[k2], %[k2], %[t2]\n\t"
        "v_pk_mul_f32 %[k3], %[k3], %[t3]\n\t"
        "v_pk_add_f32 %[k0], %[one], %[k0] neg_lo:[0,1] neg_hi:[0,1] clamp\n\t"
        "v_pk_add_f32 %[k1], %[one], %[k1] neg_lo:[0,1] neg_hi:[0,1] clamp\n\t"
        "v_pk_add_f32 %[k2], %[one], %[k2] neg_lo:[0,1] neg_hi:[0,1] clamp\n\t"
        "v_pk_add_f32 %[k3], %[one], %[k3] neg_lo:[0,1] neg_hi:[0,1] clamp\n\t"
        "v_pk_mul_f32 %[t0], %[r0], %[k0]\n\t"
        "v_pk_mul_f32 %[t1], %[r1], %[k1]\n\t"
        "v_pk_mul_f32 %[t2], %[r2], %[k2]\n\t"
        "v_pk_mul_f32 %[t3], %[r3], %[k3]"
        : [k0] "=&v"(K[0]), [k1] "=&v"(K[1]), [k2] "=&v"(K[2]), [k3] "=&v"(K[3]),
          [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3)
        : [r0] "v"(r[0]), [r1] "v"(r[1]), [r2] "v"(r[2]), [r3] "v"(r[3]), [rb] "v"(rb), [kq] "v"(kq), [one] "v"(one));
    P[0] = t0;
    P[1] = t1;
    P[2] = t2;
    P[3] = t3;
}

// Four samples (two pairs a, b) of one 3-channel pass: per channel delta, kq*delta, q; qs = (q0 + q2) + q1
// (OpenCV 3.x reduceC_); K = clamp(1 - qs); P_c = R_c * K.  30 packed instructions for what takes 60 scalar ones.
__device__ __forceinline__ void mean_shift_pk_rgb_quad(const f2 (&ra)[3], const f2 (&rb)[3], const f2 (&rbar)[3], f2 kq,
                                                       f2 (&Pa)[3], f2 (&Pb)[3], f2& Ka, f2& Kb)
{
    const f2 one = {1.0f, 1.0f};
    f2 da0, da1, da2, db0, db1, db2, ta0, ta1, ta2, tb0, tb1, tb2;
    asm("v_pk_add_f32 %[da0], %[ra0], %[m0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %[da1], %[ra1], %[m1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %[da2], %[ra2], %[m2] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %[db0], %[rb0], %[m0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %[db1], %[rb1], %[m1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %[db2], %[rb2], %[m2] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_mul_f32 %[ta0], %[kq], %[da0]\n\t"
        "v_pk_mul_f32 %[ta1], %[kq], %[da1]\n\t"
        "v_pk_mul_f32 %[ta2], %[kq], %[da2]\n\t"
        "v_pk_mul_f32 %[tb0], %[kq], %[db0]\n\t"
        "v_pk_mul_f32 %[tb1], %[kq], %[db1]\n\t"
        "v_pk_mul_f32 %[tb2], %[kq], %[db2]\n\t"
        "v_pk_mul_f32 %[da0], %[da0], %[ta0]\n\t"
        "v_pk_mul_f32 %[da2], %[da2], %[ta2]\n\t"
        "v_pk_mul_f32 %[db0], %[db0], %[tb0]\n\t"
        "v_pk_mul_f32 %[db2], %[db2], %[tb2]\n\t"
        "v_pk_mul_f32 %[da1], %[da1], %[ta1]\n\t"
        "v_pk_mul_f32 %[db1], %[db1], %[tb1]\n\t"
        "v_pk_add_f32 %[da0], %[da0], %[da2]\n\t"
        "v_pk_add_f32 %[db0], %[db0], %[db2]\n\t"
        "v_pk_add_f32 %[da0], %[da0], %[da1]\n\t"
        "v_pk_add_f32 %[db0], %[db0], %[db1]\n\t"
        "v_pk_add_f32 %[ka], %[one], %[da0] neg_lo:[0,1] neg_hi:[0,1] clamp\n\t"
        "v_pk_add_f32 %[kb], %[one], %[db0] neg_lo:[0,1] neg_hi:[0,1] clamp\n\t"
        "v_pk_mul_f32 %[ta0], %[ra0], %[ka]\n\t"
        "v_pk_mul_f32 %[ta1], %[ra1], %[ka]\n\t"
        "v_pk_mul_f32 %[ta2], %[ra2], %[ka]\n\t"
        "v_pk_mul_f32 %[tb0], %[rb0], %[kb]\n\t"
        "v_pk_mul_f32 %[tb1], %[rb1], %[kb]\n\t"
        "v_pk_mul_f32 %[tb2], %[rb2], %[kb]"
        : [da0] "=&v"(da0), [da1] "=&v"(da1), [da2] "=&v"(da2), [db0] "=&v"(db0), [db1] "=&v"(db1), [db2] "=&v"(db2),
          [ta0] "=&v"(ta0), [ta1] "=&v"(ta1), [ta2] "=&v"(ta2), [tb0] "=&v"(tb0), [tb1] "=&v"(tb1), [tb2] "=&v"(tb2),
          [ka] "=&v"(Ka), [kb] "=&v"(Kb)
        : [ra0] "v"(ra[0]), [ra1] "v"(ra[1]), [ra2] "v"(ra[2]), [rb0] "v"(rb[0]), [rb1] "v"(rb[1]), [rb2] "v"(rb[2]),
          [m0] "v"(rbar[0]), [m1] "v"(rbar[1]), [m2] "v"(rbar[2]), [kq] "v"(kq), [one] "v"(one));
    Pa[0] = ta0;
    Pa[1] = ta1;
    Pa[2] = ta2;
    Pb[0] = tb0;
    Pb[1] = tb1;
    Pb[2] = tb2;
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

// The same for ROW tiles, balanced (round 4): XCD x takes the scanlines x, x + 8, x + 16, ... of the launch -- every scanline's
// `row_blocks` logical blocks (tiles x hypothesis groups) in a row, so a scanline's tiles still share one L2 -- instead of
// one contiguous eighth of the scanlines.  A contiguous eighth is as good while every scanline holds the same work; an
// edge mask does not (textured and flat regions of an image; the bands of a sweep's later visits), and the launch then
// lasts as long as its busiest XCD: 15.4 ms against 12.3 for the same pixel count spread evenly
// (profiles/r04_k2_variants.md section 7).  Scanlines do not share EPI rows, so nothing is lost by interleaving them.
// Returns a logical block >= rows * row_blocks for the surplus blocks of a launch (they return at once).
__device__ __forceinline__ int xcd_logical_block_rows(int b, int row_blocks)
{
    const int slot = b >> 3;
    const int k = slot / row_blocks;
    return ((b & 7) + 8 * k) * row_blocks + (slot - k * row_blocks);
}

}  // namespace rslf
