// K3: selective median filter, any window size.
//
// rslf::selective_median_filter (include/rslf_depth_computation_core.hpp:663-718): at every pixel of the edge mask, the
// n/2-th order statistic (std::nth_element, :713) of the depths of the window pixels that are also in the mask and whose
// radiance at s_hat is within epsilon (norm<>, src/rslf_types.cpp:80-91) of the centre's; 0 elsewhere (:678-679).  The
// window is width = (a_size - 1) / 2 pixels either side (:686): an even size is the next smaller odd window, and any
// size runs -- the report's parameter table documents 11 (report/rs_report.tex:388).
//
// One workgroup = 256 consecutive pixels of one scanline.  It loads the rows v-w..v+w x columns u0-w..u0+255+w of the
// raw depths and the s_hat radiances ONCE into LDS (coalesced row loads; a pixel outside the image or the mask holds a
// NaN radiance, which fails every test exactly as the reference's own `mask && norm < eps` does), and every thread takes
// its window from there -- (2w+1)^2 * (2 + C) per-lane global loads per pixel were what the kernel waited for.
// Selection, by window side W = 2w + 1 (plan::median_plan):
//   W <= 11   the W*W slots stay in registers (+inf = free slot), Batcher's odd-even merge sort with min / max pairs,
//             answer = slot n/2
//   W <= 31   one predicate bit per window pixel (a word per window row), then a radix select over the depths' order-
//             preserving integer keys in LDS: 32 counting passes at most, fewer when the candidates share high bits
//   beyond    the same radix select straight from global memory, predicate recomputed per pass (no tile would fit):
//             slow, there so that no size the reference runs is refused
// The radiance test itself is one compare: plan::norm_threshold turns `norm<T>(x) < eps` (through double, with a sqrt
// for three channels) into the exactly equivalent |x| < a1 / x.x < s3 on the host.
// Reads rows v-w..v+w, so it runs as its own launch after K2 (a <2 us boundary).
#pragma once

#include "rslf_device.hpp"
#include "rslf_plan.hpp"   // median_plan, NormThreshold

namespace rslf {

constexpr int kMedianBlock = plan::kMedianBlock;   // pixels per workgroup

// norm<T>(df) < eps  (core.hpp:703-706, :1116) with the threshold form of plan::norm_threshold; NaN fails
template <int C>
__device__ __forceinline__ bool norm_below(const float (&df)[C], const plan::NormThreshold& t)
{
    if (C == 1)
        return fabsf(df[0]) < t.a1;
    double s = (double)df[0] * (double)df[0];
    s += (double)df[C > 1 ? 1 : 0] * (double)df[C > 1 ? 1 : 0];
    s += (double)df[C > 2 ? 2 : 0] * (double)df[C > 2 ? 2 : 0];
    return s < t.s3;
}

// float <-> unsigned key with the same order (-0 sorts just below +0: equal as floats, either is "the" median)
__device__ __forceinline__ uint32_t median_key(float x)
{
    const uint32_t b = __float_as_uint(x);
    return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float median_unkey(uint32_t k)
{
    return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}

// The workgroup's tile in LDS: dep[rows][tw], rad[C][rows][tw]; rows = 2w + 1, tw = 256 + 2w.  Thread t's window column j
// is tile column t + j: neighbouring lanes read neighbouring words, conflict-free.
struct MedianTile {
    float* dep;
    float* rad;
    int w, rows, tw;
};

__device__ __forceinline__ MedianTile median_tile(float* lds, int w)
{
    MedianTile t;
    t.w = w;
    t.rows = 2 * w + 1;
    t.tw = kMedianBlock + 2 * w;
    t.dep = lds;
    t.rad = lds + t.rows * t.tw;
    return t;
}

// One tile entry: row kc (clamped) of the planes, tile row dk, tile column x.
template <int C, bool KEYS>
__device__ __forceinline__ void median_tile_entry(const VolView& vol, const float* __restrict__ src, const uint8_t* __restrict__ mask,
                                                  const float* __restrict__ rk, long long rowo, bool kin, int u_first, int dk, int x,
                                                  const MedianTile& t)
{
    const int l = u_first - t.w + x;
    const bool in = kin && l >= 0 && l < vol.U;
    const int lc = min(max(l, 0), vol.U - 1);
    // (unconditional loads at the clamped address, gated afterwards: a load under a condition is a basic block of its own
    // with a full wait behind it)
    const uint8_t m = mask[rowo + lc];
    const float d = src[rowo + lc];
    float r[C];
#pragma unroll
    for (int c = 0; c < C; c++)
        r[c] = rk[lc * C + c];
    const bool take = in && m != 0;
    t.dep[dk * t.tw + x] = KEYS ? __uint_as_float(median_key(d)) : d;
#pragma unroll
    for (int c = 0; c < C; c++)
        t.rad[(c * t.rows + dk) * t.tw + x] = take ? r[c] : __builtin_nanf("");
}

// Fill the tile for the block of pixels (v, u_first .. u_first + 255).  WT > 0: the window side is a template constant
// and the row loop is unrolled (every row's loads in flight together); WT == 0: t.w at run time.
template <int C, bool KEYS, int WT>
__device__ __forceinline__ void median_tile_fill(const VolView& vol, const float* __restrict__ src, const uint8_t* __restrict__ mask,
                                                 int s_hat, int v, int u_first, const MedianTile& t)
{
    const int tid = threadIdx.x;
    if constexpr (WT > 0) {
#pragma unroll
        for (int dk = 0; dk < WT; dk++) {
            const int k = v - t.w + dk;
            const bool kin = k >= 0 && k < vol.V;
            const int kc = kin ? k : v;
            median_tile_entry<C, KEYS>(vol, src, mask, vol.row(kc, s_hat), (long long)kc * vol.U, kin, u_first, dk, tid, t);
        }
    } else {
        for (int dk = 0; dk < t.rows; dk++) {
            const int k = v - t.w + dk;
            const bool kin = k >= 0 && k < vol.V;
            const int kc = kin ? k : v;
            median_tile_entry<C, KEYS>(vol, src, mask, vol.row(kc, s_hat), (long long)kc * vol.U, kin, u_first, dk, tid, t);
        }
    }
    if (tid < 2 * t.w)   // the 2w columns past the block's last pixel
        for (int dk = 0; dk < t.rows; dk++) {
            const int k = v - t.w + dk;
            const bool kin = k >= 0 && k < vol.V;
            const int kc = kin ? k : v;
            median_tile_entry<C, KEYS>(vol, src, mask, vol.row(kc, s_hat), (long long)kc * vol.U, kin, u_first, dk, kMedianBlock + tid, t);
        }
    __syncthreads();
}

constexpr int median_pow2_ceil(int n)
{
    int p = 1;
    while (p < n)
        p *= 2;
    return p;
}

// W x W window, W a template constant: slots in registers, sorting network, slot n/2.  Disparities are finite, so +inf
// marks a free slot; the comparators that would touch one of the padding slots past W*W are no-ops and are dropped at
// compile time.
template <int C, int W>
__device__ __forceinline__ float median_select_net(const MedianTile& t, const plan::NormThreshold& thr)
{
    constexpr int N = median_pow2_ceil(W * W), w = W / 2, TW = kMedianBlock + 2 * w;
    const int tx = threadIdx.x;
    float ec[C];
#pragma unroll
    for (int c = 0; c < C; c++)
        ec[c] = t.rad[(c * W + w) * TW + tx + w];
    float a[N];
#pragma unroll
    for (int i = W * W; i < N; i++)
        a[i] = __builtin_inff();
    int n = 0;
#pragma unroll
    for (int dk = 0; dk < W; dk++)
#pragma unroll
        for (int j = 0; j < W; j++) {
            float df[C];
#pragma unroll
            for (int c = 0; c < C; c++)
                df[c] = ec[c] - t.rad[(c * W + dk) * TW + tx + j];
            const bool take = norm_below<C>(df, thr);
            a[dk * W + j] = take ? t.dep[dk * TW + tx + j] : __builtin_inff();
            n += take ? 1 : 0;
        }
    // Batcher's odd-even merge sort on N slots, ascending
#pragma unroll
    for (int p = 1; p < N; p *= 2)
#pragma unroll
        for (int k = p; k >= 1; k /= 2)
#pragma unroll
            for (int j = k % p; j + k < N; j += 2 * k)
#pragma unroll
                for (int i = 0; i < k; i++)
                    if ((i + j) / (2 * p) == (i + j + k) / (2 * p) && i + j + k < W * W) {
                        const float lo = __builtin_fminf(a[i + j], a[i + j + k]);
                        const float hi = __builtin_fmaxf(a[i + j], a[i + j + k]);
                        a[i + j] = lo;
                        a[i + j + k] = hi;
                    }
    // element of rank n/2 (n <= W*W: ranks 0 .. W*W/2)
    const int target = n / 2;
    float out = a[0];
#pragma unroll
    for (int r = 1; r <= (W * W) / 2; r++)
        out = (target == r) ? a[r] : out;
    return n ? out : 0.0f;
}

// Radix select: the key of rank `target` among the candidates = the largest x with #{key < x} <= target, built bit by
// bit from the first bit the candidates' keys differ in (kmin ^ kmax).  `count(trial)` = #{candidate keys < trial}.
template <class Count>
__device__ __forceinline__ uint32_t median_radix_select(uint32_t kmin, uint32_t kmax, int target, Count count)
{
    const uint32_t diff = kmin ^ kmax;
    if (diff == 0)
        return kmin;
    const int nb = 32 - __clz(diff);                              // low bits in which candidates differ
    uint32_t prefix = nb == 32 ? 0u : (kmin >> nb) << nb;
    for (int bit = nb - 1; bit >= 0; bit--) {
        const uint32_t trial = prefix | (1u << bit);
        if (count(trial) <= target)
            prefix = trial;
    }
    return prefix;
}

// Window side up to 31 at run time: predicate bits in registers (one word per window row), keys in the tile.
template <int C>
__device__ __forceinline__ float median_select_radix(const MedianTile& t, const plan::NormThreshold& thr)
{
    constexpr int R = plan::kMedianTileMaxSide;
    const int tx = threadIdx.x;
    float ec[C];
#pragma unroll
    for (int c = 0; c < C; c++)
        ec[c] = t.rad[(c * t.rows + t.w) * t.tw + tx + t.w];
    const uint32_t* keys = reinterpret_cast<const uint32_t*>(t.dep);
    uint32_t pm[R];
    uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
    int n = 0;
    // (eight window columns at a time, their LDS reads issued together: one read, one wait per candidate left the two
    // waves a SIMD holds at this register count waiting most of the time)
    constexpr int B = 8;
#pragma unroll
    for (int dk = 0; dk < R; dk++) {
        uint32_t m = 0;
        if (dk < t.rows)
            for (int j0 = 0; j0 < t.rows; j0 += B) {
                float rv[C][B];
                uint32_t kv[B];
#pragma unroll
                for (int b = 0; b < B; b++) {
                    const int j = min(j0 + b, t.rows - 1);
#pragma unroll
                    for (int c = 0; c < C; c++)
                        rv[c][b] = t.rad[(c * t.rows + dk) * t.tw + tx + j];
                    kv[b] = keys[dk * t.tw + tx + j];
                }
#pragma unroll
                for (int b = 0; b < B; b++) {
                    float df[C];
#pragma unroll
                    for (int c = 0; c < C; c++)
                        df[c] = ec[c] - rv[c][b];
                    if (j0 + b < t.rows && norm_below<C>(df, thr)) {
                        m |= 1u << (j0 + b);
                        kmin = min(kmin, kv[b]);
                        kmax = max(kmax, kv[b]);
                        n++;
                    }
                }
            }
        pm[dk] = m;
    }
    if (n == 0)
        return 0.0f;
    const uint32_t key = median_radix_select(kmin, kmax, n / 2, [&](uint32_t trial) {
        int cnt = 0;
#pragma unroll
        for (int dk = 0; dk < R; dk++)
            if (dk < t.rows) {
                const uint32_t m = pm[dk];
                const uint32_t* kr = keys + dk * t.tw + tx;
                for (int j0 = 0; j0 < t.rows; j0 += B) {
                    uint32_t kv[B];
#pragma unroll
                    for (int b = 0; b < B; b++)
                        kv[b] = kr[min(j0 + b, t.rows - 1)];
                    const uint32_t mb = m >> j0;   // (bits past the row's last column are 0)
#pragma unroll
                    for (int b = 0; b < B; b++)
                        cnt += (int)((mb >> b) & 1u) & (int)(kv[b] < trial);
                }
            }
        return cnt;
    });
    return median_unkey(key);
}

// Any window side, straight from global memory (no tile: windows past plan::kMedianTileMaxSide).
template <int C>
__device__ __forceinline__ float median_select_global(const VolView& vol, const float* __restrict__ src, const uint8_t* __restrict__ mask,
                                                      int s_hat, int w, const plan::NormThreshold& thr, int v, int u)
{
    const int U = vol.U, V = vol.V;
    const float* rc = vol.row(v, s_hat);
    float ec[C];
#pragma unroll
    for (int c = 0; c < C; c++)
        ec[c] = rc[u * C + c];
    const int k0 = max(0, v - w), k1 = (int)min((long long)V, (long long)v + w + 1);   // (w < 2^30: no overflow)
    const int l0 = max(0, u - w), l1 = (int)min((long long)U, (long long)u + w + 1);
    // walks the candidates; f(key) for each
    auto each = [&](auto f) {
        for (int k = k0; k < k1; k++) {
            const float* rk = vol.row(k, s_hat);
            const long long rowo = (long long)k * U;
            for (int l = l0; l < l1; l++) {
                if (!mask[rowo + l])
                    continue;
                float df[C];
#pragma unroll
                for (int c = 0; c < C; c++)
                    df[c] = ec[c] - rk[l * C + c];
                if (norm_below<C>(df, thr))
                    f(median_key(src[rowo + l]));
            }
        }
    };
    uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
    int n = 0;
    each([&](uint32_t k) {
        kmin = min(kmin, k);
        kmax = max(kmax, k);
        n++;
    });
    if (n == 0)
        return 0.0f;
    const uint32_t key = median_radix_select(kmin, kmax, n / 2, [&](uint32_t trial) {
        int cnt = 0;
        each([&](uint32_t k) { cnt += k < trial ? 1 : 0; });
        return cnt;
    });
    return median_unkey(key);
}

// The filtered value of the block's pixels: every thread of the workgroup calls this (the tile is filled together);
// returns 0 for a thread outside the row or the mask.  MODE = plan::MedianPlan::mode: > 0 the window side of the
// register network, 0 the tile + radix form (window half-width w at run time), < 0 the global form.
// `lds`: plan::median_lds_bytes(w, C) of dynamic LDS (none for MODE < 0).
template <int C, int MODE>
__device__ __forceinline__ float selective_median_block(const VolView& vol, const float* __restrict__ src,
                                                        const uint8_t* __restrict__ mask, int s_hat, int w,
                                                        const plan::NormThreshold& thr, int v, int u_first, float* lds)
{
    const int u = u_first + threadIdx.x;
    const bool inside = u < vol.U;
    const bool on = inside && mask[(long long)v * vol.U + (inside ? u : 0)] != 0;
    if constexpr (MODE < 0) {
        return on ? median_select_global<C>(vol, src, mask, s_hat, w, thr, v, u) : 0.0f;
    } else {
        const MedianTile t = median_tile(lds, MODE > 0 ? MODE / 2 : w);
        median_tile_fill<C, MODE == 0, MODE>(vol, src, mask, s_hat, v, u_first, t);
        if (!on)
            return 0.0f;
        if constexpr (MODE > 0)
            return median_select_net<C, MODE>(t, thr);
        else
            return median_select_radix<C>(t, thr);
    }
}

template <int C, int MODE>
__global__ __launch_bounds__(kMedianBlock) void k3_selective_median(VolView vol, const float* __restrict__ src,
                                                                   float* __restrict__ dst, const uint8_t* __restrict__ mask,
                                                                   int s_hat, int w, plan::NormThreshold thr)
{
    extern __shared__ __attribute__((aligned(16))) float s_median_tile[];
    const int v = blockIdx.y;
    const int u_first = blockIdx.x * kMedianBlock;
    const float out = selective_median_block<C, MODE>(vol, src, mask, s_hat, w, thr, v, u_first, s_median_tile);
    const int u = u_first + threadIdx.x;
    if (u < vol.U)
        dst[(long long)v * vol.U + u] = out;
}

// Host-side dispatch over the compiled modes: RSLF_MEDIAN_MODES(X, C) expands X(C, MODE) for each.
#define RSLF_MEDIAN_MODES(X, C) X(C, 1) X(C, 3) X(C, 5) X(C, 7) X(C, 9) X(C, 11) X(C, 0) X(C, -1)

}  // namespace rslf
