// K3: selective median filter.
//
// rslf::selective_median_filter (include/rslf_depth_computation_core.hpp:663-718):
// at every pixel of the edge mask, the n/2-th order statistic of the depths of
// the window pixels that are also in the mask and whose radiance at s_hat is
// within epsilon (norm<>, src/rslf_types.cpp:80-91) of the centre's; 0 elsewhere
// (core.hpp:678-679).  One thread per pixel; candidates are parked in LDS
// ([slot][thread], conflict-free; size*size*1 KiB of dynamic LDS per block, so
// the default 5x5 window leaves room for six blocks per CU) and ranked by
// counting -- value-deterministic like std::nth_element (core.hpp:713).  Reads rows v-w..v+w, so it runs as its
// own launch after K2 (a <2 us boundary).
#pragma once

#include "rslf_device.hpp"
#include "rslf_plan.hpp"   // kMedianMaxSize

namespace rslf {


// The median of one mask pixel (v, u); `cand` is the block's [size*size][256] LDS array.
template <int C>
__device__ __forceinline__ float selective_median_pixel(const VolView& vol, const float* __restrict__ src,
                                                        const uint8_t* __restrict__ mask, int s_hat, int size, float eps, int v,
                                                        int u, float (*cand)[256])
{
    const int U = vol.U, V = vol.V;
    const int w = (size - 1) / 2;
    const float* rc = vol.row(v, s_hat);
    float ec[C];
#pragma unroll
    for (int c = 0; c < C; c++)
        ec[c] = rc[u * C + c];

    int n = 0;
    const int k0 = max(0, v - w), k1 = min(V, v + w + 1);
    const int l0 = max(0, u - w), l1 = min(U, u + w + 1);
    // a window row at a time: its masks, radiances and depths are loaded together (up to 3 * 7 independent loads in
    // flight) before any is tested -- a thread's 25 neighbours one dependent load after the other is what a sparse visit's
    // median used to wait for
    for (int k = k0; k < k1; k++) {
        const float* rk = vol.row(k, s_hat);
        const long long rowo = (long long)k * U;
        uint8_t mk[kMedianMaxSize];
        float sv[kMedianMaxSize], rv[C][kMedianMaxSize];
#pragma unroll
        for (int j = 0; j < kMedianMaxSize; j++) {
            const bool in = l0 + j < l1;
            const int l = in ? l0 + j : l0;
            mk[j] = mask[rowo + l];                 // (unconditional, gated afterwards: see the 5 x 5 form)
            mk[j] = in ? mk[j] : (uint8_t)0;
            sv[j] = src[rowo + l];
#pragma unroll
            for (int c = 0; c < C; c++)
                rv[c][j] = rk[l * C + c];
        }
#pragma unroll
        for (int j = 0; j < kMedianMaxSize; j++) {
            if (!mk[j])
                continue;
            float df[C];
#pragma unroll
            for (int c = 0; c < C; c++)
                df[c] = ec[c] - rv[c][j];
            const float nr = (C == 1) ? norm1(df[0]) : norm3(df[0], df[C > 1 ? 1 : 0], df[C > 2 ? 2 : 0]);
            if (nr < eps) {
                cand[n][threadIdx.x] = sv[j];
                n++;
            }
        }
    }
    // element of rank n/2 in ascending order (ties share a value, any of them is right)
    const int target = n / 2;
    float out = 0.0f;
    for (int i = 0; i < n; i++) {
        const float x = cand[i][threadIdx.x];
        int less = 0, eq = 0;
        for (int j = 0; j < n; j++) {
            const float y = cand[j][threadIdx.x];
            less += (y < x) ? 1 : 0;
            eq += (y == x) ? 1 : 0;
        }
        if (less <= target && target < less + eq) {
            out = x;
            break;
        }
    }
    return out;
}

// The default 5 x 5 window without the LDS array: the 25 slots stay in registers (a slot that is outside the image,
// unmasked or too far in radiance holds +inf), Batcher's odd-even merge sort orders them with min / max pairs -- the
// comparators that would touch one of the 7 padding slots are no-ops and are dropped at compile time -- and the answer is
// slot n/2.  The generic form's rank count reads its candidates back from LDS n*n/2 times, one dependent read after
// the other: on a sparse visit that chain was most of the kernel's time.  Disparities are finite, so +inf marks a free slot.
template <int C>
__device__ __forceinline__ float selective_median_pixel_5x5(const VolView& vol, const float* __restrict__ src,
                                                            const uint8_t* __restrict__ mask, int s_hat, float eps, int v, int u)
{
    constexpr int W = 5, N = 32;
    const int U = vol.U, V = vol.V;
    const float* rc = vol.row(v, s_hat);
    float ec[C];
#pragma unroll
    for (int c = 0; c < C; c++)
        ec[c] = rc[u * C + c];
    float a[N];
#pragma unroll
    for (int i = W * W; i < N; i++)
        a[i] = __builtin_inff();
    int n = 0;
#pragma unroll
    for (int dk = 0; dk < W; dk++) {
        const int k = v - W / 2 + dk;
        const bool kin = k >= 0 && k < V;
        const int kc = kin ? k : v;
        const float* rk = vol.row(kc, s_hat);
        const long long rowo = (long long)kc * U;
        uint8_t mk[W];
        float sv[W], rv[C][W];
#pragma unroll
        for (int j = 0; j < W; j++) {
            const int l = u - W / 2 + j;
            const bool in = kin && l >= 0 && l < U;
            const int lc = in ? l : u;
            // (loaded unconditionally, at the clamped column, and gated afterwards: `in ? mask[..] : 0` became a branch
            // around the load and a full wait behind it -- 25 round trips one after the other per pixel)
            mk[j] = mask[rowo + lc];
            mk[j] = in ? mk[j] : (uint8_t)0;
            sv[j] = src[rowo + lc];
#pragma unroll
            for (int c = 0; c < C; c++)
                rv[c][j] = rk[lc * C + c];
        }
#pragma unroll
        for (int j = 0; j < W; j++) {
            float df[C];
#pragma unroll
            for (int c = 0; c < C; c++)
                df[c] = ec[c] - rv[c][j];
            const float nr = (C == 1) ? norm1(df[0]) : norm3(df[0], df[C > 1 ? 1 : 0], df[C > 2 ? 2 : 0]);
            const bool take = mk[j] && nr < eps;
            a[dk * W + j] = take ? sv[j] : __builtin_inff();
            n += take ? 1 : 0;
        }
    }
    // Batcher's odd-even merge sort on 32 slots, ascending
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
    // element of rank n/2 (n <= 25: ranks 0..12)
    const int target = n / 2;
    float out = a[0];
#pragma unroll
    for (int r = 1; r <= (W * W) / 2; r++)
        out = (target == r) ? a[r] : out;
    return n ? out : 0.0f;
}

// The median of the mask pixel (v, u) for any window size.
template <int C>
__device__ __forceinline__ float selective_median_any(const VolView& vol, const float* __restrict__ src,
                                                      const uint8_t* __restrict__ mask, int s_hat, int size, float eps, int v, int u,
                                                      float (*cand)[256])
{
    if (size == 5)
        return selective_median_pixel_5x5<C>(vol, src, mask, s_hat, eps, v, u);
    return selective_median_pixel<C>(vol, src, mask, s_hat, size, eps, v, u, cand);
}

template <int C>
__global__ __launch_bounds__(256) void k3_selective_median(VolView vol, const float* __restrict__ src,
                                                          float* __restrict__ dst, const uint8_t* __restrict__ mask,
                                                          int s_hat, int size, float eps)
{
    extern __shared__ __attribute__((aligned(16))) float s_median_cand[];   // [size*size][256]
    float (*cand)[256] = reinterpret_cast<float (*)[256]>(s_median_cand);
    const int v = blockIdx.y;
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= vol.U)
        return;
    const long long o = (long long)v * vol.U + u;
    dst[o] = mask[o] ? selective_median_any<C>(vol, src, mask, s_hat, size, eps, v, u, cand) : 0.0f;
}

}  // namespace rslf
