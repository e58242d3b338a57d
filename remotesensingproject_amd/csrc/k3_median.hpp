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

namespace rslf {

constexpr int kMedianMaxSize = 7;   // window side; 49 LDS slots per thread

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
    for (int k = k0; k < k1; k++) {
        const float* rk = vol.row(k, s_hat);
        for (int l = l0; l < l1; l++) {
            if (!mask[(long long)k * U + l])
                continue;
            float df[C];
#pragma unroll
            for (int c = 0; c < C; c++)
                df[c] = ec[c] - rk[l * C + c];
            const float nr = (C == 1) ? norm1(df[0]) : norm3(df[0], df[C > 1 ? 1 : 0], df[C > 2 ? 2 : 0]);
            if (nr < eps) {
                cand[n][threadIdx.x] = src[(long long)k * U + l];
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
    dst[o] = mask[o] ? selective_median_pixel<C>(vol, src, mask, s_hat, size, eps, v, u, cand) : 0.0f;
}

}  // namespace rslf
