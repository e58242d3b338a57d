// K0: normalise + copy the caller's light field into the HBM slab, and the volume's min/max.
//
// Touches each voxel once: HBM-bound, coalesced along u.  Included by rslf_core.hip only (k0_minmax_final
// is not a template: one definition per library).
#pragma once

#include "rslf_device.hpp"

namespace rslf {


// ---- K0: normalise + copy into the slab --------------------------------------
// Replaces Depth1DComputer_pile's constructor copy/convertTo
// (include/rslf_depth_computation.hpp:463-477) and, for the image-major source,
// rslf::build_epis_from_imgs (src/rslf_io.cpp:194-227).
// The slab row of (v, s) is pitch pixels x C interleaved channels (rslf_device.hpp).
//   EPI-major   source: element (v,s,u,c) at src[((v*S + s)*U + u)*C + c]
//   image-major source: element (v,s,u,c) at src[((s*V + v)*U + u)*C + c]
// One block per (v, s) row; block-level min/max partials for the volume range.
template <typename SrcT, bool IMAGE_MAJOR>
__global__ __launch_bounds__(256) void k0_pack(const SrcT* __restrict__ src, float* __restrict__ dst,
                                              int V0, int Vn, int Vsrc, int S, int U, int C, int pitch,
                                              float scale, float* __restrict__ partial_minmax)
{
    const int row = blockIdx.x;   // over Vn * S
    const int vl = row / S;       // local scanline of this chunk
    const int s = row - vl * S;
    const int v = V0 + vl;
    const long long src_row = IMAGE_MAJOR ? ((long long)s * Vsrc + vl) : ((long long)vl * S + s);
    const SrcT* in = src + src_row * (long long)U * C;
    float* out = dst + ((long long)v * S + s) * (long long)C * pitch;

    float mn = INFINITY, mx = -INFINITY;
    // source and slab rows are both pixel-major with interleaved channels: a scaled copy, then zero padding
    for (int i = threadIdx.x; i < pitch * C; i += blockDim.x) {
        float x = 0.0f;   // zero padding beyond U (a 0-weight tap must stay finite)
        if (i < U * C) {
            // dc.hpp:470 / :474: convertTo with a float scale
            x = (float)in[i] * scale;
            mn = (x != x) ? -INFINITY : fminf(mn, x);   // a NaN radiance sends the scan to the generic kernel,
            mx = fmaxf(mx, x);                          // the only variant that keeps NaNs apart (fminf would skip it)
        }
        out[i] = x;
    }
    // block reduce
    for (int o = 32; o > 0; o >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, o));
        mx = fmaxf(mx, __shfl_xor(mx, o));
    }
    __shared__ float smn[4], smx[4];
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        smn[w] = mn;
        smx[w] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < (int)(blockDim.x >> 6); i++) {
            mn = fminf(mn, smn[i]);
            mx = fmaxf(mx, smx[i]);
        }
        partial_minmax[2 * (long long)row] = mn;
        partial_minmax[2 * (long long)row + 1] = mx;
    }
}

// Image-major source with the per-EPI options of rslf::build_epis_from_imgs (src/rslf_io.cpp:194-227): the EPI
// of scanline v is E[i][x] = img_i(v, x); `transpose` makes the slab hold E^T (views = image columns, columns =
// images), `rotate_180` then turns the result by 180 degrees (both axes reversed).  src: [n_imgs][Vn][cols*C] of
// this chunk.  One block per slab row (v, s); the reads are strided when transposed -- a one-off pass.
template <typename SrcT>
__global__ __launch_bounds__(256) void k0_pack_images_xf(const SrcT* __restrict__ src, float* __restrict__ dst, int V0, int Vn,
                                                        int n_imgs, int cols, int S, int U, int C, int pitch, float scale,
                                                        int transpose, int rotate_180, float* __restrict__ partial_minmax)
{
    const int row = blockIdx.x;   // over Vn * S
    const int vl = row / S;
    const int s = row - vl * S;
    const int v = V0 + vl;
    float* out = dst + ((long long)v * S + s) * (long long)C * pitch;
    float mn = INFINITY, mx = -INFINITY;
    for (int i = threadIdx.x; i < pitch * C; i += blockDim.x) {
        const int u = i / C, c = i - u * C;
        float x = 0.0f;
        if (u < U) {
            const int sr = rotate_180 ? S - 1 - s : s;      // position in the un-rotated EPI
            const int ur = rotate_180 ? U - 1 - u : u;
            const int img = transpose ? ur : sr;            // E^T[s][u] = E[u][s]
            const int col = transpose ? sr : ur;
            x = (float)src[(((long long)img * Vn + vl) * cols + col) * C + c] * scale;
            mn = (x != x) ? -INFINITY : fminf(mn, x);   // a NaN radiance sends the scan to the generic kernel,
            mx = fmaxf(mx, x);                          // the only variant that keeps NaNs apart (fminf would skip it)
        }
        out[i] = x;
    }
    for (int o = 32; o > 0; o >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, o));
        mx = fmaxf(mx, __shfl_xor(mx, o));
    }
    __shared__ float smn[4], smx[4];
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        smn[w] = mn;
        smx[w] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < (int)(blockDim.x >> 6); i++) {
            mn = fminf(mn, smn[i]);
            mx = fmaxf(mx, smx[i]);
        }
        partial_minmax[2 * (long long)row] = mn;
        partial_minmax[2 * (long long)row + 1] = mx;
    }
}

// Folds the per-row partials into minmax[0..1] (running values, so chunks chain).
__global__ __launch_bounds__(256) void k0_minmax_final(const float* __restrict__ partial, int n, float* __restrict__ minmax)
{
    float mn = INFINITY, mx = -INFINITY;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        mn = fminf(mn, partial[2 * i]);
        mx = fmaxf(mx, partial[2 * i + 1]);
    }
    for (int o = 32; o > 0; o >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, o));
        mx = fmaxf(mx, __shfl_xor(mx, o));
    }
    __shared__ float smn[4], smx[4];
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        smn[w] = mn;
        smx[w] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 4; i++) {
            mn = fminf(mn, smn[i]);
            mx = fmaxf(mx, smx[i]);
        }
        minmax[0] = fminf(minmax[0], mn);
        minmax[1] = fmaxf(minmax[1], mx);
    }
}

}  // namespace rslf
