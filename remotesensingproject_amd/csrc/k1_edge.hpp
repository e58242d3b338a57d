// K1 (edge confidence, optional opening of its mask) and the confident-pixel compaction.  (K0: k0_pack.hpp.)
// Included by rslf_pile.hip only (the non-template kernels: one definition per library).
//
// These touch each voxel / pixel a constant number of times: HBM-bound,
// coalesced along u, negligible next to the scan.
#pragma once

#include "rslf_device.hpp"
#include "rslf_plan.hpp"   // MorphElement
#include "k_compact.hpp"

namespace rslf {

// ---- K1: edge confidence ---------------------------------------------------
// rslf::compute_1D_edge_confidence (core.hpp:426-478) for every scanline
// (core.hpp:728-757).  One thread per pixel, grid (ceil(U/256), V).
//   C_e(u) += sum_{j != centre} sum_c (E_c[u] - E_c[refl101(u + j - centre)])^2
// accumulated in that order into the caller's plane (core.cpp:6-23), shadow cut
// (core.hpp:464-474), mask = C_e > threshold (core.hpp:476).
template <int C>
__global__ __launch_bounds__(256) void k1_edge_confidence(VolView vol, int s, EdgeConsts ec,
                                                         float* __restrict__ Ce, uint8_t* __restrict__ mask)
{
    const int v = blockIdx.y;
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= vol.U)
        return;
    const float* r0 = vol.row(v, s);
    const long long o = (long long)v * vol.U + u;
    const float ce = edge_confidence_pixel<C>(vol, r0, u, ec, Ce[o]);
    Ce[o] = ce;
    mask[o] = (ce > ec.edge_thr) ? 255 : 0;
}

// compute_2D_edge_confidence (core.hpp:918-934) in ONE launch: blockIdx.z is the view, planes are [S][V][U].
template <int C>
__global__ __launch_bounds__(256) void k1_edge_confidence_views(VolView vol, EdgeConsts ec, float* __restrict__ Ce_svu,
                                                               uint8_t* __restrict__ mask_svu)
{
    const int s = blockIdx.z, v = blockIdx.y;
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= vol.U)
        return;
    const float* r0 = vol.row(v, s);
    const long long o = ((long long)s * vol.V + v) * vol.U + u;
    const float ce = edge_confidence_pixel<C>(vol, r0, u, ec, Ce_svu[o]);
    Ce_svu[o] = ce;
    mask_svu[o] = (ce > ec.edge_thr) ? 255 : 0;
}

// One pixel's edge confidence, shadow cut applied (core.hpp:449-474); `ce` is what the caller's plane held.
template <int C>
__device__ __forceinline__ float edge_confidence_pixel(const VolView& vol, const float* __restrict__ r0, int u, const EdgeConsts& ec,
                                                       float ce)
{
    const int centre = (ec.filter_size - 1) / 2;
    float e[C];
#pragma unroll
    for (int c = 0; c < C; c++)
        e[c] = r0[u * C + c];
    for (int j = 0; j < ec.filter_size; j++) {
        if (j == centre)
            continue;
        const int q = reflect101(u + j - centre, vol.U);
#pragma unroll
        for (int c = 0; c < C; c++) {
            const float t = e[c] - r0[q * C + c];
            const float t2 = t * t;
            ce = ce + t2;
        }
    }
    if (ec.cut_shadows) {
        float n;
        if (C == 1)
            n = norm1(e[0]);
        else
            n = norm3(e[0], e[C > 1 ? 1 : 0], e[C > 2 ? 2 : 0]);
        if (n < ec.shadow_level)
            ce = 0.0f;
    }
    return ce;
}

// K1 and the compaction of the scan mask in ONE launch (the pile step: Depth1DComputer_pile::run, dc.hpp:538-547, when
// the edge mask is not opened and the caller passes no scan mask -- then the scan mask IS the edge mask, core.hpp:513).
// One block per scanline: it computes the scanline's C_e and mask 256 pixels at a time and appends the confident u to
// the scanline's list in the same sweep (the list the scan's tiles are cut from).
template <int C>
__global__ __launch_bounds__(256) void k1_edge_confidence_compact(VolView vol, int s, EdgeConsts ec, float* __restrict__ Ce,
                                                                 uint8_t* __restrict__ mask, int* __restrict__ list,
                                                                 int* __restrict__ count, unsigned long long* __restrict__ total)
{
    const int v = blockIdx.x;
    const int U = vol.U;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float* r0 = vol.row(v, s);
    __shared__ int wave_tot[4];
    __shared__ int base_s;
    if (threadIdx.x == 0)
        base_s = 0;
    __syncthreads();
    for (int u0 = 0; u0 < U; u0 += 256) {
        const int u = u0 + threadIdx.x;
        bool f = false;
        if (u < U) {
            const long long o = (long long)v * U + u;
            const float ce = edge_confidence_pixel<C>(vol, r0, u, ec, Ce[o]);
            Ce[o] = ce;
            f = ce > ec.edge_thr;                 // core.hpp:476
            mask[o] = f ? 255 : 0;
        }
        const unsigned long long b = __ballot(f);
        const int rank = __popcll(b & ((1ull << lane) - 1ull));
        if (lane == 0)
            wave_tot[w] = __popcll(b);
        __syncthreads();
        int off = base_s;
        for (int i = 0; i < w; i++)
            off += wave_tot[i];
        if (f)
            list[(long long)v * U + off + rank] = u;
        __syncthreads();
        if (threadIdx.x == 0)
            base_s += wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        count[v] = base_s;
        atomicAdd(total, (unsigned long long)base_s);
    }
}

// ---- optional morphological opening of the edge mask -----------------------
// core.hpp:759-768: cv::morphologyEx(mask, mask, MORPH_OPEN, getStructuringElement(type, Size(k, k))).
// One pass = erosion (min) or dilation (max) over the element's set pixels around the anchor (k/2, k/2);
// pixels outside the plane never win (morphologyDefaultBorderValue).  The element arrives as one bit row
// per kernel row (k <= 31).  HBM-bound byte work: k*k mask reads per pixel, served by L1/L2.
__global__ __launch_bounds__(256) void k1_morph_pass(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int V, int U,
                                                    MorphElement el, int dilate)
{
    const int v = blockIdx.y;
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= U)
        return;
    const int a = el.k / 2;
    int acc = dilate ? 0 : 255;
    for (int i = 0; i < el.k; i++) {
        const int y = v + i - a;
        if (y < 0 || y >= V)
            continue;
        const unsigned bits = el.rows[i];
        for (int j = 0; j < el.k; j++) {
            const int x = u + j - a;
            if (!((bits >> j) & 1u) || x < 0 || x >= U)
                continue;
            const int val = src[(long long)y * U + x];
            acc = dilate ? max(acc, val) : min(acc, val);
        }
    }
    dst[(long long)v * U + u] = (uint8_t)acc;
}

// ---- compaction of the scan mask -------------------------------------------
// core.hpp:510-516: scan mask = edge mask (& caller mask, written back in
// place), then findNonZero.  One block per scanline writes the ascending list
// of confident u into list[v][0..count[v]) so that scan wavefronts stay full
// when the mask is sparse.  Also adds count[v] to *total.
__global__ __launch_bounds__(256) void k_compact_mask(const uint8_t* __restrict__ edge_mask, uint8_t* __restrict__ scan_mask,
                                                     int U, int* __restrict__ list, int* __restrict__ count,
                                                     unsigned long long* __restrict__ total)
{
    const int v = blockIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __shared__ int wave_tot[4];
    __shared__ int base_s;
    if (threadIdx.x == 0)
        base_s = 0;
    __syncthreads();
    for (int u0 = 0; u0 < U; u0 += 256) {
        const int u = u0 + threadIdx.x;
        bool f = false;
        if (u < U) {
            uint8_t m = edge_mask[(long long)v * U + u];
            if (scan_mask) {
                m &= scan_mask[(long long)v * U + u];
                scan_mask[(long long)v * U + u] = m;
            }
            f = m != 0;
        }
        const unsigned long long b = __ballot(f);
        const int rank = __popcll(b & ((1ull << lane) - 1ull));
        if (lane == 0)
            wave_tot[w] = __popcll(b);
        __syncthreads();
        int off = base_s;
        for (int i = 0; i < w; i++)
            off += wave_tot[i];
        if (f)
            list[(long long)v * U + off + rank] = u;
        __syncthreads();
        if (threadIdx.x == 0)
            base_s += wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        count[v] = base_s;
        atomicAdd(total, (unsigned long long)base_s);
    }
}

// (compact_row_packed: k_compact.hpp -- the 2-D sweep's apply pass lists the next visit's pixels with it too)
__global__ __launch_bounds__(256) void k_compact_mask_packed(const uint8_t* __restrict__ edge_mask, uint8_t* __restrict__ scan_mask,
                                                            int U, int* __restrict__ list, int* __restrict__ count,
                                                            unsigned long long* __restrict__ total, int* __restrict__ packed_n,
                                                            int* __restrict__ rowbase)
{
    compact_row_packed(blockIdx.x, edge_mask, scan_mask, U, list, count, total, packed_n, rowbase);
}

}  // namespace rslf
