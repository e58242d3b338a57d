// The packed confident-pixel list of sparse scan launches: a device function shared by K1's compaction kernel
// (k1_edge.hpp) and the 2-D sweep's apply pass (k4_propagate.hpp), which lists the NEXT visit's pixels.
#pragma once

#include "rslf_device.hpp"

namespace rslf {

// Packed variant for sparse launches: ONE list of pixel indices v*U + u over all scanlines, so that a
// scan wavefront is full even when a scanline holds two or three pixels.  A block counts its row, claims
// a range of the list with one atomic (rows land in arrival order; a pixel's result does not depend on
// where in the list it sits), then writes the row's ascending u.  *packed_n must be 0 on entry.
// `rowbase` (nullable): where the row's entries start in the list -- a row's entries are contiguous, so a scan can also
// take the rows that hold many pixels as ROW tiles straight from the packed list (k2_scan.hpp, ScanArgs::rowbase).
__device__ __forceinline__ void compact_row_packed(int v, const uint8_t* __restrict__ edge_mask, uint8_t* scan_mask, int U,
                                                   int* __restrict__ list, int* __restrict__ count,
                                                   unsigned long long* __restrict__ total, int* __restrict__ packed_n,
                                                   int* __restrict__ rowbase = nullptr)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __shared__ int wave_tot[4];
    __shared__ int base_s;
    int mine = 0;
    for (int u = threadIdx.x; u < U; u += 256) {
        uint8_t m = edge_mask[(long long)v * U + u];
        if (scan_mask)
            m &= scan_mask[(long long)v * U + u];
        mine += m != 0;
    }
    for (int o = 32; o > 0; o >>= 1)
        mine += __shfl_xor(mine, o);
    if (lane == 0)
        wave_tot[w] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int row = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
        count[v] = row;
        base_s = row ? atomicAdd(packed_n, row) : 0;
        if (rowbase)
            rowbase[v] = base_s;
        if (row)
            atomicAdd(total, (unsigned long long)row);
    }
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
            list[off + rank] = (int)((unsigned)v * (unsigned)U + (unsigned)u);
        __syncthreads();
        if (threadIdx.x == 0)
            base_s += wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
        __syncthreads();
    }
}

}  // namespace rslf
