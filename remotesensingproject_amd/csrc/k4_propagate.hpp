// K4: disparity propagation of the 2-D sweep ("next" row).
//
// rslf::compute_2D_depth_epi, propagation part
// (include/rslf_depth_computation_core.hpp:1088-1129, default build: gated by the edge mask,
// :1099-1103).  For every confident pixel u of the visited view s_hat (ascending u within a
// scanline), the reference paints its disparity into every view s at column
// u + round(d * (s_hat - s) * slope) if that pixel is still in the running mask and its radiance is
// within propagation_epsilon of rbar(u); painting clears the mask, so the FIRST u (smallest) that
// qualifies wins a target pixel.  Qualifying does not depend on the order (the mask test only asks
// "not painted yet in this or an earlier visit"), so the sequential loop equals:
//   claim:  every source does atomicMin(winner[target], u) on the targets it qualifies for
//   apply:  every claimed target takes the values of its winner and leaves the mask.
// Both passes are HBM-bound streaming kernels (a few bytes per (s, v, u)).
#pragma once

#include "k_compact.hpp"
#include "k3_median.hpp"
#include "rslf_device.hpp"

#ifndef RSLF_CLAIM_BATCH
#define RSLF_CLAIM_BATCH 8   // views whose running-mask bytes a source pixel has in flight together
#endif

namespace rslf {

constexpr int kNoWinner = 0x7F7F7F7F;   // what hipMemset(0x7F) leaves; >= any column index

// One source pixel's claims (core.hpp:1105-1125): `cur` is its filtered disparity.
template <int C>
__device__ __forceinline__ void propagate_claim_pixel(const VolView& vol, int s_hat, int v, int u, float cur,
                                                      const float* __restrict__ rbar_vu, const uint8_t* __restrict__ mask_svu,
                                                      int* __restrict__ winner_svu, uint8_t* __restrict__ dirty, float slope,
                                                      float prop_eps)
{
    const int nseg = (vol.U + 255) >> 8;
    const long long o = (long long)v * vol.U + u;
    float rb[C];
#pragma unroll
    for (int c = 0; c < C; c++)
        rb[c] = rbar_vu[o * C + c];
    const long long plane = (long long)vol.V * vol.U;
    const long long row = (long long)v * vol.U;
    // eight views at a time: their running-mask bytes, then the radiances of those still unpainted, are loaded together
    // before the first test -- the claims do not depend on one another (atomicMin), only the loads' latency did add up
    constexpr int B = RSLF_CLAIM_BATCH;
    for (int s0 = 0; s0 < vol.S; s0 += B) {
        int ri[B];
        bool live[B];
#pragma unroll
        for (int j = 0; j < B; j++) {
            const int s = s0 + j;
            // core.hpp:1109: u + (int)std::round(depth * (s_hat - s) * slope_factor)
            float off = cur * (float)(s_hat - s);
            off = off * slope;
            ri[j] = u + (int)roundf(off);
            live[j] = s < vol.S && ri[j] >= 0 && ri[j] < vol.U;
            if (live[j])
                live[j] = mask_svu[(long long)s * plane + row + ri[j]] != 0;
        }
        float e[B][C];
#pragma unroll
        for (int j = 0; j < B; j++) {
            if (live[j]) {
                const float* er = vol.row(v, s0 + j);
#pragma unroll
                for (int c = 0; c < C; c++)
                    e[j][c] = er[ri[j] * C + c];
            }
        }
#pragma unroll
        for (int j = 0; j < B; j++) {
            if (!live[j])
                continue;
            float df[C];
#pragma unroll
            for (int c = 0; c < C; c++)
                df[c] = e[j][c] - rb[c];
            const float nr = (C == 1) ? norm1(df[0]) : norm3(df[0], df[C > 1 ? 1 : 0], df[C > 2 ? 2 : 0]);
            if (nr < prop_eps) {   // core.hpp:1116
                atomicMin(&winner_svu[(long long)(s0 + j) * plane + row + ri[j]], u);
                dirty[((long long)(s0 + j) * vol.V + v) * nseg + (ri[j] >> 8)] = 1;   // this 256-column segment holds a claim
            }
        }
    }
}

// K3 + claim in one launch (a sweep visit): a pixel's median needs its neighbours' RAW depths only (what the scan
// left), and its claims need its own median only -- so the thread that filters a pixel also makes its claims.  The
// filtered plane is still written: the apply pass reads the winners' values from it.
template <int C>
__global__ __launch_bounds__(256) void k34_median_claim(VolView vol, int s_hat, const float* __restrict__ raw_vu,
                                                       float* __restrict__ filtered_vu, const uint8_t* __restrict__ edge_mask_vu,
                                                       int size, float eps, const float* __restrict__ rbar_vu,
                                                       const uint8_t* __restrict__ mask_svu, int* __restrict__ winner_svu,
                                                       uint8_t* __restrict__ dirty, float slope, float prop_eps,
                                                       const float* __restrict__ gate_Cd_vu, float disp_thr, int* __restrict__ reset)
{
    // `reset`: the packed list's length, which the scan before this launch was the last to read and the apply pass after
    // it counts up again from 0
    if (reset && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)
        *reset = 0;
    extern __shared__ __attribute__((aligned(16))) float s_median_cand[];   // [size*size][256]
    float (*cand)[256] = reinterpret_cast<float (*)[256]>(s_median_cand);
    const int v = blockIdx.y;
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= vol.U)
        return;
    const long long o = (long long)v * vol.U + u;
    // core.hpp:678-679, :881-892: the median over the edge mask, 0 elsewhere
    const float cur = edge_mask_vu[o] ? selective_median_any<C>(vol, raw_vu, edge_mask_vu, s_hat, size, eps, v, u, cand) : 0.0f;
    filtered_vu[o] = cur;
    if (gate_Cd_vu ? !(gate_Cd_vu[o] > disp_thr) : !edge_mask_vu[o])   // core.hpp:1097-1103
        return;
    propagate_claim_pixel<C>(vol, s_hat, v, u, cur, rbar_vu, mask_svu, winner_svu, dirty, slope, prop_eps);
}

// The apply pass of a visit (core.hpp:1119-1127) and, in the same launch, the pixel list of the NEXT visit's scan: one
// workgroup per (view, scanline) row of the planes; the rows of view s_next come first and, once their claims are
// applied (every write to a row's running mask comes from the row's own workgroup), compact that row of
// edge_mask(s_next) & running mask(s_next) into the packed list exactly as k_compact_mask_packed would have -- one dense
// pass and one launch fewer per visit.  *packed_n must be 0 on entry (k34_median_claim zeroes it: the scan before it
// was its last reader).  s_next < 0: apply only.
// `dirty` has one byte per 256-column segment of every row, set by the claims: after the first visits most segments hold
// no claim, and the pass reads the flags (S*V*ceil(U/256) bytes) instead of the winners (4 bytes per cell of the volume).
// One 256-column segment of row r = s*V + v of the planes.
__device__ __forceinline__ void apply_segment(long long r, int seg, int V, int U, const float* __restrict__ filtered_vu,
                                              const float* __restrict__ Cd_hat_vu, float* __restrict__ depth_svu,
                                              float* __restrict__ Cd_svu, uint8_t* mask_svu, int* __restrict__ winner_svu)
{
    const int u = (seg << 8) + threadIdx.x;
    if (u >= U)
        return;
    const long long t = r * U + u;
    const int w = winner_svu[t];
    if (w >= U)
        return;
    const long long src = (r % V) * U + w;     // (v, w)
    // s == s_hat: w == u, both assignments are self-assignments (core.hpp:1119-1121)
    depth_svu[t] = filtered_vu[src];
    Cd_svu[t] = Cd_hat_vu[src];
    mask_svu[t] = 0;
    winner_svu[t] = kNoWinner;
}

constexpr int kApplyRowsPerBlock = 4;     // rows a workgroup of the general part takes (their flags: one load)
constexpr int kApplyFlagSlots = 1024;     // >= kApplyRowsPerBlock * ceil(U / 256) for U <= 65536

__global__ __launch_bounds__(256) void k4_propagate_apply(int S, int V, int U, int s_hat, const float* __restrict__ filtered_vu,
                                                         const float* __restrict__ Cd_hat_vu, float* __restrict__ depth_svu,
                                                         float* __restrict__ Cd_svu, uint8_t* mask_svu, int* __restrict__ winner_svu,
                                                         uint8_t* __restrict__ dirty, int s_next, const uint8_t* __restrict__ edge_mask_next_vu,
                                                         int* __restrict__ list, int* __restrict__ count,
                                                         unsigned long long* __restrict__ total, int* __restrict__ packed_n)
{
    __shared__ uint8_t s_flags[kApplyFlagSlots];
    const int nseg = (U + 255) >> 8;
    const int lead = s_next >= 0 ? V : 0;       // workgroups [0, lead): the rows of view s_next, one each
    if ((int)blockIdx.x < lead) {
        const int v = blockIdx.x;
        const long long r = (long long)s_next * V + v;
        uint8_t* flags = dirty + r * nseg;
        for (int seg = 0; seg < nseg; seg++)
            if (flags[seg])   // the same byte for the whole workgroup
                apply_segment(r, seg, V, U, filtered_vu, Cd_hat_vu, depth_svu, Cd_svu, mask_svu, winner_svu);
        __syncthreads();      // every thread has read the flags, and this row's mask writes are the workgroup's own
        if ((int)threadIdx.x < nseg)
            flags[threadIdx.x] = 0;
        compact_row_packed(v, edge_mask_next_vu, mask_svu + (long long)s_next * V * U, U, list, count, total, packed_n);
        return;
    }
    // every other row, kApplyRowsPerBlock at a time: one coalesced load of their flags, then only the segments that hold
    // a claim are touched
    const long long nrows = (long long)S * V;
    const long long r0 = (long long)(blockIdx.x - lead) * kApplyRowsPerBlock;
    const long long r1 = r0 + kApplyRowsPerBlock < nrows ? r0 + kApplyRowsPerBlock : nrows;
    const int nfl = (int)(r1 - r0) * nseg;
    for (int i = threadIdx.x; i < nfl; i += 256) {
        const long long r = r0 + i / nseg;
        uint8_t f = 0;
        if ((int)(r / V) != s_next) {           // the lead workgroups own those rows and their flags
            f = dirty[r0 * nseg + i];
            if (f)
                dirty[r0 * nseg + i] = 0;
        }
        s_flags[i] = f;
    }
    __syncthreads();
    for (int i = 0; i < nfl; i++)
        if (s_flags[i])
            apply_segment(r0 + i / nseg, i % nseg, V, U, filtered_vu, Cd_hat_vu, depth_svu, Cd_svu, mask_svu, winner_svu);
}

}  // namespace rslf
