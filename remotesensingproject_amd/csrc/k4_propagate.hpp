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

constexpr int kClaimMaxViews = 1024;    // views whose "anything left to paint?" bits a claim workgroup keeps in LDS
constexpr int kNoWinner = 0x7F7F7F7F;   // what hipMemset(0x7F) leaves; >= any column index

// One source pixel's claims (core.hpp:1105-1125): `cur` is its filtered disparity.
// `live_views` (nullable; LDS, one bit per view, claim_live_views): a view none of whose segments this workgroup's sources
// can land in has a pixel left in its running mask is skipped for the whole workgroup -- after the first visits that is
// most views for most workgroups, where every source used to read one running-mask byte per view (the whole [S][V][U]
// volume, 209 MB at c3, on each of the 101 visits).
template <int C>
__device__ __forceinline__ void propagate_claim_pixel(const VolView& vol, int s_hat, int v, int u, float cur, bool source,
                                                      const float* __restrict__ rbar_vu, const uint8_t* __restrict__ mask_svu,
                                                      int* __restrict__ winner_svu, uint8_t* __restrict__ dirty, float slope,
                                                      const plan::NormThreshold& prop_thr, const unsigned long long* live_views)
{
    const int nseg = (vol.U + 255) >> 8;
    const long long o = (long long)v * vol.U + u;
    float rb[C];
#pragma unroll
    for (int c = 0; c < C; c++)
        rb[c] = source ? rbar_vu[o * C + c] : 0.0f;
    const long long plane = (long long)vol.V * vol.U;
    const long long row = (long long)v * vol.U;
    // eight views at a time: their running-mask bytes, then the radiances of those still unpainted, are loaded together
    // before the first test -- the claims do not depend on one another (atomicMin), only the loads' latency did add up
    constexpr int B = RSLF_CLAIM_BATCH;
    for (int s0 = 0; s0 < vol.S; s0 += B) {
        int ri[B];
        bool live[B];
        static_assert(64 % B == 0, "a batch of views sits in one word of the live-view bits");
        if (live_views && ((live_views[s0 >> 6] >> (s0 & 63)) & ((1ull << B) - 1ull)) == 0)
            continue;   // workgroup-uniform: nothing left to paint in these views within the workgroup's reach
        // (every load below is UNCONDITIONAL, its address clamped into the row: a load under `if (live)` sits in a basic
        // block of its own and hipcc waits for it there -- eight round trips one after the other per batch where this
        // was written to have eight in flight; k34_median_claim 217 -> 160 us per visit of a c3 sweep)
        uint8_t mb[B];
#pragma unroll
        for (int j = 0; j < B; j++) {
            const int s = s0 + j;
            // core.hpp:1109: u + (int)std::round(depth * (s_hat - s) * slope_factor)
            float off = cur * (float)(s_hat - s);
            off = off * slope;
            ri[j] = u + (int)roundf(off);
            live[j] = source && s < vol.S && ri[j] >= 0 && ri[j] < vol.U;
            ri[j] = min(max(ri[j], 0), vol.U - 1);
            mb[j] = mask_svu[(long long)min(s, vol.S - 1) * plane + row + ri[j]];
        }
        bool any = false;
#pragma unroll
        for (int j = 0; j < B; j++) {
            live[j] = live[j] && mb[j] != 0;
            any |= live[j];
        }
        if (!__any(any))
            continue;   // wave-uniform: nobody in this wave has an unpainted target in these views
        float e[B][C];
#pragma unroll
        for (int j = 0; j < B; j++) {
            // (a lane without a live target in view s0 + j reads its own column: the line its neighbours read)
            const float* er = vol.row(v, min(s0 + j, vol.S - 1));
            const int col = live[j] ? ri[j] : min(u, vol.U - 1);   // (a lane past the row's end: the last column, not the pitch's padding)
#pragma unroll
            for (int c = 0; c < C; c++)
                e[j][c] = er[col * C + c];
        }
#pragma unroll
        for (int j = 0; j < B; j++) {
            if (!live[j])
                continue;
            float df[C];
#pragma unroll
            for (int c = 0; c < C; c++)
                df[c] = e[j][c] - rb[c];
            if (norm_below<C>(df, prop_thr)) {   // core.hpp:1116: norm<T>(..) < par_propagation_epsilon
                atomicMin(&winner_svu[(long long)(s0 + j) * plane + row + ri[j]], u);
                dirty[((long long)(s0 + j) * vol.V + v) * nseg + (ri[j] >> 8)] = 1;   // this 256-column segment holds a claim
            }
        }
    }
}

// K3 + claim in one launch (a sweep visit): a pixel's median needs its neighbours' RAW depths only (what the scan
// left), and its claims need its own median only -- so the thread that filters a pixel also makes its claims.  The
// filtered plane is still written: the apply pass reads the winners' values from it.
template <int C, int MODE>
__global__ __launch_bounds__(256) void k34_median_claim(VolView vol, int s_hat, const float* __restrict__ raw_vu,
                                                       float* __restrict__ filtered_vu, const uint8_t* __restrict__ edge_mask_vu,
                                                       int w, plan::NormThreshold median_thr, const float* __restrict__ rbar_vu,
                                                       const uint8_t* __restrict__ mask_svu, int* __restrict__ winner_svu,
                                                       uint8_t* __restrict__ dirty, float slope, plan::NormThreshold prop_thr,
                                                       const float* __restrict__ gate_Cd_vu, float disp_thr, int* __restrict__ reset,
                                                       const int* __restrict__ remain)
{
    // `reset`: the packed list's length, which the scan before this launch was the last to read and the apply pass after
    // it counts up again from 0
    if (reset && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)
        *reset = 0;
    static_assert(kMedianBlock == 256, "one median tile per claim workgroup");
    extern __shared__ __attribute__((aligned(16))) float s_median_tile[];   // the window tile (k3_median.hpp)
    const int v = blockIdx.y;
    const int u_first = blockIdx.x * blockDim.x;
    const int u = u_first + threadIdx.x;
    const bool inside = u < vol.U;
    const long long o = (long long)v * vol.U + (inside ? u : vol.U - 1);
    // core.hpp:678-679, :881-892: the median over the edge mask, 0 elsewhere
    const float cur = selective_median_block<C, MODE>(vol, raw_vu, edge_mask_vu, s_hat, w, median_thr, v, u_first, s_median_tile);
    if (inside)
        filtered_vu[o] = cur;
    const bool source = inside && (gate_Cd_vu ? gate_Cd_vu[o] > disp_thr : edge_mask_vu[o] != 0);   // core.hpp:1097-1103
    // the range of the disparities the workgroup's sources hold (a non-finite one: everywhere)
    __shared__ float s_lo[4], s_hi[4];
    const bool finite = fabsf(cur) < 1.0e9f;   // false for NaN too
    float lo = source ? (finite ? cur : -1.0e9f) : INFINITY, hi = source ? (finite ? cur : 1.0e9f) : -INFINITY;
    for (int off = 32; off > 0; off >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, off));
        hi = fmaxf(hi, __shfl_xor(hi, off));
    }
    if ((threadIdx.x & 63) == 0) {
        s_lo[threadIdx.x >> 6] = lo;
        s_hi[threadIdx.x >> 6] = hi;
    }
    const bool any_source = __syncthreads_or(source);
    if (!any_source)
        return;
    const float cur_lo = fminf(fminf(s_lo[0], s_lo[1]), fminf(s_lo[2], s_lo[3]));
    const float cur_hi = fmaxf(fmaxf(s_hi[0], s_hi[1]), fmaxf(s_hi[2], s_hi[3]));
    // one thread per view decides whether the view is worth visiting at all (kClaimMaxViews views: more, and every view is visited)
    __shared__ unsigned long long s_live[kClaimMaxViews / 64];
    const bool skipping = remain != nullptr && vol.S <= kClaimMaxViews;
    if (skipping) {
        const int nseg = (vol.U + 255) >> 8;
        for (int s0 = 0; s0 < vol.S; s0 += 256) {
            const int s = s0 + threadIdx.x;
            bool any = false;
            if (s < vol.S) {
                // targets: u + round(fl(fl(cur * float(s_hat - s)) * slope)), monotone in cur; 1.5 covers the roundings
                const float k = (float)(s_hat - s) * slope;
                const float a0 = cur_lo * k, a1 = cur_hi * k;
                const int lo = max(0, (int)floorf((float)u_first + fminf(a0, a1) - 1.5f)) >> 8;
                const int hi = min(vol.U - 1, max(0, (int)ceilf((float)(u_first + 255) + fmaxf(a0, a1) + 1.5f))) >> 8;
                // eight segments' counts at a time, every load issued before the first is looked at
                const int* rr = remain + ((long long)s * vol.V + v) * nseg;
                for (int g0 = lo; g0 <= hi; g0 += 8) {
                    int r8[8];
#pragma unroll
                    for (int k = 0; k < 8; k++)
                        r8[k] = rr[min(g0 + k, hi)];
#pragma unroll
                    for (int k = 0; k < 8; k++)
                        any |= r8[k] != 0;
                }
            }
            const unsigned long long b = __ballot(any);
            if ((threadIdx.x & 63) == 0)
                s_live[(s0 + threadIdx.x) >> 6] = b;
        }
        __syncthreads();
    }
    propagate_claim_pixel<C>(vol, s_hat, v, u, cur, source, rbar_vu, mask_svu, winner_svu, dirty, slope, prop_thr, skipping ? s_live : nullptr);
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
                                              float* __restrict__ Cd_svu, uint8_t* mask_svu, int* __restrict__ winner_svu,
                                              int* __restrict__ remain)
{
    const int u = (seg << 8) + threadIdx.x;
    const long long t = r * U + (u < U ? u : U - 1);
    const int w = u < U ? winner_svu[t] : U;
    // every claimed target was in the running mask when it was claimed and leaves it now: the segment's count follows
    const unsigned long long leaving = __ballot(w < U);
    if (remain && leaving && (threadIdx.x & 63) == 0)
        atomicSub(&remain[r * ((U + 255) >> 8) + seg], __popcll(leaving));
    if (w >= U)
        return;
    const long long src = (r % V) * U + w;     // (v, w)
    // s == s_hat: w == u, both assignments are self-assignments (core.hpp:1119-1121)
    depth_svu[t] = filtered_vu[src];
    Cd_svu[t] = Cd_hat_vu[src];
    mask_svu[t] = 0;
    winner_svu[t] = kNoWinner;
}

// remain[row][segment] = pixels of the running mask in that 256-column segment, once per sweep (rslf_sweep_begin); the
// apply passes keep it in step.  (A compaction only ever clears pixels outside the edge mask, which the running masks --
// clones of the edge masks, core.hpp:958-965 -- do not hold: should a scan drop a pixel from its edge mask, core.hpp:653-657,
// the count stays one too high, which costs a skipped skip, never a missed claim.)
__global__ __launch_bounds__(256) void k4_count_segments(const uint8_t* __restrict__ mask_svu, long long rows, int U, int* __restrict__ remain)
{
    const int nseg = (U + 255) >> 8;
    const long long item = blockIdx.x;   // one workgroup per (row, segment)
    if (item >= rows * nseg)
        return;
    const long long r = item / nseg;
    const int seg = (int)(item - r * nseg);
    const int u = (seg << 8) + threadIdx.x;
    const unsigned long long b = __ballot(u < U && mask_svu[r * U + (u < U ? u : 0)] != 0);
    __shared__ int s_n[4];
    if ((threadIdx.x & 63) == 0)
        s_n[threadIdx.x >> 6] = __popcll(b);
    __syncthreads();
    if (threadIdx.x == 0)
        remain[item] = s_n[0] + s_n[1] + s_n[2] + s_n[3];
}

constexpr int kApplyRowsPerBlock = 4;     // rows a workgroup of the general part takes (their flags: one load)
constexpr int kApplyFlagSlots = 1024;     // >= kApplyRowsPerBlock * ceil(U / 256) for U <= 65536

__global__ __launch_bounds__(256) void k4_propagate_apply(int S, int V, int U, int s_hat, const float* __restrict__ filtered_vu,
                                                         const float* __restrict__ Cd_hat_vu, float* __restrict__ depth_svu,
                                                         float* __restrict__ Cd_svu, uint8_t* mask_svu, int* __restrict__ winner_svu,
                                                         uint8_t* __restrict__ dirty, int s_next, const uint8_t* __restrict__ edge_mask_next_vu,
                                                         int* __restrict__ list, int* __restrict__ count,
                                                         unsigned long long* __restrict__ total, int* __restrict__ packed_n,
                                                         int* __restrict__ remain, int* __restrict__ rowbase)
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
                apply_segment(r, seg, V, U, filtered_vu, Cd_hat_vu, depth_svu, Cd_svu, mask_svu, winner_svu, remain);
        __syncthreads();      // every thread has read the flags, and this row's mask writes are the workgroup's own
        if ((int)threadIdx.x < nseg)
            flags[threadIdx.x] = 0;
        compact_row_packed(v, edge_mask_next_vu, mask_svu + (long long)s_next * V * U, U, list, count, total, packed_n, rowbase);
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
            apply_segment(r0 + i / nseg, i % nseg, V, U, filtered_vu, Cd_hat_vu, depth_svu, Cd_svu, mask_svu, winner_svu, remain);
}

}  // namespace rslf
