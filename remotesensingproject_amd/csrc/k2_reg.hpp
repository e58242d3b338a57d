// K2, register variant (k2_scan_reg / k2_scan_reg_packed): every sample of a unit held in registers.
// Common definitions, work mapping and the epilogue: k2_scan.hpp.
#pragma once

#include "k2_scan.hpp"

namespace rslf {

// ---------------------------------------------------------------------------
// Register variant: C*SPAD sample registers, every radiance in [0, 1e6].
//
// The S samples (x C channels) of one (pixel, hypothesis) are gathered once into
// VGPRs and the mean-shift passes run out of registers, no memory instruction:
//   C = 1:  delta = R - rbar ; t = k1*delta ; q = t*delta ; K = clamp(1 - q)
//           P = R*K ; A += P ; B += K                                  ( 7 VALU / sample / pass)
//   C = 3:  per channel delta, t = inv_h2*delta, q = t*delta ; qs = (q0+q2)+q1 ; K = clamp(1 - qs)
//           per channel P = R*K, A += P ; B += K                       (19 VALU / sample / pass)
// Out-of-range samples (the reference's NaN, interp.hpp:189) and the padding
// slots s >= S hold kSentinel = 1e30 in every channel: then q = +inf, K = max(-inf, 0) = 0
// and P = 1e30 * 0 = 0 exactly, so they add +0 to every sum -- bit-identical to the
// reference's "NaN -> K = 0, R0 = 0" without a second register per sample.
// Needs R == max(R, 0), hence the non-negative-volume precondition checked by
// the host (rslf_pile.hip: choose_scan).
// ---------------------------------------------------------------------------
// samples whose loads are in flight together (2 registers per sample and channel while they are)
#ifndef RSLF_REG_GB104
#define RSLF_REG_GB104 8    // loads in flight per gather round of the 104-slot (c3) kernel.  8: no scratch, HBM traffic 1.03x the
                            // algorithmic bytes.  13 (eight rounds per hypothesis instead of thirteen; round 3's default) is 0.86 % faster
                            // -- four same-box alternations, 65.67-65.84 vs 66.24-66.41 ms -- at the price of 28 B/lane of per-tile
                            // scratch that reaches HBM: 1.41x the algorithmic bytes.  Below the 1 % bar: the scratch-free build ships
                            // (profiles/r04_k2_variants.md).
#endif
constexpr int gather_batch(int c) { return c == 1 ? 8 : 4; }
constexpr int kPadSlack = 16;     // compiled slot counts step by at most this: only the last kPadSlack slots can be padding

// BORDER:    some sample line of this wave may leave [0, U-1]: test validity per sample.
// UNIFORM_D: every pixel shares the hypothesis grid (no per-pixel dmin/dmax planes), so the
//            view offset fl(fl(float(s_hat - s) * D[d]) * slope) is the same for all 64 lanes:
//            the wave computes the S offsets of a hypothesis once (2-4 lane-parallel rounds),
//            parks them in LDS and every sample starts from one broadcast ds_read -- 3 VALU
//            instructions fewer per sample than recomputing them per lane.
// PK:        the samples live in register PAIRS (s, s+1) and the mean-shift pass uses packed fp32
//            instructions on them; the sums still take one sample at a time, in ascending s.  For the
//            variants that run at one wave per SIMD (rslf_device.hpp, f2).
// GB:        samples whose loads are in flight together, 0 = the default (gather_batch).  The packed kernel of long
//            one-channel units asks for a quarter of the unit at once: on a sparse launch a wave has its SIMD nearly to
//            itself, and a hypothesis costs it one memory round trip per batch.
// LANE_D:    the lanes of the wave own HYPOTHESES of one pixel instead of pixels (k2_scan_reg_px): lane `dlane` scores
//            d0 + dlane, d0 + dlane + dstep, ... below d1 (a lane past the end repeats d1 - 1 and offers nothing).
template <int SPAD, int C, bool BORDER, bool UNIFORM_D, bool PK, int GB = 0, class BestT = Best<C>, bool LANE_D = false>
__device__ __forceinline__ void scan_reg_body(const ScanArgs& a, int v, int u, int d0, int d1, BestT& best,
                                              float* __restrict__ otab, int dlane = 0, int dstep = 1)
{
    static_assert(!LANE_D || !UNIFORM_D, "hypotheses in the lanes: every lane has its own view offsets");
    // 104 slots (the c3 shape) have 15 registers to spare at three waves per SIMD: 13 loads in flight instead of 8
    // (8 batches instead of 13 per hypothesis) measured 0.5 % faster
    constexpr int kGatherBatch = GB > 0 ? GB : (C == 1 && !PK && SPAD == 104 && RSLF_REG_GB104 == 13) ? 13 : gather_batch(C);
    static_assert(SPAD % kGatherBatch == 0 && (!PK || kGatherBatch % 2 == 0), "whole batches, whole pairs");
    static_assert(SPAD % 8 == 0, "slot counts are multiples of 8");
    const VolView& vol = a.vol;
    const float* epi = vol.row(v, 0);
    const float uf = (float)u;
    const int Um1 = vol.U - 1;
    const unsigned Um1_bits = __float_as_uint((float)Um1);
    const int S = vol.S;
    const int lane = threadIdx.x & 63;
    const long long o = (long long)v * vol.U + u;
    const float dmin = a.dmin_vu ? a.dmin_vu[o] : a.dmin;
    const float dmax = a.dmax_vu ? a.dmax_vu[o] : a.dmax;
    const float range = dmax - dmin;
    const float denom = (float)(a.dim_d - 1);
    const float kq = (C == 1) ? a.k.k1 : a.k.inv_h2;   // kernels.cpp:21 / :43
    const float slope = a.k.slope;
    const int stride_s = (int)vol.stride_s;
    // core.hpp:577: rbar starts from R[s_hat] = E[s_hat][u] for every hypothesis
    float centre[C];
#pragma unroll
    for (int c = 0; c < C; c++)
        centre[c] = epi[(long long)a.s_hat * vol.stride_s + u * C + c];

#pragma unroll 1
    for (int dk = d0; dk < d1; dk += (LANE_D ? dstep : 1)) {
        const int d = LANE_D ? min(dk + dlane, d1 - 1) : dk;
        const float Dd = hypothesis(dmin, range, denom, d);
        float R[PK ? 1 : C][PK ? 1 : SPAD];
        f2 R2[PK ? C : 1][PK ? SPAD / 2 : 1];
        int card = BORDER ? 0 : S;
        // The gather is fully unrolled (R[] must be register-indexed).  Everything in it that
        // does not depend on d would otherwise be hoisted out of the d loop -- SPAD row
        // pointers and SPAD float(s_hat - s) values pinned in registers for the whole kernel.
        // Two values re-made opaque per hypothesis keep that state to two registers: the
        // view offset of s = 0 as a float, and the running row offset.
        float Ss0 = (float)a.s_hat;
        asm volatile("" : "+v"(Ss0));
        int rowoff = 0;
        asm volatile("" : "+s"(rowoff));
        if (UNIFORM_D) {
#pragma unroll
            for (int s0 = 0; s0 < SPAD; s0 += 64) {
                const int s = s0 + lane;
                float off = (float)(a.s_hat - s) * Dd;   // float(s_hat - s) * D[d]   core.hpp:542,550
                off = off * slope;                       // core.hpp:551
                if (SPAD % 64 == 0 || s < SPAD)
                    otab[s] = off;
            }
            // same wave, LDS is in order: the broadcast reads below see these writes
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int g = 0; g < SPAD / kGatherBatch; g++) {
            float tt[kGatherBatch], e0[C][kGatherBatch], e1[C][kGatherBatch];
            bool ok[kGatherBatch];
            // issue the batch's loads back to back, then blend
#pragma unroll
            for (int j = 0; j < kGatherBatch; j++) {
                const int s = g * kGatherBatch + j;
                tt[j] = 0.0f;
                ok[j] = false;
#pragma unroll
                for (int c = 0; c < C; c++) {
                    e0[c][j] = kSentinel;
                    e1[c][j] = 0.0f;
                }
                if (s < SPAD - kPadSlack || s < S) {
                    float x;
                    if (UNIFORM_D) {
                        x = otab[s];                   // one broadcast read for the wave
                    } else {
                        x = (Ss0 - (float)s) * Dd;     // float(s_hat - s) * D[d]   core.hpp:542,550
                        x = x * slope;                 // core.hpp:551
                    }
                    x = x + uf;                        // core.hpp:552
                    tt[j] = lerp_weight(x);            // interp.hpp:181
                    int i0 = floor_to_int(x);          // interp.hpp:179
                    ok[j] = true;
                    if (BORDER) {
                        // interp.hpp:182: floor(x) >= 0 <=> x >= 0 and ceil(x) <= U-1 <=> x <= U-1.
                        // x is never -0 (u >= +0 is added last), so both tests are ONE unsigned compare
                        // of the bit patterns: negative floats have the sign bit set and compare high.
                        ok[j] = __float_as_uint(x) <= Um1_bits;
                        i0 = ok[j] ? i0 : 0;           // keep the address inside the row
                    }
                    // 32-bit byte offset off the EPI's scalar base
                    const unsigned byteoff = (unsigned)(i0 * C + rowoff) << 2;
                    // both taps of every channel are 2*C consecutive floats of the row (interleaved slab).  For
                    // integral x the reference reads the first tap twice with weights 1 and 0; 0 * (second tap)
                    // is the same +0 (rows are zero padded, so the second tap is finite)
                    const float* p = (const float*)((const char*)epi + byteoff);
#pragma unroll
                    for (int c = 0; c < C; c++) {
                        e0[c][j] = p[c];
                        e1[c][j] = p[C + c];
                    }
                }
                rowoff += stride_s;
            }
#pragma unroll
            for (int j = 0; j < kGatherBatch; j++) {
                const int s = g * kGatherBatch + j;
                const float omt = 1.0f - tt[j];
#pragma unroll
                for (int c = 0; c < C; c++) {
                    const float m0 = omt * e0[c][j];   // interp.hpp:184
                    const float m1 = tt[j] * e1[c][j];
                    const float r = m0 + m1;
                    float val;
                    if (BORDER)
                        val = ok[j] ? r : kSentinel;
                    else
                        val = (s < SPAD - kPadSlack || s < S) ? r : kSentinel;
                    if constexpr (PK) {
                        if (s & 1)
                            R2[c][s >> 1].y = val;
                        else
                            R2[c][s >> 1].x = val;
                    } else {
                        R[c][s] = val;
                    }
                }
                if (BORDER)
                    card += ok[j] ? 1 : 0;
            }
            // Pin this batch: its results must exist here, and the next batch's address state is
            // re-made opaque here, so the compiler cannot turn the unrolled gather into "all
            // loads first, all blends last" (which parks 2*C*SPAD loaded values in scratch).
            {
                const int b = g * kGatherBatch;
#pragma unroll
                for (int c = 0; c < C; c++) {
                    if constexpr (PK) {
                        asm volatile("" : "+v"(R2[c][b / 2]), "+v"(R2[c][b / 2 + 1]));
                        if (kGatherBatch == 8)
                            asm volatile("" : "+v"(R2[c][b / 2 + 2]), "+v"(R2[c][b / 2 + 3]));
                        static_assert(!PK || kGatherBatch == 4 || kGatherBatch == 8, "pair pinning is written for 4 and 8");
                    } else if (kGatherBatch != 4 && kGatherBatch != 8) {
#pragma unroll
                        for (int j = 0; j < kGatherBatch; j++)
                            asm volatile("" : "+v"(R[c][b + j]));
                    } else {
                        asm volatile("" : "+v"(R[c][b + 0]), "+v"(R[c][b + 1]), "+v"(R[c][b + 2]), "+v"(R[c][b + 3]));
                        if (kGatherBatch == 8)
                            asm volatile("" : "+v"(R[c][b + kGatherBatch - 4]), "+v"(R[c][b + kGatherBatch - 3]),
                                              "+v"(R[c][b + kGatherBatch - 2]), "+v"(R[c][b + kGatherBatch - 1]));
                    }
                }
                if (BORDER)
                    asm volatile("" : "+s"(rowoff), "+v"(Ss0), "+v"(card));
                else
                    asm volatile("" : "+s"(rowoff), "+v"(Ss0));
            }
        }

        float rbar[C];
#pragma unroll
        for (int c = 0; c < C; c++)
            rbar[c] = centre[c];
        float B = 0.0f;
#pragma unroll 1
        for (int it = 0; it < a.k.n_iter; it++) {   // core.hpp:584-610
            float A[C];
#pragma unroll
            for (int c = 0; c < C; c++)
                A[c] = 0.0f;
            B = 0.0f;
            if constexpr (PK) {
                // a pair of samples per step: delta, kq*delta, q, K and R*K as packed instructions on both
                // samples, then the sums take sample s and sample s+1 in turn (core.hpp:602-603 order)
                f2 rb2[C];
                const f2 kq2 = {kq, kq};
#pragma unroll
                for (int c = 0; c < C; c++)
                    rb2[c] = f2{rbar[c], rbar[c]};
                // hand-scheduled blocks of 8 (C = 1) / 4 (C = 3) samples; the slots of a block that lie beyond S are
                // padding (K = P = +0 exactly), whole blocks beyond S are skipped (wave-uniform)
                constexpr int kBlk = (C == 1) ? 8 : 4;
#pragma unroll
                for (int s0 = 0; s0 < SPAD; s0 += kBlk) {
                    if (!(s0 < SPAD - kPadSlack || s0 < S))
                        continue;
                    if constexpr (C == 1) {
                        f2 P[4], K[4];
                        const f2 r4[4] = {R2[0][s0 / 2], R2[0][s0 / 2 + 1], R2[0][s0 / 2 + 2], R2[0][s0 / 2 + 3]};
                        mean_shift_pk_octet(r4, rb2[0], kq2, P, K);
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            A[0] = A[0] + P[j].x;                // core.hpp:602, ascending s
                            B = B + K[j].x;                      // core.hpp:603
                            A[0] = A[0] + P[j].y;
                            B = B + K[j].y;
                        }
                    } else {
                        f2 Pa[3], Pb[3], Ka, Kb;
                        const f2 ra[3] = {R2[0][s0 / 2], R2[1][s0 / 2], R2[C - 1][s0 / 2]};
                        const f2 rb[3] = {R2[0][s0 / 2 + 1], R2[1][s0 / 2 + 1], R2[C - 1][s0 / 2 + 1]};
                        const f2 m[3] = {rb2[0], rb2[C > 1 ? 1 : 0], rb2[C - 1]};
                        mean_shift_pk_rgb_quad(ra, rb, m, kq2, Pa, Pb, Ka, Kb);
#pragma unroll
                        for (int c = 0; c < C; c++)
                            A[c] = A[c] + Pa[c].x;
                        B = B + Ka.x;
#pragma unroll
                        for (int c = 0; c < C; c++)
                            A[c] = A[c] + Pa[c].y;
                        B = B + Ka.y;
#pragma unroll
                        for (int c = 0; c < C; c++)
                            A[c] = A[c] + Pb[c].x;
                        B = B + Kb.x;
#pragma unroll
                        for (int c = 0; c < C; c++)
                            A[c] = A[c] + Pb[c].y;
                        B = B + Kb.y;
                    }
                }
            } else if (C == 1) {
                // hand-scheduled, four samples per block (rslf_device.hpp).  Only the last kPadSlack slots can
                // be padding: there a wave-uniform test skips what lies beyond S (a padded slot would add +0
                // to both sums, so skipping it changes nothing but the instruction count).
#pragma unroll
                for (int s0 = 0; s0 < SPAD; s0 += 4) {
                    if (s0 + 4 <= SPAD - kPadSlack || s0 + 4 <= S) {
                        mean_shift_group4(R[0][s0], R[0][s0 + 1], R[0][s0 + 2], R[0][s0 + 3], rbar[0], kq, A[0], B);
                    } else {
#pragma unroll
                        for (int j = 0; j < 3; j++)
                            if (s0 + j < S)
                                mean_shift_group1(R[0][s0 + j], rbar[0], kq, A[0], B);
                    }
                }
            } else {
#pragma unroll
                for (int s = 0; s < SPAD; s++) {
                    if (!(s < SPAD - kPadSlack || s < S))   // wave-uniform: padding slot
                        continue;
                    float q[C];
#pragma unroll
                    for (int c = 0; c < C; c++) {
                        const float delta = R[c][s] - rbar[c];   // core.hpp:591
                        const float tq = kq * delta;             // kernels.cpp:21 / :43
                        q[c] = tq * delta;
                    }
                    float qs = q[0];
                    if (C == 3) {
                        qs = q[0] + q[C - 1];                    // OpenCV 3.x reduceC_: (q0 + q2) + q1
                        qs = qs + q[C > 1 ? 1 : 0];
                    }
                    const float K = kernel_weight(qs);           // kernels.cpp:23-25 / :51-53
#pragma unroll
                    for (int c = 0; c < C; c++) {
                        const float pr = R[c][s] * K;            // core.cpp:28 / :36
                        A[c] = A[c] + pr;                        // core.hpp:602
                    }
                    B = B + K;                                   // core.hpp:603
                }
            }
#pragma unroll
            for (int c = 0; c < C; c++) {
                const float qd = (B != 0.0f) ? (A[c] / B) : 0.0f;   // core.cpp:42 / :50, OpenCV 3.x: /0 -> 0
                rbar[c] = (qd > 0.0f) ? qd : 0.0f;                  // core.hpp:609
            }
        }
        const float cardf = (float)card;
        float sc = (card != 0) ? (B / cardf) : 0.0f;   // core.hpp:616-620: the last pass's sum of K
        sc = (sc > 0.0f) ? sc : 0.0f;                  // core.hpp:622
        if (!LANE_D || dk + dlane < d1)
            best.offer(sc, d, Dd, rbar);
    }
}

// Waves per SIMD the register budget allows: C*SPAD sample registers + ~64 working registers
// (a batch of in-flight samples, the per-pixel state, SGPR overflow lanes), in the hardware's
// 8-register granules, 512 registers per SIMD lane.
constexpr int scan_reg_waves(int spad, int c)
{
    const int regs = ((c * spad + (c == 1 ? 64 : 96) + 7) / 8) * 8;
    const int w = 512 / regs;
    // Measured exceptions (profiles/r01_k2_variants.md): one more wave per SIMD than the budget above allows, the
    // compiler keeping a few dozen sample registers in scratch (coalesced per lane, cheap), wins 4-13 % here ...
    if (c == 1 && spad > 104 && spad <= 144) return 3;
    if (c == 1 && spad >= 80 && spad <= 88) return 4;
    if (c == 3 && spad >= 32 && spad <= 40) return 3;
    // ... and where the working set is smaller than the 64 / 96 assumed, the extra wave costs no scratch at all
    if (c == 1 && (spad == 48 || spad == 40)) return 5;   // 40: c2 (33 views) +2 %
    if (c == 3 && spad == 24) return 4;
    if (c == 1 && spad == 16) return 7;    // c1 (9 views) +2.5 %
    return w > 8 ? 8 : (w < 1 ? 1 : w);
}

// The variants above that run a wave more than their registers allow keep the per-pixel running result in LDS, not in
// registers (BestLds, k2_scan.hpp): nothing is left for hipcc to spill.
#ifndef RSLF_REG_BEST_LDS
#define RSLF_REG_BEST_LDS 1
#endif
constexpr bool scan_reg_best_in_lds(int spad, int c)
{
    if (!RSLF_REG_BEST_LDS)
        return false;
    if (c == 1)
        return spad == 16 || spad == 40 || spad == 48 || (spad >= 80 && spad <= 88) || (spad >= 104 && spad <= 144);
    return spad == 24 || spad == 40;
}

// One wave per SIMD: a wave issues a VALU instruction every ~5 clocks whatever it is (tools/ubench_valu.hip),
// so packed fp32 halves the issue slots of the mean-shift pass.  With two or more waves the SIMD is already
// saturated by scalar instructions and packed ones run at half rate.
constexpr bool scan_reg_packed_math(int spad, int c) { return scan_reg_waves(spad, c) == 1; }

template <int SPAD, int C, class BestT>
__device__ __forceinline__ void scan_reg_rows(const ScanArgs& a, int v, int u, int d0, int d1, BestT& best, float* otab)
{
    constexpr bool PK = scan_reg_packed_math(SPAD, C);
    if (a.dmin_vu) {
        scan_reg_body<SPAD, C, true, false, PK, 0, BestT>(a, v, u, d0, d1, best, otab);
        return;
    }
    // the validity test is decided per hypothesis, as in the streaming kernel (scan_stream_rows): runs of hypotheses whose
    // sample lines stay inside the row for every lane take the form without it, in ascending order
    const float max_ds = (float)max(a.s_hat, a.vol.S - 1 - a.s_hat);
    const float range = a.dmax - a.dmin, denom = (float)(a.dim_d - 1);
    const float uf = (float)u, Um1 = (float)(a.vol.U - 1);
    auto interior = [&](int d) -> bool {
        const float reach = max_ds * fabsf(hypothesis(a.dmin, range, denom, d)) * fabsf(a.k.slope) + 2.0f;
        return __all((uf - reach >= 0.0f) && (uf + reach <= Um1));
    };
    int d = d0;
    while (d < d1) {
        const bool in = interior(d);
        int e = d + 1;
        while (e < d1 && interior(e) == in)
            e++;
        if (in)
            scan_reg_body<SPAD, C, false, true, PK, 0, BestT>(a, v, u, d, e, best, otab);
        else
            scan_reg_body<SPAD, C, true, true, PK, 0, BestT>(a, v, u, d, e, best, otab);
        d = e;
    }
}

template <int SPAD, int C>
__global__ __launch_bounds__(64 * kScanWaves) __attribute__((amdgpu_waves_per_eu(scan_reg_waves(SPAD, C), scan_reg_waves(SPAD, C))))
void k2_scan_reg(ScanArgs a)
{
    __shared__ float s_otab[kScanWaves][SPAD];
    float* otab = s_otab[__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)];
    constexpr bool kEpiDyn = false;
    float* const epi_lds = nullptr;
    const int epi_stride = 0;
    if constexpr (scan_reg_best_in_lds(SPAD, C)) {
        __shared__ float s_best[kScanWaves][BestLds<C>::kFloats];
        BestLds<C> running(s_best[__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)]);
        RSLF_SCAN_ROW_TILE((running.init(), scan_reg_rows<SPAD, C>(a, v, u, d0, d1, running, otab), running.load(best)))
    } else {
        RSLF_SCAN_ROW_TILE((scan_reg_rows<SPAD, C>(a, v, u, d0, d1, best, otab)))
    }
}

// Packed tiles: its own kernel, because per-lane EPI bases cost address registers the row kernel's
// budget does not have (and must not pay for).
#ifndef RSLF_PACKED_LONG_GB
#define RSLF_PACKED_LONG_GB 4   // divisor: a quarter of the unit per gather batch
#endif
constexpr bool packed_long_unit(int spad, int c) { return c == 1 && spad >= 80 && spad <= 128 && spad % RSLF_PACKED_LONG_GB == 0; }
constexpr int packed_waves(int spad, int c) { return packed_long_unit(spad, c) ? 2 : scan_reg_waves(spad + 24, c); }
constexpr int packed_gather_batch(int spad, int c) { return packed_long_unit(spad, c) ? spad / RSLF_PACKED_LONG_GB : 0; }

template <int SPAD, int C>
__global__ __launch_bounds__(64 * kScanWaves) __attribute__((amdgpu_waves_per_eu(packed_waves(SPAD, C), packed_waves(SPAD, C))))
void k2_scan_reg_packed(ScanArgs a)
{
    // (Tiles whose 64 entries sit on one scanline -- most of them on a visit that scans many pixels -- were also given
    // the row kernel's forms, scalar EPI base and shared offset table: no gain, not kept.)
    constexpr bool kEpiDyn = false;
    float* const epi_lds = nullptr;
    const int epi_stride = 0;
    RSLF_SCAN_PACKED_LOOP((scan_reg_body<SPAD, C, true, false, packed_waves(SPAD, C) == 1, packed_gather_batch(SPAD, C)>(a, v, u, d0, d1, best, nullptr)))
}

// Sparse launches, lanes own HYPOTHESES (k2_scan_reg_px).  A sparse visit's pixels sit two or three to a scanline: with a
// pixel per lane (k2_scan_reg_packed) every load of the gather touches up to 64 scanlines -- 64 cache lines for 512 useful
// bytes -- and the kernel runs at 1.0 G units/s on a list of scattered pixels where the dense kernel makes 8 (c3 shape,
// tools/probe_sparse.py).  Here a wave owns ONE pixel and its lanes score 64 hypotheses of it: a load's 64 taps lie within
// (d_63 - d_0) * |s_hat - s| pixels of one EPI row, a handful of lines, the EPI base is a scalar, and nothing is shared
// between workgroups -- no hypothesis groups, no records, no tickets.  px_waves = 1 / 2 / 4 waves share a pixel's
// hypotheses (lane slot + k * 64 * px_waves; plan::px_waves picks by lane use), so a workgroup holds 4 / 2 / 1 pixels.
// The arithmetic of a (pixel, hypothesis) unit is scan_reg_body's per-lane form, the one per-pixel [dmin, dmax] planes
// take: same operations, same bits.  What changes is the reduction over hypotheses, now across lanes (scan_px_finish,
// k2_scan.hpp).
// Occupancy and gather batch: the ROW kernel's (scalar EPI base, near-coalesced loads: three waves per SIMD and 13 loads in
// flight at 104 slots), not the packed kernel's (two waves, a quarter of the unit in flight): same-box A/B, c3 sweep 118.0
// vs 120.0 ms, SkysatLR-like fine-to-coarse 257 vs 278 ms (profiles/r03_k2_variants.md section 5).  0 builds the other.
#ifndef RSLF_PX_ROWLIKE
#define RSLF_PX_ROWLIKE 1
#endif
constexpr int px_kernel_waves(int spad, int c) { return RSLF_PX_ROWLIKE ? scan_reg_waves(spad, c) : packed_waves(spad, c); }
template <int SPAD, int C>
__global__ __launch_bounds__(64 * kScanWaves) __attribute__((amdgpu_waves_per_eu(px_kernel_waves(SPAD, C), px_kernel_waves(SPAD, C))))
void k2_scan_reg_px(ScanArgs a)
{
    __shared__ double s_sum[kScanWaves];
    __shared__ float s_rec[kScanWaves][kPxRecFloats];
    const int n = *a.packed_n;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wpp = a.px_waves, ppw = kScanWaves / wpp;   // waves per pixel, pixels per workgroup
    const int items = (n + ppw - 1) / ppw;
    const int sub = wave % wpp;
    const int dlane = sub * 64 + lane, dstep = 64 * wpp;
    for (int item = blockIdx.x; item < items; item += gridDim.x) {
        const int e = item * ppw + wave / wpp;
        const bool listed = e < n;   // (a workgroup's last pixels may be missing: those waves shadow the list's last entry and write nothing)
        const unsigned o = (unsigned)__builtin_amdgcn_readfirstlane(a.list[listed ? e : n - 1]);
        const int v = (int)(o / (unsigned)a.vol.U);
        const int u = (int)(o - (unsigned)v * (unsigned)a.vol.U);
        const bool have = listed && scan_px_owns(a, v);   // (wave-uniform; a row with many pixels is the row-tile launch's)
        Best<C> best;
        best.init();
        // the validity test (interp.hpp:182) can go where every sample line of every hypothesis stays inside the row
        const float dlo = a.dmin_vu ? a.dmin_vu[o] : a.dmin, dhi = a.dmax_vu ? a.dmax_vu[o] : a.dmax;
        const float reach = (float)max(a.s_hat, a.vol.S - 1 - a.s_hat) * fmaxf(fabsf(dlo), fabsf(dhi)) * fabsf(a.k.slope) + 2.0f;
        const bool interior = (float)u - reach >= 0.0f && (float)u + reach <= (float)(a.vol.U - 1);   // wave-uniform
        constexpr int kGB = RSLF_PX_ROWLIKE ? 0 : packed_gather_batch(SPAD, C);
        if (!have) {
            // nothing to scan: the finish below still runs (its barriers are the workgroup's)
        } else if (interior)
            scan_reg_body<SPAD, C, false, false, false, kGB, Best<C>, true>(a, v, u, 0, a.dim_d, best, nullptr, dlane, dstep);
        else
            scan_reg_body<SPAD, C, true, false, false, kGB, Best<C>, true>(a, v, u, 0, a.dim_d, best, nullptr, dlane, dstep);
        scan_px_finish<C>(a, o, have, best, wave, lane, wpp, s_rec, s_sum);   // across the lanes, then the waves that share the pixel
    }
}

}  // namespace rslf
