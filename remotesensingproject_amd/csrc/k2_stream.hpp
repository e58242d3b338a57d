// K2, streaming variant (k2_scan_stream): more views than the register file of two waves per SIMD holds.
// Common definitions, work mapping and the epilogue: k2_scan.hpp.
#pragma once

#include "k2_scan.hpp"

namespace rslf {

// ---------------------------------------------------------------------------
// Streaming variant: any S, C in {1,3}, radiances in [0, 1e6].  For view counts beyond the register variants
// (C = 1 above 192 views, RGB above 48 -- 100-view RGB fields, BASELINE.json's 201-view RGB config).  A unit's
// samples are split three ways: a resident prefix held in registers (and compiler scratch) over the passes,
// samples parked in LDS behind it, and a tail that is re-gathered on every mean-shift pass with the register
// variant's economies: hypothesis-uniform view offsets from an LDS table (one broadcast read per sample),
// validity as one unsigned compare, invalid samples as the 1e30 sentinel (K = 0, P = 0 exactly), K as one
// clamp instruction, four / eight samples in flight.  The re-gathered tail is what costs.
// ---------------------------------------------------------------------------
// Resident prefix: the first NRES samples of a unit are gathered ONCE per hypothesis and stay in registers
// over the mean-shift passes (as in the register variant); only the samples behind them are re-gathered every
// pass.  The kernel is bound by its gathers (the texture-address unit is 85-93 % busy, PMC), so every resident
// sample is a gather saved in nine of ten passes: c5 slice 98 -> 67 ms with 48 of 201 RGB samples resident.
// One wave per SIMD with far more residents measured slower.  Volumes with fewer views than the shortest prefix
// take NRES = 0.
// Resident-prefix lengths compiled in: the largest one not above S is used.  They exceed what the registers of
// two waves per SIMD hold -- the compiler keeps the overflow in scratch, whose per-lane accesses are coalesced and
// far cheaper than a gather (measured: more residents won up to these counts, profiles/r01_k2_variants.md).
#ifndef RSLF_STREAM_WAVES
#define RSLF_STREAM_WAVES 2   // waves per SIMD the streaming kernel is compiled for
#endif
#ifndef RSLF_STREAM_GS
#define RSLF_STREAM_GS 8      // samples per batch of the shared-tap tail (a multiple of 4)
#endif
#ifndef RSLF_STREAM_NRES_RGB
#define RSLF_STREAM_NRES_RGB 68
#endif
#ifndef RSLF_STREAM_NRES_1CH
#define RSLF_STREAM_NRES_1CH 192
#endif
__host__ __device__ constexpr int stream_resident_hi(int C) { return C == 1 ? RSLF_STREAM_NRES_1CH : RSLF_STREAM_NRES_RGB; }
#ifndef RSLF_STREAM_NRES_RGB_LO
#define RSLF_STREAM_NRES_RGB_LO 48
#endif
__host__ __device__ constexpr int stream_resident_lo(int C) { return C == 1 ? 0 : RSLF_STREAM_NRES_RGB_LO; }
__host__ __device__ constexpr int stream_resident_for(int S, int C)
{
    return S >= stream_resident_hi(C) ? stream_resident_hi(C) : (stream_resident_lo(C) > 0 && S >= stream_resident_lo(C)) ? stream_resident_lo(C) : 0;
}
// ... of the pixel-per-wave form: its gathers have no neighbour wave to share L1 lines with, and the residents past what
// the registers hold -- compiler scratch -- cost it more than they save: RGB 48 instead of 68 is 13 % faster (67.4 vs 77.9 ms
// on a dense 100-view list, profiles/r04_k2_variants.md)
__host__ __device__ constexpr int stream_px_resident_for(int S, int C)
{
    return C == 3 ? (S >= stream_resident_lo(3) ? stream_resident_lo(3) : 0) : stream_resident_for(S, C);
}

// DENSE: the tile is 63 consecutive pixels of one scanline in lanes 0..62 and lane 63 stands on the pixel after them
// (scan_stream_rows): a lane's right tap is then its neighbour's left tap, so the re-gathered tail loads ONE texel
// per lane and sample and takes the other from lane + 1 (v_mov_b32 wave_shl:1) -- half the vector-memory
// instructions of the tail, which is what bounds it (the CU's texture data path takes ~17 clocks per multi-dword
// wave-instruction whatever its width, tools/ubench_ta.hip; PMC: TD_BUSY 90 %).
// LANE_D: the lanes of the wave own HYPOTHESES of one pixel instead of pixels (k2_scan_stream_px): lane `dlane` scores
// d0 + dlane, d0 + dlane + dstep, ... below d1 (a lane past the end repeats d1 - 1 and offers nothing); v and u are then
// wave-uniform and every lane has its own view offsets (UNIFORM_D is false).
template <int C, bool BORDER, bool UNIFORM_D, int NRES, bool DENSE = false, bool LANE_D = false>
__device__ __forceinline__ void scan_stream_body(const ScanArgs& a, int v, int u, int d0, int d1, Best<C>& best,
                                                 float* __restrict__ otab, int dlane = 0, int dstep = 1)
{
    static_assert(!DENSE || (UNIFORM_D && !BORDER), "shared taps need a common hypothesis grid and no border lane");
    static_assert(!LANE_D || (!UNIFORM_D && !DENSE), "hypotheses in the lanes: every lane has its own view offsets");
    const VolView& vol = a.vol;
    const float* epi = vol.row(v, 0);
    const float uf = (float)u;
    const unsigned Um1_bits = __float_as_uint((float)(vol.U - 1));
    const int S = vol.S;
    const int lane = threadIdx.x & 63;
    const long long o = (long long)v * vol.U + u;
    const float dmin = a.dmin_vu ? a.dmin_vu[o] : a.dmin;
    const float dmax = a.dmax_vu ? a.dmax_vu[o] : a.dmax;
    const float range = dmax - dmin;
    const float denom = (float)(a.dim_d - 1);
    const float kq = (C == 1) ? a.k.k1 : a.k.inv_h2;
    const float slope = a.k.slope;
    const unsigned stride_b = (unsigned)vol.stride_s << 2;
    float centre[C];
#pragma unroll
    for (int c = 0; c < C; c++)
        centre[c] = epi[(long long)a.s_hat * vol.stride_s + u * C + c];

    for (int dk = d0; dk < d1; dk += (LANE_D ? dstep : 1)) {
        const int d = LANE_D ? min(dk + dlane, d1 - 1) : dk;
        const float Dd = hypothesis(dmin, range, denom, d);
        bool shared_taps = false;
        if (UNIFORM_D) {
            bool odd = false;
            for (int s = lane; s < S; s += 64) {
                float off = (float)(a.s_hat - s) * Dd;   // core.hpp:542,550
                off = off * slope;                       // core.hpp:551
                otab[s] = off;
                // positions are off + (integer u): all lanes floor alike unless the sum rounds up to the next
                // integer in some of them, which takes a fraction within one ulp of 1
                if (DENSE)
                    odd |= __builtin_amdgcn_fractf(off) > a.stream_frac_max;
            }
            __builtin_amdgcn_wave_barrier();
            if (DENSE)
                shared_taps = !__any(odd);               // wave-uniform, per hypothesis
        }
        float rbar[C];
#pragma unroll
        for (int c = 0; c < C; c++)
            rbar[c] = centre[c];                         // core.hpp:577
        float B = 0.0f;
        int card = BORDER ? 0 : S;
        // resident prefix: samples [0, NRES) (the kernel picks NRES = stream_resident_for(S, C))
        float Rres[C][NRES > 0 ? NRES : 1];
        int card_res = 0;
        // `shared_tag` (DENSE tiles, regular hypothesis): one texel load per sample, the right tap from lane + 1, as in the tail
        auto gather_resident = [&](auto shared_tag) {
            constexpr bool SH = decltype(shared_tag)::value;
            constexpr int GR = (C == 1) ? 8 : 4;
            unsigned rowb = 0;
#pragma unroll
            for (int g = 0; g < NRES / GR; g++) {
                float tt[GR], e0[C][GR], e1[C][GR];
                bool ok[GR];
#pragma unroll
                for (int j = 0; j < GR; j++) {
                    const int s = g * GR + j;
                    float x;
                    if (UNIFORM_D) {
                        x = otab[s];
                    } else {
                        x = (float)(a.s_hat - s) * Dd;
                        x = x * slope;
                    }
                    x = x + uf;
                    tt[j] = lerp_weight(x);
                    int i0 = floor_to_int(x);
                    ok[j] = true;
                    if (BORDER) {
                        ok[j] = __float_as_uint(x) <= Um1_bits;
                        i0 = ok[j] ? i0 : 0;
                    }
                    const float* p = (const float*)((const char*)epi + (((unsigned)(i0 * C) << 2) + rowb));
                    rowb += stride_b;
                    if constexpr (SH && C == 3) {
                        typedef float f3u __attribute__((ext_vector_type(3), aligned(4)));
                        const f3u t3 = *(const f3u*)p;
                        e0[0][j] = t3.x, e0[1][j] = t3.y, e0[C - 1][j] = t3.z;
                    } else {
#pragma unroll
                        for (int c = 0; c < C; c++) {
                            e0[c][j] = p[c];
                            if (!SH)
                                e1[c][j] = p[C + c];
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < GR; j++) {
                    const float omt = 1.0f - tt[j];
#pragma unroll
                    for (int c = 0; c < C; c++) {
                        // 0x130 = wave_shl:1: lane i reads lane i + 1, the owner of this lane's right tap
                        const float right = SH ? __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(e0[c][j]), 0x130, 0xf, 0xf, false))
                                               : e1[c][j];
                        const float m0 = omt * e0[c][j];
                        const float m1 = tt[j] * right;
                        const float r = m0 + m1;
                        Rres[c][g * GR + j] = ok[j] ? r : kSentinel;
                    }
                    if (BORDER)
                        card_res += ok[j] ? 1 : 0;
                }
#pragma unroll
                for (int c = 0; c < C; c++)
#pragma unroll
                    for (int j = 0; j < GR; j++)
                        asm volatile("" : "+v"(Rres[c][g * GR + j]));
                asm volatile("" : "+s"(rowb));
            }
        };
        if constexpr (NRES > 0) {
            if (DENSE && shared_taps)
                gather_resident(std::true_type{});
            else
                gather_resident(std::false_type{});
        }
        // parked samples [NRES, NRES + npark): gathered once per hypothesis like the resident ones, kept in LDS
        // ([sample][channel][lane], conflict-free) -- one LDS read instead of one gather per pass
        const int npark = (NRES > 0) ? a.stream_park : 0;
        float* park = otab + ((S + 3) & ~3);
        auto gather_parked = [&](auto shared_tag) {
            constexpr bool SH = decltype(shared_tag)::value;
            constexpr int GP = (C == 1) ? 8 : 4;
            unsigned rowb = (unsigned)NRES * stride_b;
#pragma unroll 1
            for (int s0 = NRES; s0 < NRES + npark; s0 += GP) {
                float tt[GP], e0[C][GP], e1[C][GP];
                bool ok[GP];
#pragma unroll
                for (int j = 0; j < GP; j++) {
                    const int s = s0 + j;
                    float x;
                    if (UNIFORM_D) {
                        x = otab[s];
                    } else {
                        x = (float)(a.s_hat - s) * Dd;
                        x = x * slope;
                    }
                    x = x + uf;
                    tt[j] = lerp_weight(x);
                    int i0 = floor_to_int(x);
                    ok[j] = true;
                    if (BORDER) {
                        ok[j] = __float_as_uint(x) <= Um1_bits;
                        i0 = ok[j] ? i0 : 0;
                    }
                    const float* p = (const float*)((const char*)epi + (((unsigned)(i0 * C) << 2) + rowb));
                    rowb += stride_b;
                    if constexpr (SH && C == 3) {
                        typedef float f3u __attribute__((ext_vector_type(3), aligned(4)));
                        const f3u t3 = *(const f3u*)p;
                        e0[0][j] = t3.x, e0[1][j] = t3.y, e0[C - 1][j] = t3.z;
                    } else {
#pragma unroll
                        for (int c = 0; c < C; c++) {
                            e0[c][j] = p[c];
                            if (!SH)
                                e1[c][j] = p[C + c];
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < GP; j++) {
                    const float omt = 1.0f - tt[j];
#pragma unroll
                    for (int c = 0; c < C; c++) {
                        const float right = SH ? __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(e0[c][j]), 0x130, 0xf, 0xf, false))
                                               : e1[c][j];
                        const float m0 = omt * e0[c][j];
                        const float m1 = tt[j] * right;
                        const float r = m0 + m1;
                        park[((s0 - NRES + j) * C + c) * 64 + lane] = ok[j] ? r : kSentinel;
                    }
                    if (BORDER)
                        card_res += ok[j] ? 1 : 0;
                }
            }
        };
        if (NRES > 0 && npark > 0) {
            if (DENSE && shared_taps)
                gather_parked(std::true_type{});
            else
                gather_parked(std::false_type{});
        }
        for (int it = 0; it < a.k.n_iter; it++) {        // core.hpp:584-610
            float A[C];
#pragma unroll
            for (int c = 0; c < C; c++)
                A[c] = 0.0f;
            B = 0.0f;
            int ncard = card_res;
            if constexpr (NRES > 0)
#pragma unroll
            for (int s = 0; s < NRES; s++) {
                float q[C];
#pragma unroll
                for (int c = 0; c < C; c++) {
                    const float delta = Rres[c][s] - rbar[c];
                    const float tq = kq * delta;
                    q[c] = tq * delta;
                }
                float qs = q[0];
                if (C == 3) {
                    qs = q[0] + q[C - 1];
                    qs = qs + q[C > 1 ? 1 : 0];
                }
                const float K = kernel_weight(qs);
#pragma unroll
                for (int c = 0; c < C; c++) {
                    const float pr = Rres[c][s] * K;
                    A[c] = A[c] + pr;
                }
                B = B + K;
            }
            if (NRES > 0 && npark > 0) {
                constexpr int GP = (C == 1) ? 8 : 4;
#pragma unroll 1
                for (int sp = 0; sp < npark; sp += GP) {
                    float Rp[C][GP];
#pragma unroll
                    for (int j = 0; j < GP; j++)
#pragma unroll
                        for (int c = 0; c < C; c++)
                            Rp[c][j] = park[((sp + j) * C + c) * 64 + lane];
#pragma unroll
                    for (int j = 0; j < GP; j++) {
                        float q[C];
#pragma unroll
                        for (int c = 0; c < C; c++) {
                            const float delta = Rp[c][j] - rbar[c];
                            const float tq = kq * delta;
                            q[c] = tq * delta;
                        }
                        float qs = q[0];
                        if (C == 3) {
                            qs = q[0] + q[C - 1];
                            qs = qs + q[C > 1 ? 1 : 0];
                        }
                        const float K = kernel_weight(qs);
#pragma unroll
                        for (int c = 0; c < C; c++) {
                            const float pr = Rp[c][j] * K;
                            A[c] = A[c] + pr;
                        }
                        B = B + K;
                    }
                }
            }
            // The re-gathered tail, G samples per batch: all G address computations and loads are issued before the
            // first blend.  The wave's instruction stream is what this tail costs (two waves per SIMD: a wave gets an
            // issue slot every ~4.4 clocks whatever the instruction, PMC in DESIGN.md), so the loop is kept lean: the
            // G view offsets of a batch come from ONE broadcast LDS read issued a batch ahead, the gather address is
            // one v_mad_u32_u24 off a scalar row offset, and only the last, partial batch tests for slots past S.
            constexpr int G = (C == 1) ? 8 : 4;
            typedef float f4v __attribute__((ext_vector_type(4)));
            auto batch = [&](auto full_tag, int s0, const float (&xoff)[G]) {
                constexpr bool FULL = decltype(full_tag)::value;
                float tt[G], e0[C][G], e1[C][G];
                bool ok[G];
                unsigned rowb = (unsigned)s0 * stride_b;     // scalar
#pragma unroll
                for (int j = 0; j < G; j++) {
                    const bool live = FULL || s0 + j < S;    // wave-uniform
                    float x;
                    if (UNIFORM_D) {
                        x = xoff[j];
                    } else {
                        x = (float)(a.s_hat - min(s0 + j, S - 1)) * Dd;
                        x = x * slope;
                    }
                    x = x + uf;                              // core.hpp:552
                    tt[j] = lerp_weight(x);                  // interp.hpp:181
                    int i0 = floor_to_int(x);                // interp.hpp:179
                    ok[j] = live;
                    if (BORDER) {
                        ok[j] = live && (__float_as_uint(x) <= Um1_bits);   // interp.hpp:182 (x is never -0)
                        i0 = ok[j] ? i0 : 0;
                    }
                    if (!FULL)
                        i0 = live ? i0 : 0;                  // a slot past S: its offset is whatever follows the table
                    // both taps of every channel are 2*C consecutive floats of the row (interleaved slab); positions
                    // are below 2^24, so the 24-bit multiply-add is exact
                    const unsigned byteoff = __umul24((unsigned)i0, 4u * C) + (live ? rowb : 0u);
                    rowb += stride_b;
                    const float* p = (const float*)((const char*)epi + byteoff);
#pragma unroll
                    for (int c = 0; c < C; c++) {
                        e0[c][j] = p[c];
                        e1[c][j] = p[C + c];
                    }
                }
#pragma unroll
                for (int j = 0; j < G; j++) {
                    const float omt = 1.0f - tt[j];
                    float R[C], q[C];
#pragma unroll
                    for (int c = 0; c < C; c++) {
                        const float m0 = omt * e0[c][j];     // interp.hpp:184
                        const float m1 = tt[j] * e1[c][j];
                        float r = m0 + m1;
                        if (BORDER || !FULL)
                            r = ok[j] ? r : kSentinel;       // interp.hpp:189 stand-in: K = 0 and r * K = 0 exactly
                        R[c] = r;
                        const float delta = r - rbar[c];     // core.hpp:591
                        const float tq = kq * delta;         // kernels.cpp:21 / :43
                        q[c] = tq * delta;
                    }
                    float qs = q[0];
                    if (C == 3) {
                        qs = q[0] + q[C - 1];                // OpenCV 3.x reduceC_: (q0 + q2) + q1
                        qs = qs + q[C > 1 ? 1 : 0];
                    }
                    const float K = kernel_weight(qs);       // kernels.cpp:23-25 / :51-53
#pragma unroll
                    for (int c = 0; c < C; c++) {
                        const float pr = R[c] * K;           // core.cpp:28 / :36
                        A[c] = A[c] + pr;                    // core.hpp:602
                    }
                    B = B + K;                               // core.hpp:603
                    if (BORDER)
                        ncard += ok[j] ? 1 : 0;
                }
            };
            // Shared taps (DENSE, every offset of this hypothesis regular): one 12/4-byte load per lane and sample,
            // the right tap from lane + 1.  Lane 63 computes on its own left tap twice; it is never written.
            constexpr int GS = RSLF_STREAM_GS;   // samples per batch of the shared-tap form: their loads are all in flight before the first blend
            auto issue_shared = [&](int s0, const float (&xoff)[GS], float (&tt)[GS], float (&e0)[C][GS]) {
                unsigned rowb = (unsigned)s0 * stride_b;     // scalar
#pragma unroll
                for (int j = 0; j < GS; j++) {
                    const float x = xoff[j] + uf;            // core.hpp:552
                    tt[j] = lerp_weight(x);                  // interp.hpp:181
                    const int i0 = floor_to_int(x);          // interp.hpp:179
                    const unsigned byteoff = __umul24((unsigned)i0, 4u * C) + rowb;
                    rowb += stride_b;
                    const float* p = (const float*)((const char*)epi + byteoff);
                    if constexpr (C == 3) {
                        typedef float f3u __attribute__((ext_vector_type(3), aligned(4)));
                        const f3u t3 = *(const f3u*)p;
                        e0[0][j] = t3.x, e0[1][j] = t3.y, e0[C - 1][j] = t3.z;
                    } else {
                        e0[0][j] = p[0];
                    }
                }
            };
            auto consume_shared = [&](const float (&tt)[GS], const float (&e0)[C][GS]) {
#pragma unroll
                for (int j = 0; j < GS; j++) {
                    const float omt = 1.0f - tt[j];
                    float R[C], q[C];
#pragma unroll
                    for (int c = 0; c < C; c++) {
                        // 0x130 = wave_shl:1: lane i reads lane i + 1, the owner of this lane's right tap
                        const float e1 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(e0[c][j]), 0x130, 0xf, 0xf, false));
                        const float m0 = omt * e0[c][j];     // interp.hpp:184
                        const float m1 = tt[j] * e1;
                        const float r = m0 + m1;
                        R[c] = r;
                        const float delta = r - rbar[c];     // core.hpp:591
                        const float tq = kq * delta;         // kernels.cpp:21 / :43
                        q[c] = tq * delta;
                    }
                    float qs = q[0];
                    if (C == 3) {
                        qs = q[0] + q[C - 1];                // OpenCV 3.x reduceC_: (q0 + q2) + q1
                        qs = qs + q[C > 1 ? 1 : 0];
                    }
                    const float K = kernel_weight(qs);       // kernels.cpp:23-25 / :51-53
#pragma unroll
                    for (int c = 0; c < C; c++) {
                        const float pr = R[c] * K;           // core.cpp:28 / :36
                        A[c] = A[c] + pr;                    // core.hpp:602
                    }
                    B = B + K;                               // core.hpp:603
                }
            };
            // the table is 16-byte aligned and padded to a multiple of 4 floats; s_begin is a multiple of G
            auto offsets = [&](int s0, float (&xoff)[G]) {
                if (UNIFORM_D) {
#pragma unroll
                    for (int j4 = 0; j4 < G; j4 += 4) {
                        const f4v v4 = *(const f4v*)(otab + s0 + j4);
                        xoff[j4] = v4.x, xoff[j4 + 1] = v4.y, xoff[j4 + 2] = v4.z, xoff[j4 + 3] = v4.w;
                    }
                }
            };
            const int s_begin = NRES + npark;
            const int s_full = s_begin + (S - s_begin) / G * G;      // end of the full batches
            int s_gen = s_begin;   // where the general form takes over
            if (DENSE && shared_taps) {
                // With half the loads (shared taps) the memory pipeline keeps up; what is left is latency -- a wave
                // alone on its SIMD issues no faster than one instruction per four clocks, so its waits are never made
                // up for by the partner wave.  Eight samples per batch: the loads of all eight are in flight before
                // the first blend, and nothing is carried from one batch to the next (loop-carried prefetch registers
                // cost hipcc a copy of every loaded value right behind the loads, i.e. the wait it was meant to hide).
                const int s_full8 = s_begin + (S - s_begin) / GS * GS;
#pragma unroll 1
                for (int s0 = s_begin; s0 < s_full8; s0 += GS) {
                    float xo[GS], tt8[GS], e8[C][GS];
#pragma unroll
                    for (int j4 = 0; j4 < GS; j4 += 4) {
                        const f4v v4 = *(const f4v*)(otab + s0 + j4);
                        xo[j4] = v4.x, xo[j4 + 1] = v4.y, xo[j4 + 2] = v4.z, xo[j4 + 3] = v4.w;
                    }
                    issue_shared(s0, xo, tt8, e8);
                    consume_shared(tt8, e8);
                }
                s_gen = s_full8;
            }
            float xnext[G];
            offsets(s_gen < S ? s_gen : 0, xnext);
#pragma unroll 1
            for (int s0 = s_gen; s0 < s_full; s0 += G) {
                float xcur[G];
#pragma unroll
                for (int j = 0; j < G; j++)
                    xcur[j] = xnext[j];
                offsets(s0 + G < S ? s0 + G : s0, xnext);            // the next batch's offsets, a batch ahead
                batch(std::true_type{}, s0, xcur);
            }
            if (s_full < S)
                batch(std::false_type{}, s_full, xnext);
            if (BORDER)
                card = ncard;
#pragma unroll
            for (int c = 0; c < C; c++) {
                const float qd = (B != 0.0f) ? (A[c] / B) : 0.0f;   // core.cpp:42 / :50
                rbar[c] = (qd > 0.0f) ? qd : 0.0f;                  // core.hpp:609
            }
        }
        const float cardf = (float)card;
        float sc = (card != 0) ? (B / cardf) : 0.0f;     // core.hpp:616-620
        sc = (sc > 0.0f) ? sc : 0.0f;                    // core.hpp:622
        if (!LANE_D || dk + dlane < d1)
            best.offer(sc, d, Dd, rbar);
    }
}

// A wave whose every sample line stays inside [0, U-1] for every hypothesis
// needs no validity test: |x - u| <= max|s_hat - s| * max|d| * slope.
__device__ __forceinline__ bool wave_is_interior(const ScanArgs& a, int u)
{
    if (a.dmin_vu)
        return false;
    const float max_ds = (float)max(a.s_hat, a.vol.S - 1 - a.s_hat);
    const float max_d = fmaxf(fabsf(a.dmin), fabsf(a.dmax));
    const float reach = max_ds * max_d * fabsf(a.k.slope) + 2.0f;
    const float uf = (float)u;
    return __all((uf - reach >= 0.0f) && (uf + reach <= (float)(a.vol.U - 1)));
}

// Whether a tile needs the validity test is decided per HYPOTHESIS, not once per tile: the reach of the sample lines is
// max|s_hat - s| * |D[d]| * slope, and with 201 views and disparities up to 6 px/view (BASELINE.json configs[4]) the
// all-hypotheses bound makes a third of a 4096-pixel row "border" where the per-hypothesis one leaves 15 %.  The border
// form costs about twice the dense one in the re-gathered tail (two loads per sample instead of one shared tap).
// Runs of hypotheses of the same kind go to one body call, in ascending order (first maximum wins, core.hpp:636-645).
template <int C, int NRES>
__device__ __forceinline__ void scan_stream_rows(const ScanArgs& a, int v, int u, bool active, int d0, int d1, Best<C>& best,
                                                 float* otab)
{
    if (a.dmin_vu) {
        scan_stream_body<C, true, false, NRES>(a, v, u, d0, d1, best, otab);
        return;
    }
    // 63 consecutive pixels in lanes 0..62 (lane 63 is idle and shadows lane 62): lane 63 moves one pixel on -- still
    // inside the row for every sample of an interior hypothesis, which leaves two pixels of margin -- and the tail
    // shares taps between neighbours
    // (every lane is compared, not just the ends: a short or gappy list can span 62 pixels too -- idle lanes shadow
    // the last entry; found by the fuzz campaign, profiles/r02_fuzz_parity.txt)
    const int u0 = __builtin_amdgcn_readfirstlane(u), u62 = __builtin_amdgcn_readlane(u, 62);
    const int ln = threadIdx.x & 63;
    const bool consecutive = __all(ln > 62 || u == u0 + ln);
    // (a row's last tile may hold a 64th entry, scan_tile: lane 63 is then a pixel of its own and cannot lend itself out)
    const bool lane63_free = !__any(ln == 63 && active);
    const bool dense = a.tile_w == 63 && consecutive && lane63_free && NRES + a.stream_park < a.vol.S;
    const int ud = (ln == 63) ? u62 + 1 : u;
    const float max_ds = (float)max(a.s_hat, a.vol.S - 1 - a.s_hat);
    const float range = a.dmax - a.dmin, denom = (float)(a.dim_d - 1);
    const float uf = (float)u, Um1 = (float)(a.vol.U - 1);
    auto interior = [&](int d) -> bool {   // |x - u| <= max|s_hat - s| * |D[d]| * slope for every sample of hypothesis d
        const float reach = max_ds * fabsf(hypothesis(a.dmin, range, denom, d)) * fabsf(a.k.slope) + 2.0f;
        return __all((uf - reach >= 0.0f) && (uf + reach <= Um1));
    };
    int d = d0;
    while (d < d1) {
        const bool in = interior(d);
        int e = d + 1;
        while (e < d1 && interior(e) == in)
            e++;
        if (!in)
            scan_stream_body<C, true, true, NRES>(a, v, u, d, e, best, otab);
        else if (dense)
            scan_stream_body<C, false, true, NRES, true>(a, v, ud, d, e, best, otab);
        else
            scan_stream_body<C, false, true, NRES>(a, v, u, d, e, best, otab);
        d = e;
    }
}

// One kernel per resident-prefix length and launch form (round 4): a single kernel holding all three prefix lengths and both
// forms was allocated for the worst of its six paths and carried that path's scratch everywhere (544 B/lane for RGB).
// NRES = stream_resident_for(S, C); PACKED = one packed pixel list (lanes on different scanlines: per-lane offsets).
template <int C, int NRES, bool PACKED>
__global__ __launch_bounds__(64 * kScanWaves) __attribute__((amdgpu_waves_per_eu(RSLF_STREAM_WAVES, 8))) void k2_scan_stream(ScanArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float s_stream_otab[];   // [kScanWaves][stream_wave_floats]
    float* otab = s_stream_otab + (size_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * a.stream_wave_floats;
    // the waves' results for the epilogue go to the head of their own regions (EpilogueBlock): no static LDS at all
    constexpr bool kEpiDyn = true;
    float* const epi_lds = otab;
    const int epi_stride = a.stream_wave_floats;
    if constexpr (PACKED) {
        RSLF_SCAN_PACKED_LOOP((scan_stream_body<C, true, false, NRES>(a, v, u, d0, d1, best, otab)))
    } else {
        RSLF_SCAN_ROW_TILE((scan_stream_rows<C, NRES>(a, v, u, active, d0, d1, best, otab)))
    }
}

// host side: f(kernel) for the instantiation with resident prefix `nres` (one of 0, stream_resident_lo(C), stream_resident_hi(C))
template <int C, bool PACKED, class F>
inline void stream_kernel_for(int nres, F f)
{
    if (nres == stream_resident_hi(C)) {
        f(k2_scan_stream<C, stream_resident_hi(C), PACKED>);
        return;
    }
    if constexpr (stream_resident_lo(C) > 0 && stream_resident_lo(C) != stream_resident_hi(C)) {
        if (nres == stream_resident_lo(C)) {
            f(k2_scan_stream<C, stream_resident_lo(C), PACKED>);
            return;
        }
    }
    f(k2_scan_stream<C, 0, PACKED>);
}

// Sparse launches of the stream-class units (RGB above 48 views, one channel above 192), lanes own HYPOTHESES
// (k2_scan_stream_px; the register kernels' form of this is k2_scan_reg_px, k2_reg.hpp).  With a pixel per lane a sparse
// list puts a wave's 64 lanes on up to 64 scanlines: every load of the gather -- and of the tail re-gathered on every
// pass -- touches up to 64 cache lines.  Here a wave owns ONE pixel: a load's 64 taps lie within (d_63 - d_0) * |s_hat - s|
// pixels of one EPI row, the EPI base is a scalar, nothing is shared between workgroups (no hypothesis groups, records or
// tickets).  px_waves = 1 / 2 / 4 waves share a pixel's hypotheses (plan::px_waves).  A unit's arithmetic is
// scan_stream_body's per-lane-offset form -- the one per-pixel [dmin, dmax] planes and the packed tiles have always taken:
// same operations, same bits -- with the same three tiers (resident prefix, samples parked in LDS, re-gathered tail); the
// reduction over hypotheses is scan_px_finish (k2_scan.hpp).  The fine-to-coarse run of the report's MansionLR shape
// (1146 x 720, 100 views RGB, report/rs_report.tex:406,427; core.hpp:993-1028 is the caller) spends nine tenths of its time
// in these launches.
template <int C, int NRES>
__global__ __launch_bounds__(64 * kScanWaves) __attribute__((amdgpu_waves_per_eu(RSLF_STREAM_WAVES, 8))) void k2_scan_stream_px(ScanArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float s_stream_otab[];   // [kScanWaves][stream_wave_floats]
    __shared__ double s_sum[kScanWaves];
    __shared__ float s_rec[kScanWaves][kPxRecFloats];
    const int n = *a.packed_n;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* otab = s_stream_otab + (size_t)wave * a.stream_wave_floats;
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
        if (!have) {
            // nothing to scan: the finish below still runs (its barriers are the workgroup's)
        } else if (interior)
            scan_stream_body<C, false, false, NRES, false, true>(a, v, u, 0, a.dim_d, best, otab, dlane, dstep);
        else
            scan_stream_body<C, true, false, NRES, false, true>(a, v, u, 0, a.dim_d, best, otab, dlane, dstep);
        scan_px_finish<C>(a, o, have, best, wave, lane, wpp, s_rec, s_sum);
    }
}

template <int C, class F>
inline void stream_px_kernel_for(int nres, F f)
{
    if (nres == stream_resident_hi(C)) {
        f(k2_scan_stream_px<C, stream_resident_hi(C)>);
        return;
    }
    if constexpr (stream_resident_lo(C) > 0 && stream_resident_lo(C) != stream_resident_hi(C)) {
        if (nres == stream_resident_lo(C)) {
            f(k2_scan_stream_px<C, stream_resident_lo(C)>);
            return;
        }
    }
    f(k2_scan_stream_px<C, 0>);
}

}  // namespace rslf
