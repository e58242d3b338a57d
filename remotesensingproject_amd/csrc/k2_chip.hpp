// K2, on-chip variant (k2_scan_chip): RGB light fields with more views than two waves per SIMD can hold on chip -- 123 to
// 220 views; BASELINE.json's 201-view RGB config (c5) is the top rung of its ladder.
//
// The streaming variant (k2_stream.hpp) runs two waves per SIMD: 256 registers and 20 KiB of LDS per wave hold 92 of a
// unit's RGB samples, and the others are re-gathered on each of the ten mean-shift passes -- 34 vector instructions and
// a 12-byte L1 read per sample and pass against 19 for a sample that is at hand; three quarters of its time at 201 views.
// Here a wave has its SIMD to itself: 512 registers (the unified file: 256 VGPRs + 256 AGPRs) and a quarter of the CU's
// 160 KiB of LDS hold all but three samples of a unit -- in the order a pass walks them,
//     views [0, NV)               in VGPRs, as register pairs (s, s+1) per channel
//     [NV, NV + NA)               in AGPRs, laid out by hand: one v_accvgpr_read_b32 per value and pass (VALU operands cannot name an AGPR)
//     the next three              fetched AGAIN on every pass (L1 / L2 hits), their loads issued a tier ahead of their use
//     [NV + NA + 3, .. + NL)      in LDS, [pair][channel][lane] as 8-byte pairs: one conflict-free ds_read_b64 per pair and channel
//     the rest (beyond the top rung only: 202 .. 220 views) fetched again per pass too, in pairs, one pair ahead
// -- NV = 64; NA and NL are template parameters, one instantiation per RUNG of plan::kChipLadder (rslf_plan.hpp): the tiers
// fill in the order VGPRs, AGPRs, LDS as the view count grows (127 views = <60, 0> ... 151 = <84, 0> ... 201 = <84, 50>,
// 8 views apart), and a volume runs on the smallest rung that holds all its views, the missing ones PADDED with samples
// that contribute nothing (PAD, below).  At c5 the chip is full to the word: 201 x 3 samples + the wave's running result
// -- a double sum and the best score in three AGPRs, the best index and rbar in LDS -- is what 192 + 256 registers and
// 157 words of LDS per lane hold, less two samples.  A wave alone on its SIMD issues one instruction every ~5 clocks
// whatever it is (tools/ubench_valu.hip), so the pass runs in packed fp32 on sample pairs: v_pk_add / v_pk_mul do two
// samples' work per issue slot, each half the scalar instruction's IEEE operation (tools/ubench_pk.hip).  The two running
// sums still take one sample at a time in ascending s (core.hpp:602-603).  No scratch in the kernels of record, and nothing
// written but the workgroup's record: the kernel's HBM traffic is the slab read once plus 2 KB per (tile, group).
// Against the streaming kernel on a 1146 x 720 x 120-hypothesis dense step: 127 views 69.6 vs 71.9 ms, 151 82.9 vs 93.5,
// 201 109.6 vs 143.1; below 123 views two waves per SIMD win (100 views: 54.2 vs 45.5) -- profiles/r04_k2_variants.md section 9.
//
// Dense row-tile launches with one hypothesis grid for all pixels only (no per-pixel [dmin, dmax] planes, no packed
// lists): everything else stays with the streaming kernel.
#pragma once

#include "k2_scan.hpp"

namespace rslf {

// samples per tier (all even: the pass works on pairs).  NV: 3 * NV VGPRs beside ~60 of working state (60 left four more
// samples of c5 to per-pass fetches and ran 7.6 % slower, profiles/r03_k2_variants.md); NA: 3 * NA + 3 <= 256 AGPRs;
// NL: what a quarter of the CU's LDS holds behind the wave's offset table and before its running best.
// NV is fixed; NA and NL are TEMPLATE PARAMETERS of the kernel (round 4): the tiers fill in the order VGPRs, AGPRs, LDS as
// the view count grows, one instantiation per rung of plan::kChipLadder (rslf_plan.hpp), and a volume runs on the largest
// rung it fills -- what it has beyond the rung's views is fetched again per pass (TAIL).  c5's 201 views are the top rung
// exactly: <84, 50>.
#ifndef RSLF_CHIP_NV
#define RSLF_CHIP_NV 64
#endif
#ifndef RSLF_CHIP_PD
#define RSLF_CHIP_PD 3
#endif
constexpr int kChipNV = RSLF_CHIP_NV;
constexpr int kChipNAMax = plan::kChipNAMax, kChipNLMax = plan::kChipNLMax;
constexpr int kChipAhead = plan::kChipAhead;               // samples fetched again on every pass, ahead of their use
constexpr int kChipBestFloats = plan::kChipBestFloats;     // a wave's running index and rbar (ChipBest)
constexpr size_t kChipLdsBytes = plan::kChipLdsBytes;      // one workgroup per CU may take all of it
static_assert(kChipNV == plan::kChipNV && kChipNV % 4 == 0, "gather batches of four samples, pairs in the pass");
static_assert(3 * kChipNAMax + 3 <= 256, "AGPR tier and the three named registers");

// f(integral_constant<int, G>) for G = 0 .. N-1, in order: a loop unrolled by the type system -- `#pragma unroll` leaves the
// 49-batch gather rolled, and a rolled loop cannot index registers
template <int G, int N>
struct ChipUnroll {
    template <class F>
    static __device__ __forceinline__ void run(F& f)
    {
        f(std::integral_constant<int, G>{});
        ChipUnroll<G + 1, N>::run(f);
    }
};
template <int N>
struct ChipUnroll<N, N> {
    template <class F>
    static __device__ __forceinline__ void run(F&) {}
};

// The 256 AGPRs are laid out BY HAND: sample i of the AGPR tier has its channels in a[3i .. 3i + 2], a253:a254 hold a
// lane's running score sum (a double) and a255 its best score (ChipBest, below).  The register numbers are template
// constants printed into the instruction text ("a%c[n]"), and every statement that touches one lists ALL of them as
// clobbered.  A clobber list binds the allocator across that one statement only -- between statements it could still park
// a value of its own in an AGPR -- so the layout is VERIFIED PER BUILD, not guaranteed: tests/test_isa_cpu.py disassembles
// the shipped kernel and requires exactly the accumulator reads and writes written here, none moved, none named by any
// other instruction.  (Left to the allocator as "a"-constrained values, 252 + 3 of 256
// was more than it could colour: it spilled an AGPR value per hypothesis in one build and five in the next.)
#define RSLF_CHIP_AGPRS \
    "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", \
    "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", \
    "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", \
    "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", \
    "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", \
    "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", \
    "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", \
    "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", \
    "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", \
    "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", \
    "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", \
    "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", \
    "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", \
    "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", \
    "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", \
    "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255"

template <int R>
__device__ __forceinline__ void agpr_put(float v)
{
    static_assert(R >= 0 && R < 3 * kChipNAMax, "an AGPR-tier register");
    asm volatile("v_accvgpr_write_b32 a%c1, %0" : : "v"(v), "n"(R) : RSLF_CHIP_AGPRS);
}

// One pair of samples (s, s+1) of one RGB pass in packed fp32: per channel delta = R - rbar, t = kq * delta,
// q = t * delta; qs = (q0 + q2) + q1 (OpenCV 3.x reduceC_); K = clamp(1 - qs); P_c = R_c * K -- 15 packed instructions;
// rbar arrives as two register pairs {rbar0, rbar1}, {rbar2, -} and is broadcast to both halves by op_sel.
// A wave alone on its SIMD issues in order, one instruction per ~5 clocks, and an instruction that reads its
// predecessor's result waits longer: so the block is SKEWED -- it also makes the eight sequential adds of the PREVIOUS
// pair's P and K (core.hpp:602-603: sample s, then s + 1, one IEEE add each), placed where the packed chain would stall.
// `prev` = the previous pair's results (zeros before the first pair of a pass: +0 added to a sum changes nothing).
struct ChipPK {
    f2 P0, P1, P2, K;
};

// between the three subtractions and the three products R * K: the same text in every form of the block
#define RSLF_CHIP_PAIR_MID \
    "v_pk_mul_f32 %[t0], %[kq], %[d0] op_sel_hi:[0,1]\n\t" \
    "v_pk_mul_f32 %[t1], %[kq], %[d1] op_sel_hi:[0,1]\n\t" \
    "v_pk_mul_f32 %[t2], %[kq], %[d2] op_sel_hi:[0,1]\n\t" \
    "v_pk_mul_f32 %[d0], %[d0], %[t0]\n\t" \
    "v_pk_mul_f32 %[d2], %[d2], %[t2]\n\t" \
    "v_pk_mul_f32 %[d1], %[d1], %[t1]\n\t" \
    "v_add_f32 %[A0], %[A0], %[p0x]\n\t" \
    "v_add_f32 %[A1], %[A1], %[p1x]\n\t" \
    "v_pk_add_f32 %[d0], %[d0], %[d2]\n\t" \
    "v_add_f32 %[A2], %[A2], %[p2x]\n\t" \
    "v_add_f32 %[B], %[B], %[kx]\n\t" \
    "v_add_f32 %[A0], %[A0], %[p0y]\n\t" \
    "v_pk_add_f32 %[d0], %[d0], %[d1]\n\t" \
    "v_add_f32 %[A1], %[A1], %[p1y]\n\t" \
    "v_add_f32 %[A2], %[A2], %[p2y]\n\t" \
    "v_add_f32 %[B], %[B], %[ky]\n\t" \
    "v_pk_add_f32 %[k], 1.0, %[d0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1] clamp\n\t"

#define RSLF_CHIP_PAIR_OUTS(n)                                                                                        \
    [d0] "=&v"(d0), [d1] "=&v"(d1), [d2] "=&v"(d2), [t0] "=&v"(n.P0), [t1] "=&v"(n.P1), [t2] "=&v"(n.P2), [k] "=&v"(n.K), \
        [A0] "+v"(A[0]), [A1] "+v"(A[1]), [A2] "+v"(A[2]), [B] "+v"(B)
#define RSLF_CHIP_PAIR_PREV(p)                                                                                        \
    [p0x] "v"(p.P0.x), [p0y] "v"(p.P0.y), [p1x] "v"(p.P1.x), [p1y] "v"(p.P1.y), [p2x] "v"(p.P2.x), [p2y] "v"(p.P2.y),    \
        [kx] "v"(p.K.x), [ky] "v"(p.K.y)

// samples in VGPR pairs (the VGPR tier, and the LDS tier once its ds_read_b64 have landed)
__device__ __forceinline__ ChipPK chip_pair(f2 r0, f2 r1, f2 r2, f2 m01, f2 m2x, unsigned long long kq, const ChipPK& prev, float (&A)[3], float& B)
{
    ChipPK n;
    f2 d0, d1, d2;
    asm("v_pk_add_f32 %[d0], %[r0], %[m01] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %[d1], %[r1], %[m01] op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %[d2], %[r2], %[m2x] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        RSLF_CHIP_PAIR_MID
        "v_pk_mul_f32 %[t0], %[r0], %[k]\n\t"
        "v_pk_mul_f32 %[t1], %[r1], %[k]\n\t"
        "v_pk_mul_f32 %[t2], %[r2], %[k]"
        : RSLF_CHIP_PAIR_OUTS(n)
        : [r0] "v"(r0), [r1] "v"(r1), [r2] "v"(r2), [m01] "v"(m01), [m2x] "v"(m2x), [kq] "s"(kq), RSLF_CHIP_PAIR_PREV(prev));
    return n;
}

// samples in AGPRs (pair P of the tier: a[6P .. 6P + 5]): the six reads open the block (VALU operands cannot name an AGPR);
// they land in a fixed register window, v[250:255], whose halves the asm can name -- an operand the compiler allocates is
// a whole pair to the asm.
template <int P>
__device__ __forceinline__ ChipPK chip_pair_agpr(f2 m01, f2 m2x, unsigned long long kq, const ChipPK& prev, float (&A)[3], float& B)
{
    static_assert(P >= 0 && 2 * P + 1 < kChipNAMax, "a pair of the AGPR tier");
    ChipPK n;
    f2 d0, d1, d2;
    asm volatile("v_accvgpr_read_b32 v250, a%c[a0x]\n\t"
                 "v_accvgpr_read_b32 v251, a%c[a0y]\n\t"
                 "v_accvgpr_read_b32 v252, a%c[a1x]\n\t"
                 "v_accvgpr_read_b32 v253, a%c[a1y]\n\t"
                 "v_accvgpr_read_b32 v254, a%c[a2x]\n\t"
                 "v_accvgpr_read_b32 v255, a%c[a2y]\n\t"
                 "v_pk_add_f32 %[d0], v[250:251], %[m01] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                 "v_pk_add_f32 %[d1], v[252:253], %[m01] op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                 "v_pk_add_f32 %[d2], v[254:255], %[m2x] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                 RSLF_CHIP_PAIR_MID
                 "v_pk_mul_f32 %[t0], v[250:251], %[k]\n\t"
                 "v_pk_mul_f32 %[t1], v[252:253], %[k]\n\t"
                 "v_pk_mul_f32 %[t2], v[254:255], %[k]"
                 : RSLF_CHIP_PAIR_OUTS(n)
                 : [a0x] "n"(6 * P), [a1x] "n"(6 * P + 1), [a2x] "n"(6 * P + 2), [a0y] "n"(6 * P + 3), [a1y] "n"(6 * P + 4), [a2y] "n"(6 * P + 5),
                   [m01] "v"(m01), [m2x] "v"(m2x), [kq] "s"(kq), RSLF_CHIP_PAIR_PREV(prev)
                 : "v250", "v251", "v252", "v253", "v254", "v255", RSLF_CHIP_AGPRS);
    return n;
}

// the last pair's P and K, once no pair follows
__device__ __forceinline__ void chip_flush(const ChipPK& p, float (&A)[3], float& B)
{
    A[0] = A[0] + p.P0.x;
    A[1] = A[1] + p.P1.x;
    A[2] = A[2] + p.P2.x;
    B = B + p.K.x;
    A[0] = A[0] + p.P0.y;
    A[1] = A[1] + p.P1.y;
    A[2] = A[2] + p.P2.y;
    B = B + p.K.y;
}

// NOTHING a lane owns is carried through a hypothesis by the compiler's choice: with 192 of the 256 registers holding
// samples, every per-lane value that lives across the gather and the passes -- the pixel's column, its float, its byte
// offset, the lane's LDS addresses -- was spilled once per workgroup and re-read once per hypothesis (84 bytes of scratch
// per lane; dirty scratch lines pushed out of the L2 by the streaming reads were four fifths of the kernel's HBM writes).
// So a tile is known by wave-UNIFORM values only (scalar registers), and what a lane needs is made again where it is
// used: its number from v_mbcnt (an asm the optimiser cannot hoist or merge), its pixel from the scanline's list
// (K1's output, an L2 hit) -- or, on a dense tile, first pixel + lane.
__device__ __forceinline__ int chip_lane()
{
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\t"
                 "v_mbcnt_hi_u32_b32 %0, -1, %0"
                 : "=v"(l));
    return l;
}
struct ChipTile {
    const int* row;   // the scanline's pixel list
    int e0, last;     // the tile's first entry; the last entry it may read (idle lanes shadow it, scan_tile)
    int u0;           // the first entry's pixel
    __device__ __forceinline__ int pixel(int lane) const { return row[min(e0 + lane, last)]; }
};

// A wave's running result over its hypotheses, kept OUT of the allocator's vector registers.  With 192 of the 256
// holding samples hipcc spilled Best (eight values per lane) to scratch and wrote it back after every hypothesis: 32 bytes
// per lane and hypothesis, 4.3 GB of HBM writes per 64 scanlines of c5 against 0.64 GB of algorithmic traffic (the
// streaming reads push the dirty lines out of the L2 as fast as they are made).  (Index and rbar in scratch by hand,
// written only where a lane's best improves, still wrote 1.1 GB: scores rise smoothly towards a pixel's disparity, so
// half the hypotheses improve on their predecessor.)  What every hypothesis touches -- the double sum of the scores
// (cv::mean, core.hpp:641) and the best score -- sits in a253:a254 and a255; index and rbar sit in LDS, [4][64] behind the
// wave's sample tier, which gave up two samples for them (kChipAhead).  Same operations in the same order as Best<3>
// (core.hpp:636-645: strictly greater, first maximum).
struct ChipBest {
    float* wave_blk;   // [index (bits), rbar0, rbar1, rbar2][64 lanes]
    __device__ __forceinline__ void init(float* wave_block)
    {
        wave_blk = wave_block;
        float* blk = wave_blk + chip_lane();
        blk[0] = __int_as_float(0);
        blk[64] = blk[128] = blk[192] = 0.0f;
        asm volatile("v_accvgpr_write_b32 a253, 0\n\t"
                     "v_accvgpr_write_b32 a254, 0\n\t"
                     "v_accvgpr_write_b32 a255, -1.0"
                     :
                     :
                     : RSLF_CHIP_AGPRS);
    }
    __device__ __forceinline__ void offer(float sc, int d, const float (&rb)[3])
    {
        unsigned lo, hi;
        float cur;
        asm volatile("v_accvgpr_read_b32 %0, a253\n\t"
                     "v_accvgpr_read_b32 %1, a254\n\t"
                     "v_accvgpr_read_b32 %2, a255"
                     : "=&v"(lo), "=&v"(hi), "=&v"(cur)
                     :
                     : RSLF_CHIP_AGPRS);
        const double sum = __hiloint2double((int)hi, (int)lo) + (double)sc;
        asm volatile("v_accvgpr_write_b32 a253, %0\n\t"
                     "v_accvgpr_write_b32 a254, %1"
                     :
                     : "v"(__double2loint(sum)), "v"(__double2hiint(sum))
                     : RSLF_CHIP_AGPRS);
        if (sc > cur) {
            asm volatile("v_accvgpr_write_b32 a255, %0" : : "v"(sc) : RSLF_CHIP_AGPRS);
            float* blk = wave_blk + chip_lane();
            blk[0] = __int_as_float(d);
            blk[64] = rb[0];
            blk[128] = rb[1];
            blk[192] = rb[2];
        }
    }
    __device__ __forceinline__ void finish(const ScanArgs& a, Best<3>& b) const
    {
        unsigned lo, hi;
        asm volatile("v_accvgpr_read_b32 %0, a253\n\t"
                     "v_accvgpr_read_b32 %1, a254\n\t"
                     "v_accvgpr_read_b32 %2, a255"
                     : "=&v"(lo), "=&v"(hi), "=&v"(b.score)
                     :
                     : RSLF_CHIP_AGPRS);
        b.sum = __hiloint2double((int)hi, (int)lo);
        const float* blk = wave_blk + chip_lane();
        b.d = __float_as_int(blk[0]);
        b.D = hypothesis(a.dmin, a.dmax - a.dmin, (float)(a.dim_d - 1), b.d);   // the scan's own operations (core.hpp:545-548)
        b.rbar[0] = blk[64];
        b.rbar[1] = blk[128];
        b.rbar[2] = blk[192];
    }
};

// r = (view W of S is missing) ? sentinel : r, for the padded rungs.  By hand: left to hipcc, the thirteen wave-uniform
// answers were hoisted out of the hypothesis loop as lane masks (26 scalar registers: the scalar file overflowed into vector
// registers and those into scratch) and the sentinel sat in a vector register of its own; here the answer is made where it
// is used and the constant lives for four instructions (v_cndmask_b32_e32: D = VCC ? src1 : src0).
template <int W>
__device__ __forceinline__ void chip_pad_select(float (&r)[3], int S)
{
    static_assert(kSentinel == 1e30f, "the literal below");
    float sent;
    asm volatile("s_cmp_gt_i32 %4, %5\n\t"            // SCC = S > W: the view exists
                 "s_cselect_b64 vcc, -1, 0\n\t"
                 "v_mov_b32_e32 %3, 0x7149f2ca\n\t"   // (VCC and a literal in one instruction are two reads of the constant bus)
                 "v_cndmask_b32_e32 %0, %3, %0, vcc\n\t"
                 "v_cndmask_b32_e32 %1, %3, %1, vcc\n\t"
                 "v_cndmask_b32_e32 %2, %3, %2, vcc"
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "=&v"(sent)
                 : "s"(S), "n"(W)
                 : "vcc", "scc");
}

// SHARED = false: the general form -- every lane loads both taps of a sample (16 + 8 bytes) and tests the sample's validity
// (interp.hpp:182).  SHARED = true: lanes 0..62 hold 63 consecutive pixels, lane 63 stands on the pixel after them, every
// sample line of every lane stays inside the row and all lanes floor alike (scan_chip_rows): a lane's right tap is its
// neighbour's left tap, so a sample costs ONE 12-byte load, the right tap comes from lane + 1 (v_mul_f32_dpp wave_shl:1),
// and with 4 registers per sample in flight instead of 7 the gather runs TWO batches ahead of its blends -- a wave alone on
// its SIMD has nobody to hide an L2 round trip behind (the general form, one batch ahead, waits a third of its gather).
// TAIL = false: exactly the rung's views (c5's 201 on the top rung): the per-pass fetches of further views are not compiled in (their
// registers cost the 201-view kernel 0.5 %).
// PAD = true: the volume has FEWER views than the rung holds (at most kChipPadMax fewer: plan::chip_rung_for): the missing
// ones are samples that contribute nothing -- the sentinel, K = 0 and R K = 0 exactly, added to the sums as +0 -- read from
// the last view's row at offset 0 (in bounds), not counted in the cardinal.  A view of padding costs a quarter of what a
// view fetched again per pass does, which is why a volume runs on the rung ABOVE its view count, not the one below.
template <bool SHARED, bool TAIL, int NA, int NL, bool PAD>
__device__ __forceinline__ void scan_chip_body(const ScanArgs& a, int v, const ChipTile& t, int d0, int d1, ChipBest& best, float* __restrict__ otab)
{
    constexpr int C = 3, NV = kChipNV, GB = 4;
    static_assert(NA % 4 == 0 && NA <= kChipNAMax && NL % 2 == 0 && NL <= kChipNLMax, "a rung of the ladder");
    constexpr int NO = NV + NA + NL, NB = (NO + GB - 1) / GB;   // on-chip samples; gather batches (the last may be half full)
    constexpr int T = NO + kChipAhead;                          // the rung's views
    static_assert(!(PAD && TAIL), "a volume is padded up to a rung or has views beyond the top one");
    // gather index -> view (the gather walks the LDS tier, then the AGPR tier, then the VGPR tier)
    auto view_of = [](int i) constexpr { return i < NL ? NV + NA + kChipAhead + i : i < NL + NA ? NV + (i - NL) : i - NL - NA; };
    // can view w be padding?  (compile-time: only the rung's last kChipPadMax views)
    auto may_pad = [](int w) constexpr { return PAD && w >= T - plan::kChipPadMax; };
    constexpr bool BORDER = !SHARED;
    constexpr int PD = SHARED ? RSLF_CHIP_PD : 2;   // batches of loads in flight + 1
    const VolView& vol = a.vol;
    const float* epi = vol.row(v, 0);
    // Values every hypothesis needs but that cost an instruction or two to make are made per hypothesis: hipcc otherwise
    // hoists them out of the loop into vector registers and spills them (ChipTile, above).
    const unsigned Um1_bits = __float_as_uint((float)(vol.U - 1));
    const int S = vol.S;
    // kq for the packed multiplies: a scalar-register PAIR whose low half op_sel broadcasts (no vector registers held for it)
    const unsigned long long kq2 = (unsigned long long)__float_as_uint(a.k.inv_h2);

#pragma unroll 1
    for (int d = d0; d < d1; d++) {
    unsigned stride_b = (unsigned)vol.stride_s << 2;
    asm volatile("" : "+s"(stride_b));   // (its multiples are made where they are used, not carried in scalar registers)
    const int lane = chip_lane();
    const int u = SHARED ? t.u0 + lane : t.pixel(lane);   // (dense tiles: lanes 0..62 consecutive, lane 63 on the pixel after them)
    const float uf = (float)u;
    const unsigned centre_off = ((unsigned)(a.s_hat * (int)vol.stride_s) + (unsigned)(u * C)) << 2;   // 32-bit: one EPI is < 2 GiB
    // LDS tier: [pair][channel][lane] as f2 behind the offset table
    f2* park = reinterpret_cast<f2*>(otab + plan::chip_table_floats(S, T)) + lane;
    unsigned row_last = 0;   // PAD: the row every missing view reads
    if constexpr (PAD) {
        row_last = (unsigned)(S - 1) * stride_b;
        asm volatile("" : "+s"(row_last));
    }

    // one sample's two taps and its blend (interp.hpp:179-190); `rowb` = byte offset of view s's row in the EPI
    // `xoff` = fl(fl(float(s_hat - s) * D[d]) * slope), from the wave's offset table (one broadcast read serves four samples)
    auto taps = [&](float xoff, unsigned rowb, float (&e0)[C], float (&e1)[C], float& tt, bool& ok) {
        float x = xoff + uf;                     // core.hpp:552
        tt = lerp_weight(x);                     // interp.hpp:181
        int i0 = floor_to_int(x);                // interp.hpp:179
        ok = true;
        if (BORDER) {
            ok = __float_as_uint(x) <= Um1_bits; // interp.hpp:182 (x is never -0)
            i0 = ok ? i0 : 0;
        }
        const float* p = (const float*)((const char*)epi + (__umul24((unsigned)i0, 4u * C) + rowb));
        if constexpr (SHARED) {
            typedef float f3u __attribute__((ext_vector_type(3), aligned(4)));
            const f3u t3 = *(const f3u*)p;
            e0[0] = t3.x, e0[1] = t3.y, e0[2] = t3.z;
        } else {
#pragma unroll
            for (int c = 0; c < C; c++) {
                e0[c] = p[c];
                e1[c] = p[C + c];
            }
        }
    };
    auto blend = [&](const float (&e0)[C], const float (&e1)[C], float tt, bool ok, float (&r)[C]) {
        const float omt = 1.0f - tt;
#pragma unroll
        for (int c = 0; c < C; c++) {
            const float m0 = omt * e0[c];        // interp.hpp:184
            // 0x130 = wave_shl:1: lane i reads lane i + 1, the owner of this lane's right tap
            const float right = SHARED ? __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(e0[c]), 0x130, 0xf, 0xf, false)) : e1[c];
            const float m1 = tt * right;
            const float rr = m0 + m1;
            r[c] = (BORDER && !ok) ? kSentinel : rr;   // interp.hpp:189 stand-in: K = 0 and r * K = 0 exactly
        }
    };

    {
        float dmin_s = a.dmin, dmax_s = a.dmax, slope = a.k.slope;
        int dim_s = a.dim_d;
        asm volatile("" : "+s"(dmin_s), "+s"(dmax_s), "+s"(slope), "+s"(dim_s));   // opaque: re-derived here, not hoisted
        const float range = dmax_s - dmin_s;
        const float denom = (float)(dim_s - 1);
        const float Dd = hypothesis(dmin_s, range, denom, d);
        for (int s = lane; s < (PAD ? T : S); s += 64) {
            float off = (float)(a.s_hat - s) * Dd;   // core.hpp:542,550
            off = off * slope;                       // core.hpp:551
            if (PAD && s >= S)
                off = 0.0f;                          // a missing view: any in-bounds address will do
            // the table is in the order of USE: the on-chip samples as the gather walks them (LDS tier, AGPR tier, VGPR
            // tier), then the three fetched ahead, then the ragged tail
            const int pos = s < NV                        ? NL + NA + s
                            : s < NV + NA                 ? NL + (s - NV)
                            : s < NV + NA + kChipAhead    ? NO + (s - NV - NA)
                            : s < NO + kChipAhead         ? s - NV - NA - kChipAhead
                                                          : s;
            otab[pos] = off;
        }
        __builtin_amdgcn_wave_barrier();
        int card = BORDER ? 0 : S;

        f2 Rv[C][NV / 2];
        // ---- gather, once per hypothesis: batches of four samples, software-pipelined -- the loads of the batches to come
        // are issued before batch g is blended, because a wave alone on its SIMD has nobody to hide an L2 round trip
        // behind (PMC on the first version: a quarter of the wave's cycles spent in s_waitcnt).  Fully unrolled: the
        // buffers rotate by a compile-time index, so nothing is copied.  The VGPR tier is gathered LAST: until then its 192
        // registers are free, and as they fill up the loads in flight run out -- gathered first, the tier kept the file
        // full for two thirds of the gather and hipcc spilled four of its registers once per hypothesis.
        // (The AGPR tier stays in the middle: gathered first, hipcc spills five of ITS values.)
        // gather index i: [0, NL) LDS tier (views NV + NA + 3 ..), [NL, NL + NA) AGPR tier (views NV ..), then the VGPR tier
        // (a rung without an LDS tier starts in the AGPR tier)
        unsigned rowb = (unsigned)(NL > 0 ? NV + NA + kChipAhead : NV) * stride_b;
        if (may_pad(NL > 0 ? NV + NA + kChipAhead : NV))
            rowb = min(rowb, row_last);
        asm volatile("" : "+s"(rowb));
        typedef float f4v __attribute__((ext_vector_type(4)));
        float e0[PD][GB][C], e1[PD][GB][C], tt[PD][GB];
        bool ok[PD][GB];
        f4v xo[PD];   // the offsets of a batch, read a batch before its loads are issued (the table is padded to a multiple of 4)
        static_assert((RSLF_CHIP_PD - 1) * GB <= (NL > 0 ? NL : NA) && (NA + NL) % 2 == 0, "prologue inside the first tier gathered; pairs do not straddle tiers");
#pragma unroll
        for (int b = 0; b < PD; b++)
            xo[b] = *(const f4v*)(otab + b * GB);
#pragma unroll
        for (int b = 0; b < PD - 1; b++)
#pragma unroll
            for (int j = 0; j < GB; j++) {
                taps(xo[b][j], rowb, e0[b][j], e1[b][j], tt[b][j], ok[b][j]);
                rowb += stride_b;
                if (may_pad(view_of(b * GB + j) + 1))
                    rowb = min(rowb, row_last);
            }
        auto batch = [&](auto gc) {
            constexpr int g = decltype(gc)::value;
            constexpr int cur = g % PD, nxt = (g + PD - 1) % PD;   // batch g is blended; batch g + PD - 1's loads are issued
            if (g + PD - 1 < NB) {
#pragma unroll
                for (int j = 0; j < GB; j++) {
                    const int i = (g + PD - 1) * GB + j;
                    if (i < NO) {
                        if (i == NL)
                            rowb = (unsigned)NV * stride_b;   // (the views fetched ahead on every pass are not gathered here)
                        if (i == NL + NA)
                            rowb = 0;
                        taps(xo[nxt][j], rowb, e0[nxt][j], e1[nxt][j], tt[nxt][j], ok[nxt][j]);
                        rowb += stride_b;
                        if (may_pad(view_of(i) + 1))
                            rowb = min(rowb, row_last);
                    }
                }
            }
            if (g + PD < NB)
                xo[cur] = *(const f4v*)(otab + (g + PD) * GB);     // batch g + PD's offsets take the place of batch g's
            float r[GB][C];
#pragma unroll
            for (int j = 0; j < GB; j++) {
                if (g * GB + j < NO) {
                    blend(e0[cur][j], e1[cur][j], tt[cur][j], ok[cur][j], r[j]);
                    if (BORDER)
                        card += ok[cur][j] ? 1 : 0;   // (a missing view reads an in-bounds address: counted here, taken off below)
                }
            }
            auto pad = [&](auto jc) {
                constexpr int j = decltype(jc)::value, i = g * GB + j;
                if constexpr (i < NO && may_pad(view_of(i)))
                    chip_pad_select<view_of(i)>(r[j], S);
            };
            if constexpr (PAD)
                ChipUnroll<0, GB>::run(pad);
#pragma unroll
            for (int j = 0; j < GB; j++) {
                const int i = g * GB + j;
#pragma unroll
                for (int c = 0; c < C; c++) {
                    if (i >= NO) {
                    } else if (i < NL) {
                        if ((j & 1) == 0)
                            park[((i >> 1) * C + c) * 64] = f2{r[j][c], r[j + 1][c]};   // pairs go out as 8-byte stores
                    } else if (i < NL + NA) {
                        // (below: the register number must be a constant expression)
                    } else if ((i - NA - NL) & 1) {
                        Rv[c][(i - NA - NL) >> 1].y = r[j][c];
                    } else {
                        Rv[c][(i - NA - NL) >> 1].x = r[j][c];
                    }
                }
            }
            auto to_agpr = [&](auto kc) {
                constexpr int k = decltype(kc)::value, j = k / C, c = k % C, i = g * GB + j;
                if constexpr (i >= NL && i < NL + NA)
                    agpr_put<(i - NL) * C + c>(r[j][c]);
            };
            ChipUnroll<0, GB * C>::run(to_agpr);
            // pin the batch: its values exist here, and the address state of the batches to come is opaque -- else hipcc
            // turns the unrolled gather into "all loads, then all blends" and parks the texels in scratch (k2_reg.hpp)
#pragma unroll
            for (int j = 0; j < GB; j += 2) {
                const int i = g * GB + j;
                if (i >= NA + NL && i < NO) {
#pragma unroll
                    for (int c = 0; c < C; c++)
                        asm volatile("" : "+v"(Rv[c][(i - NA - NL) >> 1]));
                }
            }
            asm volatile("" : "+s"(rowb), "+v"(card));
        };
        ChipUnroll<0, NB>::run(batch);
        // Three samples (views NV + NA .. + 2) have no place to stay -- registers, AGPRs and LDS are full to the word, and
        // the running best needs a corner of LDS -- so every pass fetches them again (L1 / L2 hits), but AHEAD: their
        // loads go out before the AGPR tier's 42 blocks and their blends follow it (the ragged tail below waits for each
        // of its loads instead).
        float rbar[C];
#pragma unroll
        for (int c = 0; c < C; c++)
            rbar[c] = ((const float*)((const char*)epi + centre_off))[c];   // core.hpp:577: R[s_hat] = E[s_hat][u] exactly (re-read: L1 / L2)
        float B = 0.0f;
        int ncard = card;
#pragma unroll 1
        for (int it = 0; it < a.k.n_iter; it++) {    // core.hpp:584-610
            float A[C] = {0.0f, 0.0f, 0.0f};
            B = 0.0f;
            const f2 m01 = {rbar[0], rbar[1]}, m2x = {rbar[2], rbar[2]};
            ChipPK pk;
            pk.P0 = pk.P1 = pk.P2 = pk.K = f2{0.0f, 0.0f};
            // VGPR tier
#pragma unroll
            for (int p = 0; p < NV / 2; p++)
                pk = chip_pair(Rv[0][p], Rv[1][p], Rv[2][p], m01, m2x, kq2, pk, A, B);
            float xe0[kChipAhead][C], xe1[kChipAhead][C], xtt[kChipAhead];
            bool xok[kChipAhead];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < kChipAhead; k++) {
                unsigned rb = (unsigned)(NV + NA + k) * stride_b;
                if (may_pad(NV + NA + k))
                    rb = min(rb, row_last);
                taps(otab[NO + k], rb, xe0[k], xe1[k], xtt[k], xok[k]);
            }
            __builtin_amdgcn_sched_barrier(0);
            // AGPR tier
            auto agpr_pair = [&](auto pc) { pk = chip_pair_agpr<decltype(pc)::value>(m01, m2x, kq2, pk, A, B); };
            ChipUnroll<0, NA / 2>::run(agpr_pair);
            // the three fetched ahead: a pair, and one with a sentinel for its partner (K = 0, P = 0 exactly)
            {
                float rx[kChipAhead][C];
#pragma unroll
                for (int k = 0; k < kChipAhead; k++) {
                    blend(xe0[k], xe1[k], xtt[k], xok[k], rx[k]);
                }
                auto pad_ahead = [&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    if constexpr (may_pad(NV + NA + k))
                        chip_pad_select<NV + NA + k>(rx[k], S);
                };
                if constexpr (PAD)
                    ChipUnroll<0, kChipAhead>::run(pad_ahead);
                pk = chip_pair(f2{rx[0][0], rx[1][0]}, f2{rx[0][1], rx[1][1]}, f2{rx[0][2], rx[1][2]}, m01, m2x, kq2, pk, A, B);
                pk = chip_pair(f2{rx[2][0], kSentinel}, f2{rx[2][1], kSentinel}, f2{rx[2][2], kSentinel}, m01, m2x, kq2, pk, A, B);
            }
            // LDS tier, fully unrolled (immediate offsets): the three ds_read_b64 of pair p + 1 are issued before pair p's
            // block, which then covers their latency
            f2 q[2][C];
            if constexpr (NL > 0) {
#pragma unroll
                for (int c = 0; c < C; c++)
                    q[0][c] = park[c * 64];
            }
#pragma unroll
            for (int p = 0; p < NL / 2; p++) {
                const int cur = p & 1, nxt = cur ^ 1;
                if (p + 1 < NL / 2) {
#pragma unroll
                    for (int c = 0; c < C; c++)
                        q[nxt][c] = park[((p + 1) * C + c) * 64];
                }
                // (left to itself hipcc sinks the reads to just before their use and waits for them there)
                __builtin_amdgcn_sched_barrier(0);
                pk = chip_pair(q[cur][0], q[cur][1], q[cur][2], m01, m2x, kq2, pk, A, B);
                __builtin_amdgcn_sched_barrier(0);
            }
            ncard = card;
            if (BORDER) {
#pragma unroll
                for (int k = 0; k < kChipAhead; k++)
                    ncard += xok[k] ? 1 : 0;
                if (PAD)
                    ncard -= T - S;   // the missing views' in-bounds stand-ins
            }
            // What is left (none at c5) is fetched again on every pass, two views at a time and one pair AHEAD: the loads of
            // views s + 2, s + 3 go out before the pair (s, s + 1) is blended and run through the same packed block as the
            // tiers.  One view at a time with each load waited for cost 2.6 % of a hypothesis per view and the streaming
            // kernel overtook this one at 215 views; a pipelined pair costs 2.0 % per view, the crossover is at 220, and
            // chip_takes leaves more views than that to the streaming kernel (profiles/r03_k2_variants.md section 7).
            constexpr int T0 = T;
            if (TAIL && S > T0) {   // wave-uniform
                float ae0[2][C], ae1[2][C], att[2], be0[2][C], be1[2][C], btt[2];
                bool aok[2], bok[2];
                auto issue = [&](int s, float (&e0)[2][C], float (&e1)[2][C], float (&tt)[2], bool (&ok)[2]) {
#pragma unroll
                    for (int k = 0; k < 2; k++) {
                        const int sk = min(s + k, S - 1);   // a missing partner repeats the last view (its result is dropped)
                        taps(otab[sk], (unsigned)sk * stride_b, e0[k], e1[k], tt[k], ok[k]);
                    }
                };
                auto consume = [&](int s, const float (&e0)[2][C], const float (&e1)[2][C], const float (&tt)[2], const bool (&ok)[2]) {
                    float r[2][C];
                    blend(e0[0], e1[0], tt[0], ok[0], r[0]);
                    blend(e0[1], e1[1], tt[1], ok[1], r[1]);
                    const bool partner = s + 1 < S;   // wave-uniform
#pragma unroll
                    for (int c = 0; c < C; c++)
                        r[1][c] = partner ? r[1][c] : kSentinel;
                    pk = chip_pair(f2{r[0][0], r[1][0]}, f2{r[0][1], r[1][1]}, f2{r[0][2], r[1][2]}, m01, m2x, kq2, pk, A, B);
                    if (BORDER)
                        ncard += (ok[0] ? 1 : 0) + ((partner && ok[1]) ? 1 : 0);
                };
                if constexpr (SHARED) {
                    // (two buffers; a third -- two pairs ahead -- does not fit the registers: 28 B/lane of scratch written per hypothesis)
                    issue(T0, ae0, ae1, att, aok);
#pragma unroll 1
                    for (int s = T0; s < S; s += 4) {
                        if (s + 2 < S)
                            issue(s + 2, be0, be1, btt, bok);
                        __builtin_amdgcn_sched_barrier(0);
                        consume(s, ae0, ae1, att, aok);
                        if (s + 2 < S) {
                            if (s + 4 < S)
                                issue(s + 4, ae0, ae1, att, aok);
                            __builtin_amdgcn_sched_barrier(0);
                            consume(s + 2, be0, be1, btt, bok);
                        }
                    }
                } else {
                    // (border and ragged tiles: seven registers per view in flight -- a pair at a time, nothing ahead)
#pragma unroll 1
                    for (int s = T0; s < S; s += 2) {
                        issue(s, ae0, ae1, att, aok);
                        consume(s, ae0, ae1, att, aok);
                    }
                }
            }
            chip_flush(pk, A, B);
#pragma unroll
            for (int c = 0; c < C; c++) {
                const float qd = (B != 0.0f) ? (A[c] / B) : 0.0f;   // core.cpp:50, OpenCV 3.x: /0 -> 0
                rbar[c] = (qd > 0.0f) ? qd : 0.0f;                  // core.hpp:609
            }
        }
        const float cardf = (float)ncard;
        float sc = (ncard != 0) ? (B / cardf) : 0.0f;   // core.hpp:616-620: the last pass's sum of K
        sc = (sc > 0.0f) ? sc : 0.0f;                   // core.hpp:622
        best.offer(sc, d, rbar);
    }
    }
}

// Which form a hypothesis takes is decided per HYPOTHESIS (as scan_stream_rows does): the shared-tap form needs a dense
// tile -- 63 consecutive pixels in lanes 0..62, lane 63 free to stand on the pixel after them -- sample lines that stay
// inside the row for every lane (two pixels of margin, which covers lane 63's extra pixel), and view offsets none of whose
// fractions is within an ulp of 1 (positions are offset + integer u: the lanes then all floor alike).  Runs of
// hypotheses of one kind go to one body call, in ascending order: first maximum wins (core.hpp:636-645).
template <bool TAIL, int NA, int NL, bool PAD>
__device__ __forceinline__ void scan_chip_rows(const ScanArgs& a, int v, int e0, int width, int n, int d0, int d1, Best<3>& result, float* otab)
{
    ChipBest best;
    best.init(otab + plan::chip_table_floats(a.vol.S, kChipNV + NA + NL + kChipAhead) + NL * 3 * 64);
    const int S = a.vol.S;
    ChipTile t;
    t.row = a.list + (long long)v * a.vol.U;
    t.e0 = e0;
    t.last = min(e0 + width, n) - 1;
    bool dense;
    {
        const int ln = chip_lane();
        const int u = t.pixel(ln);
        const bool active = ln < width && e0 + ln < n;   // scan_tile
        const int u0 = __builtin_amdgcn_readfirstlane(u);
        t.u0 = u0;
        // (every lane is compared, not just the ends: a short or gappy list can span 62 pixels too -- idle lanes shadow the last entry)
        const bool consecutive = __all(ln > 62 || u == u0 + ln);
        // (a row's last tile may hold a 64th entry, scan_tile: lane 63 is then a pixel of its own and cannot lend itself out)
        const bool lane63_free = !__any(ln == 63 && active);
        dense = a.tile_w == 63 && consecutive && lane63_free;
    }
    // (a scalar: as a bool hipcc carried it through the hypotheses as a 0 / 1 per lane, and with every vector register
    // spoken for that was a dword of scratch on some rungs)
    int gappy_s = __builtin_amdgcn_readfirstlane(dense ? 0 : 1);
    asm volatile("" : "+s"(gappy_s));   // (opaque from here on: seen through, it is the per-lane bool again)
    // (uniform FLOATS are made in the vector ALU and would be carried in vector registers: they are made per call, from
    // scalars the optimiser cannot see through)
    auto shared_form = [&](int d) -> bool {
        if (gappy_s)
            return false;
        float dmin_s = a.dmin, dmax_s = a.dmax, slope_s = a.k.slope;
        int dim_s = a.dim_d, reach_s = max(a.s_hat, S - 1 - a.s_hat), last_s = a.vol.U - 1;
        asm volatile("" : "+s"(dmin_s), "+s"(dmax_s), "+s"(slope_s), "+s"(dim_s), "+s"(reach_s), "+s"(last_s));
        const float max_ds = (float)reach_s, Um1 = (float)last_s;
        const float Dd = hypothesis(dmin_s, dmax_s - dmin_s, (float)(dim_s - 1), d);
        const float reach = max_ds * fabsf(Dd) * fabsf(slope_s) + 2.0f;
        const int ln = chip_lane();
        const float uf = (float)min(t.u0 + ln, t.u0 + 62);   // dense: lanes 0..62 hold u0 .. u0 + 62 (lane 63's extra pixel is within the margin)
        if (!__all((uf - reach >= 0.0f) && (uf + reach <= Um1)))
            return false;
        bool odd = false;
        for (int s = ln; s < S; s += 64) {
            float off = (float)(a.s_hat - s) * Dd;   // core.hpp:542,550 -- the very operations the body's table holds
            off = off * slope_s;                     // core.hpp:551
            odd |= __builtin_amdgcn_fractf(off) > a.stream_frac_max;
        }
        return !__any(odd);
    };
    int d = d0;
    while (d < d1) {
        const bool sh = shared_form(d);
        int e = d + 1;
        while (e < d1 && shared_form(e) == sh)
            e++;
        if (sh)
            scan_chip_body<true, TAIL, NA, NL, PAD>(a, v, t, d, e, best, otab);
        else
            scan_chip_body<false, TAIL, NA, NL, PAD>(a, v, t, d, e, best, otab);
        d = e;
    }
    best.finish(a, result);
}

// This wave's hypotheses: scan_chunk's contiguous slices in wave order, but differing by at most one hypothesis (the first
// dim_d % slices take the extra one).  scan_chunk's ceil-sized chunks leave the last waves of a tile idle when the count
// does not divide -- 120 hypotheses over 8 groups: 30 slices of 4 and two idle waves -- which costs the other kernels
// nothing (the SIMD's other waves take up the slots, profiles/r04_k2_variants.md section 8) and this one, a wave per SIMD, the
// idle SIMDs.
__device__ __forceinline__ void chip_chunk(const ScanArgs& a, int group, int wave, int& d0, int& d1)
{
    const int slices = kScanWaves * a.groups, slice = group * kScanWaves + wave;
    const int q = a.dim_d / slices, r = a.dim_d - q * slices;
    d0 = slice * q + min(slice, r);
    d1 = d0 + q + (slice < r ? 1 : 0);
}

template <bool TAIL, int NA, int NL, bool PAD>
__global__ __launch_bounds__(64 * kScanWaves) __attribute__((amdgpu_waves_per_eu(1, 1))) void k2_scan_chip(ScanArgs a)
{
    constexpr int C = 3;
    extern __shared__ __attribute__((aligned(16))) float s_chip_lds[];   // [kScanWaves][stream_wave_floats]
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (a scalar: threadIdx.x itself is not kept)
    float* otab = s_chip_lds + (size_t)wave * a.stream_wave_floats;
    constexpr bool kEpiDyn = true;   // the waves' results for the epilogue go to the head of their own regions
    float* const epi_lds = otab;
    const int epi_stride = a.stream_wave_floats;
    // RSLF_SCAN_ROW_TILE, with the tile's per-lane state (pixel, active) derived AGAIN after the scan instead of carried
    // through it (ChipTile): `lb` is made opaque in between, so the second derivation is not merged with the first
    Best<C> best;
    int v, d0, d1, e0, width, n;
    int lb = RSLF_XCD_ROW_INTERLEAVE ? xcd_logical_block_rows(blockIdx.x, a.tiles_per_row * a.groups) : xcd_logical_block(blockIdx.x, a.per_xcd);
    if (!scan_tile_span(a, lb, v, e0, width, n))
        return;
    chip_chunk(a, lb % a.groups, wave, d0, d1);
    scan_chip_rows<TAIL, NA, NL, PAD>(a, v, e0, width, n, d0, d1, best, otab);
    asm volatile("" : "+s"(lb));
    int u;
    bool active;
    const int lane = chip_lane();
    scan_tile(a, lb, v, u, active, lane);
    scan_epilogue<C, kEpiDyn>(a, lb, v, u, active, best, epi_lds, epi_stride, lane, wave);
}

// ---- host side: one launcher per translation unit of instantiations (rslf_chip_a/b/c.hip) -------------------------------
template <bool TAIL, int NA, int NL, bool PAD>
int launch_chip_rung(const ScanArgs& a, dim3 grid, size_t lds_bytes, hipStream_t stream)
{
    // (more than the 64 KiB a kernel gets without asking)
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k2_scan_chip<TAIL, NA, NL, PAD>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)kChipLdsBytes));
    hipLaunchKernelGGL((k2_scan_chip<TAIL, NA, NL, PAD>), grid, dim3(64 * kScanWaves), lds_bytes, stream, a);
    return RSLF_OK;
}
// the padded form of one of this unit's rungs; RSLF_ERR_UNSUPPORTED = not one of this unit's
#define RSLF_CHIP_PART_CASE_(NA, NL) \
    if (na == NA && nl == NL)        \
        return launch_chip_rung<false, NA, NL, true>(a, grid, lds_bytes, stream);
#define RSLF_CHIP_PART_LAUNCHER(NAME, LIST)                                                                  \
    int NAME(int na, int nl, const ScanArgs& a, dim3 grid, size_t lds_bytes, hipStream_t stream)             \
    {                                                                                                        \
        LIST(RSLF_CHIP_PART_CASE_)                                                                           \
        return RSLF_ERR_UNSUPPORTED;                                                                         \
    }
int launch_chip_part_a(int na, int nl, const ScanArgs& a, dim3 grid, size_t lds_bytes, hipStream_t stream);
int launch_chip_part_b(int na, int nl, const ScanArgs& a, dim3 grid, size_t lds_bytes, hipStream_t stream);
int launch_chip_part_c(int na, int nl, const ScanArgs& a, dim3 grid, size_t lds_bytes, hipStream_t stream);

}  // namespace rslf
