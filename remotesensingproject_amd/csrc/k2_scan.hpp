// K2: the (pixel, hypothesis) EPI-slope scan.
//
// rslf::compute_1D_depth_epi (include/rslf_depth_computation_core.hpp:480-661)
// for every scanline (core.hpp:799-854), with Interpolation1DLinear
// (include/rslf_interpolation.hpp:155-193) and BandwidthKernel
// (src/rslf_kernels.cpp:16-54) inlined.
//
// Variants: k2_scan_reg (samples held in registers; scalar, or packed fp32 at one wave per SIMD),
// k2_scan_stream (samples re-gathered every pass), k2_scan_generic (any input, nearest-neighbour modes),
// k2_kernel_column (the optional K output); hypothesis groups merge their records in the epilogue (combine_tile).
//
// Work mapping (every variant)
//   workgroup = one tile: 64 consecutive entries of one scanline's confident-pixel list
//               (sparse launches: of ONE list over all scanlines, and `groups` workgroups per tile)
//   wavefront = one quarter of the tile's (or the group's) hypothesis range
//   lane      = one pixel u; it walks its wave's hypotheses itself
// A wave's 64 gathers per (s, d) are 64 neighbouring floats of one EPI row
// (coalesced along u), and the sum over s is the reference's sequential float
// sum.  The argmax / mean over d stays in the lane within a wave; the four
// waves' partial results meet once in LDS (first maximum wins in hypothesis
// order, as cv::minMaxLoc does).  Putting the four waves on ONE tile rather
// than on four tiles quarters the number of EPIs a CU has in flight, so an
// XCD's working set (~3 EPIs) stays inside its 4 MiB L2; logical workgroups are
// dealt to XCDs in contiguous scanline ranges (rslf_device.hpp).
//
// The kernel is FP32-VALU bound: per (pixel, hypothesis, view,
// mean-shift pass) the register variant issues 7 vector instructions and no
// memory instruction.
#pragma once

#include <type_traits>

#include "rslf_device.hpp"

namespace rslf {

constexpr int kScanWaves = 4;   // waves per workgroup = hypothesis chunks per tile
#ifndef RSLF_XCD_ROW_INTERLEAVE
#define RSLF_XCD_ROW_INTERLEAVE 1   // row-tile launches: scanlines dealt to the XCDs in turn (rslf_device.hpp); 0: contiguous eighths
#endif

struct ScanArgs {
    VolView vol;
    const int* list;        // [V][U] ascending confident u per scanline
    const int* count;       // [V]
    const float* dmin_vu;   // nullable => dmin/dmax scalars
    const float* dmax_vu;
    float dmin, dmax;
    int dim_d, s_hat;
    ScanConsts k;
    float* Ce;              // [V][U] in/out
    uint8_t* Ce_mask;       // [V][U] in/out
    float* Cd;              // [V][U]
    float* depth;           // [V][U]
    float* rbar;            // [V][U][C]
    int32_t* idx;           // nullable
    float* score;           // nullable
    int tile_w;             // list entries per row tile: 64, or 63 where lane 63 only carries its neighbour's right tap
    int tiles_per_row;      // ceil(U / tile_w)
    int logical_blocks;     // V * tiles_per_row * groups
    int per_xcd;            // ceil(logical_blocks / 8)
    // Hypothesis groups: `groups` workgroups share one tile, each taking a contiguous slice of the
    // hypothesis range (its 4 waves a quarter of the slice each).  groups == 1: the workgroup writes the
    // pixel itself.  groups > 1 (launches expected to be sparse, where the launch lasts as long as ONE
    // wave's walk over its hypotheses): each workgroup leaves a partial record and the last one to finish merges
    // them in hypothesis order (scan_epilogue / combine_tile).
    int groups;
    struct Partial* partial;   // records of grouped launches: [tile * groups + group][word 0..7][lane 0..63] (word-major)
    int* ticket;               // [tile], zero between launches: the group that draws the last ticket merges the tile
    int v0;                    // row tiles: first scanline of this launch (grouped dense launches go by row blocks)
    // Packed tiles (sparse launches): `list` is ONE list of pixel indices v*U + u over all scanlines,
    // *packed_n long, and a tile is 64 consecutive entries of it -- lanes of a wave then sit on different
    // scanlines.  The grid is fixed and every workgroup strides over the (tile, group) items, because only
    // the device knows how many there are.
    int packed;
    const int* packed_n;
    int packed_adapt;          // 1: the kernel settles the group count from the list length (packed_groups)
    // streaming kernel: samples per lane parked in LDS behind the register-resident prefix (a multiple of the
    // gather batch; 0 = none), sized by the host to what the dynamic-LDS limit leaves after the offset table
    int stream_park;
    // streaming kernel: floats of dynamic LDS per wave = [view offsets, S rounded up to 4][parked samples]
    int stream_wave_floats;
    // streaming kernel, shared-tap tail: a view offset whose fraction is above this could round a position up to
    // the next integer in some lanes and not in others (1 - ulp of the largest position, host-computed)
    float stream_frac_max;
    // pixel-per-wave kernel (k2_scan_reg_px, sparse launches): waves that share one pixel's hypotheses, 1 / 2 / 4; 0 = not that kernel
    int px_waves;
    // Row split of a packed list (sparse visits of stream-class volumes).  A row's entries are contiguous in the packed
    // list (compact_row_packed), rowbase[v] = where they start.  Rows that hold at least `row_min` pixels are scanned as
    // ROW tiles straight from the packed list -- 64 lanes on one scanline, a scalar EPI base, the tile's waves sharing
    // their L1 lines: 11.6 ms for 180 k pixels of a banded 28 % mask where a pixel per wave takes 15.1
    // (tools/probe_density.py) -- and the pixel-per-wave launch of the same list leaves those rows out (it wins below
    // ~8 % density).  rowbase == nullptr: per-row lists at list[v * U] (dense launches); row_min == 0: no split.
    const int* rowbase;
    int row_min;
};

// Laid out WORD-major in memory ([item][word][lane], record_word): a wave's store of one word is then 256 contiguous bytes.
// Lane-major 32-byte structs made every dword store touch sixteen 128-byte lines in part, and the write-through (sc1) stores
// left the L2 as sector writes: 274 MB of HBM writes per c2 launch for 33.5 MB of records (profiles/r03_c2_n1_pmc.json).
struct Partial {   // one lane's merged result over one group's hypotheses
    float score, D;
    int d;
    float rbar[3];
    double sum;
};

// One lane's running result over the hypotheses it has scored (core.hpp:630-644).
template <int C>
struct Best {
    float score;   // -1 before any hypothesis: scores are >= 0, so the first one always takes the lead
    int d;
    float D;
    float rbar[C];
    double sum;    // of all scores, for cv::mean (core.hpp:641)
    __device__ __forceinline__ void init()
    {
        score = -1.0f;
        d = 0;
        D = 0.0f;
#pragma unroll
        for (int c = 0; c < C; c++)
            rbar[c] = 0.0f;
        sum = 0.0;
    }
    __device__ __forceinline__ void offer(float sc, int dd, float Dd, const float (&rb)[C])
    {
        sum += (double)sc;
        if (sc > score) {   // strict: first maximum wins (cv::minMaxLoc, core.hpp:634)
            score = sc;
            d = dd;
            D = Dd;
#pragma unroll
            for (int c = 0; c < C; c++)
                rbar[c] = rb[c];
        }
    }
};

// The same running result kept in LDS, [field][lane] in the wave's own block: for the register kernels that run one wave
// per SIMD more than their registers allow -- the six (C = 1) or eight values of Best are touched once per hypothesis
// (a few thousand instructions apart) and were what hipcc spilled to scratch there: 330 MB of HBM writes per c2 launch
// against 41 MB of algorithmic traffic (profiles/r02_c2_n1_pmc.json).  Same operations, same order, same bits.
template <int C>
struct BestLds {
    float* blk;   // this wave's block: score[64], d[64], D[64], rbar[C][64], sum (double)[64]
    static constexpr int kFloats = (3 + C + 2) * 64;
    __device__ __forceinline__ explicit BestLds(float* wave_block) : blk(wave_block + (threadIdx.x & 63)) {}
    __device__ __forceinline__ double* sum_ptr() const
    {
        return reinterpret_cast<double*>(blk - (threadIdx.x & 63) + (3 + C) * 64) + (threadIdx.x & 63);
    }
    __device__ __forceinline__ void init()
    {
        blk[0] = -1.0f;
        reinterpret_cast<int*>(blk)[64] = 0;
        blk[128] = 0.0f;
#pragma unroll
        for (int c = 0; c < C; c++)
            blk[(3 + c) * 64] = 0.0f;
        *sum_ptr() = 0.0;
    }
    __device__ __forceinline__ void offer(float sc, int dd, float Dd, const float (&rb)[C])
    {
        double* sp = sum_ptr();
        *sp = *sp + (double)sc;
        if (sc > blk[0]) {   // strict: first maximum wins (cv::minMaxLoc, core.hpp:634)
            blk[0] = sc;
            reinterpret_cast<int*>(blk)[64] = dd;
            blk[128] = Dd;
#pragma unroll
            for (int c = 0; c < C; c++)
                blk[(3 + c) * 64] = rb[c];
        }
    }
    __device__ __forceinline__ void load(Best<C>& b) const
    {
        b.score = blk[0];
        b.d = reinterpret_cast<const int*>(blk)[64];
        b.D = blk[128];
#pragma unroll
        for (int c = 0; c < C; c++)
            b.rbar[c] = blk[(3 + c) * 64];
        b.sum = *sum_ptr();
    }
};

// Which tile does this workgroup own, which pixel this lane?  Block-uniform result
// (every wave of the block takes the same branch, so the later barrier is safe).
// `lb` = tile * groups + group.
// (the uniform part: scanline, the tile's first entry, its width, the scanline's count)
__device__ __forceinline__ bool scan_tile_span(const ScanArgs& a, int lb, int& v, int& e0, int& width, int& n)
{
    if (lb >= a.logical_blocks)
        return false;
    const int tile = lb / a.groups;
    const int vr = tile / a.tiles_per_row;
    const int j = tile - vr * a.tiles_per_row;
    v = vr + a.v0;
    n = a.count[v];
    if (j * a.tile_w >= n || n < a.row_min)   // (row_min: a row with few pixels belongs to the pixel-per-wave launch)
        return false;
    // 63-entry tiles (streaming kernel, lane 63 left to the shared taps): the row's LAST tile takes up to 64 entries, so
    // that a row of 63 k + 1 pixels (4096 = 65 * 63 + 1) does not end in a tile of one
    width = a.tile_w;
    if (a.tile_w == 63) {
        const int T = max(1, (n + 61) / 63);      // tiles of this row: 63 entries each, the last one 1..64
        if (j >= T)
            return false;
        if (j == T - 1)
            width = n - 63 * j;
    }
    e0 = j * a.tile_w;
    return true;
}
// (`lane`, and `wave` below: a kernel that cannot afford to keep threadIdx.x alive hands in its own -- k2_chip.hpp)
__device__ __forceinline__ bool scan_tile(const ScanArgs& a, int lb, int& v, int& u, bool& active, int lane = threadIdx.x & 63)
{
    int e0, width, n;
    if (!scan_tile_span(a, lb, v, e0, width, n))
        return false;
    const int e = e0 + lane;
    active = lane < width && e < n;
    // idle lanes shadow the tile's last pixel so their addresses stay valid
    const int idx = active ? e : min(e0 + width, n) - 1;
    if (a.rowbase)   // the row's stretch of a packed list: entries are pixel indices v * U + u
        u = a.list[a.rowbase[v] + idx] - v * a.vol.U;
    else
        u = a.list[(long long)v * a.vol.U + idx];
    return true;
}

// Packed tiles: entry e of the flat list is pixel index v*U + u; `n` = *packed_n > tile * 64.
__device__ __forceinline__ void scan_tile_packed(const ScanArgs& a, int item, int n, int& v, int& u, bool& active)
{
    const int lane = threadIdx.x & 63;
    const int e = (item / a.groups) * 64 + lane;
    active = e < n;
    const unsigned o = (unsigned)a.list[active ? e : n - 1];
    v = (int)(o / (unsigned)a.vol.U);
    u = (int)(o - (unsigned)v * (unsigned)a.vol.U);
}

// This wave's hypotheses [d0, d1): contiguous quarters, so "first maximum" = lowest wave first.
__device__ __forceinline__ void scan_chunk(const ScanArgs& a, int group, int& d0, int& d1)
{
    // wave-uniform by construction; readfirstlane lets the compiler keep it in an SGPR
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int slices = kScanWaves * a.groups;
    const int chunk = (a.dim_d + slices - 1) / slices;
    d0 = min((group * kScanWaves + wave) * chunk, a.dim_d);
    d1 = min(d0 + chunk, a.dim_d);
}

// core.hpp:545-548: D[d] = dmin + d * (dmax - dmin) / (dim_d - 1)
__device__ __forceinline__ float hypothesis(float dmin, float range, float denom, int d)
{
    const float num = (float)d * range;
    const float quo = num / denom;
    return dmin + quo;
}

// core.hpp:636-657 for one pixel once every hypothesis is scored.
template <int C>
__device__ __forceinline__ void write_pixel(const ScanArgs& a, long long o, float best, int best_d, float best_D,
                                            const float (&best_rbar)[C], double sum)
{
    if ((double)best > (double)a.k.raw_thr) {   // core.hpp:636
        a.depth[o] = best_D;
        const double mean = sum / (double)a.dim_d;
        a.Cd[o] = (float)((double)a.Ce[o] * fabs((double)best - mean));   // core.hpp:641
#pragma unroll
        for (int c = 0; c < C; c++)
            a.rbar[o * C + c] = best_rbar[c];
        if (a.idx)
            a.idx[o] = best_d;
        if (a.score)
            a.score[o] = best;
    } else {   // core.hpp:653-657
        a.Ce[o] = 0.0f;
        a.Ce_mask[o] = 0;
    }
}

// Pixel-per-wave launches (k2_scan_reg_px, k2_scan_stream_px): the lanes of `wpp` waves hold one pixel's hypotheses, each
// lane its own running result.  Across the lanes the best score wins and the LOWEST hypothesis among equal scores (first
// maximum, cv::minMaxLoc, core.hpp:634); the winner's disparity and rbar come from the one lane that scored it; the score
// sum is a double (cv::mean, core.hpp:641 -- C_d is held to 1e-5; sums of <= 4096 floats in [0, 1] are exact in a double
// unless a score is below 2^-21, so in practice the same bits in any order).  Then across the waves that share the pixel
// through `rec` / `wsum` (LDS, one row per wave of the workgroup), and the pixel is written by the first of them.
constexpr int kPxRecFloats = 3 + 3;   // score, hypothesis (bits), disparity, rbar[<= 3]
// Does this pixel-per-wave launch scan the pixels of scanline v?  Not where the row split hands the row to the row-tile launch.
__device__ __forceinline__ bool scan_px_owns(const ScanArgs& a, int v)
{
    return a.row_min == 0 || a.count[v] < a.row_min;
}

template <int C>
__device__ __forceinline__ void scan_px_finish(const ScanArgs& a, unsigned o, bool have, const Best<C>& best, int wave, int lane, int wpp,
                                               float (*rec)[kPxRecFloats], double* wsum)
{
    const int sub = wave % wpp;
    float bs = best.score;                        // -1 where a lane had no hypothesis
    int bd = bs < 0.0f ? 0x7fffffff : best.d;
    double sum = best.sum;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const float os = __shfl_xor(bs, m);
        const int od = __shfl_xor(bd, m);
        const bool take = os > bs || (os == bs && od < bd);
        bs = take ? os : bs;
        bd = take ? od : bd;
        sum += __shfl_xor(sum, m);
    }
    const int owner = __ffsll((unsigned long long)__ballot(best.score == bs && best.d == bd)) - 1;   // exactly one lane scored bd
    float bD = __shfl(best.D, owner);
    float br[C];
#pragma unroll
    for (int c = 0; c < C; c++)
        br[c] = __shfl(best.rbar[c], owner);
    if (wpp > 1) {
        if (lane == 0) {
            rec[wave][0] = bs;
            rec[wave][1] = __int_as_float(bd);
            rec[wave][2] = bD;
#pragma unroll
            for (int c = 0; c < C; c++)
                rec[wave][3 + c] = br[c];
            wsum[wave] = sum;
        }
        __syncthreads();
        if (sub == 0) {
            for (int w = wave + 1; w < wave + wpp; w++) {
                const float os = rec[w][0];
                const int od = __float_as_int(rec[w][1]);
                sum += wsum[w];
                if (os > bs || (os == bs && od < bd)) {
                    bs = os;
                    bd = od;
                    bD = rec[w][2];
#pragma unroll
                    for (int c = 0; c < C; c++)
                        br[c] = rec[w][3 + c];
                }
            }
        }
        __syncthreads();   // the next item's records may overwrite these
    }
    if (have && sub == 0 && lane == 0)
        write_pixel<C>(a, (long long)o, bs, bd, bD, br, sum);
}

// Word accesses that are coherent at agent scope by themselves (sc1: to / from the memory side), for data handed from one
// workgroup to another that may run on a different XCD -- no cache-wide write-back or invalidate.
// The hand-off built on them (scan_epilogue) leans on gfx9 behaviour, not on the language memory model: stores count in
// vmcnt (on gfx10+ they are tracked by vscnt, and the s_waitcnt below would no longer wait for them) and sc1 accesses go
// past the per-XCD L2.  Measured valid on gfx950 (MI355X_MICROARCH.md, "Valid forms": sc1 stores, every storing wave's
// vmcnt(0), one lane's agent-scope atomic add, the last adder reads with sc1 loads); any other target must not build it.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "k2_scan.hpp: the cross-workgroup record hand-off is written for gfx950 (gfx9 vmcnt / sc1 semantics); use agent-scope release/acquire fences on other targets"
#endif
__device__ __forceinline__ void store_coherent(unsigned* p, unsigned v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned load_coherent(const unsigned* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

constexpr int kRecordWords = 8;

template <int C>
__device__ __forceinline__ void combine_tile(const ScanArgs& a, int tile, int v, int u, int lane);

// Merge the waves' partial results in hypothesis order (first maximum wins, cv::minMaxLoc) and either
// write the pixel (groups == 1) or leave this group's record -- the last group to finish merges the records.
// Where the waves leave their results for wave 0: one block of 64 doubles (score sums) + (3 + C) x 64 floats per wave,
// 2 KB for RGB.  DYN = false: a static array.  DYN = true (the streaming kernel, whose two workgroups per CU want every
// byte of the 160 KiB for parked samples): the head of the wave's own dynamic region -- the offset table, dead once the
// wave's hypotheses are done -- `wave_lds`, the regions `wave_stride` floats apart.
template <int C>
struct EpilogueBlock {
    static constexpr int kDoubles = 64 + (3 + C) * 32;
    double* sum;
    float *score, *D, *rbar;
    int* d;
    __device__ __forceinline__ explicit EpilogueBlock(double* base)
        : sum(base), score(reinterpret_cast<float*>(base + 64)), D(score + 64), rbar(D + 64), d(reinterpret_cast<int*>(rbar + C * 64)) {}
};

template <int C, bool DYN = false>
__device__ __forceinline__ void scan_epilogue(const ScanArgs& a, int lb, int v, int u, bool active, const Best<C>& mine,
                                              float* wave_lds = nullptr, int wave_stride = 0, int lane = threadIdx.x & 63,
                                              int wave = threadIdx.x >> 6)
{
    __shared__ double s_static[DYN ? 1 : kScanWaves][DYN ? 1 : EpilogueBlock<C>::kDoubles];
    auto block_of = [&](int w) {
        return EpilogueBlock<C>(DYN ? reinterpret_cast<double*>(wave_lds + (long long)(w - wave) * wave_stride) : s_static[DYN ? 0 : w]);
    };
    {
        const EpilogueBlock<C> me = block_of(wave);
        me.score[lane] = mine.score;
        me.D[lane] = mine.D;
        me.d[lane] = mine.d;
        me.sum[lane] = mine.sum;
#pragma unroll
        for (int c = 0; c < C; c++)
            me.rbar[c * 64 + lane] = mine.rbar[c];
    }
    __syncthreads();
    if (wave != 0) {
        __syncthreads();   // wave 0 has read the arrays: the next item's epilogue may overwrite them
        return;            // ... and these waves go on to it while wave 0 hands over this item's record
    }

    float best = mine.score, best_D = mine.D;
    int best_d = mine.d;
    float best_rbar[C];
#pragma unroll
    for (int c = 0; c < C; c++)
        best_rbar[c] = mine.rbar[c];
    double sum = mine.sum;
#pragma unroll
    for (int w = 1; w < kScanWaves; w++) {
        const EpilogueBlock<C> o = block_of(w);
        const float sc = o.score[lane];
        sum += o.sum[lane];
        // first maximum in hypothesis order: the lower wave holds the lower hypotheses (the index settles a tie
        // all the same).  A wave that scored nothing holds -1 and never wins
        if (sc > best || (sc == best && o.d[lane] < best_d)) {
            best = sc;
            best_d = o.d[lane];
            best_D = o.D[lane];
#pragma unroll
            for (int c = 0; c < C; c++)
                best_rbar[c] = o.rbar[c * 64 + lane];
        }
    }
    __syncthreads();       // (pairs with the other waves' second barrier)
    if (a.groups > 1) {
        // The group that finishes LAST merges the tile's records (no combine launch).  The groups of a tile may sit on
        // different XCDs, whose L2s are not coherent.  A release / acquire fence pair at agent scope would be the textbook
        // hand-off, but on gfx950 it is `buffer_wbl2 sc1` + `buffer_inv sc1` -- a write-back and an invalidate of the
        // XCD's whole L2, i.e. of the EPI lines every other workgroup there is gathering from (measured: +12 % on the
        // packed scans of a 100-view fine-to-coarse run).  So the records alone are made coherent, access by access:
        // written with agent-scope atomic stores (sc1: through to the memory side), the wave waits for them to complete
        // (s_waitcnt vmcnt(0)) before lane 0 draws the ticket, and the group that draws the last one reads the records
        // with agent-scope atomic loads (sc1: past the caches), which its branch on the ticket orders after the draw.
        unsigned* w = reinterpret_cast<unsigned*>(a.partial) + ((long long)lb * kRecordWords * 64 + lane);   // word k at w[k * 64]
        static_assert(sizeof(Partial) == 4 * kRecordWords, "eight words per record");
        const unsigned long long sb = (unsigned long long)__double_as_longlong(sum);
        store_coherent(w + 0 * 64, __float_as_uint(best));
        store_coherent(w + 1 * 64, __float_as_uint(best_D));
        store_coherent(w + 2 * 64, (unsigned)best_d);
#pragma unroll
        for (int c = 0; c < C; c++)   // (one-channel records leave words 4 and 5 unwritten: nobody reads them)
            store_coherent(w + (3 + c) * 64, __float_as_uint(best_rbar[c]));
        store_coherent(w + 6 * 64, (unsigned)sb);
        store_coherent(w + 7 * 64, (unsigned)(sb >> 32));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (a workgroup-scope release fence emits nothing here)
        const int tile = lb / a.groups;
        int drawn = 0;
        if (lane == 0)
            drawn = atomicAdd(&a.ticket[tile], 1);
        drawn = __builtin_amdgcn_readfirstlane(drawn);
        if (drawn != a.groups - 1)
            return;
        asm volatile("" ::: "memory");
        if (active)
            combine_tile<C>(a, tile, v, u, lane);
        if (lane == 0)
            a.ticket[tile] = 0;   // clean for the next launch (kernel boundary orders it)
        return;
    }
    if (active)
        write_pixel<C>(a, (long long)v * a.vol.U + u, best, best_d, best_D, best_rbar, sum);
}

// groups > 1: the wave that drew the tile's last ticket merges the groups' records in hypothesis order and writes the pixels.
template <int C>
__device__ __forceinline__ void combine_tile(const ScanArgs& a, int tile, int v, int u, int lane)
{
    const unsigned* pr = reinterpret_cast<const unsigned*>(a.partial) + ((long long)tile * a.groups * kRecordWords * 64 + lane);
    float best = -1.0f, best_D = 0.0f;
    int best_d = -1;
    float best_rbar[C];
#pragma unroll
    for (int c = 0; c < C; c++)
        best_rbar[c] = 0.0f;
    double sum = 0.0;
    // eight groups' records at a time, every word's load issued before the first is used: the loads go past the caches
    // (a round trip to memory each), and this wave is the last thing its tile waits for
    constexpr int GB = 8, W = kRecordWords;
    for (int g0 = 0; g0 < a.groups; g0 += GB) {
        unsigned w[GB][W];
#pragma unroll
        for (int j = 0; j < GB; j++) {
            const unsigned* q = pr + (long long)min(g0 + j, a.groups - 1) * 64 * W;
#pragma unroll
            for (int k = 0; k < W; k++)
                if (k < 3 + C || k >= 6)
                    w[j][k] = load_coherent(q + k * 64);
        }
#pragma unroll
        for (int j = 0; j < GB; j++) {
            const int g = g0 + j;
            if (g >= a.groups)
                break;
            const float q_score = __uint_as_float(w[j][0]);
            const double q_sum = __longlong_as_double((long long)((unsigned long long)w[j][6] | ((unsigned long long)w[j][7] << 32)));
            sum = g == 0 ? q_sum : sum + q_sum;
            if (g == 0 || q_score > best) {   // first maximum in hypothesis order: the lower group holds the lower hypotheses
                best = q_score;
                best_D = __uint_as_float(w[j][1]);
                best_d = (int)w[j][2];
#pragma unroll
                for (int c = 0; c < C; c++)
                    best_rbar[c] = __uint_as_float(w[j][3 + c]);
            }
        }
    }
    write_pixel<C>(a, (long long)v * a.vol.U + u, best, best_d, best_D, best_rbar, sum);
}

// ---------------------------------------------------------------------------
// Generic variant: any S, C in {1,3}, negative radiances allowed.  Nothing is
// kept between mean-shift passes: every pass re-gathers its samples from the
// slab (L1/L2 hits).  ~3x the instructions of the register variant.
// ---------------------------------------------------------------------------
// `Kcol` (nullable): where K(r - rbar) of the LAST pass goes, element s at Kcol[s * kstride] -- the optional
// K_r_m_rbar column of core.hpp:647-651, filled by k2_kernel_column for one hypothesis per pixel.
template <int C>
__device__ __forceinline__ void scan_generic_body(const ScanArgs& a, int v, int u, int d0, int d1, Best<C>& best,
                                                  float* __restrict__ Kcol = nullptr, long long kstride = 0)
{
    const VolView& vol = a.vol;
    const float* epi = vol.row(v, 0);
    const float uf = (float)u;
    const int Um1 = vol.U - 1;
    const long long o = (long long)v * vol.U + u;
    const float dmin = a.dmin_vu ? a.dmin_vu[o] : a.dmin;
    const float dmax = a.dmax_vu ? a.dmax_vu[o] : a.dmax;
    const float range = dmax - dmin;
    const float denom = (float)(a.dim_d - 1);

    for (int d = d0; d < d1; d++) {
        const float Dd = hypothesis(dmin, range, denom, d);
        float rbar[C];
#pragma unroll
        for (int c = 0; c < C; c++) {   // core.hpp:577: R[s_hat] is E[s_hat][u] exactly (position u + 0*d) ...
            rbar[c] = epi[(long long)a.s_hat * vol.stride_s + u * C + c];
            if (a.k.interp == 2 && u != 0)   // ... except as built (interp.hpp:118): bits(float(u)) is no column unless u = 0
                rbar[c] = NAN;
        }
        float B = 0.0f, card = 0.0f;
        for (int it = 0; it < a.k.n_iter; it++) {
            float A[C];
#pragma unroll
            for (int c = 0; c < C; c++)
                A[c] = 0.0f;
            B = 0.0f;
            card = 0.0f;
            for (int s = 0; s < vol.S; s++) {
                float x = (float)(a.s_hat - s) * Dd;   // I = S * D          core.hpp:550
                x = x * a.k.slope;                     // I *= slope_factor  core.hpp:551
                x = x + uf;                            // I += u             core.hpp:552
                const float fl = floorf(x);
                int i0 = (int)fl;                      // interp.hpp:179-181
                int i1 = (int)ceilf(x);
                float t = x - fl;
                bool valid = !(i0 < 0 || i1 > Um1);    // interp.hpp:182
                if (a.k.interp != 0) {
                    // Interpolation1DNearestNeighbour::interpolate_mat (interp.hpp:94-131): one tap.  Both taps
                    // below read E[r] with weights 1 and 0, and 1*E + 0*E == E exactly for every finite E.
                    int r;
                    if (a.k.interp == 2) {
                        r = __float_as_int(x);         // interp.hpp:118: the float index read through an int pointer
                        valid = r > -1 && r < vol.U;   // interp.hpp:122
                    } else {
                        r = (int)roundf(x);            // interp.hpp:121: std::round, halves away from zero
                        valid = fabsf(x) < 2.0e9f && r > -1 && r < vol.U;
                    }
                    i0 = i1 = r;
                    t = 0.0f;
                }
                const int j0 = min(max(i0, 0), Um1), j1 = min(max(i1, 0), Um1);
                const float* row = epi + (long long)s * vol.stride_s;
                const float omt = 1.0f - t;
                float K;
                float R0[C];
                if (C == 1) {
                    const float m0 = omt * row[j0];
                    const float m1 = t * row[j1];
                    float R = m0 + m1;                      // interp.hpp:184
                    R = valid ? R : NAN;                    // interp.hpp:189
                    R0[0] = (R > 0.0f) ? R : 0.0f;          // cv::max(R, 0), NaN -> 0   core.hpp:580
                    const float delta = R - rbar[0];
                    const float q = (a.k.k1 * delta) * delta;
                    K = kernel_weight(q);                   // max(1 - q, 0), NaN -> 0
                } else {
                    float q[C];
#pragma unroll
                    for (int c = 0; c < C; c++) {
                        const float m0 = omt * row[j0 * C + c];
                        const float m1 = t * row[j1 * C + c];
                        float R = m0 + m1;
                        R = valid ? R : NAN;
                        R0[c] = (R > 0.0f) ? R : 0.0f;
                        const float delta = R - rbar[c];
                        q[c] = (a.k.inv_h2 * delta) * delta;
                    }
                    float qs = q[0] + q[C > 2 ? 2 : 0];     // OpenCV 3.x reduceC_: (q0 + q2) + q1
                    qs = qs + q[C > 1 ? 1 : 0];
                    K = kernel_weight(qs);
                }
#pragma unroll
                for (int c = 0; c < C; c++) {
                    const float pr = R0[c] * K;
                    A[c] = A[c] + pr;
                }
                B = B + K;
                card = card + (valid ? 1.0f : 0.0f);
                if (Kcol && it == a.k.n_iter - 1)
                    Kcol[(long long)s * kstride] = K;              // core.hpp:650
            }
#pragma unroll
            for (int c = 0; c < C; c++) {
                const float q = (B != 0.0f) ? (A[c] / B) : 0.0f;   // OpenCV 3.x divide: /0 -> 0
                rbar[c] = (q > 0.0f) ? q : 0.0f;
            }
        }
        float sc = (card != 0.0f) ? (B / card) : 0.0f;   // core.hpp:620
        sc = (sc > 0.0f) ? sc : 0.0f;
        best.offer(sc, d, Dd, rbar);
    }
}

// Packed launches: only the device knows how many pixels the list holds.  Few tiles: every hypothesis group the host
// allowed (the launch lasts as long as one wave's walk over its hypotheses).  Many tiles: the groups only multiply the
// records to merge -- 64 000 pixels of the c2 shape take 530 us with one group and 920 us with sixteen
// (tools/probe_sparse.py) -- so the groups are halved until the items are about two per workgroup of the fixed grid.
constexpr int kPackedItemTarget = 2048;
// Not for the streaming kernel: its groups are also what keeps the tiles an XCD works on at any one time few enough for
// their EPI lines to stay in its L2 (a 100-view RGB fine-to-coarse run: +9 % with the groups cut) -- `adapt` is 0 there.
__device__ __forceinline__ int packed_groups(int groups, int tiles, int adapt)
{
    while (adapt && groups > 1 && tiles * groups > kPackedItemTarget)
        groups >>= 1;
    return groups;
}

// The launch shapes every variant shares.  Row tiles: one (tile, group) item per workgroup, dealt to XCDs
// in scanline order.  Packed tiles: a fixed grid strides over the items the device-side count yields;
// every wave of a workgroup makes the same trips, and the epilogue's second barrier separates one item's
// merge in LDS from the next item's.
#define RSLF_SCAN_PACKED_LOOP(PACKED_CALL) RSLF_SCAN_PACKED_LOOP_(scan_chunk, PACKED_CALL)
#define RSLF_SCAN_ROW_TILE(ROWS_CALL) RSLF_SCAN_ROW_TILE_(scan_chunk, ROWS_CALL)
#define RSLF_SCAN_PACKED_LOOP_(CHUNK, PACKED_CALL)                                      \
    {                                                                                   \
        const int n = *a.packed_n;                                                      \
        ScanArgs a_items = a;                                                           \
        a_items.groups = packed_groups(a.groups, (n + 63) >> 6, a.packed_adapt);        \
        {                                                                               \
            const ScanArgs& a = a_items;   /* shadows the kernel argument */            \
            Best<C> best;                                                               \
            int v, u, d0, d1;                                                           \
            bool active;                                                                \
            const int items = ((n + 63) >> 6) * a.groups;                               \
            for (int item = blockIdx.x; item < items; item += gridDim.x) {              \
                scan_tile_packed(a, item, n, v, u, active);                             \
                CHUNK(a, item % a.groups, d0, d1);                                      \
                best.init();                                                            \
                PACKED_CALL;                                                            \
                scan_epilogue<C, kEpiDyn>(a, item, v, u, active, best, epi_lds, epi_stride); \
            }                                                                           \
        }                                                                               \
    }
#define RSLF_SCAN_ROW_TILE_(CHUNK, ROWS_CALL)                                           \
    {                                                                                   \
        Best<C> best;                                                                   \
        int v, u, d0, d1;                                                               \
        bool active;                                                                    \
        const int lb = RSLF_XCD_ROW_INTERLEAVE ? xcd_logical_block_rows(blockIdx.x, a.tiles_per_row * a.groups) \
                                               : xcd_logical_block(blockIdx.x, a.per_xcd); \
        if (!scan_tile(a, lb, v, u, active))                                            \
            return;                                                                     \
        CHUNK(a, lb % a.groups, d0, d1);                                                \
        best.init();                                                                    \
        ROWS_CALL;                                                                      \
        scan_epilogue<C, kEpiDyn>(a, lb, v, u, active, best, epi_lds, epi_stride);      \
    }
#define RSLF_SCAN_KERNEL_BODY_(CHUNK, ROWS_CALL, PACKED_CALL)                           \
    if (a.packed) {                                                                     \
        RSLF_SCAN_PACKED_LOOP_(CHUNK, PACKED_CALL)                                      \
        return;                                                                         \
    }                                                                                   \
    RSLF_SCAN_ROW_TILE_(CHUNK, ROWS_CALL)
#define RSLF_SCAN_KERNEL_BODY(ROWS_CALL, PACKED_CALL) RSLF_SCAN_KERNEL_BODY_(scan_chunk, ROWS_CALL, PACKED_CALL)

template <int C>
__global__ __launch_bounds__(64 * kScanWaves) void k2_scan_generic(ScanArgs a)
{
    constexpr bool kEpiDyn = false;
    float* const epi_lds = nullptr;
    const int epi_stride = 0;
    RSLF_SCAN_KERNEL_BODY((scan_generic_body<C>(a, v, u, d0, d1, best)), (scan_generic_body<C>(a, v, u, d0, d1, best)))
}

// The optional last output of compute_1D_depth_epi (core.hpp:266, :647-651): for every pixel that received a
// disparity, the column K(r - rbar)[:, d*] of its winning hypothesis, S values.  One thread per pixel re-runs
// that ONE hypothesis with the generic arithmetic (the same operations as every scan variant, so the same
// bits) and stores the last pass's K; 1/D of the scan's work.  Pixels without a disparity are left untouched,
// as in the reference.  K_vsu is [V][S][U].
template <int C>
__global__ __launch_bounds__(256) void k2_kernel_column(ScanArgs a, const int32_t* __restrict__ idx_vu, float* __restrict__ K_vsu)
{
    const int v = blockIdx.y;
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= a.vol.U)
        return;
    const int d = idx_vu[(long long)v * a.vol.U + u];
    if (d < 0)
        return;
    Best<C> best;
    best.init();
    scan_generic_body<C>(a, v, u, d, d + 1, best, K_vsu + (long long)v * a.vol.S * a.vol.U + u, (long long)a.vol.U);
}

// The on-chip kernel's instantiations (k2_chip.hpp, one per rung of plan::kChipLadder) are translation units of their own
// (rslf_chip_a/b/c.hip): the hot path's unit launches them through this.  Returns an rslf error code.
int launch_scan_chip(const ScanArgs& a, dim3 grid, size_t lds_bytes, hipStream_t stream);

}  // namespace rslf
