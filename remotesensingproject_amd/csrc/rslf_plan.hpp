// Host-side planning of the gfx950 EPI depth scan: pure functions, no HIP, no device state.
//
// Everything the library decides on the host before it queues work lives here, so that it can be compiled with g++
// alone and unit-tested under AddressSanitizer / UBSan on a box without a GPU (tests/cpp/test_plan.cpp,
// tests/test_plan_cpu.py): which scanlines a device and a chunk take and with what halo, in which order a sweep
// visits the views, how many hypothesis groups and row blocks a scan launch uses and how many records that costs,
// how the streaming kernel's LDS is shared out, which copies a chunk's upload needs, how the pyramid's levels are
// sized, and in which order the devices of a sharded sweep wait for one another.  The .hip translation units only
// turn these plans into launches and copies.
#pragma once

#include <stddef.h>
#include <stdint.h>

#include <string.h>

#include <algorithm>
#include <cmath>
#include <vector>

namespace rslf {

// cv::getStructuringElement result as one bit row per kernel row (k <= 31); the device side is k1_morph_pass
struct MorphElement {
    int k;
    unsigned rows[31];
};

namespace plan {

// ---- constants of the launch shapes -------------------------------------------------------------------------------

constexpr int kScanWavesPerTile = 4;                               // waves of a scan workgroup = hypothesis chunks per tile
constexpr size_t kPartialRecordBytes = 32;                         // sizeof(rslf::Partial), checked in k2_scan.hpp
constexpr size_t kPartialBudget = (size_t)256 << 20;               // bytes of (tile, group, lane) records per grouped scan launch
constexpr size_t kAutoGroupBudget = (size_t)96 << 20;              // ... of which the AUTOMATIC groups of short register launches may take
constexpr int kPackedItemTarget = 2048;                            // packed launches: (tile, group) items the device aims for
constexpr int kSweepGroups = 32;                                   // workgroups sharing a packed tile on a sweep's sparse visits
constexpr int kStreamGroups = 16;                                  // dense launches of the streaming kernel: workgroups per tile
constexpr size_t kStreamSmallEpiBytes = (size_t)2 << 20;           // ... 2 or 3 where a scanline's EPI is at most this (half an XCD's L2)
constexpr int kRowSplitMin = 64;                                   // packed launches of stream-class volumes: rows with at least this many pixels go as row tiles
constexpr int kChipGroups = 8;                                     // ... of the on-chip kernel: half the records, same speed (profiles/r03_k2_variants.md)
constexpr size_t kStreamLdsBytes = (size_t)80 << 10;               // dynamic LDS of one streaming workgroup (two per CU)
constexpr size_t kStagingBudget = (size_t)256 << 20;               // device staging buffer of the chunked host upload
constexpr int kMinSpatialDim = 10;                                 // _MIN_SPATIAL_DIM, rslf_fine_to_coarse.hpp:8

// ---- the on-chip kernel's ladder (k2_chip.hpp) -----------------------------------------------------------------------
// A rung = how many views sit in the AGPR tier (NA) and the LDS tier (NL) behind the 64 of the VGPR tier; three more are
// fetched ahead on every pass.  The tiers fill in that order as the view count grows.  A volume runs on the SMALLEST rung
// that holds all its views, the missing ones padded with samples that contribute nothing (~1 % of a hypothesis each; a
// view beyond a rung, fetched again on every pass, costs 3.5 %) -- so the rungs are 8 views apart.  The top rung is
// BASELINE.json's c5 (201 views), which has its own unpadded instantiation, and one with a ragged tail for 202..220.
// The lists are per translation unit (rslf_chip_a/b/c.hip: each instantiation is ~20 s of hipcc).
constexpr int kChipNV = 64, kChipNAMax = 84, kChipNLMax = 50, kChipAhead = 3;
constexpr int kChipPadMax = 10;                                    // a rung holds at most this many views more than the volume has
constexpr int kChipBestFloats = 4 * 64;                            // a wave's running index and rbar (ChipBest)
constexpr size_t kChipLdsBytes = (size_t)160 << 10;                // one workgroup per CU may take all of it
constexpr int kChipMaxS = 220;                                     // beyond, the per-pass fetches cost more than the streaming kernel's tail
#ifndef RSLF_CHIP_LADDER_A
#define RSLF_CHIP_LADDER_A(X) X(84, 32) X(84, 40) X(84, 50)
#define RSLF_CHIP_LADDER_B(X) X(84, 0) X(84, 8) X(84, 16) X(84, 24)
#define RSLF_CHIP_LADDER_C(X) X(60, 0) X(68, 0) X(76, 0)
#endif
#define RSLF_CHIP_LADDER(X) RSLF_CHIP_LADDER_C(X) RSLF_CHIP_LADDER_B(X) RSLF_CHIP_LADDER_A(X)
#ifndef RSLF_CHIP_FIRST_S
#define RSLF_CHIP_FIRST_S 123   // fewer views: the streaming kernel's two waves per SIMD are faster (profiles/r04_k2_variants.md section 9)
#endif
struct ChipRung {
    int na, nl;
    constexpr int views() const { return kChipNV + na + kChipAhead + nl; }
};
#define RSLF_CHIP_RUNG_(NA, NL) ChipRung{NA, NL},
constexpr ChipRung kChipLadder[] = {RSLF_CHIP_LADDER(RSLF_CHIP_RUNG_)};   // ascending
#undef RSLF_CHIP_RUNG_
constexpr int kChipRungs = (int)(sizeof(kChipLadder) / sizeof(kChipLadder[0]));
constexpr int kChipTopS = kChipLadder[kChipRungs - 1].views();
constexpr int kChipMinS = RSLF_CHIP_FIRST_S > kChipLadder[0].views() - kChipPadMax ? RSLF_CHIP_FIRST_S : kChipLadder[0].views() - kChipPadMax;
constexpr bool chip_ladder_ok()
{
    for (int i = 1; i < kChipRungs; i++)
        if (kChipLadder[i].views() <= kChipLadder[i - 1].views() || kChipLadder[i].views() - kChipLadder[i - 1].views() > kChipPadMax)
            return false;
    return true;
}
static_assert(chip_ladder_ok(), "rungs ascend, and a volume is never padded by more than kChipPadMax views");

// the rung a volume of S views runs on: the smallest that holds them all; beyond the top rung, the top rung (-1: none)
inline int chip_rung_for(int S)
{
    if (S < kChipMinS || S > kChipMaxS)
        return -1;
    for (int i = 0; i < kChipRungs; i++)
        if (kChipLadder[i].views() >= S)
            return i;
    return kChipRungs - 1;
}
// the wave's table of view offsets: one entry per view of the volume or of the rung, whichever is more
constexpr int chip_table_floats(int S, int rung_views) { return ((S > rung_views ? S : rung_views) + 3) & ~3; }
// floats of dynamic LDS per wave: [view offsets][NL samples x 3 channels x 64 lanes][running best, 4 x 64] -- and no less
// than the epilogue's block, which reuses the region's head
constexpr int chip_wave_floats(int S, const ChipRung& r)
{
    return chip_table_floats(S, r.views()) + r.nl * 3 * 64 + kChipBestFloats < 2 * (64 + 6 * 32)
               ? 2 * (64 + 6 * 32)
               : chip_table_floats(S, r.views()) + r.nl * 3 * 64 + kChipBestFloats;
}
// the kernel takes RGB volumes that have a rung and whose per-wave LDS share fits
inline bool chip_takes(int S, int C)
{
    if (C != 3)
        return false;
    const int r = chip_rung_for(S);
    return r >= 0 && (size_t)chip_wave_floats(S, kChipLadder[r]) * 4 * kScanWavesPerTile <= kChipLdsBytes;
}

// Register-variant slot counts compiled into the library (multiples of 8), per channel count: those that run at two or
// more waves per SIMD.  Beyond them the streaming / on-chip kernels take over (k2_scan.hpp).
#ifndef RSLF_SPAD_LIST_1CH
#define RSLF_SPAD_LIST_1CH(X) X(8) X(16) X(24) X(32) X(40) X(48) X(56) X(64) X(72) X(80) X(88) X(96) X(104) X(112) X(120) X(128) \
    X(144) X(160) X(176) X(192)
#endif
#ifndef RSLF_SPAD_LIST_3CH
#define RSLF_SPAD_LIST_3CH(X) X(8) X(16) X(24) X(32) X(40) X(48)
#endif

// Smallest compiled slot count >= S (0 = none: another kernel runs).
inline int pick_spad(int S, int C)
{
    int best = 0;
#define RSLF_PICK(N) \
    if (N >= S && best == 0) best = N;
    if (C == 1) {
        RSLF_SPAD_LIST_1CH(RSLF_PICK)
    } else if (C == 3) {
        RSLF_SPAD_LIST_3CH(RSLF_PICK)
    }
#undef RSLF_PICK
    return best;
}

// ---- small reference rules ----------------------------------------------------------------------------------------

// dc.hpp:490-494 (pile) / :303-311 (single EPI): an s_hat outside the views means the middle one
inline int resolve_s_hat(int s_hat, int S)
{
    if (s_hat < 0 || s_hat > S - 1)
        return (int)std::floor((0.0 + S) / 2);
    return s_hat;
}

// core.hpp:584: `for (int i = 0; i < par_mean_shift_max_iter; i++)` with a FLOAT bound
inline int mean_shift_passes(float max_iter)
{
    int n = 0;
    while ((float)n < max_iter && n < (1 << 20))
        n++;
    return n;
}

// core.hpp:981-990: the centre view, then outwards, alternating (an even view count never reaches view 0)
inline std::vector<int> sweep_order(int S)
{
    std::vector<int> order;
    if (S < 1)
        return order;
    const int s_mid = (int)std::floor(S / 2.0);
    order.push_back(s_mid);
    for (int off = 1; off < S - s_mid; off++) {
        order.push_back(s_mid + off);
        if (s_mid - off > -1)
            order.push_back(s_mid - off);
    }
    return order;
}

inline int sweep_view_after(int S, int s_hat)
{
    const std::vector<int> order = sweep_order(S);
    for (size_t i = 0; i + 1 < order.size(); i++)
        if (order[i] == s_hat)
            return order[i + 1];
    return -1;
}

// cv::getStructuringElement(shape, Size(k, k)) with the default anchor, as OpenCV 3.x builds it (imgproc/src/morph.cpp):
// RECT every column; CROSS the anchor row entirely, elsewhere the anchor column; ELLIPSE the columns
// [c - dx, c + dx + 1), dx = cvRound(c * sqrt((r*r - dy*dy) / (r*r))), r = c = k/2, dy = i - r.
inline MorphElement structuring_element(int shape, int k)
{
    MorphElement el;
    el.k = k;
    const int r = k / 2, c = k / 2;
    const double inv_r2 = r ? 1.0 / ((double)r * r) : 0.0;
    for (int i = 0; i < 31; i++)
        el.rows[i] = 0;
    for (int i = 0; i < k && i < 31; i++) {
        int j1 = 0, j2 = 0;
        if (shape == 0 || (shape == 1 && i == k / 2)) {
            j2 = k;
        } else if (shape == 1) {
            j1 = k / 2;
            j2 = j1 + 1;
        } else {
            const int dy = i - r;
            if (std::abs(dy) <= r) {
                const int dx = (int)std::lrint(c * std::sqrt((r * r - dy * dy) * inv_r2));
                j1 = std::max(c - dx, 0);
                j2 = std::min(c + dx + 1, k);
            }
        }
        for (int j = j1; j < j2 && j < 31; j++)
            el.rows[i] |= 1u << j;
    }
    return el;
}

// cvRound(x * 0.5): ties to even (fine_to_coarse_core.cpp:47-48)
inline void f2c_level_dims(int V, int U, int* V2, int* U2)
{
    *V2 = (int)std::lrint(V * 0.5);
    *U2 = (int)std::lrint(U * 0.5);
}

struct LevelDims {
    int V, U;
};

// The pyramid of FineToCoarse's constructor (rslf_fine_to_coarse.hpp:130-154): levels while both sides exceed
// _MIN_SPATIAL_DIM and the depth limit allows (max_depth < 1: no limit).
inline std::vector<LevelDims> f2c_pyramid(int V, int U, int max_depth)
{
    std::vector<LevelDims> levels;
    if (max_depth < 1)
        max_depth = 1 << 30;
    int dim_v = V, dim_u = U, counter = 0;
    while (dim_v > kMinSpatialDim && dim_u > kMinSpatialDim && counter < max_depth) {
        counter++;
        levels.push_back(LevelDims{dim_v, dim_u});
        int v2, u2;
        f2c_level_dims(dim_v, dim_u, &v2, &u2);
        if (v2 < 1 || u2 < 1)
            break;
        dim_v = v2;
        dim_u = u2;
    }
    return levels;
}

// ---- the selective median (core.hpp:663-718; k3_median.hpp) -------------------------------------------------------

constexpr int kMedianBlock = 256;          // pixels of one scanline per workgroup
constexpr int kMedianNetMaxSide = 11;      // window sides up to this sort their slots in registers
constexpr int kMedianTileMaxSide = 31;     // ... up to this keep the window tile in LDS (one predicate word per window row)
constexpr int kMedianMaxSize = 1 << 30;    // sizes past this are refused as invalid (v + width must stay an int)

// core.hpp:686: `int width = (a_size-1)/2;` -- C++ division truncates, so size 0 is the 1 x 1 window too
inline int median_width(int size) { return (size - 1) / 2; }

// floats of dynamic LDS a median workgroup needs for window half-width w: depths + C radiance planes of the tile
inline size_t median_lds_bytes(int w, int C)
{
    return (size_t)(1 + C) * (size_t)(2 * w + 1) * (size_t)(kMedianBlock + 2 * w) * sizeof(float);
}

struct MedianPlan {
    int w;             // window half-width
    int mode;          // > 0: window side of the register network; 0: LDS tile + radix select; < 0: global radix select
    size_t lds_bytes;
};

inline MedianPlan median_plan(int size, int C)
{
    MedianPlan m;
    m.w = std::max(0, median_width(size));
    const int side = 2 * m.w + 1;
    m.mode = side <= kMedianNetMaxSide ? side : side <= kMedianTileMaxSide ? 0 : -1;
    m.lds_bytes = m.mode >= 0 ? median_lds_bytes(m.w, C) : 0;
    return m;
}

// norm<T>(x) < eps (src/rslf_types.cpp:80-91; core.hpp:703-706, :1116) as ONE compare, exactly.
//   1 channel:  float(double(|x|) * 1.73205080757) < eps   -- non-decreasing in |x|  <=>  |x| < a1
//   3 channels: float(sqrt(s)) < eps, s = the double sum of squares  -- non-decreasing in s  <=>  s < s3
// a1 / s3 = the smallest non-negative float / double for which the test FAILS, found by bisection over bit patterns
// (ordered like the values).  eps <= 0 or NaN: nothing passes (a1 = s3 = 0).  The kernels then need no double multiply
// (1 channel) and no double square root (3 channels) per window pixel.
struct NormThreshold {
    float a1;
    double s3;
};

inline bool norm1_below(float ax, float eps) { return (float)((double)ax * 1.73205080757) < eps; }
inline bool norm3_below(double s, float eps) { return (float)std::sqrt(s) < eps; }

inline NormThreshold norm_threshold(float eps)
{
    NormThreshold t;
    t.a1 = 0.0f;
    t.s3 = 0.0;
    if (norm1_below(0.0f, eps)) {
        uint32_t lo = 0u, hi = 0x7F800000u;   // test(lo) true; test(+inf) false: float(inf) < eps never holds
        while (hi - lo > 1) {
            const uint32_t mid = lo + (hi - lo) / 2;
            float x;
            memcpy(&x, &mid, sizeof x);
            if (norm1_below(x, eps))
                lo = mid;
            else
                hi = mid;
        }
        memcpy(&t.a1, &hi, sizeof hi);
    }
    if (norm3_below(0.0, eps)) {
        uint64_t lo = 0u, hi = 0x7FF0000000000000ull;
        while (hi - lo > 1) {
            const uint64_t mid = lo + (hi - lo) / 2;
            double x;
            memcpy(&x, &mid, sizeof x);
            if (norm3_below(x, eps))
                lo = mid;
            else
                hi = mid;
        }
        memcpy(&t.s3, &hi, sizeof hi);
    }
    return t;
}

// ---- scanline partitions ------------------------------------------------------------------------------------------

// rows either side of a block that must be recomputed (pile path) or exchanged (sweep) for the block's own rows to come
// out exact: the median reads +-(size-1)/2 rows (core.hpp:686), and through the optional opening +-2*(k/2) rows more
// (core.hpp:759-768)
inline int median_halo(int median_filter_size) { return std::max(0, median_width(median_filter_size)); }
inline int halo_rows(int median_filter_size, int opening_size)
{
    return median_halo(median_filter_size) + (opening_size > 1 ? 2 * (opening_size / 2) : 0);
}

struct RowBlock {
    int a, b;     // owned rows [a, b) of the whole field
    int lo, hi;   // held / computed rows: owned + halo, clipped to the field
};

// block i of n over V scanlines: contiguous, sizes differ by at most one, together they cover [0, V) exactly
inline RowBlock row_block(int V, int i, int n, int halo)
{
    RowBlock r;
    r.a = (int)((long long)V * i / n);
    r.b = (int)((long long)V * (i + 1) / n);
    r.lo = std::max(0, r.a - halo);
    r.hi = std::min(V, r.b + halo);
    return r;
}

// A sharded sweep's blocks must be able to fill their neighbours' halo rows from rows they own: fewer devices otherwise.
inline int sweep_devices_for(int V, int n_devices, int halo)
{
    int nd = std::max(1, n_devices);
    while (nd > 1 && V / nd < std::max(1, halo))
        nd--;
    return nd;
}

// Chunks of one device's rows [r0, r1) on the pipelined host path.  A given size: uniform.  Automatic (chunk_rows <= 0):
// a short first chunk so that the kernels start early (its upload is the one copy nothing hides), then two large ones
// (stacked input; scattered input, which is gathered into pinned memory first, takes a middling second chunk and pieces
// of about rows/3.5) -- a chunk's scan is a grid of its own, and a grid of 5.4 rounds of workgroups pays for 6.
inline std::vector<RowBlock> chunk_plan(int r0, int r1, int V, int halo, int chunk_rows, bool scattered)
{
    std::vector<RowBlock> chunks;
    const int rows = r1 - r0;
    if (rows <= 0)
        return chunks;
    std::vector<int> sizes;
    if (chunk_rows > 0) {
        for (int a = 0; a < rows; a += chunk_rows)
            sizes.push_back(std::min(chunk_rows, rows - a));
    } else {
        int left = rows;
        const int first = std::min(left, std::max(32, (rows + 15) / 16));
        sizes.push_back(first);
        left -= first;
        if (left > 0 && scattered) {   // gathered through pinned memory first: smaller steps keep the kernels fed
            const int second = std::min(left, std::max(32, (rows + 7) / 8));
            sizes.push_back(second);
            left -= second;
        }
        if (left > 0) {
            // the chunk before has to cover the next one's upload with its scan
            const int n = scattered ? std::max(1, (int)std::lround(left / (rows / 3.5))) : (left > 2 * first ? 2 : 1);
            for (int i = 0; i < n; i++) {
                const int sz = (left + (n - i) - 1) / (n - i);
                sizes.push_back(sz);
                left -= sz;
            }
        }
    }
    int a = r0;
    for (int sz : sizes) {
        if (sz <= 0)
            continue;
        RowBlock c;
        c.a = a;
        c.b = a + sz;
        c.lo = std::max(0, c.a - halo);
        c.hi = std::min(V, c.b + halo);
        chunks.push_back(c);
        a += sz;
    }
    return chunks;
}

inline int max_held_rows(const std::vector<RowBlock>& chunks)
{
    int m = 0;
    for (const RowBlock& c : chunks)
        m = std::max(m, c.hi - c.lo);
    return m;
}

// worker t of nt over `rows` items: [first, last)
inline void split_range(int rows, int t, int nt, int* first, int* last)
{
    *first = (int)((long long)rows * t / nt);
    *last = (int)((long long)rows * (t + 1) / nt);
}

// ---- host copies --------------------------------------------------------------------------------------------------

// Number of maximal runs of EPIs that follow one another in host memory among ptrs[0..rows): each run is one 1-D copy.
// Rows with a stride (row_stride != row_bytes) are a run each.
inline int count_runs(const void* const* ptrs, int rows, size_t row_stride_bytes, size_t row_bytes, size_t epi_bytes)
{
    if (rows <= 0)
        return 0;
    int runs = 1;
    for (int i = 1; i < rows; i++)
        if (row_stride_bytes != row_bytes || (const char*)ptrs[i] != (const char*)ptrs[i - 1] + epi_bytes)
            runs++;
    return runs;
}

// Are the EPIs [r0, r1) scattered over the heap (a Vec<Mat>) rather than stacked in one array?
inline bool epis_scattered(const void* const* ptrs, int r0, int r1, size_t row_stride_bytes, size_t row_bytes, size_t epi_bytes)
{
    if (row_stride_bytes != row_bytes)
        return true;
    for (int i = r0 + 1; i < r1; i++)
        if (ptrs[i] && ptrs[i - 1] && (const char*)ptrs[i] != (const char*)ptrs[i - 1] + epi_bytes)
            return true;
    return false;
}

// A chunk goes through the pinned staging buffer when it is broken into more than 8 runs AND the buffer holds it
// (the buffer is sized from the scattered-ness of the device's own rows; a chunk's halo rows can add breaks of their
// own, so the capacity is checked per chunk -- ADVICE r2).
inline bool use_pinned_gather(int runs, int rows, size_t epi_bytes, size_t pin_cap)
{
    return runs > 8 && pin_cap > 0 && (size_t)rows * epi_bytes <= pin_cap;
}

// rows per pass of the chunked host upload through the bounded device staging buffer
inline int staging_chunk_rows(size_t epi_bytes, int V, size_t budget = kStagingBudget)
{
    const size_t c = std::max<size_t>(1, budget / std::max<size_t>(1, epi_bytes));
    return (int)std::min<size_t>(c, (size_t)std::max(1, V));
}

// ---- result planes of one chunk, carved from one block ------------------------------------------------------------

struct PlaneLayout {   // byte offsets into the block, for n pixels and C channels
    size_t Ce, Cd, depth, raw, score, rbar, idx, mask, counts, bytes;
};

inline PlaneLayout plane_layout(size_t n, int C, int count_rows)
{
    PlaneLayout q;
    q.Ce = 0;
    q.Cd = q.Ce + n * 4;
    q.depth = q.Cd + n * 4;
    q.raw = q.depth + n * 4;
    q.score = q.raw + n * 4;
    q.rbar = q.score + n * 4;
    q.idx = q.rbar + n * 4 * (size_t)C;
    q.mask = q.idx + n * 4;
    q.counts = (q.mask + n + 15) & ~(size_t)15;
    q.bytes = q.counts + (size_t)count_rows * sizeof(int);
    return q;
}

// ---- scan launches ------------------------------------------------------------------------------------------------

// largest fraction below 1 that cannot round a position up to the next integer: 1 - ulp(largest position the scan forms)
inline float stream_frac_max(int U)
{
    int e = 0;
    (void)std::frexp((float)U + 2.0f, &e);   // U + 2 < 2^e
    return 1.0f - std::ldexp(1.0f, std::max(e - 24, -24));
}

// The pixel-per-wave kernel of sparse launches (k2_reg.hpp, k2_scan_reg_px): a pixel's hypotheses sit in the LANES of 1, 2
// or 4 waves (lane slot + k * 64 * waves), so what matters is how many lanes a pixel keeps busy.  Returns the number of
// waves per pixel with the best lane use (the larger on ties: finer items), or 0 when no choice reaches 60 %.
inline int px_waves(int dim_d)
{
    int best = 0;
    double best_use = 0.0;
    for (int w = 1; w <= kScanWavesPerTile; w *= 2) {
        const int lanes = 64 * w, iters = (dim_d + lanes - 1) / lanes;
        const double use = (double)dim_d / ((double)iters * lanes);
        if (use >= best_use - 1e-12) {
            best_use = use;
            best = w;
        }
    }
    return best_use >= 0.6 ? best : 0;
}

struct ScanRequest {
    int V, U, S, C, dim_d;
    int spad;              // register kernel's slot count, 0 = none
    bool use_stream;       // streaming kernel
    bool use_chip;         // on-chip kernel (k2_chip.hpp): one workgroup per CU, row tiles as the streaming kernel's, never packed
    int chip_wave_floats;  // its dynamic LDS per wave, in floats
    int reg_waves;         // waves per SIMD of the register kernel (scan_reg_waves), 0 if unknown
    int num_cus;           // compute units of the device, 0 if unknown
    int ctx_groups;        // groups the caller asked for (the sweep's sparse visits), >= 1
    bool ctx_packed;       // the caller asked for one packed pixel list
    int precompacted;      // 0: the scan compacts; 1: row lists are in place (K1); 2: the packed list is (sweep)
    int force_groups;      // debug hooks: 0 / -1 = automatic
    int force_packed;
    int px_mode;           // pixel-per-wave kernel for packed launches of a register kernel: -1 automatic, 0 never, 1 whenever it can run
    int stream_groups;     // 0 = kStreamGroups
    int stream_share;      // 63-pixel row tiles (shared taps): 0 never, 1 where the re-gathered tail is long enough to pay for them, 2 always
    size_t stream_lds_bytes;
};

struct ScanPlan {
    int groups;            // workgroups sharing one tile's hypotheses
    bool packed;           // one packed list over all scanlines
    bool packed_adapt;     // the device settles the group count from the list length
    int px_waves;          // > 0: the pixel-per-wave kernel takes the packed list, this many waves per pixel (no groups, no records)
    int tile_w;            // 63 or 64 entries per row tile
    int tiles_per_row;
    int rows_per_launch;   // grouped row-tile launches go by blocks of scanlines
    size_t records;        // 32-byte records the launches need (0 when groups == 1)
    size_t tickets;        // one int per tile of a launch
    // streaming kernel
    int stream_nres;       // resident prefix of the instantiation that runs (k2_stream.hpp)
    int stream_park;       // samples per lane parked in LDS
    int stream_wave_floats;
    size_t lds_bytes;
};

inline int stream_resident_hi(int C, int nres_1ch, int nres_rgb) { return C == 1 ? nres_1ch : nres_rgb; }

// The streaming kernel's LDS per wave: the S view offsets (rounded up to 4) and as many batches of parked samples as the
// workgroup's share leaves room for (never past the end of the views).  Returns the parked samples per lane.
inline int stream_park_for(int S, int C, int nres, size_t stream_lds_bytes)
{
    if (nres <= 0)
        return 0;
    const int batch = C == 1 ? 8 : 4;
    const size_t s4 = ((size_t)S + 3) & ~(size_t)3;
    size_t room = stream_lds_bytes / kScanWavesPerTile / sizeof(float);   // floats per wave
    room -= std::min(room, s4);
    int park = (int)(room / ((size_t)C * 64));
    park = std::min(park, S - nres);
    park = std::max(park, 0);
    return park - park % batch;
}

// Shared taps (63-pixel row tiles, one texel load per sample, the right tap from lane + 1) halve the loads of the samples
// that are gathered again on EVERY pass -- and cost a lane per tile.  They pay where that tail is long: 109 of 201 views at
// c5 (+9 %); at 100 views RGB the tail is 8 samples and plain 64-pixel tiles are 6 % faster (47.5 vs 44.7 ms,
// profiles/r04_k2_variants.md).  Rule: the tail is at least a quarter of the views.
inline bool stream_shares_taps(int S, int nres, int park) { return 4 * (S - nres - park) >= S; }

// `nres` = the streaming kernel's resident prefix for this volume (stream_resident_for), needed only with use_stream;
// `nres_px` = the prefix of its pixel-per-wave form (stream_px_resident_for; -1: the same).
inline ScanPlan plan_scan(const ScanRequest& r, int nres, int nres_px = -1)
{
    ScanPlan p;
    const size_t n = (size_t)r.V * r.U;
    int groups = std::max(1, r.ctx_groups);
    bool packed = r.ctx_packed;
    // the streaming kernel's dense launches share tiles so that what an XCD's workgroups gather from fits its L2
    if ((r.use_stream || r.use_chip) && groups == 1 && !packed) {
        groups = r.stream_groups > 0 ? r.stream_groups : kStreamGroups;
        // the on-chip kernel reads a tile's samples once per workgroup whatever their number; fewer workgroups per tile leave
        // fewer records -- unless the launch is then only a few rounds of workgroups
        const long long tiles = (long long)r.V * ((r.U + 62) / 63);
        if (r.use_chip && r.stream_groups <= 0 && (r.num_cus <= 0 || tiles * kChipGroups >= 8LL * r.num_cus))
            groups = kChipGroups;
        // The streaming kernel's groups are there for the XCD's L2.  Where one scanline's EPI (all its views) is a small part
        // of it and the launch has workgroups to spare, two or three groups are enough and every group fewer is a record
        // not written and merged: 100 views RGB x 1146 px (1.4 MB), 120 hypotheses, dense: 8 groups 44.9 ms, 4 43.4, 3 41.0,
        // 2 41.6, 1 43.3 (profiles/r04_k2_variants.md section 12).  Three where the hypotheses divide evenly over its 12 waves.
        if (r.use_stream && r.stream_groups <= 0 && (size_t)r.S * r.U * r.C * sizeof(float) <= kStreamSmallEpiBytes) {
            const int few = r.dim_d % (3 * kScanWavesPerTile) == 0 ? 3 : 2;
            if (r.num_cus <= 0 || tiles * few >= 16LL * r.num_cus)
                groups = few;
        }
    }
    // A dense launch of a register kernel whose grid is only a few rounds of workgroups pays for its last, partly empty
    // round: sharing each tile's hypotheses among 2-8 workgroups makes the rounds shorter and more numerous -- as long
    // as a wave keeps at least eight hypotheses and ~256 (hypothesis, view) pairs.
    if (r.spad && groups == 1 && !packed && r.num_cus > 0 && r.reg_waves > 0) {
        const long long tiles = (long long)r.V * ((r.U + 63) / 64);
        const long long resident = (long long)r.num_cus * r.reg_waves;
        // (and the records stay a modest scratch: the groups are a convenience here, not the kernel's locality)
        while (groups < 8 && tiles * groups < 40 * resident && r.dim_d >= 8 * kScanWavesPerTile * 2 * groups &&
               (long long)(r.dim_d / (kScanWavesPerTile * 2 * groups)) * r.S >= 256 &&
               (size_t)tiles * (size_t)(groups * 2) * 64 * kPartialRecordBytes <= kAutoGroupBudget)
            groups *= 2;
    }
    if (r.force_groups > 0)
        groups = std::min(64, r.force_groups);
    if (r.force_packed >= 0)
        packed = r.force_packed != 0;
    if (n > (size_t)INT32_MAX || r.precompacted == 1)
        packed = false;   // entry counts are ints; precompacted: the row lists are what K1 wrote
    if (r.precompacted == 2)
        packed = true;    // the previous visit's apply pass left the packed list and its length
    // enough hypotheses to share out?
    while (groups > 1 && r.dim_d < 2 * kScanWavesPerTile * groups)
        groups /= 2;

    // Packed launches of a register or streaming kernel: lanes own hypotheses, a wave owns a pixel (the gather of a sparse
    // list is then 64 neighbouring taps of one scanline instead of 64 scanlines) -- no hypothesis groups, no records
    int px_w = 0;
    if (packed && (r.spad || r.use_stream) && !r.use_chip && r.px_mode != 0) {
        px_w = px_waves(r.dim_d);
        if (px_w == 0 && r.px_mode == 1)
            px_w = 1;
    }
    if (px_w && r.use_stream && nres_px >= 0)
        nres = nres_px;
    p.stream_nres = r.use_stream ? nres : 0;
    const int park_early = r.use_stream ? stream_park_for(r.S, r.C, nres, r.stream_lds_bytes) : 0;
    const bool share = r.use_chip ? r.stream_share != 0
                                  : r.use_stream && (r.stream_share == 2 || (r.stream_share == 1 && stream_shares_taps(r.S, nres, park_early)));
    p.tile_w = (share && !packed) ? 63 : 64;
    // 63-entry tiles: a row's last tile takes up to 64 entries (scan_tile)
    p.tiles_per_row = p.tile_w == 63 ? std::max(1, (r.U + 61) / 63) : (r.U + p.tile_w - 1) / p.tile_w;
    p.packed_adapt = packed && !r.use_stream && !r.use_chip;
    p.rows_per_launch = r.V;
    if (groups > 1 && !packed) {
        const size_t per_row = (size_t)p.tiles_per_row * groups * 64 * kPartialRecordBytes;
        p.rows_per_launch = (int)std::min<size_t>((size_t)r.V, std::max<size_t>(1, kPartialBudget / per_row));
    }
    if (packed && (r.use_stream || r.use_chip))
        while (groups > 1 && ((n + 63) / 64) * groups * 64 * kPartialRecordBytes > kPartialBudget)
            groups /= 2;
    p.px_waves = px_w;
    if (px_w) {
        groups = 1;
        p.packed_adapt = false;
    }
    p.groups = groups;
    p.packed = packed;
    p.records = 0;
    p.tickets = 0;
    if (groups > 1) {
        const size_t tiles_all = (n + 63) / 64;
        const size_t tiles_max = !packed ? (size_t)p.rows_per_launch * p.tiles_per_row
                                         : p.packed_adapt ? std::min<size_t>(tiles_all, kPackedItemTarget / 2) : tiles_all;
        p.records = (packed && p.packed_adapt) ? std::min<size_t>(tiles_all * groups, kPackedItemTarget) * 64 : tiles_max * groups * 64;
        p.tickets = tiles_max;
    }
    p.stream_park = 0;
    p.stream_wave_floats = 0;
    p.lds_bytes = 0;
    if (r.use_stream) {
        // LDS per wave: the S view offsets (rounded up to 4) and as many batches of parked samples as the workgroup's
        // share leaves room for (never past the end of the views)
        const size_t s4 = ((size_t)r.S + 3) & ~(size_t)3;
        const int park = park_early;
        p.stream_park = park;
        p.stream_wave_floats = (int)(s4 + (size_t)park * r.C * 64);
        p.stream_wave_floats = std::max(p.stream_wave_floats, 2 * (64 + (3 + r.C) * 32));   // room for the wave's EpilogueBlock
        p.lds_bytes = (size_t)kScanWavesPerTile * p.stream_wave_floats * sizeof(float);
    }
    if (r.use_chip) {
        p.stream_wave_floats = r.chip_wave_floats;
        p.lds_bytes = (size_t)kScanWavesPerTile * r.chip_wave_floats * sizeof(float);
    }
    return p;
}

// The records a sweep's sparse visits will need, sized before the first visit (no allocation in the middle of the sequence)
inline void sweep_record_plan(size_t n_pixels, int dim_d, bool stream, size_t* records, size_t* tickets)
{
    int g = kSweepGroups;
    while (g > 1 && dim_d < 2 * kScanWavesPerTile * g)
        g /= 2;
    const size_t tiles_all = (n_pixels + 63) / 64;
    if (stream)
        while (g > 1 && tiles_all * g * 64 * kPartialRecordBytes > kPartialBudget)
            g /= 2;
    *records = 0;
    *tickets = 0;
    if (g > 1 && n_pixels <= (size_t)INT32_MAX) {
        if (stream) {
            *records = tiles_all * g * 64;
            *tickets = tiles_all;
        } else {
            *records = std::min<size_t>(tiles_all * g, kPackedItemTarget) * 64;
            *tickets = std::min<size_t>(tiles_all, kPackedItemTarget / 2);
        }
    }
}

// ---- one visit of a sweep sharded over several devices of one process ---------------------------------------------
//
// scan on every device; every device fetches its neighbours' boundary rows (needs the neighbour's scan); every device
// finishes (median + propagation) once BOTH neighbours have fetched its rows, because its apply pass rewrites them
// (core.hpp:1119-1121).  One host thread queues the ops in this order; `waits` are the cross-device event waits an op's
// stream makes before it runs (same-device order is the stream's own).
struct VisitOp {
    enum Kind { SCAN, FETCH, FINISH } kind;
    int dev;
    int neighbour;                    // FETCH: the device whose rows are read, else -1
    std::vector<int> wait_scan_of;    // devices whose SCAN event this op waits for
    std::vector<int> wait_fetch_of;   // devices whose (last) FETCH event this op waits for
};

inline std::vector<VisitOp> sweep_visit_schedule(int nd, int h_med)
{
    std::vector<VisitOp> ops;
    for (int i = 0; i < nd; i++)
        ops.push_back(VisitOp{VisitOp::SCAN, i, -1, {}, {}});
    for (int i = 0; i < nd; i++)
        for (int side = 0; side < 2 && h_med > 0; side++) {
            const int k = side == 0 ? i - 1 : i + 1;
            if (k < 0 || k >= nd)
                continue;
            ops.push_back(VisitOp{VisitOp::FETCH, i, k, {k}, {}});
        }
    for (int i = 0; i < nd; i++) {
        VisitOp f{VisitOp::FINISH, i, -1, {}, {}};
        if (i > 0)
            f.wait_fetch_of.push_back(i - 1);
        if (i + 1 < nd)
            f.wait_fetch_of.push_back(i + 1);
        ops.push_back(f);
    }
    return ops;
}

// Local row numbers of a boundary-row fetch: device `d` (block bd) reads h rows owned by its neighbour (block bo).
// side 0: the h rows above d's block = the last h own rows of the block before; side 1: the first h own rows of the next.
inline void fetch_rows(const RowBlock& bd, const RowBlock& bo, int side, int h, int* dst_row, int* src_row)
{
    *dst_row = side == 0 ? (bd.a - bd.lo) - h : (bd.b - bd.lo);
    *src_row = side == 0 ? (bo.b - bo.lo) - h : (bo.a - bo.lo);
}

}  // namespace plan
}  // namespace rslf
