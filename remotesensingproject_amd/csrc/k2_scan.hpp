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
    struct Partial* partial;   // [tile][group][64]
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
};

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

// Which tile does this workgroup own, which pixel this lane?  Block-uniform result
// (every wave of the block takes the same branch, so the later barrier is safe).
// `lb` = tile * groups + group.
__device__ __forceinline__ bool scan_tile(const ScanArgs& a, int lb, int& v, int& u, bool& active)
{
    const int lane = threadIdx.x & 63;
    if (lb >= a.logical_blocks)
        return false;
    const int tile = lb / a.groups;
    const int vr = tile / a.tiles_per_row;
    const int j = tile - vr * a.tiles_per_row;
    v = vr + a.v0;
    const int n = a.count[v];
    if (j * a.tile_w >= n)
        return false;
    // 63-entry tiles (streaming kernel, lane 63 left to the shared taps): the row's LAST tile takes up to 64 entries, so
    // that a row of 63 k + 1 pixels (4096 = 65 * 63 + 1) does not end in a tile of one
    int width = a.tile_w;
    if (a.tile_w == 63) {
        const int T = max(1, (n + 61) / 63);      // tiles of this row: 63 entries each, the last one 1..64
        if (j >= T)
            return false;
        if (j == T - 1)
            width = n - 63 * j;
    }
    const int e = j * a.tile_w + lane;
    active = lane < width && e < n;
    // idle lanes shadow the tile's last pixel so their addresses stay valid
    u = a.list[(long long)v * a.vol.U + (active ? e : min(j * a.tile_w + width, n) - 1)];
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

// Word accesses that are coherent at agent scope by themselves (sc1: to / from the memory side), for data handed from one
// workgroup to another that may run on a different XCD -- no cache-wide write-back or invalidate.
__device__ __forceinline__ void store_coherent(unsigned* p, unsigned v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned load_coherent(const unsigned* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int C>
__device__ __forceinline__ void combine_tile(const ScanArgs& a, int tile, int v, int u);

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
                                              float* wave_lds = nullptr, int wave_stride = 0)
{
    __shared__ double s_static[DYN ? 1 : kScanWaves][DYN ? 1 : EpilogueBlock<C>::kDoubles];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
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
        unsigned* w = reinterpret_cast<unsigned*>(a.partial + ((long long)lb * 64 + lane));
        static_assert(sizeof(Partial) == 32, "eight words per record");
        const unsigned long long sb = (unsigned long long)__double_as_longlong(sum);
        store_coherent(w + 0, __float_as_uint(best));
        store_coherent(w + 1, __float_as_uint(best_D));
        store_coherent(w + 2, (unsigned)best_d);
#pragma unroll
        for (int c = 0; c < 3; c++)
            store_coherent(w + 3 + c, c < C ? __float_as_uint(best_rbar[c < C ? c : 0]) : 0u);
        store_coherent(w + 6, (unsigned)sb);
        store_coherent(w + 7, (unsigned)(sb >> 32));
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
            combine_tile<C>(a, tile, v, u);
        if (lane == 0)
            a.ticket[tile] = 0;   // clean for the next launch (kernel boundary orders it)
        return;
    }
    if (active)
        write_pixel<C>(a, (long long)v * a.vol.U + u, best, best_d, best_D, best_rbar, sum);
}

// groups > 1: the wave that drew the tile's last ticket merges the groups' records in hypothesis order and writes the pixels.
template <int C>
__device__ __forceinline__ void combine_tile(const ScanArgs& a, int tile, int v, int u)
{
    const unsigned* pr = reinterpret_cast<const unsigned*>(a.partial + (long long)tile * a.groups * 64 + (threadIdx.x & 63));
    float best = -1.0f, best_D = 0.0f;
    int best_d = -1;
    float best_rbar[C];
#pragma unroll
    for (int c = 0; c < C; c++)
        best_rbar[c] = 0.0f;
    double sum = 0.0;
    // eight groups' records at a time, every word's load issued before the first is used: the loads go past the caches
    // (a round trip to memory each), and this wave is the last thing its tile waits for
    constexpr int GB = 8, W = sizeof(Partial) / sizeof(unsigned);
    for (int g0 = 0; g0 < a.groups; g0 += GB) {
        unsigned w[GB][W];
#pragma unroll
        for (int j = 0; j < GB; j++) {
            const unsigned* q = pr + (long long)min(g0 + j, a.groups - 1) * 64 * W;
#pragma unroll
            for (int k = 0; k < W; k++)
                if (k < 3 + C || k >= 6)
                    w[j][k] = load_coherent(q + k);
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
        const int lb = xcd_logical_block(blockIdx.x, a.per_xcd);                        \
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
__host__ __device__ constexpr int stream_resident_lo(int C) { return C == 1 ? 0 : 48; }
__host__ __device__ constexpr int stream_resident_for(int S, int C)
{
    return S >= stream_resident_hi(C) ? stream_resident_hi(C) : (stream_resident_lo(C) > 0 && S >= stream_resident_lo(C)) ? stream_resident_lo(C) : 0;
}

// DENSE: the tile is 63 consecutive pixels of one scanline in lanes 0..62 and lane 63 stands on the pixel after them
// (scan_stream_rows): a lane's right tap is then its neighbour's left tap, so the re-gathered tail loads ONE texel
// per lane and sample and takes the other from lane + 1 (v_mov_b32 wave_shl:1) -- half the vector-memory
// instructions of the tail, which is what bounds it (the CU's texture data path takes ~17 clocks per multi-dword
// wave-instruction whatever its width, tools/ubench_ta.hip; PMC: TD_BUSY 90 %).
template <int C, bool BORDER, bool UNIFORM_D, int NRES, bool DENSE = false>
__device__ __forceinline__ void scan_stream_body(const ScanArgs& a, int v, int u, int d0, int d1, Best<C>& best,
                                                 float* __restrict__ otab)
{
    static_assert(!DENSE || (UNIFORM_D && !BORDER), "shared taps need a common hypothesis grid and no border lane");
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

    for (int d = d0; d < d1; d++) {
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

template <int C>
__global__ __launch_bounds__(64 * kScanWaves) __attribute__((amdgpu_waves_per_eu(RSLF_STREAM_WAVES, 8))) void k2_scan_stream(ScanArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float s_stream_otab[];   // [kScanWaves][stream_wave_floats]
    float* otab = s_stream_otab + (size_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * a.stream_wave_floats;
    // the waves' results for the epilogue go to the head of their own regions (EpilogueBlock): no static LDS at all
    constexpr bool kEpiDyn = true;
    float* const epi_lds = otab;
    const int epi_stride = a.stream_wave_floats;
    // packed tiles: lanes sit on different scanlines, offsets are per lane (the <true, false> body)
    if (a.vol.S >= stream_resident_hi(C)) {
        RSLF_SCAN_KERNEL_BODY((scan_stream_rows<C, stream_resident_hi(C)>(a, v, u, active, d0, d1, best, otab)),
                              (scan_stream_body<C, true, false, stream_resident_hi(C)>(a, v, u, d0, d1, best, otab)))
    } else if (stream_resident_lo(C) > 0 && a.vol.S >= stream_resident_lo(C)) {
        RSLF_SCAN_KERNEL_BODY((scan_stream_rows<C, stream_resident_lo(C)>(a, v, u, active, d0, d1, best, otab)),
                              (scan_stream_body<C, true, false, stream_resident_lo(C)>(a, v, u, d0, d1, best, otab)))
    } else {
        RSLF_SCAN_KERNEL_BODY((scan_stream_rows<C, 0>(a, v, u, active, d0, d1, best, otab)),
                              (scan_stream_body<C, true, false, 0>(a, v, u, d0, d1, best, otab)))
    }
}

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
// the host (rslf_abi.hip: rslf_depth_epi_pile).
// ---------------------------------------------------------------------------
// samples whose loads are in flight together (2 registers per sample and channel while they are)
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
template <int SPAD, int C, bool BORDER, bool UNIFORM_D, bool PK, int GB = 0>
__device__ __forceinline__ void scan_reg_body(const ScanArgs& a, int v, int u, int d0, int d1, Best<C>& best,
                                              float* __restrict__ otab)
{
    // 104 slots (the c3 shape) have 15 registers to spare at three waves per SIMD: 13 loads in flight instead of 8
    // (8 batches instead of 13 per hypothesis) measured 0.5 % faster
    constexpr int kGatherBatch = GB > 0 ? GB : (C == 1 && !PK && SPAD == 104) ? 13 : gather_batch(C);
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
    for (int d = d0; d < d1; d++) {
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

// One wave per SIMD: a wave issues a VALU instruction every ~5 clocks whatever it is (tools/ubench_valu.hip),
// so packed fp32 halves the issue slots of the mean-shift pass.  With two or more waves the SIMD is already
// saturated by scalar instructions and packed ones run at half rate.
constexpr bool scan_reg_packed_math(int spad, int c) { return scan_reg_waves(spad, c) == 1; }

template <int SPAD, int C>
__device__ __forceinline__ void scan_reg_rows(const ScanArgs& a, int v, int u, int d0, int d1, Best<C>& best, float* otab)
{
    constexpr bool PK = scan_reg_packed_math(SPAD, C);
    if (a.dmin_vu) {
        scan_reg_body<SPAD, C, true, false, PK>(a, v, u, d0, d1, best, otab);
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
            scan_reg_body<SPAD, C, false, true, PK>(a, v, u, d, e, best, otab);
        else
            scan_reg_body<SPAD, C, true, true, PK>(a, v, u, d, e, best, otab);
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
    RSLF_SCAN_ROW_TILE((scan_reg_rows<SPAD, C>(a, v, u, d0, d1, best, otab)))
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

}  // namespace rslf
