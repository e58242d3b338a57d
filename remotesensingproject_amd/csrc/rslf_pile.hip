// librslf_hip.so, unit 2 of 9: the hot path -- edge confidence (K1), the scan (K2), the selective median (K3) and the
// Depth1DComputer / Depth1DComputer_pile drivers over them.  C-ABI: include/rslf_hip.h.
#include "rslf_internal.hpp"

#include <algorithm>
#include <cmath>

#include "k1_edge.hpp"
#include "k2_scan.hpp"
#include "k2_reg.hpp"
#include "k2_stream.hpp"
#include "k3_median.hpp"

using namespace rslf;

static_assert(sizeof(Partial) == plan::kPartialRecordBytes, "rslf_plan.hpp sizes the record scratch");
static_assert(kScanWaves == plan::kScanWavesPerTile, "rslf_plan.hpp shares the hypotheses out over this many waves");

// ---- hot path -------------------------------------------------------------

extern "C" int rslf_edge_confidence_pile(rslf_ctx* ctx, const rslf_volume* vol, int s, const rslf_params* p,
                                         float* d_Ce_vu, uint8_t* d_Ce_mask_vu) RSLF_API_TRY
{
    if (!ctx || !vol || !d_Ce_vu || !d_Ce_mask_vu)
        return fail(RSLF_ERR_INVALID_ARG, "NULL argument");
    int rc = check_params(p);
    if (rc)
        return rc;
    if (s < 0 || s >= vol->S)
        return fail(RSLF_ERR_INVALID_ARG, "s=%d outside [0,%d)", s, vol->S);
    if (!vol->filled)
        return fail(RSLF_ERR_INVALID_ARG, "volume has not been filled");
    HIP_TRY(hipSetDevice(ctx->device));
    EdgeConsts ec;
    ec.filter_size = p->edge_confidence_filter_size;
    ec.cut_shadows = p->cut_shadows;
    ec.shadow_level = p->shadow_level;
    ec.edge_thr = p->edge_score_threshold;
    const dim3 grid((vol->U + 255) / 256, vol->V);
    if (vol->C == 1)
        hipLaunchKernelGGL(k1_edge_confidence<1>, grid, dim3(256), 0, ctx->stream, view_of(vol), s, ec, d_Ce_vu, d_Ce_mask_vu);
    else
        hipLaunchKernelGGL(k1_edge_confidence<3>, grid, dim3(256), 0, ctx->stream, view_of(vol), s, ec, d_Ce_vu, d_Ce_mask_vu);
    HIP_TRY(hipGetLastError());
    if (p->edge_confidence_opening_size > 1) {   // core.hpp:759-768
        rc = ensure_plane_scratch(ctx, vol->V, vol->U);
        if (rc)
            return rc;
        const MorphElement el = plan::structuring_element(p->edge_confidence_opening_type, p->edge_confidence_opening_size);
        uint8_t* tmp = reinterpret_cast<uint8_t*>(ctx->depth_tmp);   // V*U floats: room for a byte plane
        hipLaunchKernelGGL(k1_morph_pass, grid, dim3(256), 0, ctx->stream, d_Ce_mask_vu, tmp, vol->V, vol->U, el, 0);   // erode
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(k1_morph_pass, grid, dim3(256), 0, ctx->stream, tmp, d_Ce_mask_vu, vol->V, vol->U, el, 1);   // dilate
        HIP_TRY(hipGetLastError());
    }
    return RSLF_OK;
}
RSLF_API_CATCH

// Edge confidence of every view in one launch (the 2-D sweep's first step, core.hpp:918-934)
extern "C" int rslf_edge_confidence_2d(rslf_ctx* ctx, const rslf_volume* vol, const rslf_params* p, float* d_Ce_svu,
                                       uint8_t* d_Ce_mask_svu) RSLF_API_TRY
{
    if (!ctx || !vol || !d_Ce_svu || !d_Ce_mask_svu)
        return fail(RSLF_ERR_INVALID_ARG, "NULL argument");
    const size_t n = (size_t)vol->V * vol->U;
    if (p && p->edge_confidence_opening_size <= 1 && vol->S <= 65535 && vol->V <= 65535) {   // every view in one launch
        int rc = check_params(p);
        if (rc)
            return rc;
        if (!vol->filled)
            return fail(RSLF_ERR_INVALID_ARG, "volume has not been filled");
        HIP_TRY(hipSetDevice(ctx->device));
        EdgeConsts ec;
        ec.filter_size = p->edge_confidence_filter_size;
        ec.cut_shadows = p->cut_shadows;
        ec.shadow_level = p->shadow_level;
        ec.edge_thr = p->edge_score_threshold;
        const dim3 grid((vol->U + 255) / 256, vol->V, vol->S);
        if (vol->C == 1)
            hipLaunchKernelGGL(k1_edge_confidence_views<1>, grid, dim3(256), 0, ctx->stream, view_of(vol), ec, d_Ce_svu, d_Ce_mask_svu);
        else
            hipLaunchKernelGGL(k1_edge_confidence_views<3>, grid, dim3(256), 0, ctx->stream, view_of(vol), ec, d_Ce_svu, d_Ce_mask_svu);
        HIP_TRY(hipGetLastError());
        return RSLF_OK;
    }
    for (int s = 0; s < vol->S; s++) {   // core.hpp:918-934
        int rc = rslf_edge_confidence_pile(ctx, vol, s, p, d_Ce_svu + (size_t)s * n, d_Ce_mask_svu + (size_t)s * n);
        if (rc)
            return rc;
    }
    return RSLF_OK;
}
RSLF_API_CATCH

extern "C" int rslf_selective_median(rslf_ctx* ctx, const rslf_volume* vol, const float* d_src_vu, float* d_dst_vu,
                                     int s_hat, int size, const uint8_t* d_mask_vu, float epsilon) RSLF_API_TRY
{
    if (!ctx || !vol || !d_src_vu || !d_dst_vu || !d_mask_vu)
        return fail(RSLF_ERR_INVALID_ARG, "NULL argument");
    if (d_src_vu == d_dst_vu)
        return fail(RSLF_ERR_INVALID_ARG, "selective median cannot run in place");
    if (size < 0 || size > plan::kMedianMaxSize)
        return fail(RSLF_ERR_INVALID_ARG, "median size %d: must be in [0, %d] (width = (size - 1) / 2, core.hpp:686)", size, plan::kMedianMaxSize);
    if (s_hat < 0 || s_hat >= vol->S)
        return fail(RSLF_ERR_INVALID_ARG, "s_hat=%d outside [0,%d)", s_hat, vol->S);
    HIP_TRY(hipSetDevice(ctx->device));
    const plan::MedianPlan mp = plan::median_plan(size, vol->C);
    const plan::NormThreshold thr = plan::norm_threshold(epsilon);
    const dim3 grid((vol->U + kMedianBlock - 1) / kMedianBlock, vol->V);
    bool launched = false;
#define RSLF_K3_CASE(CC, MODE)                                                                                                   \
    if (!launched && vol->C == CC && mp.mode == MODE) {                                                                           \
        if (mp.lds_bytes > ((size_t)64 << 10))                                                                                    \
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k3_selective_median<CC, MODE>),                           \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)mp.lds_bytes));                         \
        hipLaunchKernelGGL((k3_selective_median<CC, MODE>), grid, dim3(kMedianBlock), mp.lds_bytes, ctx->stream, view_of(vol), d_src_vu, \
                           d_dst_vu, d_mask_vu, s_hat, mp.w, thr);                                                                \
        launched = true;                                                                                                          \
    }
    RSLF_MEDIAN_MODES(RSLF_K3_CASE, 1)
    RSLF_MEDIAN_MODES(RSLF_K3_CASE, 3)
#undef RSLF_K3_CASE
    if (!launched)
        return fail(RSLF_ERR_INTERNAL, "no selective-median kernel for %d channels, mode %d", vol->C, mp.mode);
    HIP_TRY(hipGetLastError());
    return RSLF_OK;
}
RSLF_API_CATCH

// (the slot counts compiled in: RSLF_SPAD_LIST_* in rslf_plan.hpp)
static int launch_scan_reg(int spad, int C, const ScanArgs& a, dim3 grid, hipStream_t stream)
{
#define RSLF_CASE(N)                                                                                        \
    case N:                                                                                                 \
        if (a.packed && a.px_waves)                                                                         \
            hipLaunchKernelGGL((k2_scan_reg_px<N, RSLF_C>), grid, dim3(64 * kScanWaves), 0, stream, a);     \
        else if (a.packed)                                                                                  \
            hipLaunchKernelGGL((k2_scan_reg_packed<N, RSLF_C>), grid, dim3(64 * kScanWaves), 0, stream, a); \
        else                                                                                                \
            hipLaunchKernelGGL((k2_scan_reg<N, RSLF_C>), grid, dim3(64 * kScanWaves), 0, stream, a);        \
        return RSLF_OK;
    if (C == 1) {
        switch (spad) {
#define RSLF_C 1
            RSLF_SPAD_LIST_1CH(RSLF_CASE)
#undef RSLF_C
        default:
            break;
        }
    } else if (C == 3) {
        switch (spad) {
#define RSLF_C 3
            RSLF_SPAD_LIST_3CH(RSLF_CASE)
#undef RSLF_C
        default:
            break;
        }
    }
#undef RSLF_CASE
    return fail(RSLF_ERR_UNSUPPORTED, "no register scan kernel with %d slots x %d channels", spad, C);
}

// Radiances in [0, 1e6] (max(R,0) == R, and the 1e30 sentinel dwarfs them) and an offset table that fits the LDS
static bool scan_takes_lds_kernel(const rslf_volume* vol)
{
    return vol->min_value >= 0.0f && vol->max_value <= 1.0e6f && (size_t)kScanWaves * vol->S * sizeof(float) <= (size_t)48 << 10;
}

bool rslf::scan_takes_stream(const rslf_volume* vol)
{
    return scan_takes_lds_kernel(vol) && plan::pick_spad(vol->S, vol->C) == 0;
}

// Which kernel?  Register variant: S within the compiled slot counts, and radiances in [0, 1e6] so that max(R,0) == R and
// the 1e30 sentinel dwarfs them.  Streaming variant: same precondition, any S whose offset table fits the LDS.
// Otherwise the generic kernel.
struct ScanChoice {
    int spad;          // register kernel's slot count, 0 = not the register kernel
    bool stream_ok;    // an LDS kernel could take this volume
    bool use_stream;
    bool use_chip;     // the on-chip kernel (k2_chip.hpp) -- only ever set by choose_scan's caller-visible conditions
};

// `dense_uniform`: a row-tile launch with one hypothesis grid for all pixels (what the on-chip kernel is written for)
static ScanChoice choose_scan(const rslf_ctx* ctx, int S, int C, bool in_range, int interpolation, bool dense_uniform = false)
{
    ScanChoice c;
    c.spad = in_range ? plan::pick_spad(S, C) : 0;
    c.stream_ok = in_range && (size_t)kScanWaves * S * sizeof(float) <= (size_t)48 << 10;
    if (interpolation != RSLF_INTERP_LINEAR) {      // nearest-neighbour sampling: generic kernel only
        c.spad = 0;
        c.stream_ok = false;
    } else if (ctx->force_scan == 1) {              // parity tests exercise every variant on small cases
        c.spad = 0;
        c.stream_ok = false;
    } else if (ctx->force_scan == 2) {
        c.spad = 0;
    }
    c.use_stream = !c.spad && c.stream_ok;
    // more views than two waves per SIMD hold on chip (RGB, 123 to 220 views: plan::chip_takes): one wave per SIMD with every sample at hand
    c.use_chip = c.use_stream && dense_uniform && plan::chip_takes(S, C) && ctx->force_scan != 2;
    if (c.use_chip)
        c.use_stream = false;
    return c;
}

static plan::ScanRequest scan_request(const rslf_ctx* ctx, int V, int U, int S, int C, int dim_d, const ScanChoice& ch, int precompacted)
{
    plan::ScanRequest rq;
    rq.V = V, rq.U = U, rq.S = S, rq.C = C, rq.dim_d = dim_d;
    rq.spad = ch.spad;
    rq.use_stream = ch.use_stream;
    rq.use_chip = ch.use_chip;
    rq.chip_wave_floats = ch.use_chip ? plan::chip_wave_floats(S, plan::kChipLadder[plan::chip_rung_for(S)]) : 0;
    rq.reg_waves = ch.spad ? scan_reg_waves(ch.spad, C) : 0;
    rq.num_cus = ctx->num_cus;
    rq.ctx_groups = ctx->scan_groups;
    rq.ctx_packed = ctx->scan_packed;
    rq.precompacted = precompacted;
    rq.force_groups = ctx->force_groups;
    rq.force_packed = ctx->force_packed;
    rq.px_mode = ctx->px_mode;
    rq.stream_groups = ctx->stream_groups;
    rq.stream_share = ctx->stream_share;
    rq.stream_lds_bytes = ctx->stream_lds_bytes;
    return rq;
}

// Size the scan's scratch once for pile steps over each of the given scanline counts of an S x U x C volume (the chunks
// of the pipelined host path), assuming radiances in range -- an out-of-range volume runs the generic kernel, whose
// launches take no records.
int rslf::scan_presize(rslf_ctx* ctx, int S, int U, int C, int dim_d, const rslf_params* p, const int* rows, int n_rows)
{
    const ScanChoice ch = choose_scan(ctx, S, C, true, p ? p->interpolation : RSLF_INTERP_LINEAR,
                                      !ctx->scan_packed && ctx->force_packed != 1);
    const bool fused = p && p->edge_confidence_opening_size <= 1 && ctx->force_packed != 1 && !ctx->scan_packed;
    int max_rows = 0;
    size_t recs = 0, tickets = 0;
    for (int i = 0; i < n_rows; i++) {
        max_rows = std::max(max_rows, rows[i]);
        const plan::ScanPlan sp = plan::plan_scan(scan_request(ctx, rows[i], U, S, C, dim_d, ch, fused ? 1 : 0),
                                                  ch.use_stream ? stream_resident_for(S, C) : 0, ch.use_stream ? stream_px_resident_for(S, C) : -1);
        recs = std::max(recs, sp.records);
        tickets = std::max(tickets, sp.tickets);
    }
    if (max_rows < 1)
        return RSLF_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t ticket_cap = ctx->ticket_cap;
    int rc = ensure_plane_scratch(ctx, max_rows, U);
    if (rc == RSLF_OK && recs)
        rc = ensure_group_scratch(ctx, recs, tickets);
    // fresh tickets are zeroed on the context's CURRENT stream; the caller's launches may go to another one
    if (rc == RSLF_OK && ctx->ticket_cap != ticket_cap)
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    return rc;
}

void rslf::fill_stats(rslf_ctx* ctx, unsigned long long tot, int dim_d, rslf_stats* stats)
{
    stats->pixels_scanned = (int64_t)tot;
    stats->units = (int64_t)tot * dim_d;
    stats->scan_kernel = ctx->last_kernel;
    stats->s_pad = ctx->last_spad;
}

extern "C" int rslf_depth_epi_scan(rslf_ctx* ctx, const rslf_volume* vol, const float* d_dmin_vu, const float* d_dmax_vu,
                                   float dmin, float dmax, int dim_d, int s_hat, float* d_Ce_vu, uint8_t* d_Ce_mask_vu,
                                   float* d_Cd_vu, float* d_depth_vu, float* d_rbar_vu, const rslf_params* p,
                                   uint8_t* d_mask_vu, int32_t* d_idx_vu, float* d_score_vu, rslf_stats* stats) RSLF_API_TRY
{
    if (!ctx || !vol || !d_Ce_vu || !d_Ce_mask_vu || !d_Cd_vu || !d_depth_vu || !d_rbar_vu)
        return fail(RSLF_ERR_INVALID_ARG, "NULL argument");
    int rc = check_params(p);
    if (rc)
        return rc;
    if ((d_dmin_vu == nullptr) != (d_dmax_vu == nullptr))
        return fail(RSLF_ERR_INVALID_ARG, "d_dmin_vu and d_dmax_vu must both be given or both be NULL");
    if (dim_d < 2)
        return fail(RSLF_ERR_INVALID_ARG, "dim_d=%d: the hypothesis grid divides by dim_d-1 (core.hpp:548)", dim_d);
    if (s_hat < 0 || s_hat >= vol->S)
        return fail(RSLF_ERR_INVALID_ARG, "s_hat=%d outside [0,%d)", s_hat, vol->S);
    if (!vol->filled)
        return fail(RSLF_ERR_INVALID_ARG, "volume has not been filled");
    HIP_TRY(hipSetDevice(ctx->device));
    rc = ensure_plane_scratch(ctx, vol->V, vol->U);
    if (rc)
        return rc;

    const size_t n = (size_t)vol->V * vol->U;
    hipStream_t st = ctx->stream;
    if (d_idx_vu)
        HIP_TRY(hipMemsetAsync(d_idx_vu, 0xFF, n * sizeof(int32_t), st));   // -1
    if (d_score_vu)
        HIP_TRY(hipMemsetAsync(d_score_vu, 0, n * sizeof(float), st));
    const int precompacted = ctx->precompacted;   // 1: rslf_depth1d_pile_run's K1 left row lists and total; 2: packed list (sweep)
    ctx->precompacted = 0;
    if (!ctx->keep_total && !precompacted)
        HIP_TRY(hipMemsetAsync(ctx->total, 0, sizeof(unsigned long long), st));

    const bool dense_uniform = !d_dmin_vu && !ctx->scan_packed && ctx->force_packed != 1 && precompacted != 2;
    const ScanChoice ch = choose_scan(ctx, vol->S, vol->C, vol->min_value >= 0.0f && vol->max_value <= 1.0e6f, p->interpolation,
                                      dense_uniform);
    const int spad = ch.spad;
    const bool stream_ok = ch.stream_ok, use_stream = ch.use_stream, use_chip = ch.use_chip;
    // Launch shape (rslf_plan.hpp, plan_scan): hypothesis groups per tile, packed or row tiles, 63- or 64-entry tiles, row
    // blocks and records of grouped launches, the streaming kernel's LDS split -- pure host logic, unit-tested on the CPU.
    const plan::ScanRequest rq = scan_request(ctx, vol->V, vol->U, vol->S, vol->C, dim_d, ch, precompacted);
    const plan::ScanPlan sp = plan::plan_scan(rq, use_stream ? stream_resident_for(vol->S, vol->C) : 0,
                                              use_stream ? stream_px_resident_for(vol->S, vol->C) : -1);
    const int groups = sp.groups;
    const bool packed = sp.packed;

    int* packed_n = reinterpret_cast<int*>(ctx->total + 1);
    if (precompacted) {
        // nothing to compact
    } else if (packed) {
        if (!ctx->packed_n_clean)
            HIP_TRY(hipMemsetAsync(packed_n, 0, sizeof(int), st));
        hipLaunchKernelGGL(k_compact_mask_packed, dim3(vol->V), dim3(256), 0, st, d_Ce_mask_vu, d_mask_vu, vol->U, ctx->list,
                           ctx->count, ctx->total, packed_n, ctx->count + ctx->count_cap);
    } else {
        hipLaunchKernelGGL(k_compact_mask, dim3(vol->V), dim3(256), 0, st, d_Ce_mask_vu, d_mask_vu, vol->U, ctx->list,
                           ctx->count, ctx->total);
    }
    HIP_TRY(hipGetLastError());

    ScanArgs a;
    a.vol = view_of(vol);
    a.list = ctx->list;
    a.count = ctx->count;
    a.dmin_vu = d_dmin_vu;
    a.dmax_vu = d_dmax_vu;
    a.dmin = dmin;
    a.dmax = dmax;
    a.dim_d = dim_d;
    a.s_hat = s_hat;
    a.k = make_scan_consts(p);
    a.Ce = d_Ce_vu;
    a.Ce_mask = d_Ce_mask_vu;
    a.Cd = d_Cd_vu;
    a.depth = d_depth_vu;
    a.rbar = d_rbar_vu;
    a.idx = d_idx_vu;
    a.score = d_score_vu;
    a.tile_w = sp.tile_w;                 // the streaming kernel's row tiles leave lane 63 to its neighbour's right tap (DENSE)
    a.tiles_per_row = sp.tiles_per_row;
    a.stream_frac_max = plan::stream_frac_max(vol->U);
    a.packed = packed ? 1 : 0;
    a.packed_n = packed_n;
    a.packed_adapt = sp.packed_adapt ? 1 : 0;
    a.px_waves = sp.px_waves;
    a.stream_park = sp.stream_park;
    a.stream_wave_floats = sp.stream_wave_floats;
    a.partial = nullptr;
    a.ticket = nullptr;
    a.v0 = 0;
    a.groups = groups;
    a.rowbase = nullptr;
    a.row_min = 0;
    // Row split (sparse visits of stream-class volumes): the rows of the packed list that hold many pixels go as ROW tiles --
    // the streaming kernel's dense form, reading its tiles straight from the packed list -- and the pixel-per-wave launch
    // takes the rest.  Both are queued; which rows each scans is settled on the device, from the rows' counts.
    const bool row_split = packed && use_stream && sp.px_waves > 0 && ctx->row_split != 0 && precompacted != 1;
    plan::ScanPlan spr = sp;
    if (row_split) {
        plan::ScanRequest rr = rq;
        rr.ctx_packed = false;
        rr.ctx_groups = 1;
        rr.precompacted = 1;      // the lists are in place
        rr.force_packed = 0;
        if (rr.stream_groups <= 0)
            rr.stream_groups = plan::kStreamGroups;   // (sparse rows: the groups are the launch's parallelism, not the dense rule's few)
        spr = plan::plan_scan(rr, stream_resident_for(vol->S, vol->C));
        a.row_min = ctx->row_split > 1 ? ctx->row_split : plan::kRowSplitMin;   // (hook: 1 = the default threshold, larger = that many pixels)
    }
    // Grouped launches leave one 32-byte record per (tile, group, lane) for the tile's last group to merge (k2_scan.hpp)
    const int rows_per_launch = sp.rows_per_launch;
    if (groups > 1) {
        rc = ensure_group_scratch(ctx, sp.records, sp.tickets);
        if (rc)
            return rc;
        a.partial = ctx->scan_partial;
        a.ticket = ctx->scan_ticket;
    }
    if (row_split && spr.groups > 1) {
        rc = ensure_group_scratch(ctx, spr.records, spr.tickets);
        if (rc)
            return rc;
    }
    const size_t lds = sp.lds_bytes;
    // (more than the 64 KiB a kernel gets without asking: set on the instantiation about to be launched, below)
    HIP_TRY(hipGetLastError());   // anything an earlier enqueue left behind is not this launch's fault
    ctx->last_spad = spad;
    ctx->last_kernel = spad ? (sp.px_waves ? RSLF_SCAN_REG_PX : RSLF_SCAN_REG)
                       : use_chip ? RSLF_SCAN_CHIP
                       : stream_ok ? (sp.px_waves ? RSLF_SCAN_STREAM_PX : RSLF_SCAN_STREAM) : RSLF_SCAN_GENERIC;
    // The events that time K2 are marker packets of their own: ~5.6 us each before the next kernel starts (measured,
    // tools/probe_gaps.py) -- nothing beside a 66 ms scan, a tenth of a sweep's sparse visit.  A sweep times its first
    // (dense) visit only.
    const bool timed = !ctx->sweep_open || ctx->sweep_first;
    hipEvent_t pool0 = nullptr, pool1 = nullptr;
    if (ctx->time_all && ctx->ev_used + 2 <= ((size_t)1 << 16)) {   // every launch sequence gets a pair of its own (rslf_scan_time_total_ms;
                                                                      // 32 768 untimed-for launches are the pool's end: later ones go untimed)
        while (ctx->ev_pool.size() < ctx->ev_used + 2) {
            hipEvent_t e = nullptr;
            HIP_TRY(hipEventCreate(&e));
            ctx->ev_pool.push_back(e);
        }
        pool0 = ctx->ev_pool[ctx->ev_used];
        pool1 = ctx->ev_pool[ctx->ev_used + 1];
        ctx->ev_used += 2;
        HIP_TRY(hipEventRecord(pool0, st));
    }
    if (timed)
        HIP_TRY(hipEventRecord(ctx->ev0, st));
    if (row_split) {   // the rows with many pixels, as row tiles of the packed list
        ScanArgs ar = a;
        ar.packed = 0;
        ar.px_waves = 0;
        ar.packed_adapt = 0;
        ar.rowbase = ctx->count + ctx->count_cap;
        ar.groups = spr.groups;
        ar.tile_w = spr.tile_w;
        ar.tiles_per_row = spr.tiles_per_row;
        ar.stream_park = spr.stream_park;
        ar.stream_wave_floats = spr.stream_wave_floats;
        ar.partial = spr.groups > 1 ? ctx->scan_partial : nullptr;
        ar.ticket = spr.groups > 1 ? ctx->scan_ticket : nullptr;
        for (int v0 = 0; v0 < vol->V; v0 += spr.rows_per_launch) {
            const int rows = std::min(spr.rows_per_launch, vol->V - v0);
            const long long tiles = (long long)rows * ar.tiles_per_row;
            if (tiles * ar.groups > (long long)1 << 30)
                return fail(RSLF_ERR_UNSUPPORTED, "%lld tiles x %d groups exceeds the grid limit", tiles, ar.groups);
            ar.v0 = v0;
            ar.logical_blocks = (int)(tiles * ar.groups);
            ar.per_xcd = ((rows + 7) / 8) * ar.tiles_per_row * ar.groups;
            const dim3 rgrid((unsigned)(ar.per_xcd * 8));
            hipError_t rattr = hipSuccess;
            auto launch_rows = [&](auto kernel) {
                rattr = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)ctx->stream_lds_bytes);
                if (rattr == hipSuccess)
                    hipLaunchKernelGGL(kernel, rgrid, dim3(64 * kScanWaves), spr.lds_bytes, st, ar);
            };
            if (vol->C == 1)
                stream_kernel_for<1, false>(spr.stream_nres, launch_rows);
            else
                stream_kernel_for<3, false>(spr.stream_nres, launch_rows);
            HIP_TRY(rattr);
            HIP_TRY(hipGetLastError());
        }
    }
    for (int v0 = 0; v0 < vol->V; v0 += rows_per_launch) {
        const int rows = std::min(rows_per_launch, vol->V - v0);
        // row tiles: ceil(U/64) per scanline; packed tiles: at most ceil(V*U/64), the device knows how many
        const long long tiles = packed ? (long long)((n + 63) / 64) : (long long)rows * a.tiles_per_row;
        if (tiles * groups > (long long)1 << 30)
            return fail(RSLF_ERR_UNSUPPORTED, "%lld tiles x %d groups exceeds the grid limit", tiles, groups);
        a.v0 = v0;
        a.logical_blocks = (int)(tiles * groups);   // `groups` workgroups per tile, their waves split the hypotheses
        // (row tiles: the scanlines are dealt to the XCDs in turn, every XCD ceil(rows / 8) of them: xcd_logical_block_rows)
        a.per_xcd = packed ? (a.logical_blocks + 7) / 8 : ((rows + 7) / 8) * a.tiles_per_row * groups;
        // packed: a fixed grid strides over the items (k2_scan.hpp); ~4 workgroups per CU cover any occupancy
        // (the pixel-per-wave kernel's items are 4 / px_waves pixels each)
        const long long px_items = sp.px_waves ? ((long long)n * sp.px_waves + kScanWaves - 1) / kScanWaves : 0;
        const dim3 grid(sp.px_waves ? (unsigned)std::min<long long>(px_items, 2048)
                        : packed    ? (unsigned)std::min<long long>(tiles * groups, 1024)
                                    : (unsigned)(a.per_xcd * 8));
        if (spad) {
            rc = launch_scan_reg(spad, vol->C, a, grid, st);
            if (rc)
                return rc;
        } else if (use_chip) {
            rc = launch_scan_chip(a, grid, lds, st);   // rslf_chip_a.hip: the rung that holds this view count
            if (rc)
                return rc;
        } else if (use_stream) {
            // one instantiation per channel count, resident-prefix length and launch form (k2_stream.hpp)
            hipError_t attr = hipSuccess;
            auto launch = [&](auto kernel) {
                attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)ctx->stream_lds_bytes);   // (more than the 64 KiB a kernel gets without asking)
                if (attr == hipSuccess)
                    hipLaunchKernelGGL(kernel, grid, dim3(64 * kScanWaves), lds, st, a);
            };
            if (sp.px_waves) {
                if (vol->C == 1)
                    stream_px_kernel_for<1>(sp.stream_nres, launch);
                else
                    stream_px_kernel_for<3>(sp.stream_nres, launch);
            } else if (packed) {
                if (vol->C == 1)
                    stream_kernel_for<1, true>(sp.stream_nres, launch);
                else
                    stream_kernel_for<3, true>(sp.stream_nres, launch);
            } else {
                if (vol->C == 1)
                    stream_kernel_for<1, false>(sp.stream_nres, launch);
                else
                    stream_kernel_for<3, false>(sp.stream_nres, launch);
            }
            HIP_TRY(attr);
        } else if (vol->C == 1) {
            hipLaunchKernelGGL(k2_scan_generic<1>, grid, dim3(64 * kScanWaves), 0, st, a);
        } else {
            hipLaunchKernelGGL(k2_scan_generic<3>, grid, dim3(64 * kScanWaves), 0, st, a);
        }
        HIP_TRY(hipGetLastError());   // grouped launches merge their records themselves (scan_epilogue): no combine launch
        if (packed)
            break;   // one launch covers the packed list
    }
    if (timed) {
        HIP_TRY(hipEventRecord(ctx->ev1, st));
        ctx->ev_valid = true;
    }
    if (pool1)
        HIP_TRY(hipEventRecord(pool1, st));

    if (stats) {
        unsigned long long tot = 0;
        HIP_TRY(hipMemcpyAsync(&tot, ctx->total, sizeof(tot), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        fill_stats(ctx, tot, dim_d, stats);
    }
    return RSLF_OK;
}
RSLF_API_CATCH

extern "C" int rslf_kernel_columns_pile(rslf_ctx* ctx, const rslf_volume* vol, const float* d_dmin_vu, const float* d_dmax_vu,
                                        float dmin, float dmax, int dim_d, int s_hat, const rslf_params* p,
                                        const int32_t* d_idx_vu, float* d_K_vsu) RSLF_API_TRY
{
    if (!ctx || !vol || !d_idx_vu || !d_K_vsu)
        return fail(RSLF_ERR_INVALID_ARG, "NULL argument");
    int rc = check_params(p);
    if (rc)
        return rc;
    if ((d_dmin_vu == nullptr) != (d_dmax_vu == nullptr))
        return fail(RSLF_ERR_INVALID_ARG, "d_dmin_vu and d_dmax_vu must both be given or both be NULL");
    if (dim_d < 2)
        return fail(RSLF_ERR_INVALID_ARG, "dim_d=%d: the hypothesis grid divides by dim_d-1 (core.hpp:548)", dim_d);
    if (s_hat < 0 || s_hat >= vol->S)
        return fail(RSLF_ERR_INVALID_ARG, "s_hat=%d outside [0,%d)", s_hat, vol->S);
    if (!vol->filled)
        return fail(RSLF_ERR_INVALID_ARG, "volume has not been filled");
    HIP_TRY(hipSetDevice(ctx->device));
    ScanArgs a = {};
    a.vol = view_of(vol);
    a.dmin_vu = d_dmin_vu;
    a.dmax_vu = d_dmax_vu;
    a.dmin = dmin;
    a.dmax = dmax;
    a.dim_d = dim_d;
    a.s_hat = s_hat;
    a.k = make_scan_consts(p);
    a.groups = 1;
    const dim3 grid((vol->U + 255) / 256, vol->V);
    if (vol->C == 1)
        hipLaunchKernelGGL(k2_kernel_column<1>, grid, dim3(256), 0, ctx->stream, a, d_idx_vu, d_K_vsu);
    else
        hipLaunchKernelGGL(k2_kernel_column<3>, grid, dim3(256), 0, ctx->stream, a, d_idx_vu, d_K_vsu);
    HIP_TRY(hipGetLastError());
    return RSLF_OK;
}
RSLF_API_CATCH

extern "C" int rslf_depth_epi_pile(rslf_ctx* ctx, const rslf_volume* vol, const float* d_dmin_vu, const float* d_dmax_vu,
                                   float dmin, float dmax, int dim_d, int s_hat, float* d_Ce_vu, uint8_t* d_Ce_mask_vu,
                                   float* d_Cd_vu, float* d_depth_vu, float* d_rbar_vu, const rslf_params* p,
                                   uint8_t* d_mask_vu, int32_t* d_idx_vu, float* d_score_vu, float* d_depth_raw_vu,
                                   rslf_stats* stats) RSLF_API_TRY
{
    if (!ctx || !vol || !d_depth_vu)
        return fail(RSLF_ERR_INVALID_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = ensure_plane_scratch(ctx, vol->V, vol->U);
    if (rc)
        return rc;
    const size_t n = (size_t)vol->V * vol->U;
    hipStream_t st = ctx->stream;
    // core.hpp:799-854: the scan of every EPI writes the RAW disparities -- into the caller's raw plane if one is
    // wanted (over the zeros best_depth starts from, dc.hpp:507), else into scratch, where no background is needed:
    // the median reads the raw plane at mask pixels only, and every mask pixel has been written by the scan ...
    // With a caller's scan mask, mask pixels that are NOT scanned now keep the disparity the plane came in with
    // (a_best_depth_v_u is in/out, core.hpp:305), and the median reads them: the raw plane then starts as a copy.
    float* raw = d_depth_raw_vu ? d_depth_raw_vu : ctx->depth_tmp;
    if (d_mask_vu)
        HIP_TRY(hipMemcpyAsync(raw, d_depth_vu, n * sizeof(float), hipMemcpyDeviceToDevice, st));
    else if (d_depth_raw_vu)
        HIP_TRY(hipMemsetAsync(d_depth_raw_vu, 0, n * sizeof(float), st));
    rc = rslf_depth_epi_scan(ctx, vol, d_dmin_vu, d_dmax_vu, dmin, dmax, dim_d, s_hat, d_Ce_vu, d_Ce_mask_vu, d_Cd_vu, raw,
                             d_rbar_vu, p, d_mask_vu, d_idx_vu, d_score_vu, nullptr);
    if (rc)
        return rc;
    // ... then core.hpp:881-892: median over the EDGE mask, result replaces best_depth -- written straight into the
    // caller's plane (every pixel: 0 where the mask is 0, core.hpp:678-679), so no plane is copied
    rc = rslf_selective_median(ctx, vol, raw, d_depth_vu, s_hat, p->median_filter_size, d_Ce_mask_vu, p->median_filter_epsilon);
    if (rc)
        return rc;

    if (stats) {
        unsigned long long tot = 0;
        HIP_TRY(hipMemcpyAsync(&tot, ctx->total, sizeof(tot), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        fill_stats(ctx, tot, dim_d, stats);
    }
    return RSLF_OK;
}
RSLF_API_CATCH

extern "C" int rslf_depth1d_pile_run(rslf_ctx* ctx, const rslf_volume* vol, float dmin, float dmax, int dim_d, int s_hat,
                                     const rslf_params* p, float* d_Ce_vu, uint8_t* d_Ce_mask_vu, float* d_Cd_vu,
                                     float* d_depth_vu, float* d_rbar_vu, int32_t* d_idx_vu, float* d_score_vu,
                                     float* d_depth_raw_vu, rslf_stats* stats) RSLF_API_TRY
{
    if (!ctx || !vol || !d_Ce_vu || !d_Ce_mask_vu || !d_Cd_vu || !d_depth_vu || !d_rbar_vu)
        return fail(RSLF_ERR_INVALID_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(ctx->device));
    s_hat = plan::resolve_s_hat(s_hat, vol->S);
    const size_t n = (size_t)vol->V * vol->U;
    hipStream_t st = ctx->stream;
    // dc.hpp:501-510 (C_e and C_d are uninitialised there; zero is the intended start)
    HIP_TRY(hipMemsetAsync(d_Ce_vu, 0, n * sizeof(float), st));
    HIP_TRY(hipMemsetAsync(d_Cd_vu, 0, n * sizeof(float), st));
    HIP_TRY(hipMemsetAsync(d_depth_vu, 0, n * sizeof(float), st));
    HIP_TRY(hipMemsetAsync(d_rbar_vu, 0, n * vol->C * sizeof(float), st));
    int rc = check_params(p);
    if (rc)
        return rc;
    // dc.hpp:538 + the findNonZero of dc.hpp:547's callee (core.hpp:513-516) in ONE launch when nothing sits between
    // them: no opening of the mask (core.hpp:759-768) and row tiles.  A pile step is then three launches -- edge
    // confidence + compaction, scan, selective median -- and no plane is copied.
    const bool fuse = p->edge_confidence_opening_size <= 1 && ctx->force_packed != 1 && !ctx->scan_packed && vol->filled &&
                      (size_t)vol->V * vol->U <= (size_t)INT32_MAX;
    if (fuse) {
        rc = ensure_plane_scratch(ctx, vol->V, vol->U);
        if (rc)
            return rc;
        EdgeConsts ec;
        ec.filter_size = p->edge_confidence_filter_size;
        ec.cut_shadows = p->cut_shadows;
        ec.shadow_level = p->shadow_level;
        ec.edge_thr = p->edge_score_threshold;
        if (!ctx->keep_total)
            HIP_TRY(hipMemsetAsync(ctx->total, 0, sizeof(unsigned long long), st));
        if (vol->C == 1)
            hipLaunchKernelGGL(k1_edge_confidence_compact<1>, dim3(vol->V), dim3(256), 0, st, view_of(vol), s_hat, ec, d_Ce_vu,
                               d_Ce_mask_vu, ctx->list, ctx->count, ctx->total);
        else
            hipLaunchKernelGGL(k1_edge_confidence_compact<3>, dim3(vol->V), dim3(256), 0, st, view_of(vol), s_hat, ec, d_Ce_vu,
                               d_Ce_mask_vu, ctx->list, ctx->count, ctx->total);
        HIP_TRY(hipGetLastError());
        ctx->precompacted = 1;
    } else {
        rc = rslf_edge_confidence_pile(ctx, vol, s_hat, p, d_Ce_vu, d_Ce_mask_vu);   // dc.hpp:538
        if (rc)
            return rc;
    }
    rc = rslf_depth_epi_pile(ctx, vol, nullptr, nullptr, dmin, dmax, dim_d, s_hat, d_Ce_vu, d_Ce_mask_vu, d_Cd_vu,   // dc.hpp:547
                             d_depth_vu, d_rbar_vu, p, nullptr, d_idx_vu, d_score_vu, d_depth_raw_vu, stats);
    ctx->precompacted = 0;
    return rc;
}
RSLF_API_CATCH

extern "C" int rslf_depth1d_run(rslf_ctx* ctx, const rslf_volume* vol, float dmin, float dmax, int dim_d, int s_hat,
                                const rslf_params* p, float* d_Ce_vu, uint8_t* d_Ce_mask_vu, float* d_Cd_vu, float* d_depth_vu,
                                float* d_rbar_vu, int32_t* d_idx_vu, float* d_score_vu, rslf_stats* stats) RSLF_API_TRY
{
    if (!ctx || !vol || !d_Ce_vu || !d_Ce_mask_vu || !d_Cd_vu || !d_depth_vu || !d_rbar_vu)
        return fail(RSLF_ERR_INVALID_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(ctx->device));
    s_hat = plan::resolve_s_hat(s_hat, vol->S);   // dc.hpp:303-311
    const size_t n = (size_t)vol->V * vol->U;
    hipStream_t st = ctx->stream;
    // dc.hpp:313-322: zero-initialised outputs
    HIP_TRY(hipMemsetAsync(d_Ce_vu, 0, n * sizeof(float), st));
    HIP_TRY(hipMemsetAsync(d_Cd_vu, 0, n * sizeof(float), st));
    HIP_TRY(hipMemsetAsync(d_depth_vu, 0, n * sizeof(float), st));
    HIP_TRY(hipMemsetAsync(d_rbar_vu, 0, n * vol->C * sizeof(float), st));
    int rc = rslf_edge_confidence_pile(ctx, vol, s_hat, p, d_Ce_vu, d_Ce_mask_vu);   // dc.hpp:347
    if (rc)
        return rc;
    return rslf_depth_epi_scan(ctx, vol, nullptr, nullptr, dmin, dmax, dim_d, s_hat, d_Ce_vu, d_Ce_mask_vu, d_Cd_vu,   // dc.hpp:356
                               d_depth_vu, d_rbar_vu, p, nullptr, d_idx_vu, d_score_vu, stats);
}
RSLF_API_CATCH

extern "C" int rslf_depth1d_pile_run_host(rslf_ctx* ctx, const rslf_volume* vol, float dmin, float dmax, int dim_d, int s_hat,
                                          const rslf_params* p, float* h_Ce_vu, uint8_t* h_Ce_mask_vu, float* h_Cd_vu,
                                          float* h_depth_vu, float* h_rbar_vu, int32_t* h_idx_vu, float* h_score_vu,
                                          float* h_depth_raw_vu, rslf_stats* stats) RSLF_API_TRY
{
    if (!ctx || !vol)
        return fail(RSLF_ERR_INVALID_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t n = (size_t)vol->V * vol->U;
    // one device block: Ce, Cd, depth, raw, score | rbar (n*C) | idx | mask
    const size_t f_planes = 5 + (size_t)vol->C;
    const size_t bytes = n * (f_planes * sizeof(float) + sizeof(int32_t) + 1);
    char* blk = nullptr;
    hipError_t e = hipMalloc(&blk, bytes);
    if (e != hipSuccess)
        return fail(RSLF_ERR_ALLOC, "hipMalloc(%zu) for result planes failed: %s", bytes, hipGetErrorString(e));
    float* d_Ce = (float*)blk;
    float* d_Cd = d_Ce + n;
    float* d_depth = d_Cd + n;
    float* d_raw = d_depth + n;
    float* d_score = d_raw + n;
    float* d_rbar = d_score + n;
    int32_t* d_idx = (int32_t*)(d_rbar + n * vol->C);
    uint8_t* d_mask = (uint8_t*)(d_idx + n);
    int rc = rslf_depth1d_pile_run(ctx, vol, dmin, dmax, dim_d, s_hat, p, d_Ce, d_mask, d_Cd, d_depth, d_rbar, d_idx, d_score,
                                   d_raw, nullptr);
    hipStream_t st = ctx->stream;
    auto pull = [&](void* h, const void* d, size_t b) -> hipError_t {
        return h ? hipMemcpyAsync(h, d, b, hipMemcpyDeviceToHost, st) : hipSuccess;
    };
    if (rc == RSLF_OK) {
        hipError_t ce = pull(h_Ce_vu, d_Ce, n * 4);
        if (ce == hipSuccess) ce = pull(h_Ce_mask_vu, d_mask, n);
        if (ce == hipSuccess) ce = pull(h_Cd_vu, d_Cd, n * 4);
        if (ce == hipSuccess) ce = pull(h_depth_vu, d_depth, n * 4);
        if (ce == hipSuccess) ce = pull(h_rbar_vu, d_rbar, n * 4 * vol->C);
        if (ce == hipSuccess) ce = pull(h_idx_vu, d_idx, n * 4);
        if (ce == hipSuccess) ce = pull(h_score_vu, d_score, n * 4);
        if (ce == hipSuccess) ce = pull(h_depth_raw_vu, d_raw, n * 4);
        if (ce == hipSuccess) ce = hipStreamSynchronize(st);
        if (ce != hipSuccess)
            rc = fail(RSLF_ERR_HIP, "result download failed: %s", hipGetErrorString(ce));
    } else {
        (void)hipStreamSynchronize(st);
    }
    if (rc == RSLF_OK && stats) {
        unsigned long long tot = 0;
        if (hipMemcpy(&tot, ctx->total, sizeof(tot), hipMemcpyDeviceToHost) == hipSuccess) {
            stats->pixels_scanned = (int64_t)tot;
            stats->units = (int64_t)tot * dim_d;
            stats->scan_kernel = ctx->last_kernel;
            stats->s_pad = ctx->last_spad;
        }
    }
    (void)hipFree(blk);
    return rc;
}
RSLF_API_CATCH

extern "C" int rslf_scan_time_total_ms(rslf_ctx* ctx, float* ms, int* launches) RSLF_API_TRY
{
    if (!ctx || !ms)
        return fail(RSLF_ERR_INVALID_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(ctx->device));
    double sum = 0.0;
    for (size_t i = 0; i + 1 < ctx->ev_used; i += 2) {
        float t = 0.0f;
        HIP_TRY(hipEventSynchronize(ctx->ev_pool[i + 1]));
        HIP_TRY(hipEventElapsedTime(&t, ctx->ev_pool[i], ctx->ev_pool[i + 1]));
        sum += (double)t;
    }
    *ms = (float)sum;
    if (launches)
        *launches = (int)(ctx->ev_used / 2);
    ctx->ev_used = 0;
    return RSLF_OK;
}
RSLF_API_CATCH

extern "C" int rslf_last_scan_kernel_ms(rslf_ctx* ctx, float* ms) RSLF_API_TRY
{
    if (!ctx || !ms)
        return fail(RSLF_ERR_INVALID_ARG, "NULL argument");
    if (!ctx->ev_valid)
        return fail(RSLF_ERR_INVALID_ARG, "no scan kernel has been launched on this context");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipEventSynchronize(ctx->ev1));
    HIP_TRY(hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
    return RSLF_OK;
}
RSLF_API_CATCH
