// librslf_hip.so, unit 6 of 9: the 2-D sweep (Depth2DComputer::run, core.hpp:901-1133) -- one visit per view, scan (unit 2),
// selective median + claims, apply + the next visit's compaction (K4).  C-ABI: include/rslf_hip.h.
#include "rslf_internal.hpp"

#include <algorithm>
#include <cmath>

#include "k3_median.hpp"
#include "k4_propagate.hpp"

using namespace rslf;

// ---- "next" row: the 2-D sweep ----------------------------------------------

// Two capacities: winner / running mask hold S*V*U entries, the median plane V*U.  (A single S*V*U
// capacity once let a later volume with fewer views but larger planes overrun the plane: found by
// tools/fuzz_sweep.py.)
static int ensure_sweep_scratch(rslf_ctx* ctx, const rslf_volume* vol)
{
    const size_t n = (size_t)vol->S * vol->V * vol->U;
    if (n > ctx->sweep_cap) {
        (void)hipFree(ctx->winner);
        (void)hipFree(ctx->sweep_mask);
        ctx->winner = nullptr;
        ctx->sweep_mask = nullptr;
        ctx->sweep_cap = 0;
        HIP_TRY(hipMalloc(&ctx->winner, n * sizeof(int)));
        HIP_TRY(hipMalloc(&ctx->sweep_mask, n));
        ctx->sweep_cap = n;
        // every claim pass is undone by its apply pass, so one fill lasts
        HIP_TRY(hipMemsetAsync(ctx->winner, 0x7F, n * sizeof(int), ctx->stream));
    }
    const size_t flags = (size_t)vol->S * vol->V * ((vol->U + 255) / 256);
    if (flags > ctx->dirty_cap) {
        (void)hipFree(ctx->dirty);
        (void)hipFree(ctx->remain);
        ctx->dirty = nullptr;
        ctx->remain = nullptr;
        ctx->dirty_cap = 0;
        HIP_TRY(hipMalloc(&ctx->remain, flags * sizeof(int)));
        HIP_TRY(hipMalloc(&ctx->dirty, flags));
        ctx->dirty_cap = flags;
        HIP_TRY(hipMemsetAsync(ctx->dirty, 0, flags, ctx->stream));   // every apply pass leaves them at 0 again
    }
    const size_t plane = (size_t)vol->V * vol->U;
    if (plane > ctx->sweep_plane_cap) {
        (void)hipFree(ctx->filtered);
        ctx->filtered = nullptr;
        ctx->sweep_plane_cap = 0;
        HIP_TRY(hipMalloc(&ctx->filtered, plane * sizeof(float)));
        ctx->sweep_plane_cap = plane;
    }
    return RSLF_OK;
}

// The sweep one visit at a time (rslf_sweep_*), and rslf_depth_epi_2d on top of it.  The launch shape of the visits
// (hypothesis groups, packed tiles, running total) is context state the scan reads: rslf_sweep_end restores it, and after
// an error the winners are refilled on the next sweep (a claim pass whose apply never ran leaves them set).
// The order of the visits (core.hpp:981-990): plan::sweep_order.
static void sweep_close(rslf_ctx* ctx, bool ok)
{
    ctx->keep_total = false;
    ctx->scan_groups = 1;
    ctx->scan_packed = false;
    ctx->packed_n_clean = false;
    ctx->precompacted = 0;
    ctx->sweep_expect = -1;
    if (!ok) {
        ctx->sweep_cap = 0;   // claims without their apply pass may be left behind: fresh winners and flags next time
        ctx->dirty_cap = 0;
    }
    ctx->sweep_open = false;
}

extern "C" int rslf_sweep_begin(rslf_ctx* ctx, const rslf_volume* vol, const uint8_t* d_Ce_mask_svu, uint8_t* d_scan_mask_svu,
                                int dim_d, int v_lo, int v_hi) RSLF_API_TRY
{
    if (!ctx || !vol || !d_Ce_mask_svu)
        return fail(RSLF_ERR_INVALID_ARG, "NULL argument");
    if (v_lo < 0 || v_hi > vol->V || v_lo >= v_hi)
        return fail(RSLF_ERR_INVALID_ARG, "active scanlines [%d, %d) outside the volume's %d", v_lo, v_hi, vol->V);
    HIP_TRY(hipSetDevice(ctx->device));
    if (ctx->sweep_open)
        sweep_close(ctx, false);   // a sweep left open by a caller's error path
    int rc = ensure_sweep_scratch(ctx, vol);
    if (rc)
        return rc;
    const int S = vol->S, V = vol->V, U = vol->U;
    const size_t n = (size_t)V * U;
    hipStream_t st = ctx->stream;
    uint8_t* mask_svu = d_scan_mask_svu ? d_scan_mask_svu : ctx->sweep_mask;
    // core.hpp:958-965: running masks start as clones of the edge masks ...
    HIP_TRY(hipMemcpyAsync(mask_svu, d_Ce_mask_svu, (size_t)S * n, hipMemcpyDeviceToDevice, st));
    // ... except on halo scanlines (a sharded sweep): never scanned, never painted here -- their owner does both
    if (v_lo > 0)
        HIP_TRY(hipMemset2DAsync(mask_svu, n, 0, (size_t)v_lo * U, S, st));
    if (v_hi < V)
        HIP_TRY(hipMemset2DAsync(mask_svu + (size_t)v_hi * U, n, 0, (size_t)(V - v_hi) * U, S, st));
    HIP_TRY(hipMemsetAsync(ctx->total, 0, sizeof(unsigned long long), st));
    {   // pixels of the running masks per 256-column segment: what lets the claims of the later visits skip most views
        const long long rows = (long long)S * V;
        const long long items = rows * ((U + 255) / 256);
        if (items > (1ll << 31) - 1)
            return fail(RSLF_ERR_UNSUPPORTED, "%d views x %d scanlines x %d columns: too large for one counting launch", S, V, U);
        hipLaunchKernelGGL(k4_count_segments, dim3((unsigned)items), dim3(256), 0, st, mask_svu, rows, U, ctx->remain);
        HIP_TRY(hipGetLastError());
    }
    {   // the sparse visits' records, sized before the first visit (no allocation in the middle of the sequence).
        // (The same choice of kernel as rslf_depth_epi_scan makes for linear interpolation without debug hooks; should it
        // differ, that call sizes the records itself.)
        size_t recs = 0, tickets = 0;
        plan::sweep_record_plan(n, dim_d, scan_takes_stream(vol), &recs, &tickets);
        if (scan_takes_stream(vol)) {   // the row split of the sparse visits: row tiles of the streaming kernel's dense form, in row blocks
            const size_t tiles_per_row = (size_t)std::max(1, (U + 61) / 63);
            const size_t per_row = tiles_per_row * plan::kStreamGroups * 64;
            const size_t rows = std::min<size_t>((size_t)V, std::max<size_t>(1, plan::kPartialBudget / (per_row * plan::kPartialRecordBytes)));
            recs = std::max(recs, rows * per_row);
            tickets = std::max(tickets, rows * tiles_per_row);
        }
        if (recs) {
            rc = ensure_group_scratch(ctx, recs, tickets);
            if (rc)
                return rc;
        }
    }
    ctx->keep_total = true;
    ctx->sweep_open = true;
    ctx->sweep_first = true;
    ctx->sweep_mask_run = mask_svu;
    ctx->sweep_expect = plan::sweep_order(S)[0];
    ctx->precompacted = 0;
    return RSLF_OK;
}
RSLF_API_CATCH

extern "C" int rslf_sweep_visit_scan(rslf_ctx* ctx, const rslf_volume* vol, const float* d_dmin_svu, const float* d_dmax_svu,
                                     float dmin, float dmax, int dim_d, int s_hat, float* d_Ce_svu, uint8_t* d_Ce_mask_svu,
                                     float* d_Cd_svu, float* d_depth_svu, float* d_rbar_svu, const rslf_params* p) RSLF_API_TRY
{
    if (!ctx || !vol || !d_Ce_svu || !d_Ce_mask_svu || !d_Cd_svu || !d_depth_svu || !d_rbar_svu)
        return fail(RSLF_ERR_INVALID_ARG, "NULL argument");
    if (!ctx->sweep_open)
        return fail(RSLF_ERR_INVALID_ARG, "rslf_sweep_visit_scan without rslf_sweep_begin");
    if ((d_dmin_svu == nullptr) != (d_dmax_svu == nullptr))
        return fail(RSLF_ERR_INVALID_ARG, "d_dmin_svu and d_dmax_svu must both be given or both be NULL");
    if (s_hat < 0 || s_hat >= vol->S)
        return fail(RSLF_ERR_INVALID_ARG, "s_hat=%d outside [0,%d)", s_hat, vol->S);
    if (s_hat != ctx->sweep_expect)   // the previous visit has already listed this view's pixels (k4_propagate_apply)
        return fail(RSLF_ERR_INVALID_ARG, "the sweep visits view %d next (core.hpp:981-990), not %d", ctx->sweep_expect, s_hat);
    const size_t n = (size_t)vol->V * vol->U;
    // After the centre view, propagation has explained most pixels: a visit scans a few per scanline.
    // Pack them into one list and share each tile's hypotheses out over up to kSweepGroups workgroups (k2_scan.hpp).
    ctx->scan_groups = ctx->sweep_first ? 1 : plan::kSweepGroups;
    ctx->scan_packed = !ctx->sweep_first;
    // core.hpp:1012-1028: the pile call is the scan of every EPI followed by the selective median.  In the
    // reference the stored plane keeps the RAW depths and only the local header is rebound to the median
    // (core.hpp:892), which the propagation then paints from: so the scan writes the view's depth plane and the median
    // goes to ctx->filtered (rslf_sweep_visit_finish) -- no plane copies.
    return rslf_depth_epi_scan(ctx, vol, d_dmin_svu ? d_dmin_svu + (size_t)s_hat * n : nullptr,
                               d_dmax_svu ? d_dmax_svu + (size_t)s_hat * n : nullptr, dmin, dmax, dim_d, s_hat,
                               d_Ce_svu + (size_t)s_hat * n, d_Ce_mask_svu + (size_t)s_hat * n, d_Cd_svu + (size_t)s_hat * n,
                               d_depth_svu + (size_t)s_hat * n, d_rbar_svu + (size_t)s_hat * n * vol->C, p,
                               ctx->sweep_mask_run + (size_t)s_hat * n, nullptr, nullptr, nullptr);
}
RSLF_API_CATCH

extern "C" int rslf_sweep_visit_finish(rslf_ctx* ctx, const rslf_volume* vol, int s_hat, uint8_t* d_Ce_mask_svu, float* d_Cd_svu,
                                       float* d_depth_svu, float* d_rbar_svu, const rslf_params* p) RSLF_API_TRY
{
    if (!ctx || !vol || !d_Ce_mask_svu || !d_Cd_svu || !d_depth_svu || !d_rbar_svu || !p)
        return fail(RSLF_ERR_INVALID_ARG, "NULL argument");
    if (!ctx->sweep_open)
        return fail(RSLF_ERR_INVALID_ARG, "rslf_sweep_visit_finish without rslf_sweep_begin");
    if (s_hat != ctx->sweep_expect)
        return fail(RSLF_ERR_INVALID_ARG, "rslf_sweep_visit_finish(%d): the open visit is view %d", s_hat, ctx->sweep_expect);
    if (p->median_filter_size < 0 || p->median_filter_size > plan::kMedianMaxSize)
        return fail(RSLF_ERR_INVALID_ARG, "median_filter_size=%d: must be in [0, %d]", p->median_filter_size, plan::kMedianMaxSize);
    HIP_TRY(hipSetDevice(ctx->device));
    const int S = vol->S, V = vol->V, U = vol->U, C = vol->C;
    const size_t n = (size_t)V * U;
    hipStream_t st = ctx->stream;
    uint8_t* mask_svu = ctx->sweep_mask_run;
    const dim3 grid_vu((U + 255) / 256, V);
    if ((long long)S * V > (1ll << 31) - 1 || U > 65536)
        return fail(RSLF_ERR_UNSUPPORTED, "%d views x %d scanlines x %d columns: too large for one apply launch", S, V, U);
    const plan::MedianPlan mp = plan::median_plan(p->median_filter_size, C);   // window tile in LDS (k3_median.hpp)
    const plan::NormThreshold median_thr = plan::norm_threshold(p->median_filter_epsilon);
    const plan::NormThreshold prop_thr = plan::norm_threshold(p->propagation_epsilon);
    int* packed_n = reinterpret_cast<int*>(ctx->total + 1);
    float* depth = d_depth_svu + (size_t)s_hat * n;
    float* Cd = d_Cd_svu + (size_t)s_hat * n;
    float* rbar = d_rbar_svu + (size_t)s_hat * n * C;
    uint8_t* cem = d_Ce_mask_svu + (size_t)s_hat * n;
    // core.hpp:881-892 (selective median over the edge mask) and :1088-1129 (propagation) -- the median and the
    // claims of a pixel in one launch (k34_median_claim), then the apply pass, which also lists the pixels the NEXT
    // visit scans; a visit is three launches: scan (its groups merge their records themselves), median + claims,
    // apply + compaction
    int s_next = plan::sweep_view_after(S, s_hat);
    const int s_after = s_next;
    if (ctx->force_packed == 0 || n > (size_t)INT32_MAX)
        s_next = -1;   // that scan will not take a packed list: it compacts for itself
    bool launched = false;
#define RSLF_K34_CASE(CC, MODE)                                                                                                  \
    if (!launched && C == CC && mp.mode == MODE) {                                                                                \
        if (mp.lds_bytes > ((size_t)64 << 10))                                                                                    \
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k34_median_claim<CC, MODE>),                              \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)mp.lds_bytes));                         \
        hipLaunchKernelGGL((k34_median_claim<CC, MODE>), grid_vu, dim3(256), mp.lds_bytes, st, view_of(vol), s_hat, depth, ctx->filtered, \
                           cem, mp.w, median_thr, rbar, mask_svu, ctx->winner, ctx->dirty, p->slope_factor, prop_thr,             \
                           p->use_disp_confidence_score ? Cd : nullptr, p->disp_score_threshold, packed_n,                       \
                           ctx->claim_skip ? ctx->remain : nullptr);                                                             \
        launched = true;                                                                                                          \
    }
    RSLF_MEDIAN_MODES(RSLF_K34_CASE, 1)
    RSLF_MEDIAN_MODES(RSLF_K34_CASE, 3)
#undef RSLF_K34_CASE
    if (!launched)
        return fail(RSLF_ERR_INTERNAL, "no median + claims kernel for %d channels, mode %d", C, mp.mode);
    HIP_TRY(hipGetLastError());
    const unsigned apply_blocks = (unsigned)((s_next >= 0 ? V : 0) + ((long long)S * V + kApplyRowsPerBlock - 1) / kApplyRowsPerBlock);
    hipLaunchKernelGGL(k4_propagate_apply, dim3(apply_blocks), dim3(256), 0, st, S, V, U, s_hat, ctx->filtered, Cd, d_depth_svu,
                       d_Cd_svu, mask_svu, ctx->winner, ctx->dirty, s_next, s_next >= 0 ? d_Ce_mask_svu + (size_t)s_next * n : nullptr, ctx->list,
                       ctx->count, ctx->total, packed_n, ctx->remain, ctx->count + ctx->count_cap);
    HIP_TRY(hipGetLastError());
    ctx->packed_n_clean = s_next < 0;       // k34_median_claim zeroed the packed list's length; a listing apply pass set it again
    ctx->precompacted = s_next >= 0 ? 2 : 0;
    ctx->sweep_expect = s_after;
    ctx->sweep_first = false;
    return RSLF_OK;
}
RSLF_API_CATCH

extern "C" int rslf_sweep_end(rslf_ctx* ctx, int ok, int dim_d, rslf_stats* stats) RSLF_API_TRY
{
    if (!ctx)
        return fail(RSLF_ERR_INVALID_ARG, "ctx is NULL");
    const bool was_open = ctx->sweep_open;
    sweep_close(ctx, ok != 0 && was_open);
    if (ok && was_open && stats) {
        unsigned long long tot = 0;
        HIP_TRY(hipSetDevice(ctx->device));
        HIP_TRY(hipMemcpyAsync(&tot, ctx->total, sizeof(tot), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        stats->pixels_scanned = (int64_t)tot;
        stats->units = (int64_t)tot * dim_d;
        stats->scan_kernel = ctx->last_kernel;
        stats->s_pad = ctx->last_spad;
    }
    return RSLF_OK;
}
RSLF_API_CATCH

extern "C" int rslf_depth_epi_2d(rslf_ctx* ctx, const rslf_volume* vol, const float* d_dmin_svu, const float* d_dmax_svu,
                                 float dmin, float dmax, int dim_d, float* d_Ce_svu, uint8_t* d_Ce_mask_svu, float* d_Cd_svu,
                                 float* d_depth_svu, float* d_rbar_svu, const rslf_params* p, uint8_t* d_scan_mask_svu,
                                 rslf_stats* stats) RSLF_API_TRY
{
    if (!ctx || !vol || !d_Ce_svu || !d_Ce_mask_svu || !d_Cd_svu || !d_depth_svu || !d_rbar_svu)
        return fail(RSLF_ERR_INVALID_ARG, "NULL argument");
    int rc = check_params(p);
    if (rc)
        return rc;
    rc = rslf_sweep_begin(ctx, vol, d_Ce_mask_svu, d_scan_mask_svu, dim_d, 0, vol->V);
    if (rc)
        return rc;
    for (int s_hat : plan::sweep_order(vol->S)) {   // core.hpp:981-990
        rc = rslf_sweep_visit_scan(ctx, vol, d_dmin_svu, d_dmax_svu, dmin, dmax, dim_d, s_hat, d_Ce_svu, d_Ce_mask_svu, d_Cd_svu,
                                   d_depth_svu, d_rbar_svu, p);
        if (!rc)
            rc = rslf_sweep_visit_finish(ctx, vol, s_hat, d_Ce_mask_svu, d_Cd_svu, d_depth_svu, d_rbar_svu, p);
        if (rc) {
            const std::string msg = last_error_buffer();
            (void)rslf_sweep_end(ctx, 0, dim_d, nullptr);
            return fail(rc, "%s", msg.c_str());
        }
    }
    return rslf_sweep_end(ctx, 1, dim_d, stats);
}
RSLF_API_CATCH

extern "C" int rslf_depth2d_run(rslf_ctx* ctx, const rslf_volume* vol, float dmin, float dmax, int dim_d, const rslf_params* p,
                                float* d_Ce_svu, uint8_t* d_Ce_mask_svu, float* d_Cd_svu, float* d_depth_svu, float* d_rbar_svu,
                                uint8_t* d_scan_mask_svu, rslf_stats* stats) RSLF_API_TRY
{
    if (!ctx || !vol || !d_Ce_svu || !d_Ce_mask_svu || !d_Cd_svu || !d_depth_svu || !d_rbar_svu)
        return fail(RSLF_ERR_INVALID_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t n = (size_t)vol->S * vol->V * vol->U;
    hipStream_t st = ctx->stream;
    // dc.hpp:733-750 (C_e and C_d are uninitialised there; zero is the intended start)
    HIP_TRY(hipMemsetAsync(d_Ce_svu, 0, n * sizeof(float), st));
    HIP_TRY(hipMemsetAsync(d_Cd_svu, 0, n * sizeof(float), st));
    HIP_TRY(hipMemsetAsync(d_depth_svu, 0, n * sizeof(float), st));
    HIP_TRY(hipMemsetAsync(d_rbar_svu, 0, n * vol->C * sizeof(float), st));
    int rc = rslf_edge_confidence_2d(ctx, vol, p, d_Ce_svu, d_Ce_mask_svu);   // dc.hpp:772
    if (rc)
        return rc;
    return rslf_depth_epi_2d(ctx, vol, nullptr, nullptr, dmin, dmax, dim_d, d_Ce_svu, d_Ce_mask_svu, d_Cd_svu, d_depth_svu,   // dc.hpp:780
                             d_rbar_svu, p, d_scan_mask_svu, stats);
}
RSLF_API_CATCH
