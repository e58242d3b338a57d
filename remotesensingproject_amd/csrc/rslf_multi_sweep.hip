// librslf_hip.so, unit 9 of 9: the 2-D sweep and fine-to-coarse sharded by scanline over the devices of one process, behind
// the C-ABI -- one neighbour exchange of boundary rows per visit (the path's one real exchange step).  C-ABI: include/rslf_hip.h.
#include "rslf_internal.hpp"

#include <algorithm>
#include <cmath>
#include <deque>

using namespace rslf;

// ---- Depth2DComputer::run over several devices (dc.hpp:748-805) -----------------------------------------------------
// The 2-D sweep sharded by scanline behind the C-ABI: every device holds a block of scanlines (+ the median's halo) of the
// volume and of the [S][rows][U] planes; a visit is scan on every device, then the neighbours' boundary rows of the visited
// view's raw disparities and edge mask by peer copy (the one real exchange step of the path, as sharding.ShardedDepth2D
// does it over RCCL), then median + propagation on every device.  ONE host thread drives all devices: every call only
// queues work, the order between devices is kept by events -- a device's finish waits for its neighbours to have
// fetched its boundary rows, because the apply pass rewrites them (core.hpp:1119-1121).
namespace {

struct Sweep2DDev {
    rslf_volume* vol = nullptr;
    float *Ce = nullptr, *Cd = nullptr, *depth = nullptr, *rbar = nullptr;
    float *dmin = nullptr, *dmax = nullptr;   // per-pixel hypothesis ranges over the held rows (a fine-to-coarse level), or NULL
    uint8_t *cem = nullptr, *scan_mask = nullptr;
    int lo = 0, hi = 0, a = 0, b = 0;   // rows held [lo, hi), rows owned [a, b)
    hipEvent_t ev_scan = nullptr, ev_fetch = nullptr;
    bool begun = false;
    hipStream_t saved_stream = nullptr;
    bool stream_swapped = false;
};

void sweep2d_free(rslf_multi* m, std::vector<Sweep2DDev>& ds)
{
    for (size_t i = 0; i < ds.size(); i++) {
        Sweep2DDev& d = ds[i];
        rslf_ctx* ctx = m->devs[i].ctx;
        (void)hipSetDevice(ctx->device);
        if (d.begun)
            (void)rslf_sweep_end(ctx, 0, 2, nullptr);
        (void)hipStreamSynchronize(ctx->stream);
        if (d.stream_swapped)
            ctx->stream = d.saved_stream;
        if (d.vol)
            (void)rslf_volume_destroy(d.vol);
        // (the planes live in the device's arena, which stays)
        if (d.ev_scan)
            (void)hipEventDestroy(d.ev_scan);
        if (d.ev_fetch)
            (void)hipEventDestroy(d.ev_fetch);
    }
}

// The fine-to-coarse form of a sweep: nothing passes through host memory.  The level's RAW volume, its per-pixel ranges
// and the two planes the next steps need live on the FIRST device; every device takes the rows it holds from there and
// leaves its own rows of the results there, by peer copies (plain device copies where it is the first device itself).
struct FirstDevicePlanes {
    const float* raw_vsuc = nullptr;   // [V][S][U][C] raw values of the level (replaces the host EPIs)
    const float* dmin_svu = nullptr;   // [S][V][U] ranges, or NULL for the scalar range
    const float* dmax_svu = nullptr;
    float* Ce_svu = nullptr;           // [S][V][U] results
    float* depth_svu = nullptr;
};

int multi_depth2d(rslf_multi* m, const void* const* h_epis, bool is_u8, size_t row_stride_bytes, int V, int S, int U, int C, float scale_arg,
                  float dmin, float dmax, int dim_d, const rslf_params* p, float* h_Ce_svu, uint8_t* h_Ce_mask_svu, float* h_Cd_svu,
                  float* h_depth_svu, float* h_rbar_svu, uint8_t* h_scan_mask_svu, rslf_stats* stats,
                  const FirstDevicePlanes* first = nullptr)
{
    if (!m || (!h_epis && !(first && first->raw_vsuc)))
        return fail(RSLF_ERR_INVALID_ARG, "NULL argument");
    if (first && ((first->dmin_svu == nullptr) != (first->dmax_svu == nullptr)))
        return fail(RSLF_ERR_INVALID_ARG, "dmin_svu and dmax_svu must both be given or both be NULL");
    const bool ranges = first && first->dmin_svu;
    const int dev0 = m->devs[0].ctx->device;
    if (V < 1 || S < 1 || U < 1 || (C != 1 && C != 3))
        return fail(RSLF_ERR_INVALID_ARG, "bad dimensions V=%d S=%d U=%d C=%d", V, S, U, C);
    int rc = check_params(p);
    if (rc)
        return rc;
    for (int v = 0; h_epis && v < V; v++)
        if (!h_epis[v])
            return fail(RSLF_ERR_INVALID_ARG, "h_epis[%d] is NULL", v);
    const int h_med = plan::median_halo(p->median_filter_size);
    const int halo = plan::halo_rows(p->median_filter_size, p->edge_confidence_opening_size);
    const int nd = plan::sweep_devices_for(V, (int)m->devs.size(), halo);   // a block must be able to fill its neighbours' halo rows
    std::vector<Sweep2DDev> ds((size_t)nd);
#define S2_TRY(expr)                                  \
    do {                                              \
        int rc_ = (expr);                             \
        if (rc_ != RSLF_OK) {                         \
            const std::string msg_ = last_error_buffer();           \
            sweep2d_free(m, ds);                      \
            return fail(rc_, "%s", msg_.c_str());     \
        }                                             \
    } while (0)
#define S2_HIP(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            sweep2d_free(m, ds);                                                                       \
            return fail(RSLF_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
        }                                                                                              \
    } while (0)
    // set-up: rows, volume, planes, edge confidence, sweep state
    for (int i = 0; i < nd; i++) {
        Sweep2DDev& d = ds[(size_t)i];
        rslf_ctx* ctx = m->devs[(size_t)i].ctx;
        S2_HIP(hipSetDevice(ctx->device));
        d.saved_stream = ctx->stream;
        d.stream_swapped = true;
        ctx->stream = m->devs[(size_t)i].s_comp;
        const plan::RowBlock blk = plan::row_block(V, i, nd, halo);
        d.a = blk.a, d.b = blk.b, d.lo = blk.lo, d.hi = blk.hi;
        const int rows = d.hi - d.lo;
        const size_t n = (size_t)S * rows * U;
        S2_HIP(hipEventCreateWithFlags(&d.ev_scan, hipEventDisableTiming));
        S2_HIP(hipEventCreateWithFlags(&d.ev_fetch, hipEventDisableTiming));
        S2_TRY(rslf_volume_create(ctx, rows, S, U, C, &d.vol));
        // a copy between this device and the first one (either direction), queued on this device's stream
        auto copy01 = [&](void* dst, int dst_dev, const void* src, int src_dev, size_t bytes) -> hipError_t {
            return multi_copy(m, dst, dst_dev, src, src_dev, bytes, ctx->stream);
        };
        if (first && first->raw_vsuc) {   // the held rows of the level's raw volume: from the first device
            const size_t row_floats = (size_t)S * U * C;
            const float* src = first->raw_vsuc + (size_t)d.lo * row_floats;
            if (ctx->device != dev0) {
                S2_TRY(ensure_staging(ctx, (size_t)rows * row_floats * sizeof(float)));
                S2_HIP(copy01(ctx->staging, ctx->device, src, dev0, (size_t)rows * row_floats * sizeof(float)));
                src = (const float*)ctx->staging;
            }
            S2_TRY(rslf_volume_pack_device_f32(d.vol, src, scale_arg, nullptr));
        } else if (is_u8) {
            S2_TRY(upload_host<uint8_t>(d.vol, (const uint8_t* const*)h_epis + d.lo, row_stride_bytes, false, (float)(1.0 / 255.0)));
        } else {
            S2_TRY(upload_host<float>(d.vol, (const float* const*)h_epis + d.lo, row_stride_bytes, false, scale_of(scale_arg)));
        }
        {   // planes: one allocation per device, grown when a larger field comes (allocation calls synchronise the device)
            rslf_multi::Dev& md = m->devs[(size_t)i];
            const size_t nf = (n + 63) & ~(size_t)63;   // floats per plane, 256-byte aligned
            const size_t need = nf * sizeof(float) * (3 + (size_t)C + (ranges ? 2 : 0)) + 2 * nf;
            if (need > md.arena_cap) {
                (void)hipFree(md.arena);
                md.arena = nullptr;
                md.arena_cap = 0;
                S2_HIP(hipMalloc(&md.arena, need));
                md.arena_cap = need;
            }
            float* f = reinterpret_cast<float*>(md.arena);
            d.Ce = f, f += nf;
            d.Cd = f, f += nf;
            d.depth = f, f += nf;
            d.rbar = f, f += nf * C;
            if (ranges) {
                d.dmin = f, f += nf;
                d.dmax = f, f += nf;
            }
            d.cem = reinterpret_cast<uint8_t*>(f);
            d.scan_mask = d.cem + nf;
        }
        hipStream_t st = ctx->stream;
        if (ranges) {   // the held rows of every view's range planes: one run of bytes per view
            const size_t w = (size_t)rows * U;
            for (int sv = 0; sv < S; sv++) {
                S2_HIP(copy01(d.dmin + (size_t)sv * w, ctx->device, first->dmin_svu + ((size_t)sv * V + d.lo) * U, dev0, w * sizeof(float)));
                S2_HIP(copy01(d.dmax + (size_t)sv * w, ctx->device, first->dmax_svu + ((size_t)sv * V + d.lo) * U, dev0, w * sizeof(float)));
            }
        }
        S2_HIP(hipMemsetAsync(d.Ce, 0, n * sizeof(float), st));   // dc.hpp:733-750
        S2_HIP(hipMemsetAsync(d.Cd, 0, n * sizeof(float), st));
        S2_HIP(hipMemsetAsync(d.depth, 0, n * sizeof(float), st));
        S2_HIP(hipMemsetAsync(d.rbar, 0, n * C * sizeof(float), st));
        S2_TRY(rslf_edge_confidence_2d(ctx, d.vol, p, d.Ce, d.cem));                               // dc.hpp:772
        S2_TRY(rslf_sweep_begin(ctx, d.vol, d.cem, d.scan_mask, dim_d, d.a - d.lo, d.b - d.lo));  // dc.hpp:780
        d.begun = true;
    }
    // rows of plane `base` ([S][rows][U] elements of `esz` bytes) of view s_hat, local rows [r, r + h)
    auto rows_of = [&](const Sweep2DDev& d, void* base, size_t esz, int s_hat, int r) -> char* {
        return (char*)base + (((size_t)s_hat * (d.hi - d.lo) + r) * U) * esz;
    };
    // One visit = plan::sweep_visit_schedule: scan on every device; every device fetches its neighbours' boundary rows
    // (after the neighbour's scan); every device finishes once BOTH neighbours have fetched its raw rows (its apply pass
    // rewrites them, core.hpp:1119-1121).  One host thread queues the ops; the waits are events between streams.
    const std::vector<plan::VisitOp> schedule = plan::sweep_visit_schedule(nd, h_med);
    auto block_of = [&](const Sweep2DDev& d) { return plan::RowBlock{d.a, d.b, d.lo, d.hi}; };
    for (int s_hat : plan::sweep_order(S)) {   // core.hpp:981-990
        for (const plan::VisitOp& op : schedule) {
            Sweep2DDev& d = ds[(size_t)op.dev];
            rslf_ctx* ctx = m->devs[(size_t)op.dev].ctx;
            S2_HIP(hipSetDevice(ctx->device));
            for (int k : op.wait_scan_of)
                S2_HIP(hipStreamWaitEvent(ctx->stream, ds[(size_t)k].ev_scan, 0));
            for (int k : op.wait_fetch_of)
                S2_HIP(hipStreamWaitEvent(ctx->stream, ds[(size_t)k].ev_fetch, 0));
            if (op.kind == plan::VisitOp::SCAN) {
                S2_TRY(rslf_sweep_visit_scan(ctx, d.vol, d.dmin, d.dmax, dmin, dmax, dim_d, s_hat, d.Ce, d.cem, d.Cd, d.depth, d.rbar, p));
                S2_HIP(hipEventRecord(d.ev_scan, ctx->stream));
                if (h_med == 0 || nd == 1)
                    S2_HIP(hipEventRecord(d.ev_fetch, ctx->stream));   // nothing to fetch: the event the neighbours' finish waits for
            } else if (op.kind == plan::VisitOp::FETCH) {
                const Sweep2DDev& o = ds[(size_t)op.neighbour];
                rslf_ctx* octx = m->devs[(size_t)op.neighbour].ctx;
                int dst_r, src_r;
                plan::fetch_rows(block_of(d), block_of(o), op.neighbour < op.dev ? 0 : 1, h_med, &dst_r, &src_r);
                for (int pl = 0; pl < 2; pl++) {   // the visited view's raw disparities and its edge mask
                    const size_t esz = pl == 0 ? sizeof(float) : 1;
                    char* dst = rows_of(d, pl == 0 ? (void*)d.depth : (void*)d.cem, esz, s_hat, dst_r);
                    const char* src = rows_of(o, pl == 0 ? (void*)o.depth : (void*)o.cem, esz, s_hat, src_r);
                    S2_HIP(multi_copy(m, dst, ctx->device, src, octx->device, (size_t)h_med * U * esz, ctx->stream));
                }
                S2_HIP(hipEventRecord(d.ev_fetch, ctx->stream));   // re-recorded after each fetch: the LAST one is what counts
            } else {
                S2_TRY(rslf_sweep_visit_finish(ctx, d.vol, s_hat, d.cem, d.Cd, d.depth, d.rbar, p));
            }
        }
    }
    // collect: every device's own rows of every view land at their place in the caller's [S][V][U] planes
    long long scanned = 0;
    for (int i = 0; i < nd; i++) {
        Sweep2DDev& d = ds[(size_t)i];
        rslf_ctx* ctx = m->devs[(size_t)i].ctx;
        S2_HIP(hipSetDevice(ctx->device));
        rslf_stats st_i;
        memset(&st_i, 0, sizeof(st_i));
        d.begun = false;
        S2_TRY(rslf_sweep_end(ctx, 1, dim_d, &st_i));
        scanned += st_i.pixels_scanned;
        if (stats && i == 0) {
            stats->scan_kernel = st_i.scan_kernel;
            stats->s_pad = st_i.s_pad;
        }
        const int rows = d.hi - d.lo, own = d.b - d.a;
        auto pull = [&](void* h, const void* dv, size_t esz) -> hipError_t {
            if (!h)
                return hipSuccess;
            return hipMemcpy2DAsync((char*)h + (size_t)d.a * U * esz, (size_t)V * U * esz, (const char*)dv + (size_t)(d.a - d.lo) * U * esz,
                                    (size_t)rows * U * esz, (size_t)own * U * esz, S, hipMemcpyDeviceToHost, ctx->stream);
        };
        if (first && first->Ce_svu) {   // this device's own rows of the two planes the next steps read: to the first device
            const size_t w = (size_t)own * U;
            for (int sv = 0; sv < S; sv++) {
                const size_t src = ((size_t)sv * rows + (d.a - d.lo)) * U, dst = ((size_t)sv * V + d.a) * U;
                S2_HIP(multi_copy(m, first->Ce_svu + dst, dev0, d.Ce + src, ctx->device, w * sizeof(float), ctx->stream));
                S2_HIP(multi_copy(m, first->depth_svu + dst, dev0, d.depth + src, ctx->device, w * sizeof(float), ctx->stream));
            }
        }
        S2_HIP(pull(h_Ce_svu, d.Ce, sizeof(float)));
        S2_HIP(pull(h_Ce_mask_svu, d.cem, 1));
        S2_HIP(pull(h_Cd_svu, d.Cd, sizeof(float)));
        S2_HIP(pull(h_depth_svu, d.depth, sizeof(float)));
        S2_HIP(pull(h_rbar_svu, d.rbar, sizeof(float) * C));
        S2_HIP(pull(h_scan_mask_svu, d.scan_mask, 1));
    }
    for (int i = 0; i < nd; i++) {
        S2_HIP(hipSetDevice(m->devs[(size_t)i].ctx->device));
        S2_HIP(hipStreamSynchronize(m->devs[(size_t)i].ctx->stream));
    }
    sweep2d_free(m, ds);
    if (stats) {
        stats->pixels_scanned = scanned;
        stats->units = scanned * dim_d;
    }
    return RSLF_OK;
#undef S2_TRY
#undef S2_HIP
}

}  // namespace

// FineToCoarse<T> (rslf_fine_to_coarse.hpp:103-324) over the context's devices.  Where the time goes -- every level's 2-D
// sweep -- runs sharded (multi_depth2d, with the level's tightened per-pixel ranges); the pyramid, the bound tightening
// and the fusion, cheap whole-image passes with non-local footprints, run on the first device, where every level's raw
// volume, ranges, disparities and confidences stay: the devices take their rows from there and leave their results there
// by peer copies (FirstDevicePlanes).  The host sees the EPIs going up once and the fused map coming down.
extern "C" int rslf_multi_fine_to_coarse_run_host(rslf_multi* m, const void* const* h_epis, int is_u8, int V, int S, int U, int C,
                                                  size_t row_stride_bytes, float d_min, float d_max, int dim_d, float epi_scale_factor,
                                                  const rslf_params* p, int max_pyr_depth, int accept_all_last_scale,
                                                  float* h_out_map_svu, uint8_t* h_out_valid_svu, int* n_levels, rslf_stats* stats) RSLF_API_TRY
{
    if (!m || !h_epis || !h_out_map_svu || !h_out_valid_svu || V < 1 || S < 1 || U < 1 || (C != 1 && C != 3))
        return fail(RSLF_ERR_INVALID_ARG, "bad arguments");
    int rc = check_params(p);
    if (rc)
        return rc;
    rslf_ctx* ctx = m->devs[0].ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const size_t elem = is_u8 ? 1 : 4;
    const size_t row_bytes = (size_t)U * C * elem;
    if (row_stride_bytes == 0)
        row_stride_bytes = row_bytes;

    struct Level {
        int V = 0, U = 0;
        float scale = 1.0f;
        rslf_params params;
        DevBuf raw;                 // [V][S][U][C] raw values, on the first device
        DevBuf Ce, depth, valid;    // [S][V][U], on the first device
    };
    // constructor (f2c.hpp:103-159): the pyramid (plan::f2c_pyramid) on the first device, every level's RAW volume kept there
    const std::vector<plan::LevelDims> dims = plan::f2c_pyramid(V, U, max_pyr_depth);
    if (dims.empty())
        return fail(RSLF_ERR_INVALID_ARG, "light field %dx%d is not larger than _MIN_SPATIAL_DIM: no pyramid level", V, U);
    std::deque<Level> levels(dims.size());   // (a deque: the levels own device buffers and must not move)
    {
        Level& l0 = levels.front();
        HIP_TRY(l0.raw.alloc((size_t)V * S * U * C * sizeof(float)));
        DevBuf stage;
        void* dst = l0.raw.p;
        if (is_u8) {
            HIP_TRY(stage.alloc((size_t)V * S * row_bytes));
            dst = stage.p;
        }
        for (int v = 0; v < V; v++) {
            if (!h_epis[v])
                return fail(RSLF_ERR_INVALID_ARG, "h_epis[%d] is NULL", v);
            if (row_stride_bytes == row_bytes)
                HIP_TRY(hipMemcpyAsync((char*)dst + (size_t)v * S * row_bytes, h_epis[v], (size_t)S * row_bytes, hipMemcpyHostToDevice, st));
            else
                HIP_TRY(hipMemcpy2DAsync((char*)dst + (size_t)v * S * row_bytes, row_bytes, h_epis[v], row_stride_bytes, row_bytes, S,
                                         hipMemcpyHostToDevice, st));
        }
        if (is_u8) {
            rc = f2c_u8_to_f32(st, (const uint8_t*)stage.p, (float*)l0.raw.p, (size_t)V * S * U * C);
            if (rc)
                return rc;
        }
        HIP_TRY(hipStreamSynchronize(st));
    }
    for (size_t l = 0; l < dims.size(); l++) {
        Level& lv = levels[l];
        const int dim_v = dims[l].V, dim_u = dims[l].U;
        lv.V = dim_v;
        lv.U = dim_u;
        lv.params = *p;
        lv.params.slope_factor = (float)((0.0 + dim_u) / U);              // f2c.hpp:139
        lv.scale = 255.0f;                                                // dc.hpp:696-699 (uchar)
        if (!is_u8) {
            lv.scale = epi_scale_factor;
            if (lv.scale < 0) {                                           // dc.hpp:671-690: this level's own max
                rc = rslf_device_max_f32(ctx, (const float*)lv.raw.p, (size_t)dim_v * S * dim_u * C, &lv.scale);
                if (rc)
                    return rc;
            }
        }
        if (l + 1 == dims.size())
            break;
        Level& nx = levels[l + 1];
        HIP_TRY(nx.raw.alloc((size_t)dims[l + 1].V * S * dims[l + 1].U * C * sizeof(float)));   // f2c.hpp:145-147: the RAW EPIs go down
        rc = is_u8 ? rslf_downsample_epis_u8(ctx, (const float*)lv.raw.p, dim_v, S, dim_u, C, (float*)nx.raw.p)
                   : rslf_downsample_epis_f32(ctx, (const float*)lv.raw.p, dim_v, S, dim_u, C, (float*)nx.raw.p);
        if (rc)
            return rc;
    }
    const int P = (int)levels.size();

    // run(): f2c.hpp:171-299 -- the sweeps over all devices, the tightening on the first
    int64_t pixels = 0;
    rslf_stats st1;
    memset(&st1, 0, sizeof(st1));
    for (int l = 0; l < P; l++) {
        Level& lv = levels[(size_t)l];
        const size_t n = (size_t)S * lv.V * lv.U;
        HIP_TRY(hipSetDevice(ctx->device));
        HIP_TRY(lv.Ce.alloc(n * 4));
        HIP_TRY(lv.depth.alloc(n * 4));
        HIP_TRY(lv.valid.alloc(n));
        DevBuf d_lo, d_hi;
        FirstDevicePlanes first;
        first.raw_vsuc = (const float*)lv.raw.p;
        first.Ce_svu = (float*)lv.Ce.p;
        first.depth_svu = (float*)lv.depth.p;
        if (l > 0) {
            Level& up = levels[(size_t)l - 1];
            HIP_TRY(d_lo.alloc(n * 4));
            HIP_TRY(d_hi.alloc(n * 4));
            rc = f2c_fill_f32(st, (float*)d_lo.p, n, d_min);
            if (!rc)
                rc = f2c_fill_f32(st, (float*)d_hi.p, n, d_max);
            if (rc)
                return rc;
            rc = rslf_f2c_tighten_bounds(ctx, (const float*)up.depth.p, (const uint8_t*)up.valid.p, S, up.V, up.U, (float*)d_lo.p,
                                         (float*)d_hi.p, lv.V, lv.U);
            if (rc)
                return rc;
            first.dmin_svu = (const float*)d_lo.p;
            first.dmax_svu = (const float*)d_hi.p;
        }
        HIP_TRY(hipStreamSynchronize(st));   // what the other devices' streams are about to read is complete
        rc = multi_depth2d(m, nullptr, false, 0, lv.V, S, lv.U, C, lv.scale, d_min, d_max, dim_d, &lv.params, nullptr, nullptr, nullptr,
                           nullptr, nullptr, nullptr, &st1, &first);
        if (rc)
            return rc;
        pixels += st1.pixels_scanned;
        // get_valid_depths_mask_s_v_u (dc.hpp:893-915): C_e > threshold; the last level accepts everything when asked to
        HIP_TRY(hipSetDevice(ctx->device));
        const bool all = accept_all_last_scale && l == P - 1;
        rc = f2c_valid_mask(st, (const float*)lv.Ce.p, (uint8_t*)lv.valid.p, n, all ? -1.0f : p->edge_score_threshold);
        if (rc)
            return rc;
        lv.raw.release();   // the level's raw volume has been taken by every device
    }

    // get_results(): f2c.hpp:302-324 on the first device
    std::vector<const float*> dp((size_t)P);
    std::vector<const uint8_t*> vp((size_t)P);
    std::vector<int> Vp((size_t)P), Up((size_t)P);
    for (int l = 0; l < P; l++) {
        dp[(size_t)l] = (const float*)levels[(size_t)l].depth.p;
        vp[(size_t)l] = (const uint8_t*)levels[(size_t)l].valid.p;
        Vp[(size_t)l] = levels[(size_t)l].V;
        Up[(size_t)l] = levels[(size_t)l].U;
    }
    const size_t n0 = (size_t)S * V * U;
    DevBuf omap, ovalid;
    HIP_TRY(omap.alloc(n0 * 4));
    HIP_TRY(ovalid.alloc(n0));
    rc = rslf_f2c_fuse(ctx, dp.data(), vp.data(), Vp.data(), Up.data(), P, S, (float*)omap.p, (uint8_t*)ovalid.p);
    if (rc)
        return rc;
    HIP_TRY(hipMemcpyAsync(h_out_map_svu, omap.p, n0 * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(h_out_valid_svu, ovalid.p, n0, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (n_levels)
        *n_levels = P;
    if (stats) {
        *stats = st1;
        stats->pixels_scanned = pixels;
        stats->units = pixels * dim_d;
    }
    return RSLF_OK;
}
RSLF_API_CATCH

extern "C" int rslf_multi_depth2d_run_f32(rslf_multi* m, const float* const* h_epis, size_t row_stride_bytes, int V, int S, int U, int C,
                                          float epi_scale_factor, float dmin, float dmax, int dim_d, const rslf_params* p,
                                          float* h_Ce_svu, uint8_t* h_Ce_mask_svu, float* h_Cd_svu, float* h_depth_svu,
                                          float* h_rbar_svu, uint8_t* h_scan_mask_svu, rslf_stats* stats, float* scale_used) RSLF_API_TRY
{
    if (!m || !h_epis || V < 1 || S < 1 || U < 1)
        return fail(RSLF_ERR_INVALID_ARG, "NULL argument");
    const size_t row_elems = (size_t)U * C;
    const size_t stride = row_stride_bytes ? row_stride_bytes : row_elems * sizeof(float);
    for (int v = 0; v < V; v++)
        if (!h_epis[v])
            return fail(RSLF_ERR_INVALID_ARG, "h_epis[%d] is NULL", v);
    if (epi_scale_factor < 0)   // dc.hpp:671-705: the maximum over ALL EPIs, taken once
        epi_scale_factor = host_max_f32_parallel(h_epis, V, S, stride, row_elems, epi_scale_factor);
    if (scale_used)
        *scale_used = epi_scale_factor;
    return multi_depth2d(m, (const void* const*)h_epis, false, stride, V, S, U, C, epi_scale_factor, dmin, dmax, dim_d, p, h_Ce_svu,
                         h_Ce_mask_svu, h_Cd_svu, h_depth_svu, h_rbar_svu, h_scan_mask_svu, stats);
}
RSLF_API_CATCH

extern "C" int rslf_multi_depth2d_run_u8(rslf_multi* m, const uint8_t* const* h_epis, size_t row_stride_bytes, int V, int S, int U, int C,
                                         float dmin, float dmax, int dim_d, const rslf_params* p, float* h_Ce_svu,
                                         uint8_t* h_Ce_mask_svu, float* h_Cd_svu, float* h_depth_svu, float* h_rbar_svu,
                                         uint8_t* h_scan_mask_svu, rslf_stats* stats) RSLF_API_TRY
{
    return multi_depth2d(m, (const void* const*)h_epis, true, row_stride_bytes, V, S, U, C, 255.0f, dmin, dmax, dim_d, p, h_Ce_svu,
                         h_Ce_mask_svu, h_Cd_svu, h_depth_svu, h_rbar_svu, h_scan_mask_svu, stats);
}
RSLF_API_CATCH
