// librslf_hip.so, unit 8 of 9: host pointers in, host planes out -- Depth1DComputer_pile over one or several devices, the
// upload, the kernels and the download of successive scanline chunks overlapped.  C-ABI: include/rslf_hip.h.
#include "rslf_internal.hpp"

#include <algorithm>
#include <cmath>
#include <stdexcept>

using namespace rslf;

// ---- host pointers in, host planes out: pipelined, over one or several devices ------------------------------
//
// Depth1DComputer_pile's constructor + run() + getters (dc.hpp:425-565) in ONE call on host buffers (cv::Mat::data in,
// cv::Mat::data out).  Scanlines are independent up to the median's halo (DESIGN.md "Multi-GPU"), so the V EPIs are cut
// into blocks, one per device, and every block into chunks; a chunk is computed with `halo` recomputed rows either
// side and only its own rows are copied out, straight to their place in the caller's planes -- no collective.  Per
// device one host thread keeps three things in flight: the kernels of chunk k, the upload of chunk k+1 and the download
// of chunk k-1 (two volumes and two sets of result planes; copies to and from pageable host memory hold the host
// thread, never the GPU).  The result is bit-identical to the one-volume run.

void rslf::multi_free_dev(rslf_multi::Dev& d)
{
    if (!d.ctx)
        return;
    (void)hipSetDevice(d.ctx->device);
    for (int i = 0; i < 2; i++) {
        if (d.vol[i])
            (void)rslf_volume_destroy(d.vol[i]);
        (void)hipFree(d.planes[i]);
        (void)hipHostFree(d.pin[i]);
        if (i == 0)
            (void)hipFree(d.arena);
        if (d.done[i])
            (void)hipEventDestroy(d.done[i]);
    }
    if (d.s_up)
        (void)hipStreamDestroy(d.s_up);
    if (d.s_comp)
        (void)hipStreamDestroy(d.s_comp);
    if (d.s_down)
        (void)hipStreamDestroy(d.s_down);
    (void)rslf_ctx_destroy(d.ctx);
    d = rslf_multi::Dev();
}

extern "C" int rslf_multi_destroy(rslf_multi* m) RSLF_API_TRY
{
    if (!m)
        return RSLF_OK;
    for (auto& d : m->devs)
        multi_free_dev(d);
    delete m;
    return RSLF_OK;
}
RSLF_API_CATCH

extern "C" int rslf_multi_create(const int* devices, int n_devices, rslf_multi** out) RSLF_API_TRY
{
    if (!out)
        return fail(RSLF_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (n_devices < 0 || n_devices > 64 || (n_devices > 0 && !devices))
        return fail(RSLF_ERR_INVALID_ARG, "bad device list");
    rslf_multi* m = new (std::nothrow) rslf_multi();
    if (!m)
        return fail(RSLF_ERR_ALLOC, "out of host memory");
    struct Owner {   // until the object is handed over, an early return or an exception destroys it
        rslf_multi* m;
        ~Owner() { (void)rslf_multi_destroy(m); }
    } owner{m};
    const int n = n_devices > 0 ? n_devices : 1;
    m->devs.resize(n);
    for (int i = 0; i < n; i++) {
        rslf_multi::Dev& d = m->devs[i];
        int rc = rslf_ctx_create(n_devices > 0 ? devices[i] : 0, &d.ctx);   // a device may appear more than once
        hipError_t e = hipSuccess;
        if (rc == RSLF_OK) {
            // the upload stream outranks the compute stream: its blit and pack kernels then take the slots the scan's
            // workgroups free as they finish, instead of waiting behind the whole scan of the previous chunk
            int prio_lo = 0, prio_hi = 0;
            (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
            e = hipStreamCreateWithPriority(&d.s_up, hipStreamNonBlocking, prio_hi);
            if (e == hipSuccess) e = hipStreamCreateWithPriority(&d.s_comp, hipStreamNonBlocking, prio_lo);
            if (e == hipSuccess) e = hipStreamCreateWithFlags(&d.s_down, hipStreamNonBlocking);
            for (int k = 0; k < 2 && e == hipSuccess; k++)
                e = hipEventCreateWithFlags(&d.done[k], hipEventDisableTiming);
            if (e != hipSuccess)
                rc = fail(RSLF_ERR_HIP, "stream / event creation failed: %s", hipGetErrorString(e));
        }
        if (rc != RSLF_OK) {
            const std::string msg = last_error_buffer();
            return fail(rc, "%s", msg.c_str());   // (~Owner destroys what was built)
        }
    }
    // Peer access for every pair of distinct GPUs that allows it: the boundary-row exchange of the sharded sweep, the
    // device-out form of the pile path and the fine-to-coarse row transfers then go device to device over xGMI.  Where a
    // pair does not allow it hipMemcpyPeerAsync still works, staged through the host: correct, slower, and reported by
    // rslf_multi_peer_access (tools/multi_gpu_selftest.py prints the matrix).
    m->peer.assign((size_t)n * n, 0);
    int caller_device = -1;
    (void)hipGetDevice(&caller_device);   // enabling peers selects devices in turn: the caller's current one is put back
    struct RestoreDevice {
        int dev;
        ~RestoreDevice() { if (dev >= 0) (void)hipSetDevice(dev); }
    } restore{caller_device};
    for (int i = 0; i < n; i++)
        for (int k = 0; k < n; k++) {
            const int di = m->devs[(size_t)i].ctx->device, dk = m->devs[(size_t)k].ctx->device;
            if (di == dk) {
                m->peer[(size_t)i * n + k] = 1;
                continue;
            }
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, di, dk) != hipSuccess || !can)
                continue;
            HIP_TRY(hipSetDevice(di));
            const hipError_t e = hipDeviceEnablePeerAccess(dk, 0);
            if (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled)
                m->peer[(size_t)i * n + k] = 1;
            (void)hipGetLastError();   // "already enabled" is not an error to carry forward
        }
    owner.m = nullptr;
    *out = m;
    return RSLF_OK;
}
RSLF_API_CATCH

extern "C" int rslf_multi_device_count(const rslf_multi* m) RSLF_API_TRY
{
    return m ? (int)m->devs.size() : 0;
}
RSLF_API_CATCH

extern "C" int rslf_multi_peer_access(const rslf_multi* m, int from, int to) RSLF_API_TRY
{
    const int n = m ? (int)m->devs.size() : 0;
    if (!m || from < 0 || to < 0 || from >= n || to >= n)
        return fail(RSLF_ERR_INVALID_ARG, "rslf_multi_peer_access: bad index");
    return m->peer.empty() ? 0 : (int)m->peer[(size_t)from * n + to];
}
RSLF_API_CATCH

hipError_t rslf::multi_copy(rslf_multi*, void* dst, int dst_dev, const void* src, int src_dev, size_t bytes, hipStream_t st)
{
    // one device: a plain device copy; two: a peer copy -- direct over xGMI where rslf_multi_create enabled peer access,
    // staged through the host by the runtime where it could not
    return dst_dev == src_dev ? hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st)
                              : hipMemcpyPeerAsync(dst, dst_dev, src, src_dev, bytes, st);
}

extern "C" int rslf_multi_set_chunk_rows(rslf_multi* m, int rows) RSLF_API_TRY
{
    if (!m || rows < 0)
        return fail(RSLF_ERR_INVALID_ARG, "bad argument");
    m->chunk_rows = rows;
    return RSLF_OK;
}
RSLF_API_CATCH

namespace {

struct MultiJob {
    const void* const* h_epis;
    bool is_u8;
    size_t row_stride_bytes;
    int V, S, U, C;
    float scale_arg;      // f32: the divisor (already resolved, > 0 or as given); u8: unused
    float dmin, dmax;
    int dim_d, s_hat;
    const rslf_params* p;
    float* h_Ce;
    uint8_t* h_mask;
    float* h_Cd;
    float* h_depth;
    float* h_rbar;
    int32_t* h_idx;
    float* h_score;
    float* h_raw;
    int halo;
    int out_device;   // -1: the result planes are host memory; >= 0: they live on this device (peer copies)
};

typedef plan::RowBlock Chunk;   // owned rows [a, b) of the whole field, computed rows [lo, hi) = owned + halo, clipped

struct PlanePtrs {
    float *Ce, *Cd, *depth, *raw, *score, *rbar;
    int32_t* idx;
    uint8_t* mask;
};

PlanePtrs carve(char* blk, size_t n, int C)
{
    const plan::PlaneLayout o = plan::plane_layout(n, C, 0);
    PlanePtrs q;
    q.Ce = (float*)(blk + o.Ce);
    q.Cd = (float*)(blk + o.Cd);
    q.depth = (float*)(blk + o.depth);
    q.raw = (float*)(blk + o.raw);
    q.score = (float*)(blk + o.score);
    q.rbar = (float*)(blk + o.rbar);
    q.idx = (int32_t*)(blk + o.idx);
    q.mask = (uint8_t*)(blk + o.mask);
    return q;
}

// One device's share: rows [r0, r1) of the field, chunk by chunk.  Returns an rslf status; `err` receives the message.
int multi_worker(rslf_multi::Dev& d, const MultiJob& j, int r0, int r1, int chunk_rows, long long* scanned, int* kernel,
                 int* spad, std::string* err)
{
#define MW_FAIL(rc_)                  \
    do {                              \
        *err = last_error_buffer();                 \
        return (rc_);                 \
    } while (0)
#define MW_HIP(expr)                                                                                        \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess) {                                                                             \
            (void)fail(RSLF_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            MW_FAIL(RSLF_ERR_HIP);                                                                          \
        }                                                                                                   \
    } while (0)
    *scanned = 0;
    if (r1 <= r0)
        return RSLF_OK;
    rslf_ctx* ctx = d.ctx;
    MW_HIP(hipSetDevice(ctx->device));
    // Do the EPIs follow one another in host memory (a stacked array) or are they scattered over the heap (a Vec<Mat>)?
    const size_t in_row_bytes = (size_t)j.U * j.C * (j.is_u8 ? 1 : sizeof(float));
    const size_t in_epi_bytes = in_row_bytes * j.S;
    const bool scattered = plan::epis_scattered(j.h_epis, r0, r1, j.row_stride_bytes ? j.row_stride_bytes : in_row_bytes,
                                                in_row_bytes, in_epi_bytes);
    // Chunks (plan::chunk_plan).  A given size: uniform.  Automatic: a short first chunk so that the kernels start early
    // (its upload is the one copy nothing hides), then two large ones (stacked input; scattered input, which is gathered
    // into pinned memory first, takes a middling second chunk and pieces of about V/3.5) -- a chunk's scan is a grid of
    // its own, and a grid of 5.4 rounds of workgroups pays for 6 (eight equal chunks of a 1080-row field ran 16 % longer
    // than one launch over the whole field; measured with rocprofv3 on the host-in / host-out path).
    const std::vector<Chunk> chunks = plan::chunk_plan(r0, r1, j.V, j.halo, chunk_rows, scattered);
    const int max_rows = plan::max_held_rows(chunks);
    // two volumes and two sets of result planes, kept from call to call while the shape allows
    for (int k = 0; k < 2; k++) {
        if (d.vol[k] && (d.vol_rows[k] < max_rows || d.vol_S != j.S || d.vol_U != j.U || d.vol_C != j.C)) {
            (void)rslf_volume_destroy(d.vol[k]);
            d.vol[k] = nullptr;
        }
    }
    d.vol_S = j.S, d.vol_U = j.U, d.vol_C = j.C;
    for (int k = 0; k < 2; k++) {
        if (!d.vol[k]) {
            int rc = rslf_volume_create(ctx, max_rows, j.S, j.U, j.C, &d.vol[k]);
            if (rc)
                MW_FAIL(rc);
            d.vol_rows[k] = max_rows;
        }
    }
    const size_t n_max = (size_t)max_rows * j.U;
    // result planes of one chunk + a copy of the scan's per-scanline pixel counts (the context's own array is
    // rewritten by the next chunk's kernels, which are already queued when this chunk is collected)
    const plan::PlaneLayout lay = plan::plane_layout(n_max, j.C, max_rows);
    const size_t plane_bytes = lay.counts, bytes = lay.bytes;
    if (bytes > d.planes_cap) {
        for (int k = 0; k < 2; k++) {
            (void)hipFree(d.planes[k]);
            d.planes[k] = nullptr;
        }
        d.planes_cap = 0;
        for (int k = 0; k < 2; k++)
            MW_HIP(hipMalloc(&d.planes[k], bytes));
        d.planes_cap = bytes;
    }
    std::vector<int> counts((size_t)max_rows);
    // The scan's scratch -- pixel lists, and the records of grouped launches, whose count differs from chunk to chunk (a
    // short first chunk takes more hypothesis groups than the large ones) -- sized ONCE for every chunk of the plan before
    // the pipeline starts: a regrow in the middle is a hipFree + hipMalloc while the previous chunk's kernels are in
    // flight, safe only because hipFree synchronises the device, and it stalls the upload / compute overlap.
    {
        std::vector<int> rows_of;
        for (const Chunk& c : chunks)
            rows_of.push_back(c.hi - c.lo);
        int rc = scan_presize(ctx, j.S, j.U, j.C, j.dim_d, j.p, rows_of.data(), (int)rows_of.size());
        if (rc)
            MW_FAIL(rc);
    }
    // pinned staging for scattered EPIs
    const int pin_threads = std::max(1, std::min(8, (int)std::thread::hardware_concurrency() / 2));
    {
        const size_t need = (size_t)max_rows * in_epi_bytes;
        if (scattered && need > d.pin_cap) {
            for (int k = 0; k < 2; k++) {
                (void)hipHostFree(d.pin[k]);
                d.pin[k] = nullptr;
            }
            d.pin_cap = 0;
            for (int k = 0; k < 2; k++)
                MW_HIP(hipHostMalloc((void**)&d.pin[k], need, hipHostMallocDefault));
            d.pin_cap = need;
        }
    }

    // a volume object of the chunk's height over the (larger or equal) allocation: rows beyond are simply unused
    auto upload = [&](int k) -> int {
        const Chunk& c = chunks[k];
        rslf_volume* vol = d.vol[k & 1];
        vol->V = c.hi - c.lo;
        vol->bytes = (size_t)vol->V * vol->S * vol->C * vol->pitch * sizeof(float);
        ctx->stream = d.s_up;
        // EPIs that follow one another in host memory go up as they are: one pageable copy per run, at the link's rate.
        // EPIs scattered over the heap (a Vec<Mat>) would be one pageable copy each, and the runtime stages those through
        // its own bounce buffer on the calling thread at ~10 GB/s -- slower than the kernels consume them.  They are
        // gathered into a pinned buffer by a few host threads first (dense rows; ~25 GB/s per thread) and go up from there.
        const size_t esz = j.is_u8 ? 1 : sizeof(float);
        const size_t row_bytes = (size_t)j.U * j.C * esz;
        const size_t stride = j.row_stride_bytes ? j.row_stride_bytes : row_bytes;
        const size_t epi_bytes = row_bytes * j.S;
        const int rows = c.hi - c.lo;
        const int runs = plan::count_runs(j.h_epis + c.lo, rows, stride, row_bytes, epi_bytes);
        const void* const* src = j.h_epis + c.lo;
        std::vector<const void*> staged;
        size_t src_stride = j.row_stride_bytes;
        // (the pinned buffer is sized from the scattered-ness of the device's OWN rows; a chunk's halo rows can add breaks
        // of their own, so the buffer must also be seen to hold this chunk -- else the direct per-run copies below)
        if (plan::use_pinned_gather(runs, rows, epi_bytes, d.pin[k & 1] ? d.pin_cap : 0)) {
            char* pin = d.pin[k & 1];
            const int nt = std::max(1, std::min(rows, pin_threads));
            for (int i = 0; i < rows; i++)
                if (j.h_epis[c.lo + i] == nullptr)
                    return fail(RSLF_ERR_INVALID_ARG, "an EPI pointer is NULL");
            {
                JoinGuard pool;   // joined when this scope ends, however it ends
                for (int t = 0; t < nt; t++)
                    pool.run([&, t] {
                        int i0, i1;
                        plan::split_range(rows, t, nt, &i0, &i1);
                        for (int i = i0; i < i1; i++) {
                            const char* e = (const char*)j.h_epis[c.lo + i];
                            char* o = pin + (size_t)i * epi_bytes;
                            if (stride == row_bytes)
                                memcpy(o, e, epi_bytes);
                            else
                                for (int r = 0; r < j.S; r++)
                                    memcpy(o + (size_t)r * row_bytes, e + (size_t)r * stride, row_bytes);
                        }
                    });
            }
            staged.resize((size_t)rows);
            for (int i = 0; i < rows; i++)
                staged[(size_t)i] = pin + (size_t)i * epi_bytes;
            src = staged.data();
            src_stride = row_bytes;
        }
        int rc;
        if (j.is_u8)
            rc = upload_host<uint8_t>(vol, (const uint8_t* const*)src, src_stride, false, (float)(1.0 / 255.0));
        else
            rc = upload_host<float>(vol, (const float* const*)src, src_stride, false, scale_of(j.scale_arg));
        return rc;   // upload_host ends with a synchronisation of its stream (minmax_end): the pinned buffer is free again
    };
    auto compute = [&](int k) -> int {
        const Chunk& c = chunks[k];
        const size_t n = (size_t)(c.hi - c.lo) * j.U;
        const PlanePtrs q = carve(d.planes[k & 1], n, j.C);
        ctx->stream = d.s_comp;
        int rc = rslf_depth1d_pile_run(ctx, d.vol[k & 1], j.dmin, j.dmax, j.dim_d, j.s_hat, j.p, q.Ce, q.mask, q.Cd, q.depth, q.rbar,
                                       q.idx, q.score, q.raw, nullptr);
        if (rc)
            return rc;
        *kernel = ctx->last_kernel;
        *spad = ctx->last_spad;
        hipError_t e = hipMemcpyAsync(d.planes[k & 1] + plane_bytes, ctx->count, (size_t)(c.hi - c.lo) * sizeof(int),
                                      hipMemcpyDeviceToDevice, d.s_comp);
        if (e == hipSuccess)
            e = hipEventRecord(d.done[k & 1], d.s_comp);
        return e == hipSuccess ? RSLF_OK : fail(RSLF_ERR_HIP, "queueing the chunk's completion failed: %s", hipGetErrorString(e));
    };
    auto download = [&](int k) -> int {
        const Chunk& c = chunks[k];
        const int rows = c.hi - c.lo;
        const size_t n = (size_t)rows * j.U;
        const PlanePtrs q = carve(d.planes[k & 1], n, j.C);
        const size_t off = (size_t)(c.a - c.lo) * j.U, cnt = (size_t)(c.b - c.a) * j.U, dst = (size_t)c.a * j.U;
        hipError_t e = hipStreamWaitEvent(d.s_down, d.done[k & 1], 0);
        auto pull = [&](void* h, const void* dv, size_t esz, size_t mult) {
            if (e != hipSuccess || !h)
                return;
            if (j.out_device < 0)
                e = hipMemcpyAsync((char*)h + dst * esz * mult, (const char*)dv + off * esz * mult, cnt * esz * mult, hipMemcpyDeviceToHost,
                                   d.s_down);
            else   // device-out: each worker's rows go straight to their place in the planes on the output device (xGMI peer copy)
                e = multi_copy(nullptr, (char*)h + dst * esz * mult, j.out_device, (const char*)dv + off * esz * mult, ctx->device,
                               cnt * esz * mult, d.s_down);
        };
        pull(j.h_Ce, q.Ce, 4, 1);
        pull(j.h_mask, q.mask, 1, 1);
        pull(j.h_Cd, q.Cd, 4, 1);
        pull(j.h_depth, q.depth, 4, 1);
        pull(j.h_rbar, q.rbar, 4, (size_t)j.C);
        pull(j.h_idx, q.idx, 4, 1);
        pull(j.h_score, q.score, 4, 1);
        pull(j.h_raw, q.raw, 4, 1);
        // pixels scanned on the owned rows: the per-scanline counts of the scan's pixel lists
        if (e == hipSuccess)
            e = hipMemcpyAsync(counts.data(), d.planes[k & 1] + plane_bytes, (size_t)rows * sizeof(int), hipMemcpyDeviceToHost, d.s_down);
        if (e == hipSuccess)
            e = hipStreamSynchronize(d.s_down);
        if (e != hipSuccess)
            return fail(RSLF_ERR_HIP, "result download failed: %s", hipGetErrorString(e));
        for (int r = c.a - c.lo; r < c.b - c.lo; r++)
            *scanned += counts[(size_t)r];
        return RSLF_OK;
    };

    hipStream_t saved = ctx->stream;
    struct Restore {   // on every path out, exceptions included: the context's stream and the volumes' full height
        rslf_multi::Dev& d;
        hipStream_t saved;
        ~Restore()
        {
            d.ctx->stream = saved;
            for (int k = 0; k < 2; k++)
                if (d.vol[k]) {
                    d.vol[k]->V = d.vol_rows[k];
                    d.vol[k]->bytes = (size_t)d.vol[k]->V * d.vol[k]->S * d.vol[k]->C * d.vol[k]->pitch * sizeof(float);
                }
        }
    } restore{d, saved};
    struct Drain {     // an exception between queueing and collecting must not leave kernels writing freed planes
        bool armed = true;
        ~Drain()
        {
            if (armed)
                (void)hipDeviceSynchronize();
        }
    } drain;
    if (inject_hit(kInjectWorker))
        throw std::runtime_error("injected failure in a device worker (rslf_debug_inject)");
    if (inject_hit(kInjectAlloc))
        throw std::bad_alloc();
    int rc = upload(0);
    if (rc == RSLF_OK)
        rc = compute(0);
    for (int k = 0; rc == RSLF_OK && k < (int)chunks.size(); k++) {
        // chunk k's kernels are queued: feed the next chunk, then collect this one
        if (k + 1 < (int)chunks.size()) {
            rc = upload(k + 1);            // volume (k+1)&1 was last read by chunk k-1, whose download has completed
            if (rc == RSLF_OK)
                rc = compute(k + 1);       // planes (k+1)&1 likewise; queued behind chunk k on the compute stream
        }
        if (rc == RSLF_OK)
            rc = download(k);              // waits for chunk k's kernels; chunk k+1 runs meanwhile
    }
    if (rc != RSLF_OK)
        *err = last_error_buffer();
    else
        drain.armed = false;   // every chunk has been collected: nothing is in flight
    return rc;
#undef MW_FAIL
#undef MW_HIP
}

int multi_run(rslf_multi* m, MultiJob j, rslf_stats* stats)
{
    if (!m || !j.h_epis)
        return fail(RSLF_ERR_INVALID_ARG, "NULL argument");
    if (j.V < 1 || j.S < 1 || j.U < 1 || (j.C != 1 && j.C != 3))
        return fail(RSLF_ERR_INVALID_ARG, "bad dimensions V=%d S=%d U=%d C=%d", j.V, j.S, j.U, j.C);
    int rc = check_params(j.p);
    if (rc)
        return rc;
    for (int v = 0; v < j.V; v++)
        if (!j.h_epis[v])
            return fail(RSLF_ERR_INVALID_ARG, "h_epis[%d] is NULL", v);
    j.s_hat = plan::resolve_s_hat(j.s_hat, j.S);
    // rows either side of a chunk that must be recomputed for the chunk's own rows to come out exact (plan::halo_rows)
    j.halo = plan::halo_rows(j.p->median_filter_size, j.p->edge_confidence_opening_size);
    const int nd = (int)m->devs.size();
    std::vector<long long> scanned((size_t)nd, 0);
    std::vector<int> rcs((size_t)nd, RSLF_OK), kern((size_t)nd, 0), spads((size_t)nd, 0);
    std::vector<std::string> errs((size_t)nd);
    {
        JoinGuard pool;   // one host thread per device; joined on every path out of this scope
        for (int i = 0; i < nd; i++) {
            const plan::RowBlock blk = plan::row_block(j.V, i, nd, 0);
            // chunks: enough of them to overlap the copies with the kernels, large enough to keep the halo's share small
            const int chunk = m->chunk_rows > 0 ? std::max(1, std::min(m->chunk_rows, std::max(1, blk.b - blk.a))) : 0;   // 0: the graded plan
            pool.run([&, i, blk, chunk] {
                // nothing may leave a thread function by exception: a status and its text instead
                rcs[(size_t)i] = guarded_status(
                    [&] {
                        return multi_worker(m->devs[(size_t)i], j, blk.a, blk.b, chunk, &scanned[(size_t)i], &kern[(size_t)i],
                                            &spads[(size_t)i], &errs[(size_t)i]);
                    },
                    &errs[(size_t)i]);
            });
        }
    }
    for (int i = 0; i < nd; i++)
        if (rcs[(size_t)i] != RSLF_OK)
            return fail(rcs[(size_t)i], "device %d: %s", m->devs[(size_t)i].ctx->device, errs[(size_t)i].c_str());
    if (stats) {
        long long tot = 0;
        for (long long s : scanned)
            tot += s;
        stats->pixels_scanned = tot;
        stats->units = tot * j.dim_d;
        stats->scan_kernel = kern[0];
        stats->s_pad = spads[0];
    }
    return RSLF_OK;
}

}  // namespace

static int multi_pile_f32(rslf_multi* m, const float* const* h_epis, size_t row_stride_bytes, int V, int S, int U, int C,
                          float epi_scale_factor, float dmin, float dmax, int dim_d, int s_hat, const rslf_params* p, int out_device,
                          float* h_Ce_vu, uint8_t* h_Ce_mask_vu, float* h_Cd_vu, float* h_depth_vu, float* h_rbar_vu, int32_t* h_idx_vu,
                          float* h_score_vu, float* h_depth_raw_vu, rslf_stats* stats, float* scale_used)
{
    if (!m || !h_epis || V < 1)
        return fail(RSLF_ERR_INVALID_ARG, "NULL argument");
    const size_t row_elems = (size_t)U * C;
    const size_t stride = row_stride_bytes ? row_stride_bytes : row_elems * sizeof(float);
    for (int v = 0; v < V; v++)
        if (!h_epis[v])
            return fail(RSLF_ERR_INVALID_ARG, "h_epis[%d] is NULL", v);
    // dc.hpp:442-460: the default scale is the maximum over ALL EPIs -- taken once here, never per block
    if (epi_scale_factor < 0)
        epi_scale_factor = host_max_f32_parallel(h_epis, V, S, stride, row_elems, epi_scale_factor);
    if (scale_used)
        *scale_used = epi_scale_factor;
    MultiJob j = {(const void* const*)h_epis, false, stride, V, S, U, C, epi_scale_factor, dmin, dmax, dim_d, s_hat, p,
                  h_Ce_vu, h_Ce_mask_vu, h_Cd_vu, h_depth_vu, h_rbar_vu, h_idx_vu, h_score_vu, h_depth_raw_vu, 0, out_device};
    return multi_run(m, j, stats);
}

extern "C" int rslf_multi_depth1d_pile_f32(rslf_multi* m, const float* const* h_epis, size_t row_stride_bytes, int V, int S, int U,
                                           int C, float epi_scale_factor, float dmin, float dmax, int dim_d, int s_hat,
                                           const rslf_params* p, float* h_Ce_vu, uint8_t* h_Ce_mask_vu, float* h_Cd_vu,
                                           float* h_depth_vu, float* h_rbar_vu, int32_t* h_idx_vu, float* h_score_vu,
                                           float* h_depth_raw_vu, rslf_stats* stats, float* scale_used) RSLF_API_TRY
{
    return multi_pile_f32(m, h_epis, row_stride_bytes, V, S, U, C, epi_scale_factor, dmin, dmax, dim_d, s_hat, p, -1, h_Ce_vu,
                          h_Ce_mask_vu, h_Cd_vu, h_depth_vu, h_rbar_vu, h_idx_vu, h_score_vu, h_depth_raw_vu, stats, scale_used);
}
RSLF_API_CATCH

extern "C" int rslf_multi_depth1d_pile_f32_dev(rslf_multi* m, const float* const* h_epis, size_t row_stride_bytes, int V, int S, int U,
                                               int C, float epi_scale_factor, float dmin, float dmax, int dim_d, int s_hat,
                                               const rslf_params* p, int out_device, float* d_Ce_vu, uint8_t* d_Ce_mask_vu,
                                               float* d_Cd_vu, float* d_depth_vu, float* d_rbar_vu, int32_t* d_idx_vu,
                                               float* d_score_vu, float* d_depth_raw_vu, rslf_stats* stats, float* scale_used) RSLF_API_TRY
{
    if (out_device < 0)
        return fail(RSLF_ERR_INVALID_ARG, "out_device %d", out_device);
    return multi_pile_f32(m, h_epis, row_stride_bytes, V, S, U, C, epi_scale_factor, dmin, dmax, dim_d, s_hat, p, out_device, d_Ce_vu,
                          d_Ce_mask_vu, d_Cd_vu, d_depth_vu, d_rbar_vu, d_idx_vu, d_score_vu, d_depth_raw_vu, stats, scale_used);
}
RSLF_API_CATCH

extern "C" int rslf_multi_depth1d_pile_u8(rslf_multi* m, const uint8_t* const* h_epis, size_t row_stride_bytes, int V, int S, int U,
                                          int C, float dmin, float dmax, int dim_d, int s_hat, const rslf_params* p, float* h_Ce_vu,
                                          uint8_t* h_Ce_mask_vu, float* h_Cd_vu, float* h_depth_vu, float* h_rbar_vu, int32_t* h_idx_vu,
                                          float* h_score_vu, float* h_depth_raw_vu, rslf_stats* stats) RSLF_API_TRY
{
    MultiJob j = {(const void* const*)h_epis, true, row_stride_bytes, V, S, U, C, 255.0f, dmin, dmax, dim_d, s_hat, p,
                  h_Ce_vu, h_Ce_mask_vu, h_Cd_vu, h_depth_vu, h_rbar_vu, h_idx_vu, h_score_vu, h_depth_raw_vu, 0, -1};
    return multi_run(m, j, stats);
}
RSLF_API_CATCH
