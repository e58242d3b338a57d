// What the translation units of librslf_hip.so share: the error convention and the exception barrier of the C-ABI, the
// context / volume / multi-device objects, and the helpers one unit offers the others.  No kernels here -- each unit
// includes the kernel headers it launches (device code is per translation unit).
//
//   rslf_core.hip         errors, contexts, volumes, host upload / device pack (K0)
//   rslf_pile.hip         the hot path: edge confidence (K1), scan (K2), selective median (K3), Depth1DComputer(_pile)
//   rslf_chip_a/b/c.hip   the on-chip scan kernel's instantiations, one per rung of its ladder (launched by rslf_pile.hip)
//   rslf_sweep.hip        the 2-D sweep and its propagation (K4)
//   rslf_f2c.hip          fine-to-coarse: pyramid, bound tightening, fusion (K5) and the native level loop
//   rslf_multi.hip        host pointers in / host planes out, pipelined over one or several devices (pile path)
//   rslf_multi_sweep.hip  the sharded sweep and fine-to-coarse behind the C-ABI
//   rslf_plan.hpp         every host-side decision as pure functions (unit-tested on the CPU under ASan / UBSan)
#pragma once

#include "../../include/rslf_hip.h"

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <exception>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "rslf_device.hpp"
#include "rslf_plan.hpp"

// ---- errors ---------------------------------------------------------------

namespace rslf {

char* last_error_buffer();   // thread-local, 512 bytes (rslf_core.hip)

inline int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(last_error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

// Testing hook (rslf_debug_inject, include/rslf_hip.h): the named site throws / fails the next `count` times it is
// reached.  One relaxed atomic load per site visit; sites sit on host control paths only, never in a launch loop.
enum InjectSite { kInjectWorker = 0, kInjectThreadCreate = 1, kInjectAlloc = 2, kInjectSites = 3 };
bool inject_hit(InjectSite site);   // true (and one count consumed) when the site should fail now

}  // namespace rslf

#define HIP_TRY(expr)                                                                               \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            return rslf::fail(RSLF_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// The exception barrier of the C boundary (include/rslf_hip.h: "never throws across the boundary"; the reference's seam
// returns void and has no error path at all, core.hpp:279-310).  EVERY `extern "C" int rslf_*` definition is a
// function-try-block closed by this handler list -- tests/test_abi.py greps for it:
//
//     extern "C" int rslf_foo(args) RSLF_API_TRY
//     {
//         ...
//     }
//     RSLF_API_CATCH
//
// std::bad_alloc -> RSLF_ERR_ALLOC, any other std::exception -> RSLF_ERR_INTERNAL with its what() in rslf_last_error(),
// anything else -> RSLF_ERR_INTERNAL.  Threads started inside an entry point are owned by a JoinGuard, so an exception (or
// an early return) can never leave a joinable std::thread behind (std::terminate).
#define RSLF_API_TRY try
#define RSLF_API_CATCH                                                                              \
    catch (const std::bad_alloc&)                                                                   \
    {                                                                                               \
        return rslf::fail(RSLF_ERR_ALLOC, "out of host memory (std::bad_alloc)");                   \
    }                                                                                               \
    catch (const std::exception& e_)                                                                \
    {                                                                                               \
        return rslf::fail(RSLF_ERR_INTERNAL, "internal error: %s", e_.what());                      \
    }                                                                                               \
    catch (...)                                                                                     \
    {                                                                                               \
        return rslf::fail(RSLF_ERR_INTERNAL, "internal error: unknown exception");                  \
    }

namespace rslf {

// Owns the worker threads of one entry point: joins whatever is joinable when the scope ends, however it ends.
// run(f): on a new thread if one can be had, else on the calling thread (std::system_error from the constructor -- the
// GPU boxes cap a process's threads -- or an injected failure) -- the work is done either way.
class JoinGuard {
public:
    JoinGuard() = default;
    JoinGuard(const JoinGuard&) = delete;
    JoinGuard& operator=(const JoinGuard&) = delete;
    ~JoinGuard() { join_all(); }
    template <typename F>
    void run(F f)
    {
        bool started = false;
        try {
            if (!inject_hit(kInjectThreadCreate)) {
                threads_.reserve(threads_.size() + 1);   // may throw bad_alloc: before the thread exists
                threads_.emplace_back(f);
                started = true;
            }
        } catch (const std::system_error&) {
            started = false;
        }
        if (!started)
            f();   // no thread to be had: the caller does the work itself
    }
    void join_all()
    {
        for (std::thread& t : threads_)
            if (t.joinable())
                t.join();
        threads_.clear();
    }

private:
    std::vector<std::thread> threads_;
};

// Runs `f` (an `int()` returning an rslf status) and turns anything it throws into a status + message, for worker threads:
// an exception must not leave a thread function either.
template <typename F>
int guarded_status(F f, std::string* err)
{
    try {
        return f();
    } catch (const std::bad_alloc&) {
        if (err)
            *err = "out of host memory (std::bad_alloc)";
        return RSLF_ERR_ALLOC;
    } catch (const std::exception& e) {
        if (err)
            *err = std::string("internal error: ") + e.what();
        return RSLF_ERR_INTERNAL;
    } catch (...) {
        if (err)
            *err = "internal error: unknown exception";
        return RSLF_ERR_INTERNAL;
    }
}

}  // namespace rslf

// ---- objects --------------------------------------------------------------

namespace rslf {
struct Partial;   // k2_scan.hpp: one lane's merged result over one group's hypotheses (32 bytes)
}

struct rslf_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    // scratch, grown on demand (never inside a timed launch sequence after the first call)
    int* list = nullptr;
    int* count = nullptr;
    float* depth_tmp = nullptr;
    size_t plane_cap = 0;   // pixels list/depth_tmp can hold
    int count_cap = 0;
    unsigned long long* total = nullptr;   // device counter
    float* partial = nullptr;              // pack min/max partials
    size_t partial_cap = 0;
    float* minmax = nullptr;               // device [2]
    void* staging = nullptr;
    size_t staging_cap = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool ev_valid = false;
    // "time_all" (rslf_ctx_set_debug): every scan launch sequence is bracketed by its own pair of events from this pool --
    // rslf_scan_time_total_ms sums them -- so that a sweep's or a pyramid's SUMMED K2 time can be reported (bench.py's
    // roofline on the sweep2d / f2c lines).  Off by default: an event is a packet of its own in the queue (~5.6 us each).
    int time_all = 0;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    int last_spad = 0;   // register-scan slot count of the last K2 launch, 0 = none
    int last_kernel = 0; // RSLF_SCAN_* of the last K2 launch
    int num_cus = 0;           // compute units of the device (how many workgroups a launch needs to fill it)
    bool keep_total = false;   // the 2-D sweep sums the scanned pixels of all its visits
    int scan_groups = 1;       // hypothesis groups per tile for the next scan launches (the 2-D sweep raises it)
    bool scan_packed = false;  // next scan launches use one packed pixel list (sparse visits of the 2-D sweep)
    // test / tuning hooks (rslf_ctx_set_debug), per context: 0 / -1 = automatic
    int force_scan = 0;        // 1 generic kernel, 2 streaming kernel, 3 on-chip kernel
    int force_groups = 0;      // hypothesis groups per tile
    int force_packed = -1;     // 0 / 1
    int px_mode = -1;          // pixel-per-wave kernel on packed launches: -1 automatic, 0 never, 1 whenever it can run
    int stream_groups = 0;     // streaming kernel, dense launches: hypothesis groups per tile (0 = kStreamGroups)
    int stream_share = 1;      // streaming kernel: 63-pixel row tiles whose tail shares taps between neighbouring lanes: 0 never, 1 where the tail is long (plan::stream_shares_taps), 2 always
    size_t stream_lds_bytes = rslf::plan::kStreamLdsBytes;   // dynamic LDS of one streaming workgroup
    int row_split = 1;         // packed launches of stream-class volumes: rows with many pixels as row tiles of the list (0: off; A/B and tests)
    int claim_skip = 1;        // 2-D sweep: the claims skip views with nothing left to paint within reach (0: off, A/B and tests)
    rslf::Partial* scan_partial = nullptr;   // [tile][group][64] records of grouped scan launches
    size_t partial_rec_cap = 0;
    int* scan_ticket = nullptr;        // [tile] of the same launches: which group merges the tile (zero between launches)
    size_t ticket_cap = 0;
    bool packed_n_clean = false;       // the packed list's length is already 0 (the sweep's apply pass resets it)
    int precompacted = 0;              // the next scan's pixel lists and total are already in place: 1 = per-row lists (K1 +
                                       // compaction in one launch), 2 = the packed list (a sweep's apply pass made it)
    int sweep_expect = -1;             // the view the sweep visits next (core.hpp:981-990), -1 once all are done
    bool sweep_open = false;           // between rslf_sweep_begin and rslf_sweep_end
    bool sweep_first = true;           // the next visit is the sweep's first (dense) one
    uint8_t* sweep_mask_run = nullptr; // the running masks [S][V][U] of the open sweep
    // 2-D sweep scratch
    int* winner = nullptr;        // [S][V][U]
    uint8_t* dirty = nullptr;     // [S][V][ceil(U/256)]: segments of the winner rows that hold a claim (all 0 between visits)
    int* remain = nullptr;        // [S][V][ceil(U/256)]: pixels left in the running mask per segment (lets the claims skip views)
    size_t dirty_cap = 0;
    uint8_t* sweep_mask = nullptr;
    float* filtered = nullptr;    // [V][U] median of the visited view, the propagation's source
    size_t sweep_cap = 0;         // entries winner / sweep_mask can hold (S*V*U)
    size_t sweep_plane_cap = 0;   // floats `filtered` can hold (V*U)
    // grow-only scratch of the once-per-level helpers (pyramid, tightening, fusion): reused across calls, so
    // these helpers neither allocate nor free -- and so never force a device-wide synchronisation
    static constexpr int kHelperSlots = 4;
    void* helper[kHelperSlots] = {nullptr, nullptr, nullptr, nullptr};
    size_t helper_cap[kHelperSlots] = {0, 0, 0, 0};
};

struct rslf_volume {
    rslf_ctx* ctx = nullptr;
    int device = 0;   // kept here too: a volume may be destroyed after its context
    int V = 0, S = 0, U = 0, C = 0, pitch = 0;
    float* base = nullptr;
    size_t bytes = 0;
    float min_value = 0.0f, max_value = 0.0f;
    bool filled = false;
};

struct rslf_multi {
    struct Dev {
        rslf_ctx* ctx = nullptr;
        hipStream_t s_up = nullptr, s_comp = nullptr, s_down = nullptr;
        hipEvent_t done[2] = {nullptr, nullptr};
        rslf_volume* vol[2] = {nullptr, nullptr};
        int vol_rows[2] = {0, 0}, vol_S = 0, vol_U = 0, vol_C = 0;
        char* planes[2] = {nullptr, nullptr};
        size_t planes_cap = 0;
        char* pin[2] = {nullptr, nullptr};   // pinned host staging for EPIs scattered over the heap (Vec<Mat>)
        size_t pin_cap = 0;
        char* arena = nullptr;               // the sweep forms' planes, kept from call to call (and from level to level)
        size_t arena_cap = 0;
    };
    std::vector<Dev> devs;
    int chunk_rows = 0;   // 0 = automatic
    // peer access between the devices of this object (rslf_multi_create): peer[i * n + k] = device i can map device k's
    // memory (hipDeviceCanAccessPeer) AND the access is enabled; copies between devices without it stage through the host
    std::vector<unsigned char> peer;
};

namespace rslf {

inline VolView view_of(const rslf_volume* vol)
{
    VolView w;
    w.base = vol->base;
    w.V = vol->V;
    w.S = vol->S;
    w.U = vol->U;
    w.C = vol->C;
    w.pitch = vol->pitch;
    w.stride_s = (long long)vol->pitch * vol->C;
    w.stride_v = (long long)vol->S * w.stride_s;
    return w;
}

inline float scale_of(float epi_scale_factor)
{
    return (float)(1.0 / (double)epi_scale_factor);   // dc.hpp:474 through cvtScale's float scale
}

// scoped device scratch for the once-per-call helpers
struct DevBuf {
    void* p = nullptr;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
    void release()
    {
        (void)hipFree(p);
        p = nullptr;
    }
};

// rslf_core.hip
int check_params(const rslf_params* p);
ScanConsts make_scan_consts(const rslf_params* p);
int ensure_plane_scratch(rslf_ctx* ctx, int V, int U);
int ensure_group_scratch(rslf_ctx* ctx, size_t recs, size_t tiles);
int ensure_staging(rslf_ctx* ctx, size_t bytes);
int helper_scratch(rslf_ctx* ctx, int slot, size_t bytes, void** out);
float host_max_f32(const float* const* h_ptrs, int n_ptrs, int rows, size_t row_stride_bytes, size_t row_elems, float start);
float host_max_f32_parallel(const float* const* h_epis, int V, int S, size_t stride, size_t row_elems, float start);
template <typename SrcT>
int upload_host(rslf_volume* vol, const SrcT* const* h_ptrs, size_t row_stride_bytes, bool image_major, float scale);
extern template int upload_host<float>(rslf_volume*, const float* const*, size_t, bool, float);
extern template int upload_host<uint8_t>(rslf_volume*, const uint8_t* const*, size_t, bool, float);

// rslf_pile.hip
bool scan_takes_stream(const rslf_volume* vol);   // would a linear-interpolation scan of this volume run a grouped LDS kernel?
void fill_stats(rslf_ctx* ctx, unsigned long long tot, int dim_d, rslf_stats* stats);
int scan_presize(rslf_ctx* ctx, int S, int U, int C, int dim_d, const rslf_params* p, const int* rows, int n_rows);

// rslf_f2c.hip: small elementwise launches the multi-device form shares
int f2c_u8_to_f32(hipStream_t st, const uint8_t* in, float* out, size_t n);
int f2c_fill_f32(hipStream_t st, float* out, size_t n, float value);
int f2c_valid_mask(hipStream_t st, const float* Ce, uint8_t* out, size_t n, float thr);

// rslf_multi.hip
void multi_free_dev(rslf_multi::Dev& d);
// a copy between two devices of a multi object (or within one), queued on `st`: a plain device copy on one device, else a
// peer copy (direct over xGMI where rslf_multi_create enabled peer access, staged through the host by the runtime where not)
hipError_t multi_copy(rslf_multi* m, void* dst, int dst_dev, const void* src, int src_dev, size_t bytes, hipStream_t st);

}  // namespace rslf
