// rslf_hip.hpp -- C++11 host side of the MI355X EPI depth scan.
//
// Keeps the constructor / run() shape of rslf::Depth1DComputer_pile<DataType>
// (RSLightFields/include/rslf_depth_computation.hpp:93-143, :425-565) and of
// rslf::Depth1DParameters<DataType> (rslf_depth_computation_core.hpp:66-142) so
// the class drops into the reference's demos (tests/test_depth_computation_pile.cpp:49-51)
// in place of the OpenMP path.  Everything below the class goes through the
// C-ABI of rslf_hip.h; nothing here computes.
//
// OpenCV is optional: when <opencv2/core/core.hpp> is on the include path
// (the reference's own build, CMakeLists.txt:29) the cv::Mat constructor and
// getters are compiled in; otherwise the same class works on plain pointers.
// Header-only, C++11, exceptions carry the C-ABI's error text.
#ifndef RSLF_HIP_HPP
#define RSLF_HIP_HPP

#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "rslf_hip.h"

#if defined(__has_include)
#if __has_include(<opencv2/core/core.hpp>)
#include <opencv2/core/core.hpp>
#define RSLFX_HAVE_OPENCV 1
#endif
#endif

namespace rslfx {

struct Error : std::runtime_error {
    int status;
    Error(int st, const std::string& where)
        : std::runtime_error(where + ": " + rslf_status_string(st) + ": " + rslf_last_error()), status(st) {}
};

inline void check(int st, const char* where)
{
    if (st != RSLF_OK)
        throw Error(st, where);
}

// rslf::Depth1DParameters<T>: same member names (par_*), same defaults.
struct Depth1DParameters {
    float par_edge_score_threshold;
    float par_line_score_threshold;
    float par_disp_score_threshold;
    float par_raw_score_threshold;
    float par_mean_shift_max_iter;
    int par_edge_confidence_filter_size;
    int par_edge_confidence_opening_type;
    int par_edge_confidence_opening_size;
    int par_median_filter_size;
    float par_median_filter_epsilon;
    float par_propagation_epsilon;
    float par_slope_factor;
    bool par_cut_shadows;
    float par_shadow_level;
    float par_kernel_bandwidth;   // BandwidthKernel(_BANDWIDTH_KERNEL_PARAMETER), core.hpp:78
    int par_interpolation_class;  // RSLF_INTERP_*: stands for the Interpolation1DClass* of core.hpp:108 (default Linear, :76)
    bool par_use_disp_confidence_score;   // the reference's build switch _USE_DISP_CONFIDENCE_SCORE (core.hpp:35), off by default

    Depth1DParameters()
    {
        rslf_params p;
        rslf_default_params(&p);
        par_edge_score_threshold = p.edge_score_threshold;
        par_line_score_threshold = p.line_score_threshold;
        par_disp_score_threshold = p.disp_score_threshold;
        par_raw_score_threshold = p.raw_score_threshold;
        par_mean_shift_max_iter = p.mean_shift_max_iter;
        par_edge_confidence_filter_size = p.edge_confidence_filter_size;
        par_edge_confidence_opening_type = p.edge_confidence_opening_type;
        par_edge_confidence_opening_size = p.edge_confidence_opening_size;
        par_median_filter_size = p.median_filter_size;
        par_median_filter_epsilon = p.median_filter_epsilon;
        par_propagation_epsilon = p.propagation_epsilon;
        par_slope_factor = p.slope_factor;
        par_cut_shadows = p.cut_shadows != 0;
        par_shadow_level = p.shadow_level;
        par_kernel_bandwidth = p.kernel_bandwidth;
        par_interpolation_class = p.interpolation;
        par_use_disp_confidence_score = p.use_disp_confidence_score != 0;
    }

    static Depth1DParameters& get_default()
    {
        static Depth1DParameters s_default;   // core.hpp:138-142
        return s_default;
    }

    rslf_params to_c() const
    {
        rslf_params p;
        p.edge_score_threshold = par_edge_score_threshold;
        p.line_score_threshold = par_line_score_threshold;
        p.disp_score_threshold = par_disp_score_threshold;
        p.raw_score_threshold = par_raw_score_threshold;
        p.mean_shift_max_iter = par_mean_shift_max_iter;
        p.edge_confidence_filter_size = par_edge_confidence_filter_size;
        p.edge_confidence_opening_type = par_edge_confidence_opening_type;
        p.edge_confidence_opening_size = par_edge_confidence_opening_size;
        p.median_filter_size = par_median_filter_size;
        p.median_filter_epsilon = par_median_filter_epsilon;
        p.propagation_epsilon = par_propagation_epsilon;
        p.slope_factor = par_slope_factor;
        p.cut_shadows = par_cut_shadows ? 1 : 0;
        p.shadow_level = par_shadow_level;
        p.kernel_bandwidth = par_kernel_bandwidth;
        p.interpolation = par_interpolation_class;
        p.use_disp_confidence_score = par_use_disp_confidence_score ? 1 : 0;
        return p;
    }
};

// RAII handles
class Context {
public:
    explicit Context(int device = 0) : h_(nullptr) { check(rslf_ctx_create(device, &h_), "rslf_ctx_create"); }
    ~Context() { rslf_ctx_destroy(h_); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    rslf_ctx* get() const { return h_; }
    void set_stream(void* hip_stream) { check(rslf_ctx_set_stream(h_, hip_stream), "rslf_ctx_set_stream"); }
    void synchronize() { check(rslf_ctx_synchronize(h_), "rslf_ctx_synchronize"); }
    float last_scan_kernel_ms()
    {
        float ms = 0;
        check(rslf_last_scan_kernel_ms(h_, &ms), "rslf_last_scan_kernel_ms");
        return ms;
    }

private:
    rslf_ctx* h_;
};

// Several devices (or several workers on one device) behind one handle: the pile path then cuts the scanlines into one
// block per device and overlaps upload, kernels and download chunk by chunk (rslf_hip.h, rslf_multi_*).
class MultiContext {
public:
    MultiContext() : h_(nullptr) { check(rslf_multi_create(nullptr, 0, &h_), "rslf_multi_create"); }
    explicit MultiContext(const std::vector<int>& devices) : h_(nullptr)
    {
        check(rslf_multi_create(devices.empty() ? nullptr : devices.data(), (int)devices.size(), &h_), "rslf_multi_create");
    }
    ~MultiContext() { rslf_multi_destroy(h_); }
    MultiContext(const MultiContext&) = delete;
    MultiContext& operator=(const MultiContext&) = delete;
    rslf_multi* get() const { return h_; }
    int device_count() const { return rslf_multi_device_count(h_); }
    void set_chunk_rows(int rows) { check(rslf_multi_set_chunk_rows(h_, rows), "rslf_multi_set_chunk_rows"); }

private:
    rslf_multi* h_;
};

// rslf::Depth1DComputer_pile<DataType>.  DataType is float (1 channel) or a
// 3-float pixel (cv::Vec3f in the reference, dc.hpp:149-154): only its channel
// count matters here.
template <int CHANNELS>
class Depth1DComputer_pile {
public:
    // The reference's constructor (dc.hpp:97-106) on raw pointers: epis[v] points
    // at an S x U image of CHANNELS interleaved values, rows row_stride_bytes
    // apart (cv::Mat::data / cv::Mat::step; 0 = dense).  is_u8 selects the
    // uchar branch of dc.hpp:468-475.
    Depth1DComputer_pile(Context& ctx, const void* const* epis, bool is_u8, int dim_v, int dim_s, int dim_u,
                         size_t row_stride_bytes, float dmin, float dmax, int dim_d, int s_hat = -1,
                         float epi_scale_factor = -1, const Depth1DParameters& parameters = Depth1DParameters::get_default())
        : ctx_(&ctx), multi_(nullptr), vol_(nullptr), m_parameters(parameters)
    {
        init(epis, is_u8, dim_v, dim_s, dim_u, row_stride_bytes, dmin, dmax, dim_d, s_hat, epi_scale_factor);
    }

    // The same on a MultiContext: the scanlines are shared out over its devices and the host copies overlap the kernels.
    // Here the EPIs are read by run(), not copied by the constructor: the buffers must outlive it (they do in the
    // reference's demos, where the Vec<Mat> lives next to the computer).
    Depth1DComputer_pile(MultiContext& multi, const void* const* epis, bool is_u8, int dim_v, int dim_s, int dim_u,
                         size_t row_stride_bytes, float dmin, float dmax, int dim_d, int s_hat = -1,
                         float epi_scale_factor = -1, const Depth1DParameters& parameters = Depth1DParameters::get_default())
        : ctx_(nullptr), multi_(&multi), vol_(nullptr), m_parameters(parameters)
    {
        init_multi(epis, is_u8, dim_v, dim_s, dim_u, row_stride_bytes, dmin, dmax, dim_d, s_hat, epi_scale_factor);
    }

#ifdef RSLFX_HAVE_OPENCV
    Depth1DComputer_pile(MultiContext& multi, const std::vector<cv::Mat>& epis, float dmin, float dmax, int dim_d, int s_hat = -1,
                         float epi_scale_factor = -1, const Depth1DParameters& parameters = Depth1DParameters::get_default())
        : ctx_(nullptr), multi_(&multi), vol_(nullptr), m_parameters(parameters)
    {
        bool is_u8 = false;
        const std::vector<const void*> ptrs = mat_pointers(epis, is_u8);
        init_multi(ptrs.data(), is_u8, (int)epis.size(), epis[0].rows, epis[0].cols, epis[0].step[0], dmin, dmax, dim_d, s_hat,
                   epi_scale_factor);
    }
    // Exactly the reference's signature: Vec<Mat> in (dc.hpp:97-106).
    Depth1DComputer_pile(Context& ctx, const std::vector<cv::Mat>& epis, float dmin, float dmax, int dim_d, int s_hat = -1,
                         float epi_scale_factor = -1, const Depth1DParameters& parameters = Depth1DParameters::get_default())
        : ctx_(&ctx), multi_(nullptr), vol_(nullptr), m_parameters(parameters)
    {
        bool is_u8 = false;
        const std::vector<const void*> ptrs = mat_pointers(epis, is_u8);
        init(ptrs.data(), is_u8, (int)epis.size(), epis[0].rows, epis[0].cols, epis[0].step[0], dmin, dmax, dim_d, s_hat,
             epi_scale_factor);
    }
    static std::vector<const void*> mat_pointers(const std::vector<cv::Mat>& epis, bool& is_u8)
    {
        if (epis.empty())
            throw std::invalid_argument("Depth1DComputer_pile: no EPIs");
        if (epis[0].channels() != CHANNELS)
            throw std::invalid_argument("Depth1DComputer_pile: channel count does not match the instantiation");
        std::vector<const void*> ptrs(epis.size());
        for (size_t v = 0; v < epis.size(); v++) {
            if (epis[v].rows != epis[0].rows || epis[v].cols != epis[0].cols || epis[v].type() != epis[0].type())
                throw std::invalid_argument("Depth1DComputer_pile: EPIs differ in size or type");
            ptrs[v] = epis[v].data;
        }
        is_u8 = epis[0].depth() == CV_8U;
        if (!is_u8 && epis[0].depth() != CV_32F)
            throw std::invalid_argument("Depth1DComputer_pile: EPIs must be CV_8U or CV_32F");
        return ptrs;
    }
    cv::Mat get_edge_confidence() const { return cv::Mat(dim_v_, dim_u_, CV_32FC1, (void*)m_edge_confidence_v_u.data()).clone(); }
    cv::Mat get_edge_confidence_mask() const { return cv::Mat(dim_v_, dim_u_, CV_8UC1, (void*)m_edge_confidence_mask_v_u.data()).clone(); }
    cv::Mat get_disp_confidence() const { return cv::Mat(dim_v_, dim_u_, CV_32FC1, (void*)m_disp_confidence_v_u.data()).clone(); }
    cv::Mat get_best_depth() const { return cv::Mat(dim_v_, dim_u_, CV_32FC1, (void*)m_best_depth_v_u.data()).clone(); }
    cv::Mat get_rbar() const { return cv::Mat(dim_v_, dim_u_, CV_32FC(CHANNELS), (void*)m_rbar_v_u.data()).clone(); }
#endif

    ~Depth1DComputer_pile() { rslf_volume_destroy(vol_); }
    Depth1DComputer_pile(const Depth1DComputer_pile&) = delete;
    Depth1DComputer_pile& operator=(const Depth1DComputer_pile&) = delete;

    // dc.hpp:513-565: edge confidence, scan, selective median -- on the GPU -- then the
    // result planes are copied into the host members below.
    void run()
    {
        const size_t n = (size_t)dim_v_ * dim_u_;
        m_edge_confidence_v_u.assign(n, 0.f);
        m_edge_confidence_mask_v_u.assign(n, 0);
        m_disp_confidence_v_u.assign(n, 0.f);
        m_best_depth_v_u.assign(n, 0.f);
        m_rbar_v_u.assign(n * CHANNELS, 0.f);
        m_depth_idx_v_u.assign(n, -1);
        m_score_v_u.assign(n, 0.f);
        const rslf_params p = m_parameters.to_c();
        if (multi_) {
            if (is_u8_)
                check(rslf_multi_depth1d_pile_u8(multi_->get(), (const uint8_t* const*)epis_.data(), row_stride_bytes_, dim_v_, dim_s_,
                                                 dim_u_, CHANNELS, m_dmin, m_dmax, m_dim_d, m_s_hat, &p, m_edge_confidence_v_u.data(),
                                                 m_edge_confidence_mask_v_u.data(), m_disp_confidence_v_u.data(),
                                                 m_best_depth_v_u.data(), m_rbar_v_u.data(), m_depth_idx_v_u.data(),
                                                 m_score_v_u.data(), nullptr, &stats),
                      "rslf_multi_depth1d_pile_u8");
            else
                check(rslf_multi_depth1d_pile_f32(multi_->get(), (const float* const*)epis_.data(), row_stride_bytes_, dim_v_, dim_s_,
                                                  dim_u_, CHANNELS, epi_scale_arg_, m_dmin, m_dmax, m_dim_d, m_s_hat, &p,
                                                  m_edge_confidence_v_u.data(), m_edge_confidence_mask_v_u.data(),
                                                  m_disp_confidence_v_u.data(), m_best_depth_v_u.data(), m_rbar_v_u.data(),
                                                  m_depth_idx_v_u.data(), m_score_v_u.data(), nullptr, &stats, &scale_used_),
                      "rslf_multi_depth1d_pile_f32");
            return;
        }
        check(rslf_depth1d_pile_run_host(ctx_->get(), vol_, m_dmin, m_dmax, m_dim_d, m_s_hat, &p, m_edge_confidence_v_u.data(),
                                         m_edge_confidence_mask_v_u.data(), m_disp_confidence_v_u.data(),
                                         m_best_depth_v_u.data(), m_rbar_v_u.data(), m_depth_idx_v_u.data(),
                                         m_score_v_u.data(), nullptr, &stats),
              "rslf_depth1d_pile_run_host");
    }

    int get_s_hat() const { return m_s_hat; }
    int rows() const { return dim_v_; }
    int cols() const { return dim_u_; }
    float epi_scale_factor() const { return scale_used_; }

    // Results, dense row-major [V][U] (what the reference keeps as private Mats, dc.hpp:131-135)
    std::vector<float> m_edge_confidence_v_u;
    std::vector<uint8_t> m_edge_confidence_mask_v_u;
    std::vector<float> m_disp_confidence_v_u;
    std::vector<float> m_best_depth_v_u;
    std::vector<float> m_rbar_v_u;          // [V][U][CHANNELS]
    std::vector<int32_t> m_depth_idx_v_u;   // argmax index, -1 where no disparity was assigned
    std::vector<float> m_score_v_u;
    rslf_stats stats;

private:
    void init(const void* const* epis, bool is_u8, int dim_v, int dim_s, int dim_u, size_t row_stride_bytes, float dmin,
              float dmax, int dim_d, int s_hat, float epi_scale_factor)
    {
        dim_v_ = dim_v;
        dim_s_ = dim_s;
        dim_u_ = dim_u;
        m_dmin = dmin;
        m_dmax = dmax;
        m_dim_d = dim_d;
        // dc.hpp:490-498
        m_s_hat = (s_hat < 0 || s_hat > dim_s - 1) ? (int)std::floor((0.0 + dim_s) / 2) : s_hat;
        stats = rslf_stats();
        check(rslf_volume_create(ctx_->get(), dim_v, dim_s, dim_u, CHANNELS, &vol_), "rslf_volume_create");
        scale_used_ = 255.f;
        if (is_u8)
            check(rslf_volume_upload_epis_u8(vol_, (const uint8_t* const*)epis, row_stride_bytes), "rslf_volume_upload_epis_u8");
        else
            check(rslf_volume_upload_epis_f32(vol_, (const float* const*)epis, row_stride_bytes, epi_scale_factor, &scale_used_),
                  "rslf_volume_upload_epis_f32");
    }
    void init_multi(const void* const* epis, bool is_u8, int dim_v, int dim_s, int dim_u, size_t row_stride_bytes, float dmin,
                    float dmax, int dim_d, int s_hat, float epi_scale_factor)
    {
        dim_v_ = dim_v;
        dim_s_ = dim_s;
        dim_u_ = dim_u;
        m_dmin = dmin;
        m_dmax = dmax;
        m_dim_d = dim_d;
        m_s_hat = (s_hat < 0 || s_hat > dim_s - 1) ? (int)std::floor((0.0 + dim_s) / 2) : s_hat;   // dc.hpp:490-498
        stats = rslf_stats();
        epis_.assign(epis, epis + dim_v);
        is_u8_ = is_u8;
        row_stride_bytes_ = row_stride_bytes;
        epi_scale_arg_ = epi_scale_factor;
        scale_used_ = 255.f;
    }

    Context* ctx_;
    MultiContext* multi_;
    std::vector<const void*> epis_;   // multi path: the caller's buffers, read by run()
    bool is_u8_;
    size_t row_stride_bytes_;
    float epi_scale_arg_;
    rslf_volume* vol_;
    int dim_v_, dim_s_, dim_u_;
    int m_dim_d;
    float m_dmin, m_dmax;
    int m_s_hat;
    float scale_used_;
    const Depth1DParameters m_parameters;
};

// rslf::Depth2DComputer<DataType> (dc.hpp:166-225, :651-805): disparities for every view, centre outwards,
// with propagation along EPI lines.  Results are dense [S][V][U] host vectors after run().
template <int CHANNELS>
class Depth2DComputer {
public:
    Depth2DComputer(Context& ctx, const void* const* epis, bool is_u8, int dim_v, int dim_s, int dim_u, size_t row_stride_bytes,
                    float dmin, float dmax, int dim_d, float epi_scale_factor = -1,
                    const Depth1DParameters& parameters = Depth1DParameters::get_default())
        : ctx_(&ctx), multi_(nullptr), vol_(nullptr), is_u8_(is_u8), stride_(row_stride_bytes), scale_(epi_scale_factor), dim_v_(dim_v),
          dim_s_(dim_s), dim_u_(dim_u), m_dim_d(dim_d), m_dmin(dmin), m_dmax(dmax), m_parameters(parameters)
    {
        stats = rslf_stats();
        check(rslf_volume_create(ctx_->get(), dim_v, dim_s, dim_u, CHANNELS, &vol_), "rslf_volume_create");
        if (is_u8)
            check(rslf_volume_upload_epis_u8(vol_, (const uint8_t* const*)epis, row_stride_bytes), "rslf_volume_upload_epis_u8");
        else
            check(rslf_volume_upload_epis_f32(vol_, (const float* const*)epis, row_stride_bytes, epi_scale_factor, nullptr),
                  "rslf_volume_upload_epis_f32");
    }
    // The same on a MultiContext: the sweep is cut into one block of scanlines per device, the neighbours' boundary rows
    // exchanged by peer copy on every visit (rslf_multi_depth2d_run_*).  The EPIs are read by run(): they must outlive it.
    Depth2DComputer(MultiContext& multi, const void* const* epis, bool is_u8, int dim_v, int dim_s, int dim_u, size_t row_stride_bytes,
                    float dmin, float dmax, int dim_d, float epi_scale_factor = -1,
                    const Depth1DParameters& parameters = Depth1DParameters::get_default())
        : ctx_(nullptr), multi_(&multi), vol_(nullptr), epis_(epis, epis + dim_v), is_u8_(is_u8), stride_(row_stride_bytes),
          scale_(epi_scale_factor), dim_v_(dim_v), dim_s_(dim_s), dim_u_(dim_u), m_dim_d(dim_d), m_dmin(dmin), m_dmax(dmax),
          m_parameters(parameters)
    {
        stats = rslf_stats();
    }
    ~Depth2DComputer() { rslf_volume_destroy(vol_); }
    Depth2DComputer(const Depth2DComputer&) = delete;
    Depth2DComputer& operator=(const Depth2DComputer&) = delete;

    void run()   // dc.hpp:748-805
    {
        const size_t n = (size_t)dim_s_ * dim_v_ * dim_u_;
        m_edge_confidence_s_v_u.assign(n, 0.f);
        m_edge_confidence_mask_s_v_u.assign(n, 0);
        m_disp_confidence_s_v_u.assign(n, 0.f);
        m_best_depth_s_v_u.assign(n, 0.f);
        m_rbar_s_v_u.assign(n * CHANNELS, 0.f);
        const rslf_params p = m_parameters.to_c();
        if (multi_) {
            if (is_u8_)
                check(rslf_multi_depth2d_run_u8(multi_->get(), (const uint8_t* const*)epis_.data(), stride_, dim_v_, dim_s_, dim_u_, CHANNELS,
                                                m_dmin, m_dmax, m_dim_d, &p, m_edge_confidence_s_v_u.data(),
                                                m_edge_confidence_mask_s_v_u.data(), m_disp_confidence_s_v_u.data(),
                                                m_best_depth_s_v_u.data(), m_rbar_s_v_u.data(), nullptr, &stats),
                      "rslf_multi_depth2d_run_u8");
            else
                check(rslf_multi_depth2d_run_f32(multi_->get(), (const float* const*)epis_.data(), stride_, dim_v_, dim_s_, dim_u_, CHANNELS,
                                                 scale_, m_dmin, m_dmax, m_dim_d, &p, m_edge_confidence_s_v_u.data(),
                                                 m_edge_confidence_mask_s_v_u.data(), m_disp_confidence_s_v_u.data(),
                                                 m_best_depth_s_v_u.data(), m_rbar_s_v_u.data(), nullptr, &stats, nullptr),
                      "rslf_multi_depth2d_run_f32");
            return;
        }
        check(rslf_depth2d_run_host(ctx_->get(), vol_, m_dmin, m_dmax, m_dim_d, &p, m_edge_confidence_s_v_u.data(),
                                    m_edge_confidence_mask_s_v_u.data(), m_disp_confidence_s_v_u.data(),
                                    m_best_depth_s_v_u.data(), m_rbar_s_v_u.data(), &stats),
              "rslf_depth2d_run_host");
    }
    const std::vector<float>& get_depths_s_v_u() const { return m_best_depth_s_v_u; }

    std::vector<float> m_edge_confidence_s_v_u;
    std::vector<uint8_t> m_edge_confidence_mask_s_v_u;
    std::vector<float> m_disp_confidence_s_v_u;
    std::vector<float> m_rbar_s_v_u;
    std::vector<float> m_best_depth_s_v_u;
    rslf_stats stats;

private:
    Context* ctx_;
    MultiContext* multi_;
    rslf_volume* vol_;
    std::vector<const void*> epis_;
    bool is_u8_;
    size_t stride_;
    float scale_;
    int dim_v_, dim_s_, dim_u_, m_dim_d;
    float m_dmin, m_dmax;
    const Depth1DParameters m_parameters;
};

// rslf::FineToCoarse<DataType> (include/rslf_fine_to_coarse.hpp:26-81): constructor arguments as in the
// reference; run() builds the pyramid, sweeps every level and fuses; get_results() hands out the fused maps.
template <int CHANNELS>
class FineToCoarse {
public:
    FineToCoarse(Context& ctx, const void* const* epis, bool is_u8, int dim_v, int dim_s, int dim_u, size_t row_stride_bytes,
                 float d_min, float d_max, int dim_d, float epi_scale_factor = -1,
                 const Depth1DParameters& parameters = Depth1DParameters::get_default(), int max_pyr_depth = -1,
                 bool accept_all_last_scale = true)
        : ctx_(&ctx), multi_(nullptr), epis_(epis, epis + dim_v), is_u8_(is_u8), dim_v_(dim_v), dim_s_(dim_s), dim_u_(dim_u),
          stride_(row_stride_bytes), d_min_(d_min), d_max_(d_max), dim_d_(dim_d), scale_(epi_scale_factor),
          m_parameters(parameters), max_pyr_depth_(max_pyr_depth), accept_all_(accept_all_last_scale), n_levels_(0)
    {
        stats = rslf_stats();
    }
    // The same over a MultiContext's devices: every level's sweep sharded by scanline (rslf_multi_fine_to_coarse_run_host).
    FineToCoarse(MultiContext& multi, const void* const* epis, bool is_u8, int dim_v, int dim_s, int dim_u, size_t row_stride_bytes,
                 float d_min, float d_max, int dim_d, float epi_scale_factor = -1,
                 const Depth1DParameters& parameters = Depth1DParameters::get_default(), int max_pyr_depth = -1,
                 bool accept_all_last_scale = true)
        : ctx_(nullptr), multi_(&multi), epis_(epis, epis + dim_v), is_u8_(is_u8), dim_v_(dim_v), dim_s_(dim_s), dim_u_(dim_u),
          stride_(row_stride_bytes), d_min_(d_min), d_max_(d_max), dim_d_(dim_d), scale_(epi_scale_factor),
          m_parameters(parameters), max_pyr_depth_(max_pyr_depth), accept_all_(accept_all_last_scale), n_levels_(0)
    {
        stats = rslf_stats();
    }
    void run()
    {
        const size_t n = (size_t)dim_s_ * dim_v_ * dim_u_;
        out_map_s_v_u_.assign(n, 0.f);
        out_validity_s_v_u_.assign(n, 0);
        const rslf_params p = m_parameters.to_c();
        if (multi_) {
            check(rslf_multi_fine_to_coarse_run_host(multi_->get(), epis_.data(), is_u8_ ? 1 : 0, dim_v_, dim_s_, dim_u_, CHANNELS, stride_,
                                                     d_min_, d_max_, dim_d_, scale_, &p, max_pyr_depth_, accept_all_ ? 1 : 0,
                                                     out_map_s_v_u_.data(), out_validity_s_v_u_.data(), &n_levels_, &stats),
                  "rslf_multi_fine_to_coarse_run_host");
            return;
        }
        check(rslf_fine_to_coarse_run_host(ctx_->get(), epis_.data(), is_u8_ ? 1 : 0, dim_v_, dim_s_, dim_u_, CHANNELS, stride_,
                                           d_min_, d_max_, dim_d_, scale_, &p, max_pyr_depth_, accept_all_ ? 1 : 0,
                                           out_map_s_v_u_.data(), out_validity_s_v_u_.data(), &n_levels_, &stats),
              "rslf_fine_to_coarse_run_host");
    }
    void get_results(std::vector<float>& out_map_s_v_u, std::vector<uint8_t>& out_validity_s_v_u) const
    {
        out_map_s_v_u = out_map_s_v_u_;
        out_validity_s_v_u = out_validity_s_v_u_;
    }
    int pyramid_depth() const { return n_levels_; }
    rslf_stats stats;

private:
    Context* ctx_;
    MultiContext* multi_;
    std::vector<const void*> epis_;
    bool is_u8_;
    int dim_v_, dim_s_, dim_u_;
    size_t stride_;
    float d_min_, d_max_;
    int dim_d_;
    float scale_;
    const Depth1DParameters m_parameters;
    int max_pyr_depth_;
    bool accept_all_;
    int n_levels_;
    std::vector<float> out_map_s_v_u_;
    std::vector<uint8_t> out_validity_s_v_u_;
};

typedef Depth1DComputer_pile<1> Depth1DComputer_pile_1ch;   // dc.hpp:149
typedef Depth1DComputer_pile<3> Depth1DComputer_pile_3ch;   // dc.hpp:154

}  // namespace rslfx

#endif  // RSLF_HIP_HPP
