// Instantiates every cv::Mat-facing member of include/rslf_hip.hpp (the literal "cv::Mat in / cv::Mat out" signature of
// the reference, dc.hpp:97-122) against the declaration-only mock in tests/cpp/opencv_mock: compiled with -fsyntax-only.
#include "rslf_hip.hpp"
#ifndef RSLFX_HAVE_OPENCV
#error "the mock <opencv2/core/core.hpp> is not on the include path"
#endif

cv::Mat use_pile(rslfx::Context& ctx, rslfx::MultiContext& multi, const std::vector<cv::Mat>& epis)
{
    rslfx::Depth1DComputer_pile<1> a(ctx, epis, -1.f, 2.f, 16);
    a.run();
    rslfx::Depth1DComputer_pile<3> b(multi, epis, -1.f, 2.f, 16, -1, -1.f, rslfx::Depth1DParameters::get_default());
    b.run();
    cv::Mat m = a.get_edge_confidence();
    m = a.get_edge_confidence_mask();
    m = a.get_disp_confidence();
    m = b.get_rbar();
    return a.get_best_depth();
}
