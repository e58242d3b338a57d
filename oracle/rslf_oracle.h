/*
 * rslf_oracle.h -- CPU restatement of the RSLightFields 1-D EPI depth scan.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the checker the HIP path is compared
 * against; nothing under remotesensingproject_amd/ or include/ may call it.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * PARITY UNPINNED: the reference (14chanwa/remotesensingProject) holds no
 * golden vectors, known-answer tests or assertions for this path, and it
 * cannot be built here (it needs OpenCV 3.x, which this image lacks).  The
 * oracle is therefore a restatement of the reference's source, op for op,
 * pinned only by (a) an analytic known-answer case, (b) an independent numpy
 * restatement (oracle/oracle_np.py) agreeing bit for bit, and (c) the c1 anchor
 * of SURVEY.md 8c.  OpenCV 3.x primitive semantics it relies on are listed in
 * DESIGN.md.
 *
 * Layout: the light-field volume is [V][S][U][C] float32, C interleaved --
 * the reference's Vec<Mat> of V EPIs, each an S x U Mat of C channels
 * (rslf_io.cpp:194-227), stored back to back.
 *
 * Reference files followed (paths relative to RSLightFields/):
 *   include/rslf_depth_computation_core.hpp:426-478  edge confidence
 *   include/rslf_depth_computation_core.hpp:480-661  per-EPI scan
 *   include/rslf_depth_computation_core.hpp:663-718  selective median
 *   include/rslf_depth_computation_core.hpp:728-893  pile drivers
 *   include/rslf_interpolation.hpp:155-193           linear gather
 *   src/rslf_kernels.cpp:16-26,39-54                 bandwidth kernel
 *   src/rslf_depth_computation_core.cpp:6-51         channel helpers
 *   src/rslf_types.cpp:80-91                         norm<>
 *   include/rslf_depth_computation.hpp:425-565       ctor normalisation + run()
 */
#ifndef RSLF_ORACLE_H
#define RSLF_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Mirrors rslf::Depth1DParameters (rslf_depth_computation_core.hpp:66-142).
 * Field names follow the reference's par_* members. */
typedef struct oracle_params {
    float edge_score_threshold;      /* _EDGE_SCORE_THRESHOLD 0.02      core.hpp:20  */
    float raw_score_threshold;       /* _RAW_SCORE_THRESHOLD 0          core.hpp:23  */
    float mean_shift_max_iter;       /* declared float in the reference core.hpp:115 */
    int   edge_confidence_filter_size;/* 9                              core.hpp:17  */
    int   median_filter_size;        /* 5                               core.hpp:18  */
    float median_filter_epsilon;     /* 0.1                             core.hpp:19  */
    float slope_factor;              /* 1.0                             core.hpp:95  */
    int   cut_shadows;               /* true                            core.hpp:97  */
    float shadow_level;              /* 0.05*sqrt(3)                    core.hpp:31  */
    float kernel_bandwidth;          /* h = 0.2                         core.hpp:26  */
    int   interpolation;             /* par_interpolation_class (core.hpp:76-77, :108): ORACLE_INTERP_* */
    int   edge_confidence_opening_type; /* cv::MORPH_RECT 0 / MORPH_CROSS 1 / MORPH_ELLIPSE 2 (default) core.hpp:28 */
    int   edge_confidence_opening_size; /* 1 = no opening (default)            core.hpp:29, :759     */
    int   use_disp_confidence_score;    /* the reference's commented-out build switch _USE_DISP_CONFIDENCE_SCORE (core.hpp:35):
                                           the 2-D sweep's propagation is gated by C_d > disp_score_threshold (core.hpp:1097-1098)
                                           instead of the edge mask (:1102).  0 = default build */
    float disp_score_threshold;         /* _DISP_SCORE_THRESHOLD 0.01                core.hpp:22       */
} oracle_params;

/* cv::getStructuringElement(shape, Size(k, k)) with the default anchor (k/2, k/2), as OpenCV 3.x builds it:
 * element[i][j] != 0 for j in [j1, j2) of row i.  out: k*k bytes (0/1). */
void oracle_structuring_element(int shape, int k, uint8_t* out);
/* cv::morphologyEx(mask, mask, MORPH_OPEN, element): erosion then dilation, BORDER_CONSTANT with
 * morphologyDefaultBorderValue (pixels outside the image never win the min / the max).  In place, V x U. */
void oracle_morph_open(uint8_t* mask_vu, int V, int U, int shape, int k);

/* par_interpolation_class.  LINEAR = Interpolation1DLinear (interp.hpp:155-193), the default.
 * NEAREST = Interpolation1DNearestNeighbour as its scalar interpolate() states it (interp.hpp:80-92):
 * index (int)std::round(x), NaN outside [0, U-1].  NEAREST_AS_BUILT = what its interpolate_mat(), the
 * method the scan calls (core.hpp:561), really executes: it reads the FLOAT index matrix through
 * `indices.ptr<int>` (interp.hpp:118), so the "index" is the bit pattern of x -- in range only for
 * x = +0 (and denormals). */
#define ORACLE_INTERP_LINEAR            0
#define ORACLE_INTERP_NEAREST           1
#define ORACLE_INTERP_NEAREST_AS_BUILT  2

void oracle_default_params(oracle_params* p);

/* dc.hpp:442-477.  u8: x*float(1/255).  f32: x*float(1/double(scale)); when
 * scale<0 it is first replaced by the max over the whole volume. Returns the
 * scale used. */
void  oracle_normalize_u8(const uint8_t* in, float* out, size_t n);
float oracle_normalize_f32(const float* in, float* out, size_t n, float scale);

/* core.hpp:426-478 for one row.  Ce_u accumulates INTO the caller's buffer
 * (core.cpp:10) -- pass zeros.  row = U*C interleaved floats. */
void oracle_edge_confidence_row(const float* row, int U, int C,
                                float* Ce_u, uint8_t* mask_u,
                                const oracle_params* p);

/* core.hpp:728-770 (opening size 1 => no morphology). vol = [V][S][U][C]. */
void oracle_edge_confidence_pile(const float* vol, int V, int S, int U, int C,
                                 int s, float* Ce_vu, uint8_t* mask_vu,
                                 const oracle_params* p);

/* core.hpp:480-661 for one EPI.
 * epi [S][U][C]; dmin_u/dmax_u [U]; Ce_u, Ce_mask_u in/out;
 * mask_u nullable (then the edge mask is the scan mask, core.hpp:510-513);
 * when non-null it is AND-ed in place with the edge mask (core.hpp:511).
 * Extra outputs (nullable), not in the reference, used by parity tests:
 *   idx_u       argmax index d*, -1 where not scanned or rejected
 *   score_u     score[d*]
 *   K_su [S][U] K(r - rbar) column of d* (core.hpp:647-651), nullable */
void oracle_depth_epi(const float* epi, int S, int U, int C,
                      const float* dmin_u, const float* dmax_u,
                      int dim_d, int s_hat,
                      float* Ce_u, uint8_t* Ce_mask_u,
                      float* Cd_u, float* depth_u, float* rbar_u,
                      const oracle_params* p, uint8_t* mask_u,
                      int32_t* idx_u, float* score_u, float* K_su);

/* core.hpp:663-718. dst must be zero-filled by the caller (core.hpp:678-679). */
void oracle_selective_median(const float* src_vu, float* dst_vu,
                             const float* vol, int V, int S, int U, int C,
                             int s_hat, int size, const uint8_t* mask_vu,
                             float epsilon);

/* core.hpp:772-893: scan every EPI (OpenMP over v), then the selective median
 * replaces depth_vu (core.hpp:892).  depth_raw_vu (nullable) receives the
 * pre-median plane.  mask_vu nullable as above.  */
void oracle_depth_epi_pile(const float* vol, int V, int S, int U, int C,
                           const float* dmin_vu, const float* dmax_vu,
                           int dim_d, int s_hat,
                           float* Ce_vu, uint8_t* Ce_mask_vu,
                           float* Cd_vu, float* depth_vu, float* rbar_vu,
                           const oracle_params* p, uint8_t* mask_vu,
                           int32_t* idx_vu, float* score_vu,
                           float* depth_raw_vu);

/* Depth1DComputer_pile ctor + run() (dc.hpp:425-565) on an already
 * normalised volume: zero-inits outputs (fixing the reference's
 * uninitialised C_e, dc.hpp:501), constant dmin/dmax planes, s_hat<0 =>
 * floor(S/2), edge confidence, scan, median. */
void oracle_depth1d_pile_run(const float* vol, int V, int S, int U, int C,
                             float dmin, float dmax, int dim_d, int s_hat,
                             const oracle_params* p,
                             float* Ce_vu, uint8_t* Ce_mask_vu,
                             float* Cd_vu, float* depth_vu, float* rbar_vu,
                             int32_t* idx_vu, float* score_vu,
                             float* depth_raw_vu);

/* ---- "next" row: the 2-D sweep -------------------- */

/* compute_2D_edge_confidence (core.hpp:901-931): the pile edge confidence for
 * every view s.  Ce_svu [S][V][U] accumulates (pass zeros), mask_svu [S][V][U]. */
void oracle_edge_confidence_2d(const float* vol, int V, int S, int U, int C,
                               float* Ce_svu, uint8_t* mask_svu, const oracle_params* p);

/* compute_2D_depth_epi (core.hpp:933-1133), default build (neither
 * _USE_DISP_CONFIDENCE_SCORE nor _USE_LINE_CONFIDENCE_SCORE: propagation is gated
 * by the edge mask, core.hpp:1099-1103).  Views are visited s_hat, s_hat+1,
 * s_hat-1, ... (core.hpp:981-990); each visit runs the pile scan with the running
 * mask (core.hpp:1012-1028) and then paints its disparities along their EPI lines
 * into the other views (core.hpp:1088-1129).
 *   dmin_svu/dmax_svu [S][V][U]; Ce_svu, Ce_mask_svu in/out; Cd_svu, depth_svu,
 *   rbar_svu ([S][V][U][C]) in/out (pass zeros); scan_mask_svu (nullable) receives
 *   the final running masks.
 * Note core.hpp:892 rebinds only the LOCAL header best_depth_v_u to the median
 * result: the stored plane of the visited view keeps the raw arg-max depths and
 * then receives the median-filtered values the propagation paints into it. */
/* pixels scanned by the sweeps (oracle_depth_epi_2d, oracle_depth2d_run) since the last reset; reset != 0 clears the count */
long long oracle_sweep_pixels_scanned(int reset);

void oracle_depth_epi_2d(const float* vol, int V, int S, int U, int C,
                         const float* dmin_svu, const float* dmax_svu, int dim_d,
                         float* Ce_svu, uint8_t* Ce_mask_svu, float* Cd_svu,
                         float* depth_svu, float* rbar_svu, const oracle_params* p,
                         float propagation_epsilon, uint8_t* scan_mask_svu);

/* Depth2DComputer ctor + run() (dc.hpp:651-805) on a normalised volume. */
void oracle_depth2d_run(const float* vol, int V, int S, int U, int C,
                        float dmin, float dmax, int dim_d, const oracle_params* p,
                        float propagation_epsilon,
                        float* Ce_svu, uint8_t* Ce_mask_svu, float* Cd_svu,
                        float* depth_svu, float* rbar_svu, uint8_t* scan_mask_svu);

/* ---- "next" row: fine-to-coarse -------------------- */
/* The OpenCV 3.x primitives this row leans on are restated from their documented algorithms; the
 * choices that affect the last bit (accumulation order) are listed in DESIGN.md.  All images here are
 * dense float32, [rows][cols][C] interleaved. */

/* rslf::downsample_EPIs (src/rslf_fine_to_coarse_core.cpp:14-60): per view s, the V x U image is
 * smoothed with cv::GaussianBlur(7x7, sigma 0 => the fixed table {1,3.5,7,9,7,3.5,1}/32, BORDER_REFLECT)
 * and halved with cv::resize(0.5, 0.5, INTER_LINEAR) (which OpenCV runs as its 2x2 "area fast" average).
 * in [V][S][U][C] -> out [V2][S][U2][C]; V2 = cvRound(V/2), U2 = cvRound(U/2) (round half to even). */
void oracle_f2c_out_dims(int V, int U, int* V2, int* U2);
void oracle_downsample_epis(const float* in_vsuc, int V, int S, int U, int C, float* out_vsuc);
/* the same on CV_8U Mats: uchar levels (0..255) in float arrays, uchar arithmetic (fine_to_coarse_core.cpp:22-41) */
void oracle_downsample_epis_u8(const float* in, int V, int S, int U, int C, float* out);

/* FineToCoarse::run(), the bound tightening between two levels (include/rslf_fine_to_coarse.hpp:171-299):
 * for every pixel of the coarser level, the nearest valid disparities left and right of column 2u on
 * rows 2v and 2v+1 of the finer level bound its hypothesis range.  Planes [S][V][U]. */
void oracle_f2c_tighten_bounds(const float* depth_up_svu, const uint8_t* mask_up_svu, int S, int V_up, int U_up,
                               float* dmin_down_svu, float* dmax_down_svu, int V_down, int U_down);

/* rslf::fuse_disp_maps (src/rslf_fine_to_coarse_core.cpp:69-135) for ONE view: coarse-to-fine fill of the
 * invalid pixels (cv::resize INTER_LINEAR / INTER_NEAREST upscaling), then cv::medianBlur 3x3.
 * disp[p] / valid[p]: level p planes [V_p][U_p], p = 0 finest.  out_map / out_valid: [V_0][U_0]. */
void oracle_f2c_fuse(const float* const* disp, const uint8_t* const* valid, const int* Vp, const int* Up, int P,
                     float* out_map, uint8_t* out_valid);

/* What-if switches (tools/blast_radius.py): alternatives to the OpenCV 3.x readings of SURVEY.md App. B. */
#define ORACLE_ASSUME_RGB_SUM_IN_ORDER 1   /* cv::reduce over 3 columns as (q0 + q1) + q2, not (q0 + q2) + q1 */
#define ORACLE_ASSUME_MUL_SCALE_LAST   2   /* cv::multiply(a, b, scale) as scale * (a * b), not (scale * a) * b */
#define ORACLE_ASSUME_DIV0_IEEE        4   /* cv::divide by 0 as IEEE inf / NaN (OpenCV 4.x), not 0 (3.x) */
#define ORACLE_ASSUME_MAX_NAN_TAIL     8   /* cv::max(x, 0) keeps a NaN in the scalar tail (last n % 8 elements) */
/* ... and of the fine-to-coarse pyramid (rslf_fine_to_coarse_core.cpp:22-41, :69-135): */
#define ORACLE_ASSUME_GAUSS_ROW_SYMM   16  /* GaussianBlur's ROW filter in the symmetric form k[c]*x[c] + k[c+j]*(x[c+j] + x[c-j]) (as its
                                              column filter), not taps accumulated left to right */
#define ORACLE_ASSUME_GAUSS_COL_ORDER  32  /* ... and its COLUMN filter with the taps accumulated top to bottom, not the symmetric form */
#define ORACLE_ASSUME_AREA_SCALAR      64  /* the 2x2 area mean as ((S00 + S01) + S10) + S11 (resizeAreaFast's scalar tail, which the last
                                              W2 % 4 columns of a SIMD build take) for every pixel, not (S00 + S10) + (S01 + S11) */
#define ORACLE_ASSUME_SIZE_FLOOR       128 /* level sizes floor(n * 0.5) (saturate_cast of a truncating build), not cvRound */
#define ORACLE_ASSUME_RESIZE_FLOAT     256 /* the upscaling's source coordinates computed in float, not in double and then cast */
void oracle_set_assumptions(int flags);
int oracle_assumptions(void);
int oracle_num_threads(void);
void oracle_set_num_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
