/*
 * rslf_oracle.c -- CPU restatement of the RSLightFields 1-D EPI depth scan.
 *
 * TEST INFRASTRUCTURE ONLY (see rslf_oracle.h).  PARITY UNPINNED: the
 * reference ships no golden vectors and cannot be built in this image.
 *
 * Every float operation below is one IEEE-754 binary32 operation, in the order
 * the reference performs it; build with -ffp-contract=off -fno-fast-math and
 * WITHOUT an FMA-enabling -march (see oracle/Makefile).  Sums over s are
 * sequential, ascending s, as cv::reduce(...,0,REDUCE_SUM) does
 * (rslf_depth_computation_core.hpp:602-603).  Loops run d-innermost so the
 * compiler can vectorise across hypotheses without re-associating anything.
 */
#include "rslf_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define SQRT3_D 1.73205080757 /* literal used by the reference: types.cpp:84, core.hpp:31 */

void oracle_default_params(oracle_params* p)
{
    /* rslf_depth_computation_core.hpp:74-99 with the #defines of :16-31 */
    p->edge_score_threshold = (float)0.02;
    p->raw_score_threshold = (float)0;
    p->mean_shift_max_iter = (float)10;
    p->edge_confidence_filter_size = 9;
    p->median_filter_size = 5;
    p->median_filter_epsilon = (float)0.1;
    p->slope_factor = (float)1.0;
    p->cut_shadows = 1;
    p->shadow_level = (float)(0.05 * SQRT3_D);
    p->kernel_bandwidth = (float)0.2;
    p->interpolation = ORACLE_INTERP_LINEAR;   /* core.hpp:76 */
    p->edge_confidence_opening_type = 2;       /* cv::MORPH_ELLIPSE, core.hpp:28 */
    p->edge_confidence_opening_size = 1;       /* core.hpp:29 */
    p->use_disp_confidence_score = 0;          /* core.hpp:35: the #define is commented out */
    p->disp_score_threshold = (float)0.01;     /* core.hpp:22 */
}

/* Interpolation1DNearestNeighbour::interpolate_mat (interp.hpp:94-131): the sample index for position x,
 * or -1 when the reference stores NaN. */
static int nearest_index(float x, int U, int mode)
{
    int r;
    if (mode == ORACLE_INTERP_NEAREST_AS_BUILT) {
        memcpy(&r, &x, sizeof r);               /* interp.hpp:118: const int* ind_ptr = indices.ptr<int>(r) */
    } else {
        if (!(fabsf(x) < 2.0e9f))               /* (int) of an out-of-range float is undefined: never a valid index */
            return -1;
        r = (int)roundf(x);                     /* interp.hpp:121 / :86: std::round, halves away from zero */
    }
    return (r > -1 && r < U) ? r : -1;          /* interp.hpp:122 */
}

/* What-if switches for the OpenCV 3.x semantics the restatement assumes and cannot verify here:
 * 0 = the assumptions of record.  tools/blast_radius.py counts what each alternative would change; the product and
 * every parity test run with 0. */
static int g_assume = 0;
void oracle_set_assumptions(int flags) { g_assume = flags; }
int oracle_assumptions(void) { return g_assume; }

void oracle_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0)
        omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ---- dc.hpp:442-477 --------------------------------------------------- */

void oracle_normalize_u8(const uint8_t* in, float* out, size_t n)
{
    /* epi.convertTo(epi2, CV_32F, 1.0/255.0): float scale, dc.hpp:470 */
    const float sc = (float)(1.0 / 255.0);
    for (size_t i = 0; i < n; i++)
        out[i] = (float)in[i] * sc;
}

float oracle_normalize_f32(const float* in, float* out, size_t n, float scale)
{
    if (scale < 0) {
        /* dc.hpp:442-460: epi_scale_factor = max over all EPIs */
        for (size_t i = 0; i < n; i++)
            if (in[i] > scale)
                scale = in[i];
    }
    /* convertTo(..., 1.0/epi_scale_factor): double quotient, cast to float
     * inside cvtScale, dc.hpp:474 */
    const float sc = (float)(1.0 / (double)scale);
    for (size_t i = 0; i < n; i++)
        out[i] = in[i] * sc;
    return scale;
}

/* ---- types.cpp:80-91 -------------------------------------------------- */

static inline float norm_px(const float* x, int C)
{
    if (C == 1) {
        /* std::abs(x) * 1.73205080757 : float*double, returned as float */
        return (float)((double)fabsf(x[0]) * SQRT3_D);
    }
    /* cv::norm(Vec3f): sqrt of a double sum of squares, returned as float */
    double s = 0.0;
    for (int c = 0; c < C; c++)
        s += (double)x[c] * (double)x[c];
    return (float)sqrt(s);
}

/* cv::BORDER_REFLECT_101 (core.hpp:458) */
static inline int reflect101(int p, int len)
{
    if (len == 1)
        return 0;
    while (p < 0 || p >= len) {
        if (p < 0)
            p = -p;
        else
            p = 2 * len - 2 - p;
    }
    return p;
}

/* ---- core.hpp:426-478 ------------------------------------------------- */

void oracle_edge_confidence_row(const float* row, int U, int C,
                                float* Ce_u, uint8_t* mask_u,
                                const oracle_params* p)
{
    const int fs = p->edge_confidence_filter_size;
    const int center = (fs - 1) / 2; /* core.hpp:440 */

    for (int j = 0; j < fs; j++) { /* core.hpp:449 */
        if (j == center)
            continue;
        for (int u = 0; u < U; u++) {
            /* filter2D with +1 at centre, -1 at j  => E[u] - E[u+j-centre] */
            const int q = reflect101(u + j - center, U);
            /* _square_sum_channels_into: per channel, dst += t*t (core.cpp:6-23) */
            for (int c = 0; c < C; c++) {
                const float t = row[(size_t)u * C + c] - row[(size_t)q * C + c];
                const float t2 = t * t;
                Ce_u[u] = Ce_u[u] + t2;
            }
        }
    }

    if (p->cut_shadows) { /* core.hpp:464-474 */
        for (int u = 0; u < U; u++) {
            const float n = norm_px(row + (size_t)u * C, C);
            if (n < p->shadow_level)
                Ce_u[u] = 0.0f;
        }
    }

    for (int u = 0; u < U; u++) /* core.hpp:476 */
        mask_u[u] = (Ce_u[u] > p->edge_score_threshold) ? 255 : 0;
}

/* ---- core.hpp:728-770 ------------------------------------------------- */

/* cv::getStructuringElement (OpenCV 3.x imgproc/src/morph.cpp): r = k/2, c = k/2, anchor = (k/2, k/2);
 * RECT: every column; CROSS: the anchor row entirely, else the anchor column; ELLIPSE: columns
 * [c - dx, c + dx + 1) with dx = cvRound(c * sqrt((r*r - dy*dy) / (r*r))), dy = i - r, rows with |dy| <= r. */
void oracle_structuring_element(int shape, int k, uint8_t* out)
{
    const int r = k / 2, c = k / 2;
    const double inv_r2 = r ? 1.0 / ((double)r * r) : 0.0;
    memset(out, 0, (size_t)k * k);
    for (int i = 0; i < k; i++) {
        int j1 = 0, j2 = 0;
        if (shape == 0 || (shape == 1 && i == k / 2)) {
            j2 = k;
        } else if (shape == 1) {
            j1 = k / 2;
            j2 = j1 + 1;
        } else {
            const int dy = i - r;
            if (abs(dy) <= r) {
                const int dx = (int)lrint(c * sqrt((r * r - dy * dy) * inv_r2));   /* saturate_cast<int>(double) = cvRound */
                j1 = c - dx > 0 ? c - dx : 0;
                j2 = c + dx + 1 < k ? c + dx + 1 : k;
            }
        }
        for (int j = j1; j < j2; j++)
            out[(size_t)i * k + j] = 1;
    }
}

static void morph_pass(const uint8_t* src, uint8_t* dst, int V, int U, const uint8_t* el, int k, int dilate)
{
    const int a = k / 2;   /* anchor */
#pragma omp parallel for schedule(static)
    for (int v = 0; v < V; v++)
        for (int u = 0; u < U; u++) {
            int acc = dilate ? 0 : 255;
            for (int i = 0; i < k; i++)
                for (int j = 0; j < k; j++) {
                    if (!el[(size_t)i * k + j])
                        continue;
                    const int y = v + i - a, x = u + j - a;
                    if (y < 0 || y >= V || x < 0 || x >= U)
                        continue;   /* border value: +max for erosion, min for dilation -- never wins */
                    const int val = src[(size_t)y * U + x];
                    acc = dilate ? (val > acc ? val : acc) : (val < acc ? val : acc);
                }
            dst[(size_t)v * U + u] = (uint8_t)acc;
        }
}

void oracle_morph_open(uint8_t* mask_vu, int V, int U, int shape, int k)
{
    uint8_t* el = (uint8_t*)malloc((size_t)k * k);
    uint8_t* tmp = (uint8_t*)malloc((size_t)V * U);
    oracle_structuring_element(shape, k, el);
    morph_pass(mask_vu, tmp, V, U, el, k, 0);   /* erode  */
    morph_pass(tmp, mask_vu, V, U, el, k, 1);   /* dilate */
    free(tmp);
    free(el);
}

void oracle_edge_confidence_pile(const float* vol, int V, int S, int U, int C,
                                 int s, float* Ce_vu, uint8_t* mask_vu,
                                 const oracle_params* p)
{
#pragma omp parallel for schedule(static)
    for (int v = 0; v < V; v++) { /* core.hpp:743 */
        const float* row = vol + (((size_t)v * S + s) * U) * C;
        oracle_edge_confidence_row(row, U, C, Ce_vu + (size_t)v * U,
                                   mask_vu + (size_t)v * U, p);
    }
    if (p->edge_confidence_opening_size > 1)   /* core.hpp:759-768 */
        oracle_morph_open(mask_vu, V, U, p->edge_confidence_opening_type, p->edge_confidence_opening_size);
}

/* ---- core.hpp:480-661 ------------------------------------------------- */

typedef struct scan_scratch {
    float* R;    /* [C][S][DBLOCK] radiances of the current hypothesis block, NaN where out of range */
    float* R0;   /* [C][S][DBLOCK] NaN -> 0 (core.hpp:580) */
    float* Kmat; /* [S][D] last kernel values */
    float* Dv;   /* [D] hypothesis values */
    float* card; /* [D] */
    float* rbar; /* [C][D] */
    float* A;    /* [C][D] */
    float* B;    /* [D] */
    float* score;/* [D] */
    float* x;    /* [D] temp */
} scan_scratch;

static int scratch_alloc(scan_scratch* w, int S, int D, int C)
{
    size_t sd = (size_t)S * D;
    size_t sb = (size_t)S * 32; /* ORACLE_DBLOCK */
    w->R = (float*)malloc(sizeof(float) * sb * C);
    w->R0 = (float*)malloc(sizeof(float) * sb * C);
    w->Kmat = (float*)malloc(sizeof(float) * sd);
    w->Dv = (float*)malloc(sizeof(float) * D);
    w->card = (float*)malloc(sizeof(float) * D);
    w->rbar = (float*)malloc(sizeof(float) * D * C);
    w->A = (float*)malloc(sizeof(float) * D * C);
    w->B = (float*)malloc(sizeof(float) * D);
    w->score = (float*)malloc(sizeof(float) * D);
    w->x = (float*)malloc(sizeof(float) * D);
    return w->R && w->R0 && w->Kmat && w->Dv && w->card && w->rbar && w->A && w->B && w->score && w->x;
}

static void scratch_free(scan_scratch* w)
{
    free(w->R); free(w->R0); free(w->Kmat); free(w->Dv); free(w->card);
    free(w->rbar); free(w->A); free(w->B); free(w->score); free(w->x);
}

/* BandwidthKernel::evaluate_mat (kernels.cpp:16-26 / 39-54) over one s-row of
 * hypotheses; delta_c = R_c - rbar_c is formed here (core.hpp:591). */
static inline void kernel_row(const float* const* Rrow, const float* const* rbar,
                              int D, int C, float inv_h2, float k1, float* K)
{
    if (C == 1) {
        const float* r = Rrow[0];
        const float* rb = rbar[0];
        for (int d = 0; d < D; d++) {
            const float delta = r[d] - rb[d];
            /* cv::multiply(src, src, dst, 3*inv_m_h_sq): (scale*a)*b */
            const float t = k1 * delta;
            const float q = t * delta;
            /* cv::subtract(1.0, dst, dst) */
            const float o = 1.0f - q;
            /* cv::max(dst, 0.0, dst), NaN -> 0 */
            K[d] = (o > 0.0f) ? o : 0.0f;
        }
    } else {
        for (int d = 0; d < D; d++) {
            float q[3];
            for (int c = 0; c < 3; c++) {
                const float delta = Rrow[c][d] - rbar[c][d];
                const float t = inv_h2 * delta;
                q[c] = t * delta;
            }
            /* cv::reduce(dst, dst, 1, REDUCE_SUM) over 3 columns:
             * OpenCV 3.x reduceC_ keeps two accumulators: (q0 + q2) + q1 */
            float a0 = q[0] + q[2];
            a0 = a0 + q[1];
            const float o = 1.0f - a0;
            K[d] = (o > 0.0f) ? o : 0.0f;
        }
    }
}

/* Hypotheses are independent of each other, so they are scored in blocks of
 * ORACLE_DBLOCK to keep the S x block temporaries cache resident (the reference
 * streams full S x D matrices, core.hpp:584-610; the values are the same). */
#define ORACLE_DBLOCK 32

static void scan_pixel(const float* epi, int S, int U, int C, int u,
                       float dmin, float dmax, int D, int s_hat,
                       const oracle_params* p, scan_scratch* w,
                       float inv_h2, float k1, int n_iter)
{
    /* core.hpp:545-548: D[d] = dmin + d * (dmax - dmin) / (dim_d - 1) */
    const float range = dmax - dmin;
    const float denom = (float)(D - 1);
    for (int d = 0; d < D; d++) {
        const float num = (float)d * range;
        const float quo = num / denom;
        w->Dv[d] = dmin + quo;
    }
    const float uf = (float)u;

    for (int c0 = 0; c0 < D; c0 += ORACLE_DBLOCK) {
        const int nb = (D - c0 < ORACLE_DBLOCK) ? (D - c0) : ORACLE_DBLOCK;
        const size_t SD = (size_t)S * ORACLE_DBLOCK;   /* channel stride inside the block */
        const float* Dv = w->Dv + c0;
        float* card = w->card + c0;
        memset(card, 0, sizeof(float) * nb);

        /* core.hpp:550-552 + interp.hpp:155-193 */
        for (int s = 0; s < S; s++) {
            const float Ss = (float)(s_hat - s); /* core.hpp:542 */
            const float* erow = epi + (size_t)s * U * C;
            if (p->interpolation != ORACLE_INTERP_LINEAR) {
                for (int d = 0; d < nb; d++) {
                    float xi = Ss * Dv[d];          /* I = S * D (gemm, K=1)    */
                    xi = xi * p->slope_factor;      /* I *= par_slope_factor    */
                    xi = xi + uf;                   /* I += u                   */
                    const int r = nearest_index(xi, U, p->interpolation);
                    for (int c = 0; c < C; c++) {
                        const float val = (r >= 0) ? erow[(size_t)r * C + c] : NAN;   /* interp.hpp:124 / :129 */
                        w->R[c * SD + (size_t)s * ORACLE_DBLOCK + d] = val;
                        w->R0[c * SD + (size_t)s * ORACLE_DBLOCK + d] = (r >= 0 && val > 0.0f) ? val : 0.0f; /* core.hpp:580 */
                    }
                    if (r >= 0)
                        card[d] = card[d] + 1.0f;   /* interp.hpp:125 */
                }
                continue;
            }
            if (C == 1) {
                /* branch-free so the compiler can vectorise it (gathers); same values */
                float* restrict Rr = w->R + (size_t)s * ORACLE_DBLOCK;
                float* restrict R0r = w->R0 + (size_t)s * ORACLE_DBLOCK;
                const int Um1 = U - 1;
                const float slope = p->slope_factor;
                for (int d = 0; d < nb; d++) {
                    float xi = Ss * Dv[d];          /* I = S * D (gemm, K=1)    */
                    xi = xi * slope;                /* I *= par_slope_factor    */
                    xi = xi + uf;                   /* I += u                   */
                    const float fl = floorf(xi);
                    const float ce = ceilf(xi);
                    const int i0 = (int)fl;         /* interp.hpp:179           */
                    const int i1 = (int)ce;         /* interp.hpp:180           */
                    const float t = xi - (float)i0; /* interp.hpp:181           */
                    const int valid = (i0 >= 0) & (i1 <= Um1); /* interp.hpp:182 */
                    const int j0 = i0 < 0 ? 0 : (i0 > Um1 ? Um1 : i0);
                    const int j1 = i1 < 0 ? 0 : (i1 > Um1 ? Um1 : i1);
                    const float omt = 1.0f - t;
                    const float a = omt * erow[j0];
                    const float b = t * erow[j1];
                    const float r = a + b;          /* interp.hpp:184           */
                    Rr[d] = valid ? r : NAN;        /* interp.hpp:189           */
                    R0r[d] = (valid && r > 0.0f) ? r : 0.0f; /* core.hpp:580    */
                    card[d] = card[d] + (valid ? 1.0f : 0.0f);
                }
                continue;
            }
            for (int d = 0; d < nb; d++) {
                float xi = Ss * Dv[d];          /* I = S * D (gemm, K=1)    */
                xi = xi * p->slope_factor;      /* I *= par_slope_factor    */
                xi = xi + uf;                   /* I += u                   */
                const int i0 = (int)floorf(xi); /* interp.hpp:179           */
                const int i1 = (int)ceilf(xi);  /* interp.hpp:180           */
                const float t = xi - (float)i0; /* interp.hpp:181           */
                if (!(i0 < 0 || i1 > U - 1)) {  /* interp.hpp:182           */
                    const float omt = 1.0f - t;
                    for (int c = 0; c < C; c++) {
                        const float a = omt * erow[(size_t)i0 * C + c];
                        const float b = t * erow[(size_t)i1 * C + c];
                        const float r = a + b;  /* interp.hpp:184           */
                        w->R[c * SD + (size_t)s * ORACLE_DBLOCK + d] = r;
                        /* cv::max(R, 0) (core.hpp:580) */
                        w->R0[c * SD + (size_t)s * ORACLE_DBLOCK + d] = (r > 0.0f) ? r : 0.0f;
                    }
                    card[d] = card[d] + 1.0f;
                } else {
                    for (int c = 0; c < C; c++) {
                        w->R[c * SD + (size_t)s * ORACLE_DBLOCK + d] = NAN; /* interp.hpp:189 */
                        w->R0[c * SD + (size_t)s * ORACLE_DBLOCK + d] = 0.0f;
                    }
                }
            }
        }

        /* core.hpp:577: r_bar <- R[s_hat,:] */
        float* rbarp_w[3];
        const float* rbarp[3];
        float* Ap[3];
        for (int c = 0; c < C; c++) {
            rbarp_w[c] = w->rbar + (size_t)c * D + c0;
            rbarp[c] = rbarp_w[c];
            Ap[c] = w->A + (size_t)c * D + c0;
            memcpy(rbarp_w[c], w->R + c * SD + (size_t)s_hat * ORACLE_DBLOCK, sizeof(float) * nb);
        }
        float* B = w->B + c0;

        /* core.hpp:584-610 */
        for (int it = 0; it < n_iter; it++) {
            /* r*K (core.cpp:25-37) and the column sums (core.hpp:602-603): cv::reduce
             * starts from row 0 and adds the rows in order.  Starting from +0 instead is
             * the same value: every term is >= +0, and 0 + x == x. */
            for (int c = 0; c < C; c++)
                memset(Ap[c], 0, sizeof(float) * nb);
            memset(B, 0, sizeof(float) * nb);
            for (int s = 0; s < S; s++) {
                float* restrict K = w->Kmat + (size_t)s * D + c0;
                if (C == 1) {
                    /* fused single pass (the hot loop of the CPU baseline) */
                    const float* restrict r = w->R + (size_t)s * ORACLE_DBLOCK;
                    const float* restrict r0 = w->R0 + (size_t)s * ORACLE_DBLOCK;
                    const float* restrict rb = rbarp[0];
                    float* restrict A0 = Ap[0];
                    float* restrict Bb = B;
                    for (int d = 0; d < nb; d++) {
                        const float delta = r[d] - rb[d];          /* core.hpp:591 */
                        const float t = k1 * delta;                /* kernels.cpp:21 */
                        const float q = t * delta;
                        const float o = 1.0f - q;                  /* kernels.cpp:23 */
                        const float k = (o > 0.0f) ? o : 0.0f;     /* kernels.cpp:25, NaN -> 0 */
                        K[d] = k;
                        const float pr = r0[d] * k;                /* core.cpp:28 */
                        A0[d] = A0[d] + pr;                        /* core.hpp:602 */
                        Bb[d] = Bb[d] + k;                         /* core.hpp:603 */
                    }
                } else {
                    const float* Rrow[3];
                    for (int c = 0; c < C; c++)
                        Rrow[c] = w->R + c * SD + (size_t)s * ORACLE_DBLOCK;
                    kernel_row(Rrow, rbarp, nb, C, inv_h2, k1, K);
                    for (int c = 0; c < C; c++) {
                        const float* r0 = w->R0 + c * SD + (size_t)s * ORACLE_DBLOCK;
                        for (int d = 0; d < nb; d++) {
                            const float pr = r0[d] * K[d];
                            Ap[c][d] = Ap[c][d] + pr;
                        }
                    }
                    for (int d = 0; d < nb; d++)
                        B[d] = B[d] + K[d];
                }
            }
            /* _divide_multi_channel (core.cpp:39-51): OpenCV 3.x, divisor 0 -> 0;
             * then cv::max(r_bar, 0) (core.hpp:609) */
            for (int c = 0; c < C; c++) {
                for (int d = 0; d < nb; d++) {
                    float q = (B[d] != 0.0f) ? (Ap[c][d] / B[d]) : 0.0f;
                    rbarp_w[c][d] = (q > 0.0f) ? q : 0.0f;
                }
            }
        }

        /* core.hpp:616-622: K is re-evaluated on the last r - r_bar, i.e. it
         * equals the last iteration's K, and its column sum the last B. */
        for (int d = 0; d < nb; d++) {
            float sc = (card[d] != 0.0f) ? (B[d] / card[d]) : 0.0f;
            w->score[c0 + d] = (sc > 0.0f) ? sc : 0.0f;
        }
    }
}

/* cv::max(x, 0) on element `flat` of an array of n floats: the SIMD body sends NaN to 0; under
 * ORACLE_ASSUME_MAX_NAN_TAIL the scalar tail (the last n % 8 elements, std::max) keeps it. */
static inline float max0_at(float x, size_t flat, size_t n)
{
    if (x > 0.0f)
        return x;
    if ((g_assume & ORACLE_ASSUME_MAX_NAN_TAIL) && x != x && flat >= (n / 8) * 8)
        return x;
    return 0.0f;
}

/* scan_pixel under the what-if switches: the same passes over full S x D matrices, as the reference runs them
 * (core.hpp:540-625), slow and plain.  Linear interpolation only. */
static void scan_pixel_assume(const float* epi, int S, int U, int C, int u, float dmin, float dmax, int D, int s_hat,
                              const oracle_params* p, scan_scratch* w, float inv_h2, float k1, int n_iter)
{
    const size_t SD = (size_t)S * D;
    float* R = (float*)malloc(sizeof(float) * SD * C * 2 + sizeof(float) * SD);
    float* R0 = R + SD * C;
    float* K = R0 + SD * C;
    const float range = dmax - dmin, denom = (float)(D - 1), uf = (float)u;
    for (int d = 0; d < D; d++) {
        const float num = (float)d * range;
        const float quo = num / denom;
        w->Dv[d] = dmin + quo;
        w->card[d] = 0.0f;
    }
    for (int s = 0; s < S; s++) {
        const float Ss = (float)(s_hat - s);
        const float* erow = epi + (size_t)s * U * C;
        for (int d = 0; d < D; d++) {
            float xi = Ss * w->Dv[d];
            xi = xi * p->slope_factor;
            xi = xi + uf;
            const int i0 = (int)floorf(xi), i1 = (int)ceilf(xi);
            const float t = xi - (float)i0;
            const int valid = !(i0 < 0 || i1 > U - 1);
            for (int c = 0; c < C; c++) {
                float r = NAN;
                if (valid) {
                    const float omt = 1.0f - t;
                    const float a = omt * erow[(size_t)i0 * C + c];
                    const float b = t * erow[(size_t)i1 * C + c];
                    r = a + b;
                }
                R[c * SD + (size_t)s * D + d] = r;
                /* core.hpp:580: cv::max(R, 0) on the S x D (x C interleaved) matrix */
                R0[c * SD + (size_t)s * D + d] = max0_at(r, ((size_t)s * D + d) * C + c, SD * C);
            }
            if (valid)
                w->card[d] = w->card[d] + 1.0f;
        }
    }
    for (int c = 0; c < C; c++)
        memcpy(w->rbar + (size_t)c * D, R + c * SD + (size_t)s_hat * D, sizeof(float) * D);
    for (int it = 0; it < n_iter; it++) {
        for (int c = 0; c < C; c++)
            memset(w->A + (size_t)c * D, 0, sizeof(float) * D);
        memset(w->B, 0, sizeof(float) * D);
        for (int s = 0; s < S; s++) {
            for (int d = 0; d < D; d++) {
                float q[3] = {0.0f, 0.0f, 0.0f};
                const float scale = (C == 1) ? k1 : inv_h2;
                for (int c = 0; c < C; c++) {
                    const float delta = R[c * SD + (size_t)s * D + d] - w->rbar[(size_t)c * D + d];
                    if (g_assume & ORACLE_ASSUME_MUL_SCALE_LAST) {   /* scale * (a * b) */
                        const float dd = delta * delta;
                        q[c] = scale * dd;
                    } else {                                         /* (scale * a) * b */
                        const float t = scale * delta;
                        q[c] = t * delta;
                    }
                }
                float qs = q[0];
                if (C == 3) {
                    if (g_assume & ORACLE_ASSUME_RGB_SUM_IN_ORDER) {
                        qs = q[0] + q[1];
                        qs = qs + q[2];
                    } else {
                        qs = q[0] + q[2];
                        qs = qs + q[1];
                    }
                }
                const float o = 1.0f - qs;
                const float k = max0_at(o, (size_t)s * D + d, SD);   /* kernels.cpp:25 / :53 on the S x D matrix */
                K[(size_t)s * D + d] = k;
                for (int c = 0; c < C; c++) {
                    const float pr = R0[c * SD + (size_t)s * D + d] * k;
                    w->A[(size_t)c * D + d] = w->A[(size_t)c * D + d] + pr;
                }
                w->B[d] = w->B[d] + k;
            }
        }
        for (int c = 0; c < C; c++)
            for (int d = 0; d < D; d++) {
                float q;
                if (g_assume & ORACLE_ASSUME_DIV0_IEEE)
                    q = w->A[(size_t)c * D + d] / w->B[d];           /* OpenCV 4.x: IEEE inf / NaN */
                else
                    q = (w->B[d] != 0.0f) ? (w->A[(size_t)c * D + d] / w->B[d]) : 0.0f;
                w->rbar[(size_t)c * D + d] = max0_at(q, (size_t)d * C + c, (size_t)D * C);   /* core.hpp:609 */
            }
    }
    for (int d = 0; d < D; d++) {
        float sc;
        if (g_assume & ORACLE_ASSUME_DIV0_IEEE)
            sc = w->B[d] / w->card[d];
        else
            sc = (w->card[d] != 0.0f) ? (w->B[d] / w->card[d]) : 0.0f;
        w->score[d] = max0_at(sc, (size_t)d, (size_t)D);             /* core.hpp:622 */
    }
    free(R);
}

static int iter_count(float max_iter)
{
    /* for (int i=0; i < par_mean_shift_max_iter; i++) with a float bound
     * (core.hpp:584, :115) */
    int n = 0;
    while ((float)n < max_iter)
        n++;
    return n;
}

static void kernel_consts(const oracle_params* p, float* inv_h2, float* k1)
{
    /* kernels.hpp:43: inv_m_h_sq = 1.0 / (m_h_ * m_h_) -- float product,
     * double quotient, float store */
    const float h = p->kernel_bandwidth;
    const float hh = h * h;
    *inv_h2 = (float)(1.0 / (double)hh);
    /* kernels.cpp:21: 3 * inv_m_h_sq (float), passed as double scale, used as float */
    *k1 = 3.0f * (*inv_h2);
}

static void depth_epi_impl(const float* epi, int S, int U, int C,
                           const float* dmin_u, const float* dmax_u,
                           int dim_d, int s_hat,
                           float* Ce_u, uint8_t* Ce_mask_u,
                           float* Cd_u, float* depth_u, float* rbar_u,
                           const oracle_params* p, uint8_t* mask_u,
                           int32_t* idx_u, float* score_u, float* K_su,
                           scan_scratch* w)
{
    float inv_h2, k1;
    kernel_consts(p, &inv_h2, &k1);
    const int n_iter = iter_count(p->mean_shift_max_iter);
    const int D = dim_d;

    /* core.hpp:510-513 */
    if (mask_u) {
        for (int u = 0; u < U; u++)
            mask_u[u] = Ce_mask_u[u] & mask_u[u];
    }
    const uint8_t* scan_mask = mask_u ? mask_u : Ce_mask_u;

    for (int u = 0; u < U; u++) { /* core.hpp:527 (findNonZero order = ascending u) */
        if (idx_u)
            idx_u[u] = -1;
        if (score_u)
            score_u[u] = 0.0f;
        if (!scan_mask[u])
            continue;

        if (g_assume && p->interpolation == ORACLE_INTERP_LINEAR)
            scan_pixel_assume(epi, S, U, C, u, dmin_u[u], dmax_u[u], D, s_hat, p, w, inv_h2, k1, n_iter);
        else
            scan_pixel(epi, S, U, C, u, dmin_u[u], dmax_u[u], D, s_hat, p, w, inv_h2, k1, n_iter);

        /* core.hpp:630-634: minMaxLoc, first maximum */
        int best = 0;
        float bestv = w->score[0];
        for (int d = 1; d < D; d++) {
            if (w->score[d] > bestv) {
                bestv = w->score[d];
                best = d;
            }
        }
        const double maxVal = (double)bestv;
        if (maxVal > (double)p->raw_score_threshold) { /* core.hpp:636 */
            depth_u[u] = w->Dv[best];                  /* core.hpp:638 */
            /* core.hpp:641: C_e * |max - mean(score)| in double */
            double sum = 0.0;
            for (int d = 0; d < D; d++)
                sum += (double)w->score[d];
            const double mean = sum / (double)D;
            Cd_u[u] = (float)((double)Ce_u[u] * fabs(maxVal - mean));
            for (int c = 0; c < C; c++)                /* core.hpp:644 */
                rbar_u[(size_t)u * C + c] = w->rbar[(size_t)c * D + best];
            if (K_su) {                                /* core.hpp:647-651 */
                for (int s = 0; s < S; s++)
                    K_su[(size_t)s * U + u] = w->Kmat[(size_t)s * D + best];
            }
            if (idx_u)
                idx_u[u] = best;
            if (score_u)
                score_u[u] = bestv;
        } else {                                       /* core.hpp:653-657 */
            Ce_u[u] = 0.0f;
            Ce_mask_u[u] = 0;
        }
    }
}

void oracle_depth_epi(const float* epi, int S, int U, int C,
                      const float* dmin_u, const float* dmax_u,
                      int dim_d, int s_hat,
                      float* Ce_u, uint8_t* Ce_mask_u,
                      float* Cd_u, float* depth_u, float* rbar_u,
                      const oracle_params* p, uint8_t* mask_u,
                      int32_t* idx_u, float* score_u, float* K_su)
{
    scan_scratch w;
    if (!scratch_alloc(&w, S, dim_d, C))
        abort();
    depth_epi_impl(epi, S, U, C, dmin_u, dmax_u, dim_d, s_hat, Ce_u, Ce_mask_u,
                   Cd_u, depth_u, rbar_u, p, mask_u, idx_u, score_u, K_su, &w);
    scratch_free(&w);
}

/* ---- core.hpp:663-718 ------------------------------------------------- */

static int cmp_float(const void* a, const void* b)
{
    const float x = *(const float*)a, y = *(const float*)b;
    return (x > y) - (x < y);
}

void oracle_selective_median(const float* src_vu, float* dst_vu,
                             const float* vol, int V, int S, int U, int C,
                             int s_hat, int size, const uint8_t* mask_vu,
                             float epsilon)
{
    const int width = (size - 1) / 2; /* core.hpp:686 */
#pragma omp parallel for schedule(static)
    for (int v = 0; v < V; v++) {
        /* the window holds at most (2*width+1)^2 pixels and never more than the image (size 0: width 0, one pixel) */
        const long long side = 2 * (long long)(width < 0 ? 0 : width) + 1;
        const size_t side_v = side < V ? (size_t)side : (size_t)V, side_u = side < U ? (size_t)side : (size_t)U;
        float* buf = (float*)malloc(sizeof(float) * side_v * side_u);
        for (int u = 0; u < U; u++) {
            if (!mask_vu[(size_t)v * U + u]) /* core.hpp:695 */
                continue;
            const float* pc = vol + (((size_t)v * S + s_hat) * U + u) * C;
            int n = 0;
            const int k0 = v - width < 0 ? 0 : v - width;
            const int k1 = v + width + 1 > V ? V : v + width + 1;
            const int l0 = u - width < 0 ? 0 : u - width;
            const int l1 = u + width + 1 > U ? U : u + width + 1;
            for (int k = k0; k < k1; k++) {
                for (int l = l0; l < l1; l++) {
                    if (!mask_vu[(size_t)k * U + l])
                        continue;
                    const float* pn = vol + (((size_t)k * S + s_hat) * U + l) * C;
                    float df[3];
                    for (int c = 0; c < C; c++)
                        df[c] = pc[c] - pn[c];
                    if (norm_px(df, C) < epsilon) /* core.hpp:703-706 */
                        buf[n++] = src_vu[(size_t)k * U + l];
                }
            }
            /* std::nth_element(..., n/2): the n/2-th order statistic.  n == 0 -- a masked pixel whose centre radiance is NaN,
             * so that not even the pixel itself passes the test -- is undefined in the reference: core.hpp:713-714 reads
             * buffer[0] of a vector it has just cleared (in practice a stale value of the thread's previous pixel).  Defined
             * here, as in the HIP kernel (k3_median.hpp): 0, what a pixel outside the mask gets (core.hpp:678-679). */
            qsort(buf, (size_t)n, sizeof(float), cmp_float);
            dst_vu[(size_t)v * U + u] = n ? buf[n / 2] : 0.0f;
        }
        free(buf);
    }
}

/* ---- core.hpp:772-893 ------------------------------------------------- */

void oracle_depth_epi_pile(const float* vol, int V, int S, int U, int C,
                           const float* dmin_vu, const float* dmax_vu,
                           int dim_d, int s_hat,
                           float* Ce_vu, uint8_t* Ce_mask_vu,
                           float* Cd_vu, float* depth_vu, float* rbar_vu,
                           const oracle_params* p, uint8_t* mask_vu,
                           int32_t* idx_vu, float* score_vu,
                           float* depth_raw_vu)
{
#pragma omp parallel
    {
        scan_scratch w;
        if (!scratch_alloc(&w, S, dim_d, C))
            abort();
#pragma omp for schedule(dynamic, 1)
        for (int v = 0; v < V; v++) { /* core.hpp:799 */
            const size_t o = (size_t)v * U;
            depth_epi_impl(vol + (size_t)v * S * U * C, S, U, C,
                           dmin_vu + o, dmax_vu + o, dim_d, s_hat,
                           Ce_vu + o, Ce_mask_vu + o, Cd_vu + o, depth_vu + o,
                           rbar_vu + o * C, p, mask_vu ? mask_vu + o : NULL,
                           idx_vu ? idx_vu + o : NULL,
                           score_vu ? score_vu + o : NULL, NULL, &w);
        }
        scratch_free(&w);
    }

    if (depth_raw_vu)
        memcpy(depth_raw_vu, depth_vu, sizeof(float) * (size_t)V * U);

    /* core.hpp:881-892: the median reads the EDGE mask, output starts at 0,
     * and replaces best_depth */
    float* tmp = (float*)calloc((size_t)V * U, sizeof(float));
    oracle_selective_median(depth_vu, tmp, vol, V, S, U, C, s_hat,
                            p->median_filter_size, Ce_mask_vu,
                            p->median_filter_epsilon);
    memcpy(depth_vu, tmp, sizeof(float) * (size_t)V * U);
    free(tmp);
}

/* ---- dc.hpp:425-565 --------------------------------------------------- */

void oracle_depth1d_pile_run(const float* vol, int V, int S, int U, int C,
                             float dmin, float dmax, int dim_d, int s_hat,
                             const oracle_params* p,
                             float* Ce_vu, uint8_t* Ce_mask_vu,
                             float* Cd_vu, float* depth_vu, float* rbar_vu,
                             int32_t* idx_vu, float* score_vu,
                             float* depth_raw_vu)
{
    const size_t n = (size_t)V * U;
    if (s_hat < 0 || s_hat > S - 1)
        s_hat = (int)floor((0.0 + S) / 2); /* dc.hpp:490-494 */

    float* dmin_vu = (float*)malloc(sizeof(float) * n);
    float* dmax_vu = (float*)malloc(sizeof(float) * n);
    for (size_t i = 0; i < n; i++) { /* dc.hpp:486-487 */
        dmin_vu[i] = dmin;
        dmax_vu[i] = dmax;
    }
    memset(Ce_vu, 0, sizeof(float) * n); /* the reference leaves this uninitialised (dc.hpp:501) */
    memset(Cd_vu, 0, sizeof(float) * n);
    memset(depth_vu, 0, sizeof(float) * n);       /* dc.hpp:507 */
    memset(rbar_vu, 0, sizeof(float) * n * C);    /* dc.hpp:510 */

    oracle_edge_confidence_pile(vol, V, S, U, C, s_hat, Ce_vu, Ce_mask_vu, p); /* dc.hpp:538 */
    oracle_depth_epi_pile(vol, V, S, U, C, dmin_vu, dmax_vu, dim_d, s_hat,     /* dc.hpp:547 */
                          Ce_vu, Ce_mask_vu, Cd_vu, depth_vu, rbar_vu, p, NULL,
                          idx_vu, score_vu, depth_raw_vu);
    free(dmin_vu);
    free(dmax_vu);
}

/* ======================================================================
 * "next" row: compute_2D_edge_confidence / compute_2D_depth_epi
 * (core.hpp:901-1133) and Depth2DComputer (dc.hpp:651-805)
 * ====================================================================== */

void oracle_edge_confidence_2d(const float* vol, int V, int S, int U, int C,
                               float* Ce_svu, uint8_t* mask_svu, const oracle_params* p)
{
    const size_t n = (size_t)V * U;
    for (int s = 0; s < S; s++) /* core.hpp:918-934 */
        oracle_edge_confidence_pile(vol, V, S, U, C, s, Ce_svu + (size_t)s * n, mask_svu + (size_t)s * n, p);
}

/* pixels the sweeps have scanned since the last reset: the sum over visits of the pixels in both the view's edge mask and
 * its running mask on entry to the pile scan (core.hpp:515-527) -- bench.py's cpu_baseline counts its units from this */
static long long g_sweep_pixels_scanned = 0;
long long oracle_sweep_pixels_scanned(int reset)
{
    const long long v = g_sweep_pixels_scanned;
    if (reset)
        g_sweep_pixels_scanned = 0;
    return v;
}

void oracle_depth_epi_2d(const float* vol, int V, int S, int U, int C,
                         const float* dmin_svu, const float* dmax_svu, int dim_d,
                         float* Ce_svu, uint8_t* Ce_mask_svu, float* Cd_svu,
                         float* depth_svu, float* rbar_svu, const oracle_params* p,
                         float propagation_epsilon, uint8_t* scan_mask_svu)
{
    const size_t n = (size_t)V * U;
    const int s_mid = (int)floor(S / 2.0); /* core.hpp:954 */

    /* core.hpp:958-965: running masks start as clones of the edge masks */
    uint8_t* mask_svu = (uint8_t*)malloc((size_t)S * n);
    memcpy(mask_svu, Ce_mask_svu, (size_t)S * n);

    /* core.hpp:981-990: visiting order */
    int* order = (int*)malloc(sizeof(int) * (size_t)(2 * S + 2));
    int n_order = 0;
    order[n_order++] = s_mid;
    for (int off = 1; off < S - s_mid; off++) {
        order[n_order++] = s_mid + off;
        if (s_mid - off > -1)
            order[n_order++] = s_mid - off;
    }

    float* filtered = (float*)malloc(sizeof(float) * n);
    float* tmp = (float*)malloc(sizeof(float) * n);

    for (int k = 0; k < n_order; k++) {
        const int s_hat = order[k];
        float* Ce = Ce_svu + (size_t)s_hat * n;
        uint8_t* Cem = Ce_mask_svu + (size_t)s_hat * n;
        float* Cd = Cd_svu + (size_t)s_hat * n;
        float* depth = depth_svu + (size_t)s_hat * n;
        float* rbar = rbar_svu + (size_t)s_hat * n * C;
        uint8_t* mask = mask_svu + (size_t)s_hat * n;

        for (size_t i = 0; i < n; i++)
            g_sweep_pixels_scanned += (Cem[i] && mask[i]) ? 1 : 0;
        /* core.hpp:1012-1028: the pile scan writes raw depths into the stored plane ... */
        memcpy(tmp, depth, sizeof(float) * n);
        oracle_depth_epi_pile(vol, V, S, U, C, dmin_svu + (size_t)s_hat * n, dmax_svu + (size_t)s_hat * n,
                              dim_d, s_hat, Ce, Cem, Cd, tmp, rbar, p, mask, NULL, NULL, depth);
        /* ... (depth now holds the pre-median plane, as the storage does after core.hpp:819-854)
         * and core.hpp:892 leaves the median in the local header only: */
        memcpy(filtered, tmp, sizeof(float) * n);

        /* core.hpp:1088-1129: propagation, rows in parallel, u ascending within a row */
#pragma omp parallel for schedule(static)
        for (int v = 0; v < V; v++) {
            for (int u = 0; u < U; u++) {
                /* core.hpp:1097-1103: what lets a pixel paint */
                if (p->use_disp_confidence_score ? !(Cd[(size_t)v * U + u] > p->disp_score_threshold) : !Cem[(size_t)v * U + u])
                    continue;
                const float cur = filtered[(size_t)v * U + u];
                const float* rb = rbar + ((size_t)v * U + u) * C;
                for (int s = 0; s < S; s++) {
                    /* core.hpp:1109: u + (int)std::round(depth * (s_hat - s) * slope) */
                    float off = cur * (float)(s_hat - s);
                    off = off * p->slope_factor;
                    const int ri = u + (int)roundf(off);
                    if (ri > -1 && ri < U && mask_svu[(size_t)s * n + (size_t)v * U + ri]) {
                        const float* e = vol + (((size_t)v * S + s) * U + ri) * C;
                        float df[3];
                        for (int c = 0; c < C; c++)
                            df[c] = e[c] - rb[c];
                        if (norm_px(df, C) < propagation_epsilon) { /* core.hpp:1116 */
                            depth_svu[(size_t)s * n + (size_t)v * U + ri] = cur;
                            mask_svu[(size_t)s * n + (size_t)v * U + ri] = 0;
                            Cd_svu[(size_t)s * n + (size_t)v * U + ri] = Cd[(size_t)v * U + u];
                        }
                    }
                }
            }
        }
    }
    if (scan_mask_svu)
        memcpy(scan_mask_svu, mask_svu, (size_t)S * n);
    free(filtered);
    free(tmp);
    free(order);
    free(mask_svu);
}

void oracle_depth2d_run(const float* vol, int V, int S, int U, int C,
                        float dmin, float dmax, int dim_d, const oracle_params* p,
                        float propagation_epsilon,
                        float* Ce_svu, uint8_t* Ce_mask_svu, float* Cd_svu,
                        float* depth_svu, float* rbar_svu, uint8_t* scan_mask_svu)
{
    const size_t n = (size_t)S * V * U;
    float* dmin_svu = (float*)malloc(sizeof(float) * n);
    float* dmax_svu = (float*)malloc(sizeof(float) * n);
    for (size_t i = 0; i < n; i++) { /* dc.hpp:718-722 */
        dmin_svu[i] = dmin;
        dmax_svu[i] = dmax;
    }
    memset(Ce_svu, 0, sizeof(float) * n);   /* uninitialised in the reference (dc.hpp:735) */
    memset(Cd_svu, 0, sizeof(float) * n);   /* uninitialised in the reference (dc.hpp:738) */
    memset(depth_svu, 0, sizeof(float) * n);       /* dc.hpp:746 */
    memset(rbar_svu, 0, sizeof(float) * n * C);    /* dc.hpp:749 */
    oracle_edge_confidence_2d(vol, V, S, U, C, Ce_svu, Ce_mask_svu, p);   /* dc.hpp:772 */
    oracle_depth_epi_2d(vol, V, S, U, C, dmin_svu, dmax_svu, dim_d, Ce_svu, Ce_mask_svu, Cd_svu, depth_svu,
                        rbar_svu, p, propagation_epsilon, scan_mask_svu);      /* dc.hpp:780 */
    free(dmin_svu);
    free(dmax_svu);
}

/* ======================================================================
 * "next" row: fine-to-coarse (rslf_fine_to_coarse.hpp, rslf_fine_to_coarse_core.cpp)
 * ====================================================================== */

static int cv_round_half_even(double x)
{
    return (int)lrint(x); /* cvRound: the default rounding mode, ties to even */
}

void oracle_f2c_out_dims(int V, int U, int* V2, int* U2)
{
    /* cv::resize with dsize empty: Size(saturate_cast<int>(cols*fx), saturate_cast<int>(rows*fy)) */
    if (g_assume & ORACLE_ASSUME_SIZE_FLOOR) {
        *V2 = (int)floor(V * 0.5);
        *U2 = (int)floor(U * 0.5);
        return;
    }
    *V2 = cv_round_half_even(V * 0.5);
    *U2 = cv_round_half_even(U * 0.5);
}

static inline int reflect_border(int p, int len) /* cv::BORDER_REFLECT: fedcba|abcdefgh|hgfedcb */
{
    if (len == 1)
        return 0;
    while (p < 0 || p >= len)
        p = (p < 0) ? -p - 1 : 2 * len - 1 - p;
    return p;
}

/* cv::GaussianBlur(src, dst, Size(7,7), 0, 0, BORDER_REFLECT) on a float image [R][W][C]:
 * getGaussianKernel(7, sigma<=0) returns the fixed small_gaussian_tab row; sepFilter2D runs the row
 * filter (taps accumulated left to right) then the symmetric column filter
 * (centre tap, then k[c+j] * (S[y+j] + S[y-j]), j = 1..3).  No FMA. */
static void gaussian7_reflect(const float* src, float* dst, float* tmp, int R, int W, int C)
{
    static const float k[7] = {0.03125f, 0.109375f, 0.21875f, 0.28125f, 0.21875f, 0.109375f, 0.03125f};
    for (int y = 0; y < R; y++)
        for (int x = 0; x < W; x++)
            for (int c = 0; c < C; c++) {
                float s;
                if (g_assume & ORACLE_ASSUME_GAUSS_ROW_SYMM) {
                    s = k[3] * src[((size_t)y * W + x) * C + c];
                    for (int j = 1; j <= 3; j++) {
                        const float a = src[((size_t)y * W + reflect_border(x + j, W)) * C + c];
                        const float b = src[((size_t)y * W + reflect_border(x - j, W)) * C + c];
                        const float ab = a + b;
                        const float pr = k[3 + j] * ab;
                        s = s + pr;
                    }
                } else {
                    s = k[0] * src[((size_t)y * W + reflect_border(x - 3, W)) * C + c];
                    for (int j = 1; j < 7; j++) {
                        const float pr = k[j] * src[((size_t)y * W + reflect_border(x + j - 3, W)) * C + c];
                        s = s + pr;
                    }
                }
                tmp[((size_t)y * W + x) * C + c] = s;
            }
    for (int y = 0; y < R; y++)
        for (int x = 0; x < W; x++)
            for (int c = 0; c < C; c++) {
                float s;
                if (g_assume & ORACLE_ASSUME_GAUSS_COL_ORDER) {
                    s = k[0] * tmp[((size_t)reflect_border(y - 3, R) * W + x) * C + c];
                    for (int j = 1; j < 7; j++) {
                        const float pr = k[j] * tmp[((size_t)reflect_border(y + j - 3, R) * W + x) * C + c];
                        s = s + pr;
                    }
                } else {
                    s = k[3] * tmp[((size_t)y * W + x) * C + c];
                    for (int j = 1; j <= 3; j++) {
                        const float a = tmp[((size_t)reflect_border(y + j, R) * W + x) * C + c];
                        const float b = tmp[((size_t)reflect_border(y - j, R) * W + x) * C + c];
                        const float ab = a + b;
                        const float pr = k[3 + j] * ab;
                        s = s + pr;
                    }
                }
                dst[((size_t)y * W + x) * C + c] = s;
            }
}

/* cv::resize(src, dst, Size(), 0.5, 0.5, INTER_LINEAR): with an exact factor 2 OpenCV takes the
 * INTER_AREA fast path -- each output pixel is the mean of its 2x2 block, (S00 + S10) + (S01 + S11)
 * times 0.25 (vertical pairs first, the SIMD form); where the block sticks out of an odd-sized source
 * the mean runs over the pixels that exist, in row-major order, divided by their count. */
static void halve_area(const float* src, int R, int W, int C, float* dst, int R2, int W2)
{
    for (int y = 0; y < R2; y++)
        for (int x = 0; x < W2; x++)
            for (int c = 0; c < C; c++) {
                const int y0 = 2 * y, x0 = 2 * x;
                float out;
                if (y0 + 1 < R && x0 + 1 < W && (g_assume & ORACLE_ASSUME_AREA_SCALAR)) {
                    float sum = src[((size_t)y0 * W + x0) * C + c] + src[((size_t)y0 * W + x0 + 1) * C + c];
                    sum = sum + src[((size_t)(y0 + 1) * W + x0) * C + c];
                    sum = sum + src[((size_t)(y0 + 1) * W + x0 + 1) * C + c];
                    out = sum * 0.25f;
                } else if (y0 + 1 < R && x0 + 1 < W) {
                    const float a = src[((size_t)y0 * W + x0) * C + c] + src[((size_t)(y0 + 1) * W + x0) * C + c];
                    const float b = src[((size_t)y0 * W + x0 + 1) * C + c] + src[((size_t)(y0 + 1) * W + x0 + 1) * C + c];
                    const float ab = a + b;
                    out = ab * 0.25f;
                } else {
                    float sum = 0.0f;
                    int cnt = 0;
                    for (int sy = 0; sy < 2; sy++)
                        for (int sx = 0; sx < 2; sx++)
                            if (y0 + sy < R && x0 + sx < W) {
                                sum = sum + src[((size_t)(y0 + sy) * W + x0 + sx) * C + c];
                                cnt++;
                            }
                    out = cnt ? sum / (float)cnt : 0.0f;
                }
                dst[((size_t)y * W2 + x) * C + c] = out;
            }
}

void oracle_downsample_epis(const float* in, int V, int S, int U, int C, float* out)
{
    int V2, U2;
    oracle_f2c_out_dims(V, U, &V2, &U2);
#pragma omp parallel
    {
        float* img = (float*)malloc(sizeof(float) * (size_t)V * U * C);
        float* tmp = (float*)malloc(sizeof(float) * (size_t)V * U * C);
        float* blur = (float*)malloc(sizeof(float) * (size_t)V * U * C);
        float* half = (float*)malloc(sizeof(float) * (size_t)V2 * U2 * C);
#pragma omp for schedule(static)
        for (int s = 0; s < S; s++) { /* fine_to_coarse_core.cpp:28-46 */
            for (int v = 0; v < V; v++)
                memcpy(img + (size_t)v * U * C, in + (((size_t)v * S + s) * U) * C, sizeof(float) * (size_t)U * C);
            gaussian7_reflect(img, blur, tmp, V, U, C);
            halve_area(blur, V, U, C, half, V2, U2);
            for (int v = 0; v < V2; v++) /* :49-59 */
                memcpy(out + (((size_t)v * S + s) * U2) * C, half + (size_t)v * U2 * C, sizeof(float) * (size_t)U2 * C);
        }
        free(img); free(tmp); free(blur); free(half);
    }
}

/* downsample_EPIs on CV_8U Mats (fine_to_coarse_core.cpp:14-60; the reference blurs and resizes in the input's own
 * type).  Values are uchar levels 0..255, carried here in float arrays.
 *  - cv::GaussianBlur(7x7, sigma 0, BORDER_REFLECT) on 8U: the 7-tap kernel {1, 3.5, 7, 9, 7, 3.5, 1}/32 is exact in
 *    8 fractional bits ({8, 28, 56, 72, 56, 28, 8}/256), so both 8U paths of OpenCV 3.4 -- the 8-bit fixed-point
 *    separable filter up to 3.4.0 (row filter to int, column filter with FixedPtCastEx<int, uchar>(16)) and the
 *    ufixedpoint16 path from 3.4.1 on -- form the exact double sum and round it once, half up:
 *    (sum + 32768) >> 16.  The restatement therefore holds for every 3.4.x.
 *  - cv::resize(0.5, 0.5, INTER_LINEAR) with an exact factor 2 runs INTER_AREA's fast path: (S00 + S01 + S10 + S11
 *    + 2) >> 2; where the block sticks out of an odd-sized source, saturate_cast<uchar>((float)sum / count), i.e.
 *    cvRound (ties to even). */
static void gaussian7_reflect_u8(const float* src, int* dst, int* tmp, int R, int W, int C)
{
    static const int k[7] = {8, 28, 56, 72, 56, 28, 8};
    for (int y = 0; y < R; y++)
        for (int x = 0; x < W; x++)
            for (int c = 0; c < C; c++) {
                int s = 0;
                for (int j = 0; j < 7; j++)
                    s += k[j] * (int)src[((size_t)y * W + reflect_border(x + j - 3, W)) * C + c];
                tmp[((size_t)y * W + x) * C + c] = s;
            }
    for (int y = 0; y < R; y++)
        for (int x = 0; x < W; x++)
            for (int c = 0; c < C; c++) {
                int s = 0;
                for (int j = 0; j < 7; j++)
                    s += k[j] * tmp[((size_t)reflect_border(y + j - 3, R) * W + x) * C + c];
                dst[((size_t)y * W + x) * C + c] = (s + 32768) >> 16;
            }
}

static void halve_area_u8(const int* src, int R, int W, int C, float* dst, int R2, int W2)
{
    for (int y = 0; y < R2; y++)
        for (int x = 0; x < W2; x++)
            for (int c = 0; c < C; c++) {
                const int y0 = 2 * y, x0 = 2 * x;
                int out;
                if (y0 + 1 < R && x0 + 1 < W) {
                    out = (src[((size_t)y0 * W + x0) * C + c] + src[((size_t)y0 * W + x0 + 1) * C + c] +
                           src[((size_t)(y0 + 1) * W + x0) * C + c] + src[((size_t)(y0 + 1) * W + x0 + 1) * C + c] + 2) >> 2;
                } else {
                    int sum = 0, cnt = 0;
                    for (int sy = 0; sy < 2; sy++)
                        for (int sx = 0; sx < 2; sx++)
                            if (y0 + sy < R && x0 + sx < W) {
                                sum += src[((size_t)(y0 + sy) * W + x0 + sx) * C + c];
                                cnt++;
                            }
                    out = cnt ? (int)lrintf((float)sum / (float)cnt) : 0;
                }
                dst[((size_t)y * W2 + x) * C + c] = (float)out;
            }
}

void oracle_downsample_epis_u8(const float* in, int V, int S, int U, int C, float* out)
{
    int V2, U2;
    oracle_f2c_out_dims(V, U, &V2, &U2);
#pragma omp parallel
    {
        float* img = (float*)malloc(sizeof(float) * (size_t)V * U * C);
        int* tmp = (int*)malloc(sizeof(int) * (size_t)V * U * C);
        int* blur = (int*)malloc(sizeof(int) * (size_t)V * U * C);
        float* half = (float*)malloc(sizeof(float) * (size_t)V2 * U2 * C);
#pragma omp for schedule(static)
        for (int s = 0; s < S; s++) {
            for (int v = 0; v < V; v++)
                memcpy(img + (size_t)v * U * C, in + (((size_t)v * S + s) * U) * C, sizeof(float) * (size_t)U * C);
            gaussian7_reflect_u8(img, blur, tmp, V, U, C);
            halve_area_u8(blur, V, U, C, half, V2, U2);
            for (int v = 0; v < V2; v++)
                memcpy(out + (((size_t)v * S + s) * U2) * C, half + (size_t)v * U2 * C, sizeof(float) * (size_t)U2 * C);
        }
        free(img); free(tmp); free(blur); free(half);
    }
}

void oracle_f2c_tighten_bounds(const float* depth_up, const uint8_t* mask_up, int S, int V_up, int U_up,
                               float* dmin_down, float* dmax_down, int V_down, int U_down)
{
    const size_t nu = (size_t)V_up * U_up, nd = (size_t)V_down * U_down;
#pragma omp parallel for schedule(static)
    for (int s = 0; s < S; s++) { /* rslf_fine_to_coarse.hpp:202 */
        const float* dep = depth_up + (size_t)s * nu;
        const uint8_t* msk = mask_up + (size_t)s * nu;
        for (int v = 0; v < V_down; v++)
            for (int u = 0; u < U_down; u++) {
                float cand[4];
                int nc = 0;
                int v_up = (2 * v < V_up - 1) ? 2 * v : V_up - 1;   /* :212 */
                const int u_up = (2 * u < U_up - 1) ? 2 * u : U_up - 1;
                for (int line = 0; line < 2; line++) {
                    if (line == 1) {
                        if (v_up + 1 < V_up) /* :249 */
                            v_up += 1;
                        else
                            break;
                    }
                    int found_l = 0, found_r = 0;
                    float dl = 0, dr = 0;
                    int ul = u_up;
                    while (ul > 1) { /* :217-225: never looks at columns 0 */
                        ul -= 1;
                        if (msk[(size_t)v_up * U_up + ul] > 0) {
                            dl = dep[(size_t)v_up * U_up + ul];
                            found_l = 1;
                            break;
                        }
                    }
                    int ur = u_up;
                    while (ur < U_up - 1) { /* :228-236 */
                        ur += 1;
                        if (msk[(size_t)v_up * U_up + ur] > 0) {
                            dr = dep[(size_t)v_up * U_up + ur];
                            found_r = 1;
                            break;
                        }
                    }
                    if (found_l && found_r) { /* :242-246 (NaN = "not found") */
                        cand[nc++] = dl;
                        cand[nc++] = dr;
                    }
                }
                if (nc > 1) { /* :281-289: min and max of the candidates */
                    float lo = cand[0], hi = cand[0];
                    for (int i = 1; i < nc; i++) {
                        if (cand[i] < lo) lo = cand[i];
                        if (cand[i] > hi) hi = cand[i];
                    }
                    dmin_down[(size_t)s * nd + (size_t)v * U_down + u] = lo;
                    dmax_down[(size_t)s * nd + (size_t)v * U_down + u] = hi;
                }
            }
    }
}

/* cv::resize(..., dsize, 0, 0, INTER_LINEAR) of a float plane, upscaling (fine_to_coarse_core.cpp:104) */
static void resize_linear_f32(const float* src, int R, int W, float* dst, int R2, int W2)
{
    const double scale_x = 1.0 / ((double)W2 / W), scale_y = 1.0 / ((double)R2 / R);
    int* xofs = (int*)malloc(sizeof(int) * (size_t)W2);
    float* xa = (float*)malloc(sizeof(float) * 2 * (size_t)W2);
    int xmax = W2;
    const int coord_float = (g_assume & ORACLE_ASSUME_RESIZE_FLOAT) != 0;
    for (int dx = 0; dx < W2; dx++) {
        float fx = coord_float ? ((float)dx + 0.5f) * (float)scale_x - 0.5f : (float)((dx + 0.5) * scale_x - 0.5);
        int sx = (int)floor(fx);
        fx -= sx;
        if (sx < 0) {
            fx = 0;
            sx = 0;
        }
        if (sx + 1 >= W) {
            if (dx < xmax)
                xmax = dx;
            if (sx >= W - 1) {
                fx = 0;
                sx = W - 1;
            }
        }
        xofs[dx] = sx;
        xa[2 * dx] = 1.f - fx;
        xa[2 * dx + 1] = fx;
    }
    float* row0 = (float*)malloc(sizeof(float) * (size_t)W2);
    float* row1 = (float*)malloc(sizeof(float) * (size_t)W2);
    for (int dy = 0; dy < R2; dy++) {
        float fy = coord_float ? ((float)dy + 0.5f) * (float)scale_y - 0.5f : (float)((dy + 0.5) * scale_y - 0.5);
        int sy = (int)floor(fy);
        fy -= sy;
        const float b0 = 1.f - fy, b1 = fy;
        int y0 = sy < 0 ? 0 : (sy > R - 1 ? R - 1 : sy);
        int y1 = sy + 1 < 0 ? 0 : (sy + 1 > R - 1 ? R - 1 : sy + 1);
        const float* S0 = src + (size_t)y0 * W;
        const float* S1 = src + (size_t)y1 * W;
        for (int dx = 0; dx < W2; dx++) { /* HResizeLinear */
            const int sx = xofs[dx];
            if (dx < xmax) {
                const float a0 = S0[sx] * xa[2 * dx], a1 = S0[sx + 1] * xa[2 * dx + 1];
                row0[dx] = a0 + a1;
                const float c0 = S1[sx] * xa[2 * dx], c1 = S1[sx + 1] * xa[2 * dx + 1];
                row1[dx] = c0 + c1;
            } else {
                row0[dx] = S0[sx] * 1.f;
                row1[dx] = S1[sx] * 1.f;
            }
        }
        for (int dx = 0; dx < W2; dx++) { /* VResizeLinear */
            const float p0 = row0[dx] * b0, p1 = row1[dx] * b1;
            dst[(size_t)dy * W2 + dx] = p0 + p1;
        }
    }
    free(xofs); free(xa); free(row0); free(row1);
}

/* cv::resize(..., INTER_NEAREST) of a uchar plane (fine_to_coarse_core.cpp:106) */
static void resize_nearest_u8(const uint8_t* src, int R, int W, uint8_t* dst, int R2, int W2)
{
    const double ifx = 1.0 / ((double)W2 / W), ify = 1.0 / ((double)R2 / R);
    for (int y = 0; y < R2; y++) {
        int sy = (int)floor(y * ify);
        if (sy > R - 1)
            sy = R - 1;
        for (int x = 0; x < W2; x++) {
            int sx = (int)floor(x * ifx);
            if (sx > W - 1)
                sx = W - 1;
            dst[(size_t)y * W2 + x] = src[(size_t)sy * W + sx];
        }
    }
}

/* cv::medianBlur(src, dst, 3) on float: 3x3 window, BORDER_REPLICATE, exact median of 9 */
static void median3_f32(const float* src, int R, int W, float* dst)
{
    for (int y = 0; y < R; y++)
        for (int x = 0; x < W; x++) {
            float w[9];
            int n = 0;
            for (int dy = -1; dy <= 1; dy++)
                for (int dx = -1; dx <= 1; dx++) {
                    int yy = y + dy, xx = x + dx;
                    yy = yy < 0 ? 0 : (yy > R - 1 ? R - 1 : yy);
                    xx = xx < 0 ? 0 : (xx > W - 1 ? W - 1 : xx);
                    w[n++] = src[(size_t)yy * W + xx];
                }
            qsort(w, 9, sizeof(float), cmp_float);
            dst[(size_t)y * W + x] = w[4];
        }
}

void oracle_f2c_fuse(const float* const* disp, const uint8_t* const* valid, const int* Vp, const int* Up, int P,
                     float* out_map, uint8_t* out_valid)
{
    /* fine_to_coarse_core.cpp:93-127 */
    size_t n_last = (size_t)Vp[P - 1] * Up[P - 1];
    float* map_down = (float*)malloc(sizeof(float) * n_last);
    uint8_t* mask_down = (uint8_t*)malloc(n_last);
    memcpy(map_down, disp[P - 1], sizeof(float) * n_last);
    memcpy(mask_down, valid[P - 1], n_last);
    for (int p = P - 1; p > 0; p--) {
        const int R = Vp[p - 1], W = Up[p - 1];
        const size_t n = (size_t)R * W;
        float* map_up = (float*)malloc(sizeof(float) * n);
        uint8_t* mask_up = (uint8_t*)malloc(n);
        resize_linear_f32(map_down, Vp[p], Up[p], map_up, R, W);       /* :104 */
        resize_nearest_u8(mask_down, Vp[p], Up[p], mask_up, R, W);     /* :106 */
        float* nd = (float*)malloc(sizeof(float) * n);
        uint8_t* nm = (uint8_t*)malloc(n);
        for (size_t i = 0; i < n; i++) {
            const int inval = valid[p - 1][i] == 0;                     /* :117 */
            /* :119-120: setTo(0, invalid) then add(map, map_up, map, invalid) */
            nd[i] = inval ? (0.0f + map_up[i]) : disp[p - 1][i];
            nm[i] = valid[p - 1][i] | mask_up[i];                       /* :122 */
        }
        free(map_down); free(mask_down); free(map_up); free(mask_up);
        map_down = nd;
        mask_down = nm;
    }
    median3_f32(map_down, Vp[0], Up[0], out_map);                      /* :127 */
    memcpy(out_valid, mask_down, (size_t)Vp[0] * Up[0]);
    free(map_down);
    free(mask_down);
}
