// K5: the fine-to-coarse row: pyramid construction, per-pixel bound
// tightening between levels, coarse-to-fine fusion.  All HBM-bound streaming kernels over dense
// float / byte planes; the scans of every level are K1-K4.
//
// rslf::downsample_EPIs      src/rslf_fine_to_coarse_core.cpp:14-60
// FineToCoarse::run (bounds) include/rslf_fine_to_coarse.hpp:171-299
// rslf::fuse_disp_maps       src/rslf_fine_to_coarse_core.cpp:69-135
// The OpenCV 3.x primitives are restated from their algorithms (DESIGN.md lists the choices that
// touch the last bit); arithmetic is binary32, one rounding per operation, no FMA.
#pragma once

#include "rslf_device.hpp"

namespace rslf {

__device__ __forceinline__ int reflect_border(int p, int len)   // cv::BORDER_REFLECT
{
    if (len == 1)
        return 0;
    while (p < 0 || p >= len)
        p = (p < 0) ? -p - 1 : 2 * len - 1 - p;
    return p;
}

__constant__ const float kGauss7[7] = {0.03125f, 0.109375f, 0.21875f, 0.28125f, 0.21875f, 0.109375f, 0.03125f};

// Row pass of cv::GaussianBlur(7x7, sigma 0 => small_gaussian_tab, BORDER_REFLECT) on the dense raw
// volume [V][S][U][C]: along u, taps accumulated left to right.  One thread per value.
__global__ __launch_bounds__(256) void k5_gauss_rows(const float* __restrict__ in, float* __restrict__ tmp, long long rows, int U, int C)
{
    const int xblocks = (U * C + (int)blockDim.x - 1) / (int)blockDim.x;   // 1-D grid: V*S can exceed the 65535 of grid.y
    const long long row = blockIdx.x / xblocks;   // over V*S
    const int xc = (int)(blockIdx.x % xblocks) * blockDim.x + threadIdx.x;
    if (xc >= U * C || row >= rows)
        return;
    const int x = xc / C, c = xc - x * C;
    const float* r = in + row * (long long)U * C;
    float s = kGauss7[0] * r[(long long)reflect_border(x - 3, U) * C + c];
#pragma unroll
    for (int j = 1; j < 7; j++) {
        const float pr = kGauss7[j] * r[(long long)reflect_border(x + j - 3, U) * C + c];
        s = s + pr;
    }
    tmp[row * (long long)U * C + xc] = s;
}

// Column pass (symmetric form: centre tap, then k[c+j] * (S[y+j] + S[y-j])) fused with the halving of
// cv::resize(0.5, 0.5, INTER_LINEAR) = OpenCV's 2x2 area-fast mean, (S00 + S10) + (S01 + S11) times
// 0.25; where the block leaves an odd-sized image, the mean of the pixels that exist.
// tmp [V][S][U][C] -> out [V2][S][U2][C].  One thread per output value.
__device__ __forceinline__ float gauss_col(const float* __restrict__ tmp, int y, int s, int x, int c, int V, int S, int U, int C)
{
    const long long rs = (long long)S * U * C;   // stride between image rows (scanlines)
    const long long o = ((long long)s * U + x) * C + c;
    float acc = kGauss7[3] * tmp[(long long)y * rs + o];
#pragma unroll
    for (int j = 1; j <= 3; j++) {
        const float a = tmp[(long long)reflect_border(y + j, V) * rs + o];
        const float b = tmp[(long long)reflect_border(y - j, V) * rs + o];
        const float ab = a + b;
        const float pr = kGauss7[3 + j] * ab;
        acc = acc + pr;
    }
    return acc;
}

__global__ __launch_bounds__(256) void k5_gauss_cols_halve(const float* __restrict__ tmp, float* __restrict__ out, int V, int S, int U,
                                                          int C, int V2, int U2)
{
    const int xc = blockIdx.x * blockDim.x + threadIdx.x;
    const int s = blockIdx.y, y = blockIdx.z;
    if (xc >= U2 * C)
        return;
    const int x = xc / C, c = xc - x * C;
    const int y0 = 2 * y, x0 = 2 * x;
    float r;
    if (y0 + 1 < V && x0 + 1 < U) {
        const float a = gauss_col(tmp, y0, s, x0, c, V, S, U, C) + gauss_col(tmp, y0 + 1, s, x0, c, V, S, U, C);
        const float b = gauss_col(tmp, y0, s, x0 + 1, c, V, S, U, C) + gauss_col(tmp, y0 + 1, s, x0 + 1, c, V, S, U, C);
        const float ab = a + b;
        r = ab * 0.25f;
    } else {
        float sum = 0.0f;
        int cnt = 0;
        for (int sy = 0; sy < 2; sy++)
            for (int sx = 0; sx < 2; sx++)
                if (y0 + sy < V && x0 + sx < U) {
                    sum = sum + gauss_col(tmp, y0 + sy, s, x0 + sx, c, V, S, U, C);
                    cnt++;
                }
        r = cnt ? sum / (float)cnt : 0.0f;
    }
    out[(((long long)y * S + s) * U2 + x) * C + c] = r;
}

// ---- the same on CV_8U Mats (uchar levels 0..255 carried in float arrays) -----------------------------------
// The reference blurs and halves a uchar light field in uchar arithmetic (fine_to_coarse_core.cpp:22-41).
// cv::GaussianBlur(7x7, sigma 0) on 8U: the taps are exact in 8 fractional bits ({8, 28, 56, 72, 56, 28, 8}/256), so
// OpenCV 3.4's 8U paths -- the 8-bit fixed-point separable filter (<= 3.4.0) and ufixedpoint16 (>= 3.4.1) -- both form
// the exact double sum and round once, half up: (sum + 32768) >> 16.  cv::resize(0.5, INTER_LINEAR) = INTER_AREA's fast
// path, (S00 + S01 + S10 + S11 + 2) >> 2; at an odd border saturate_cast<uchar>((float)sum / count) = cvRound.
__constant__ const int kGauss7u8[7] = {8, 28, 56, 72, 56, 28, 8};

__global__ __launch_bounds__(256) void k5_gauss_rows_u8(const float* __restrict__ in, int* __restrict__ tmp, long long rows, int U, int C)
{
    const int xblocks = (U * C + (int)blockDim.x - 1) / (int)blockDim.x;
    const long long row = blockIdx.x / xblocks;   // over V*S
    const int xc = (int)(blockIdx.x % xblocks) * blockDim.x + threadIdx.x;
    if (xc >= U * C || row >= rows)
        return;
    const int x = xc / C, c = xc - x * C;
    const float* r = in + row * (long long)U * C;
    int s = 0;
#pragma unroll
    for (int j = 0; j < 7; j++)
        s += kGauss7u8[j] * (int)r[(long long)reflect_border(x + j - 3, U) * C + c];
    tmp[row * (long long)U * C + xc] = s;   // 8 fractional bits, <= 255 * 256
}

__device__ __forceinline__ int gauss_col_u8(const int* __restrict__ tmp, int y, int s, int x, int c, int V, int S, int U, int C)
{
    const long long rs = (long long)S * U * C;
    const long long o = ((long long)s * U + x) * C + c;
    int acc = 0;
#pragma unroll
    for (int j = 0; j < 7; j++)
        acc += kGauss7u8[j] * tmp[(long long)reflect_border(y + j - 3, V) * rs + o];
    return (acc + 32768) >> 16;   // the blurred uchar level
}

__global__ __launch_bounds__(256) void k5_gauss_cols_halve_u8(const int* __restrict__ tmp, float* __restrict__ out, int V, int S, int U,
                                                             int C, int V2, int U2)
{
    const int xc = blockIdx.x * blockDim.x + threadIdx.x;
    const int s = blockIdx.y, y = blockIdx.z;
    if (xc >= U2 * C)
        return;
    const int x = xc / C, c = xc - x * C;
    const int y0 = 2 * y, x0 = 2 * x;
    int r;
    if (y0 + 1 < V && x0 + 1 < U) {
        r = (gauss_col_u8(tmp, y0, s, x0, c, V, S, U, C) + gauss_col_u8(tmp, y0, s, x0 + 1, c, V, S, U, C) +
             gauss_col_u8(tmp, y0 + 1, s, x0, c, V, S, U, C) + gauss_col_u8(tmp, y0 + 1, s, x0 + 1, c, V, S, U, C) + 2) >> 2;
    } else {
        int sum = 0, cnt = 0;
        for (int sy = 0; sy < 2; sy++)
            for (int sx = 0; sx < 2; sx++)
                if (y0 + sy < V && x0 + sx < U) {
                    sum += gauss_col_u8(tmp, y0 + sy, s, x0 + sx, c, V, S, U, C);
                    cnt++;
                }
        r = cnt ? (int)rintf((float)sum / (float)cnt) : 0;   // cvRound: ties to even
    }
    out[(((long long)y * S + s) * U2 + x) * C + c] = (float)r;
}

// max over a dense float buffer (per-level epi_scale_factor, dc.hpp:671-690)
__global__ __launch_bounds__(256) void k5_max_partial(const float* __restrict__ in, long long n, float* __restrict__ partial)
{
    float mx = -INFINITY;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        mx = fmaxf(mx, in[i]);
    for (int o = 32; o > 0; o >>= 1)
        mx = fmaxf(mx, __shfl_xor(mx, o));
    __shared__ float sm[4];
    if ((threadIdx.x & 63) == 0)
        sm[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0)
        partial[blockIdx.x] = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
}

// ---- bound tightening (rslf_fine_to_coarse.hpp:202-294) -----------------------------------------
// Pass 1, one wave per finer-level row (s, v): nearest valid column strictly left (never column 0,
// the reference's `while (u_left > 1)`) and strictly right of every column, -1 if none.
__global__ __launch_bounds__(256) void k5_nearest_valid(const uint8_t* __restrict__ mask_up, long long rows, int U,
                                                       int* __restrict__ left, int* __restrict__ right)
{
    // one WAVE per row, 64 columns at a time: a ballot of the chunk's valid columns answers every lane at once (highest
    // set bit below the lane / lowest above it), and what lies beyond the chunk is carried in a scalar -- coalesced,
    // where a thread walking its own row was neither parallel nor coalesced
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows)
        return;
    const int lane = threadIdx.x & 63;
    const uint8_t* m = mask_up + row * U;
    int* L = left + row * U;
    int* R = right + row * U;
    int carry = -1;                               // nearest valid column of the chunks before (never column 0)
    for (int u0 = 0; u0 < U; u0 += 64) {
        const int u = u0 + lane;
        const bool v = u < U && u >= 1 && m[u] > 0;
        const unsigned long long b = __ballot(v);
        const unsigned long long below = b & ((1ull << lane) - 1ull);
        if (u < U)
            L[u] = below ? u0 + 63 - __clzll((long long)below) : carry;   // nearest valid column in [1, u-1]
        if (b)
            carry = u0 + 63 - __clzll((long long)b);
    }
    carry = -1;                                   // nearest valid column of the chunks after
    for (int u0 = ((U - 1) / 64) * 64; u0 >= 0; u0 -= 64) {
        const int u = u0 + lane;
        const bool v = u < U && m[u] > 0;
        const unsigned long long b = __ballot(v);
        const unsigned long long above = lane == 63 ? 0ull : (b >> (lane + 1));
        if (u < U)
            R[u] = above ? u + 1 + __ffsll((long long)above) - 1 : carry;   // nearest valid column in [u+1, U-1]
        if (b)
            carry = u0 + __ffsll((long long)b) - 1;
    }
}

// Pass 2, one thread per coarser-level pixel.
__global__ __launch_bounds__(256) void k5_tighten(const float* __restrict__ depth_up, const int* __restrict__ left,
                                                 const int* __restrict__ right, int S, int V_up, int U_up,
                                                 float* __restrict__ dmin_down, float* __restrict__ dmax_down, int V_down, int U_down)
{
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    const int v = blockIdx.y, s = blockIdx.z;
    if (u >= U_down)
        return;
    const int u_up = min(2 * u, U_up - 1);
    int v_up = min(2 * v, V_up - 1);
    float lo = 0.0f, hi = 0.0f;
    int nc = 0;
    for (int line = 0; line < 2; line++) {
        if (line == 1) {
            if (v_up + 1 < V_up)
                v_up += 1;
            else
                break;
        }
        const long long ro = ((long long)s * V_up + v_up) * U_up;
        const int l = left[ro + u_up], r = right[ro + u_up];
        if (l >= 0 && r >= 0) {
            const float dl = depth_up[ro + l], dr = depth_up[ro + r];
            if (nc == 0) {
                lo = fminf(dl, dr);
                hi = fmaxf(dl, dr);
            } else {
                lo = fminf(lo, fminf(dl, dr));
                hi = fmaxf(hi, fmaxf(dl, dr));
            }
            nc += 2;
        }
    }
    if (nc > 1) {
        const long long o = ((long long)s * V_down + v) * U_down + u;
        dmin_down[o] = lo;
        dmax_down[o] = hi;
    }
}

// ---- fusion (fine_to_coarse_core.cpp:93-131) ------------------------------------------------------
// One coarse-to-fine step for all views at once: upscale the running map (cv::resize INTER_LINEAR) and
// mask (INTER_NEAREST) from level p to level p-1 and fill level p-1's invalid pixels.
__global__ __launch_bounds__(256) void k5_fuse_step(const float* __restrict__ map_down, const uint8_t* __restrict__ mask_down, int R,
                                                   int W, const float* __restrict__ disp_fine, const uint8_t* __restrict__ valid_fine,
                                                   float* __restrict__ map_out, uint8_t* __restrict__ mask_out, int R2, int W2)
{
    const int dx = blockIdx.x * blockDim.x + threadIdx.x;
    const int dy = blockIdx.y, s = blockIdx.z;
    if (dx >= W2)
        return;
    const long long o = ((long long)s * R2 + dy) * W2 + dx;
    const uint8_t vf = valid_fine[o];
    // INTER_NEAREST: sx = min(floor(dx * (1 / (W2 / W))), W - 1)
    const double ifx = 1.0 / ((double)W2 / W), ify = 1.0 / ((double)R2 / R);
    const int nx = min((int)floor(dx * ifx), W - 1), ny = min((int)floor(dy * ify), R - 1);
    mask_out[o] = vf | mask_down[((long long)s * R + ny) * W + nx];
    if (vf) {
        map_out[o] = disp_fine[o];
        return;
    }
    // INTER_LINEAR (HResizeLinear then VResizeLinear, float weights from double coordinates)
    float fx = (float)((dx + 0.5) * ifx - 0.5);
    int sx = (int)floorf(fx);
    fx -= (float)sx;
    if (sx < 0) {
        fx = 0.0f;
        sx = 0;
    }
    const bool flat = sx + 1 >= W;
    if (sx >= W - 1) {
        fx = 0.0f;
        sx = W - 1;
    }
    float fy = (float)((dy + 0.5) * ify - 0.5);
    const int sy = (int)floorf(fy);
    fy -= (float)sy;
    const int y0 = min(max(sy, 0), R - 1), y1 = min(max(sy + 1, 0), R - 1);
    const float* S0 = map_down + ((long long)s * R + y0) * W;
    const float* S1 = map_down + ((long long)s * R + y1) * W;
    float r0, r1;
    if (!flat) {
        const float a0 = 1.0f - fx;
        const float p0 = S0[sx] * a0, p1 = S0[sx + 1] * fx;
        r0 = p0 + p1;
        const float q0 = S1[sx] * a0, q1 = S1[sx + 1] * fx;
        r1 = q0 + q1;
    } else {
        r0 = S0[sx] * 1.0f;
        r1 = S1[sx] * 1.0f;
    }
    const float b0 = 1.0f - fy;
    const float t0 = r0 * b0, t1 = r1 * fy;
    const float up = t0 + t1;
    map_out[o] = 0.0f + up;   // setTo(0, invalid) then add(..., invalid)
}

// cv::medianBlur(3) on float planes [S][R][W], BORDER_REPLICATE.
__global__ __launch_bounds__(256) void k5_median3(const float* __restrict__ src, float* __restrict__ dst, int R, int W)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y, s = blockIdx.z;
    if (x >= W)
        return;
    const float* p = src + (long long)s * R * W;
    float w[9];
    int n = 0;
#pragma unroll
    for (int dy = -1; dy <= 1; dy++)
#pragma unroll
        for (int dx = -1; dx <= 1; dx++) {
            const int yy = min(max(y + dy, 0), R - 1), xx = min(max(x + dx, 0), W - 1);
            w[n++] = p[(long long)yy * W + xx];
        }
    // median of 9 by a fixed exchange network (Devillard / Smith), min/max only: exact
#define RSLF_SORT2(a, b) { const float lo_ = fminf(w[a], w[b]); const float hi_ = fmaxf(w[a], w[b]); w[a] = lo_; w[b] = hi_; }
    RSLF_SORT2(1, 2) RSLF_SORT2(4, 5) RSLF_SORT2(7, 8) RSLF_SORT2(0, 1) RSLF_SORT2(3, 4) RSLF_SORT2(6, 7)
    RSLF_SORT2(1, 2) RSLF_SORT2(4, 5) RSLF_SORT2(7, 8) RSLF_SORT2(0, 3) RSLF_SORT2(5, 8) RSLF_SORT2(4, 7)
    RSLF_SORT2(3, 6) RSLF_SORT2(1, 4) RSLF_SORT2(2, 5) RSLF_SORT2(4, 7) RSLF_SORT2(4, 2) RSLF_SORT2(6, 4)
    RSLF_SORT2(4, 2)
#undef RSLF_SORT2
    dst[((long long)s * R + y) * W + x] = w[4];
}

}  // namespace rslf
