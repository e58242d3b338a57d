// Packed-fp32 check for the one-wave-per-SIMD scan variants.
//   1. semantics: v_pk_add_f32 (with neg / clamp) and v_pk_mul_f32 against the scalar instructions, bit for bit,
//      on values the scan meets: ordinary radiances, the 1e30 sentinel (q = +inf, K = 0), denormal products.
//   2. rate: the 1-channel mean-shift pass over 4 samples as 28 scalar instructions vs 10 packed + 8 scalar,
//      at 1, 2 and 3 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o ubench_pk tools/ubench_pk.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f2 pk_one_minus_clamp(f2 q)
{
    f2 k;
    const f2 one = {1.0f, 1.0f};
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1] clamp" : "=v"(k) : "v"(one), "v"(q));
    return k;
}
__device__ __forceinline__ float one_minus_clamp(float q)
{
    float k;
    asm("v_sub_f32_e64 %0, 1.0, %1 clamp" : "=v"(k) : "v"(q));
    return k;
}

__global__ void semantics(const float* r, const float* rbar, float kq, float* out_scalar, float* out_packed, int n)
{
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (i + 1 >= n)
        return;
    const float rb = rbar[i];
    // scalar chain
    for (int j = 0; j < 2; j++) {
        const float d = r[i + j] - rb;
        const float t = kq * d;
        const float q = t * d;
        const float K = one_minus_clamp(q);
        const float P = r[i + j] * K;
        out_scalar[(i + j) * 4 + 0] = d;
        out_scalar[(i + j) * 4 + 1] = q;
        out_scalar[(i + j) * 4 + 2] = K;
        out_scalar[(i + j) * 4 + 3] = P;
    }
    // packed chain
    const f2 R = {r[i], r[i + 1]};
    const f2 RB = {rb, rb};
    const f2 KK = {kq, kq};
    const f2 d = R - RB;
    const f2 t = KK * d;
    const f2 q = t * d;
    const f2 K = pk_one_minus_clamp(q);
    const f2 P = R * K;
    out_packed[i * 4 + 0] = d.x; out_packed[i * 4 + 1] = q.x; out_packed[i * 4 + 2] = K.x; out_packed[i * 4 + 3] = P.x;
    out_packed[i * 4 + 4] = d.y; out_packed[i * 4 + 5] = q.y; out_packed[i * 4 + 6] = K.y; out_packed[i * 4 + 7] = P.y;
}

template <int MODE>
__global__ __launch_bounds__(256) void rate(float* out, int iters, float seed)
{
    float r0 = seed + 0.001f * threadIdx.x, r1 = r0 + 0.01f, r2 = r0 + 0.02f, r3 = r0 + 0.03f;
    float A = 0.0f, B = 0.0f, rbar = r0 + 0.015f;
    const float kq = 75.0f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int rep = 0; rep < 32; rep++) {
            if (MODE == 0) {
                float t0, t1, t2, t3, u0, u1, u2, u3;
                asm volatile("v_sub_f32 %2, %10, %14\n v_sub_f32 %3, %11, %14\n v_sub_f32 %4, %12, %14\n v_sub_f32 %5, %13, %14\n"
                             "v_mul_f32 %6, %15, %2\n v_mul_f32 %7, %15, %3\n v_mul_f32 %8, %15, %4\n v_mul_f32 %9, %15, %5\n"
                             "v_mul_f32 %2, %2, %6\n v_mul_f32 %3, %3, %7\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %9\n"
                             "v_sub_f32_e64 %2, 1.0, %2 clamp\n v_sub_f32_e64 %3, 1.0, %3 clamp\n v_sub_f32_e64 %4, 1.0, %4 clamp\n v_sub_f32_e64 %5, 1.0, %5 clamp\n"
                             "v_mul_f32 %6, %10, %2\n v_mul_f32 %7, %11, %3\n v_mul_f32 %8, %12, %4\n v_mul_f32 %9, %13, %5\n"
                             "v_add_f32 %0, %0, %6\n v_add_f32 %1, %1, %2\n v_add_f32 %0, %0, %7\n v_add_f32 %1, %1, %3\n"
                             "v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %4\n v_add_f32 %0, %0, %9\n v_add_f32 %1, %1, %5"
                             : "+v"(A), "+v"(B), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(u0), "=&v"(u1), "=&v"(u2), "=&v"(u3)
                             : "v"(r0), "v"(r1), "v"(r2), "v"(r3), "v"(rbar), "s"(kq));
            } else {
                f2 R01 = {r0, r1}, R23 = {r2, r3};
                const f2 RB = {rbar, rbar}, KK = {kq, kq}, one = {1.0f, 1.0f};
                f2 d01, d23, t01, t23;
                asm volatile("v_pk_add_f32 %0, %4, %6 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %1, %5, %6 neg_lo:[0,1] neg_hi:[0,1]\n"
                             "v_pk_mul_f32 %2, %7, %0\n v_pk_mul_f32 %3, %7, %1\n"
                             "v_pk_mul_f32 %0, %0, %2\n v_pk_mul_f32 %1, %1, %3\n"
                             "v_pk_add_f32 %0, %8, %0 neg_lo:[0,1] neg_hi:[0,1] clamp\n v_pk_add_f32 %1, %8, %1 neg_lo:[0,1] neg_hi:[0,1] clamp\n"
                             "v_pk_mul_f32 %2, %4, %0\n v_pk_mul_f32 %3, %5, %1"
                             : "=&v"(d01), "=&v"(d23), "=&v"(t01), "=&v"(t23)
                             : "v"(R01), "v"(R23), "v"(RB), "v"(KK), "v"(one));
                asm volatile("v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %3" : "+v"(A), "+v"(B) : "v"(t01.x), "v"(d01.x));
                asm volatile("v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %3" : "+v"(A), "+v"(B) : "v"(t01.y), "v"(d01.y));
                asm volatile("v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %3" : "+v"(A), "+v"(B) : "v"(t23.x), "v"(d23.x));
                asm volatile("v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %3" : "+v"(A), "+v"(B) : "v"(t23.y), "v"(d23.y));
            }
        }
        rbar = (B != 0.0f) ? A / B : rbar;
        A = 0.0f;
        B = 0.0f;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = rbar;
}

template <int MODE>
static void run_rate(const char* name, int waves_per_simd, float* d_out)
{
    const int iters = 400;
    const int blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(rate<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, 10, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double groups = (double)blocks * 4 * iters * 32;   // 4-sample groups per wave, summed over waves
    const double per_simd = groups / (256.0 * 4) / (ms * 1e-3);
    printf("%-22s waves/SIMD=%d  %7.3f ms  %6.1f clk per 4-sample group per SIMD @2.4GHz\n", name, waves_per_simd, ms, 2.4e9 / per_simd);
}

int main()
{
    // ---- semantics
    std::vector<float> r, rb;
    const float specials[] = {0.0f, 0.2f, 0.5f, 0.99999994f, 1.0f, 1e30f, 1e-30f, 3e-20f, 1e-10f, 0.3333333f, 123456.0f, 1e6f};
    for (float a : specials)
        for (float b : specials) {
            if (b > 1e7f)
                continue;   // rbar is never the sentinel
            r.push_back(a); r.push_back(a * 0.75f + 1e-3f);
            rb.push_back(b); rb.push_back(b);
        }
    unsigned x = 12345;
    for (int i = 0; i < 200000; i++) {
        x = x * 1664525u + 1013904223u;
        const float a = (x >> 8) * (1.0f / 16777216.0f);
        x = x * 1664525u + 1013904223u;
        const float b = (x >> 8) * (1.0f / 16777216.0f);
        r.push_back(a); r.push_back(b);
        x = x * 1664525u + 1013904223u;
        const float c = (x >> 8) * (1.0f / 16777216.0f);
        rb.push_back(c); rb.push_back(c);
    }
    const int n = (int)r.size();
    float *d_r, *d_rb, *d_s, *d_p;
    hipMalloc(&d_r, n * 4); hipMalloc(&d_rb, n * 4); hipMalloc(&d_s, n * 16); hipMalloc(&d_p, n * 16);
    hipMemcpy(d_r, r.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_rb, rb.data(), n * 4, hipMemcpyHostToDevice);
    hipMemset(d_s, 0, n * 16); hipMemset(d_p, 0, n * 16);
    hipLaunchKernelGGL(semantics, dim3((n / 2 + 255) / 256), dim3(256), 0, 0, d_r, d_rb, 75.0f, d_s, d_p, n);
    std::vector<float> hs(n * 4), hp(n * 4);
    hipMemcpy(hs.data(), d_s, n * 16, hipMemcpyDeviceToHost);
    hipMemcpy(hp.data(), d_p, n * 16, hipMemcpyDeviceToHost);
    long bad = 0;
    for (long i = 0; i < (long)n * 4; i++)
        if (memcmp(&hs[i], &hp[i], 4) != 0) {
            if (bad < 10)
                printf("MISMATCH elem %ld field %ld: scalar %a packed %a (r=%a rbar=%a)\n", i / 4, i % 4, hs[i], hp[i], r[i / 4], rb[i / 4]);
            bad++;
        }
    printf("semantics: %d values x 4 fields, %ld mismatches\n", n, bad);

    // ---- rate
    float* d_out;
    hipMalloc(&d_out, sizeof(float) * 256 * 256 * 4);
    for (int w : {1, 2, 3}) {
        run_rate<0>("scalar group4 (28)", w, d_out);
        run_rate<1>("packed group4 (10+8)", w, d_out);
    }
    return bad ? 1 : 0;
}
