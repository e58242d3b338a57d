// Micro-benchmark: issue rate of the fp32 VALU instructions the scan kernel is made of.
// Answers one design question: do v_pk_add_f32 / v_pk_mul_f32 (two floats per lane per
// instruction) issue at the rate of v_add_f32 / v_mul_f32 on gfx950?  If yes, a lane should
// carry two hypotheses.   hipcc --offload-arch=gfx950 -O3 -o ubench_valu tools/ubench_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

#define REP8(X) X X X X X X X X
#define REP64(X) REP8(REP8(X))

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed)
{
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    f2 p4 = p0 + 1.0f, p5 = p1 + 1.0f, p6 = p2 + 1.0f, p7 = p3 + 1.0f;
    const float c = 1.0000001f;
    const f2 c2 = {c, c};
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {   // 8 independent v_mul_f32 chains x 64
            REP64(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                               "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)
        } else if (MODE == 1) {   // 8 independent v_pk_mul_f32 chains x 64
            REP64(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                               "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c2));)
        } else if (MODE == 2) {   // v_add_f32
            REP64(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                               "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)
        } else if (MODE == 3) {   // v_pk_add_f32
            REP64(asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                               "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c2));)
        } else if (MODE == 4) {   // v_fma_f32
            REP64(asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                               "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)
        } else if (MODE == 5) {   // v_pk_fma_f32
            REP64(asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n"
                               "v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8\n"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c2));)
        } else if (MODE == 6) {   // v_max_f32
            REP64(asm volatile("v_max_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n"
                               "v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_max_f32 %7, %7, %8\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)
        } else if (MODE == 7) {   // v_sub_f32 with clamp (VOP3)
            REP64(asm volatile("v_sub_f32_e64 %0, %8, %0 clamp\n v_sub_f32_e64 %1, %8, %1 clamp\n v_sub_f32_e64 %2, %8, %2 clamp\n v_sub_f32_e64 %3, %8, %3 clamp\n"
                               "v_sub_f32_e64 %4, %8, %4 clamp\n v_sub_f32_e64 %5, %8, %5 clamp\n v_sub_f32_e64 %6, %8, %6 clamp\n v_sub_f32_e64 %7, %8, %7 clamp\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)
        }
    }
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p5.y + p6.x + p7.y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE>
double run(const char* name, int waves_per_simd, float* d_out, int flops_per_lane_instr)
{
    const int iters = 200;
    const int blocks = 256 * waves_per_simd;   // 256 CUs x (4 waves/block = 1 wave per SIMD) x waves_per_simd
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, 10, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double wave_instrs = (double)blocks * 4 * iters * 64 * 8;
    const double per_simd_per_s = wave_instrs / (256.0 * 4) / (ms * 1e-3);
    const double lane_ops = wave_instrs * 64 * flops_per_lane_instr;
    printf("%-14s waves/SIMD=%d  %8.3f ms  %6.3f G wave-instr/s/SIMD  (%.2f clk/instr @2.4GHz)  %7.2f Tflop/s\n", name,
           waves_per_simd, ms, per_simd_per_s * 1e-9, 2.4e9 / per_simd_per_s, lane_ops / (ms * 1e-3) * 1e-12);
    return ms;
}

int main()
{
    float* d_out;
    hipMalloc(&d_out, sizeof(float) * 256 * 256 * 8);
    for (int w : {1, 2, 4}) {
        run<0>("v_mul_f32", w, d_out, 1);
        run<1>("v_pk_mul_f32", w, d_out, 2);
        run<2>("v_add_f32", w, d_out, 1);
        run<3>("v_pk_add_f32", w, d_out, 2);
        run<4>("v_fma_f32", w, d_out, 2);
        run<5>("v_pk_fma_f32", w, d_out, 4);
        run<6>("v_max_f32", w, d_out, 1);
        run<7>("v_sub clamp", w, d_out, 1);
    }
    return 0;
}
