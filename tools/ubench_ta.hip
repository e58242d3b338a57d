// Micro-benchmark: THROUGHPUT of the CU's vector-memory path per access shape (8 waves per SIMD, 8 loads in flight
// per wave, so latency is covered), for rows served by the L1 (a few KiB per CU) and by the L2 (1 MiB per CU's
// share).  Unit: clocks of CU time per wave-instruction (2.4 GHz assumed; the ratio between shapes is the point).
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_ta tools/ubench_ta.hip && tools/ubench_ta
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define LDS_PTR(p) ((void __attribute__((address_space(3)))*)(p))
#define GLB_PTR(p) ((const void __attribute__((address_space(1)))*)(p))
typedef float f3 __attribute__((ext_vector_type(3)));
enum Mode { G_X4, G_X2, G_X4X2, C_X3, C_X4, C_X1, DMA_X4, DMA_X1, PX_X4X2, PX_X3X3, PX_X1X6, NMODES };
static const char* kNames[NMODES] = {"gather x4 (12-B stride)", "gather x2 (12-B stride)", "gather x4 + x2", "coalesced x3",
                                     "coalesced x4", "coalesced x1", "LDS-DMA x4 coalesced", "LDS-DMA x1 coalesced",
                                     // round 4: the pixel-per-wave kernels' pattern -- lanes = hypotheses of ONE pixel, lane l reads the
                                     // texel pair at floor(l * f), f = |s_hat - s| * (hypothesis step) in [0, 1.7]: non-decreasing, irregular
                                     "px x4 + x2 (f = j/4)", "px x3 + x3 (f = j/4)", "px 6 x x1 (f = j/4)"};
constexpr int ROWB = 4160 * 12;

template <int MODE>
__global__ __launch_bounds__(256) void k(const char* __restrict__ vol, float* out, int iters, int rows_per_wave, int row_stride)
{
    __shared__ __attribute__((aligned(16))) float lds[4 * 8 * 256];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* stage = lds + wave * 8 * 256;
    const char* base = vol + (size_t)(blockIdx.x & 255) * 64 * ROWB + (size_t)wave * 1536;
    float acc = 0.0f;
    int r = 0;
    for (int it = 0; it < iters; it++) {
        float4 v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const char* row = base + (size_t)r * row_stride;
            r = (r + 1 == rows_per_wave) ? 0 : r + 1;
            v[j] = float4{0, 0, 0, 0};
            if (MODE == G_X4 || MODE == G_X4X2)
                v[j] = *(const float4*)(row + lane * 12);
            if (MODE == G_X2)
                *(float2*)&v[j] = *(const float2*)(row + lane * 12);
            if (MODE == G_X4X2) {
                const float2 b = *(const float2*)(row + lane * 12 + 16);
                v[j].x += b.x, v[j].y += b.y;
            }
            if (MODE == C_X3) {
                const f3 a = *(const f3*)(row + lane * 12);
                v[j].x = a.x, v[j].y = a.y, v[j].z = a.z;
            }
            if (MODE == C_X4)
                v[j] = *(const float4*)(row + lane * 16);
            if (MODE == C_X1)
                v[j].x = *(const float*)(row + lane * 4);
            if (MODE == PX_X4X2 || MODE == PX_X3X3 || MODE == PX_X1X6) {
                const char* p = row + (int)((float)lane * (0.25f * (float)j)) * 12;
                if (MODE == PX_X4X2) {
                    v[j] = *(const float4*)p;
                    const float2 b = *(const float2*)(p + 16);
                    v[j].x += b.x, v[j].y += b.y;
                } else if (MODE == PX_X3X3) {
                    const f3 a = *(const f3*)p, b = *(const f3*)(p + 12);
                    v[j].x = a.x + b.x, v[j].y = a.y + b.y, v[j].z = a.z + b.z;
                } else {
                    const float* q = (const float*)p;
                    v[j].x = q[0] + q[3], v[j].y = q[1] + q[4], v[j].z = q[2] + q[5];
                }
            }
            if (MODE == DMA_X4)
                __builtin_amdgcn_global_load_lds(GLB_PTR(row + lane * 16), LDS_PTR(stage + j * 256), 16, 0, 0);
            if (MODE == DMA_X1)
                __builtin_amdgcn_global_load_lds(GLB_PTR(row + lane * 4), LDS_PTR(stage + j * 256), 4, 0, 0);
        }
        if (MODE == DMA_X4 || MODE == DMA_X1)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < 8; j++)
            acc += v[j].x + v[j].y + v[j].z + v[j].w;
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc + stage[lane];
}

template <int MODE>
void run(const char* d_vol, float* d_out, int rows_per_wave, int row_stride, const char* where)
{
    const int iters = 400, blocks = 256 * 8;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d_vol, d_out, 20, rows_per_wave, row_stride);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d_vol, d_out, iters, rows_per_wave, row_stride);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double groups_per_cu = 8.0 * 4 * iters * 8;   // (wave, load group) pairs per CU
    printf("%-26s %-3s %8.3f ms  %6.1f clk of CU time per wave-level load group\n", kNames[MODE], where, ms, ms * 1e-3 * 2.4e9 / groups_per_cu);
    fflush(stdout);
}

int main()
{
    const size_t n = (size_t)256 * 64 * ROWB + (1 << 20);
    char* d_vol;
    float* d_out;
    (void)hipMalloc(&d_vol, n);
    (void)hipMemset(d_vol, 0, n);
    (void)hipMalloc(&d_out, sizeof(float) * 256 * 8 * 256);
    // L1: 2 rows per wave (each 768-1024 B): 8 blocks x 4 waves share one base per block index -> a few KiB per CU
    // L2: 64 rows per wave at the EPI row stride -> 4 waves x 64 x 1 KiB = 256 KiB per block index, 8 blocks share it
#define ALL(rows, stride, where)                                   \
    run<G_X4>(d_vol, d_out, rows, stride, where);                  \
    run<G_X2>(d_vol, d_out, rows, stride, where);                  \
    run<G_X4X2>(d_vol, d_out, rows, stride, where);                \
    run<C_X3>(d_vol, d_out, rows, stride, where);                  \
    run<C_X4>(d_vol, d_out, rows, stride, where);                  \
    run<C_X1>(d_vol, d_out, rows, stride, where);                  \
    run<DMA_X4>(d_vol, d_out, rows, stride, where);                \
    run<DMA_X1>(d_vol, d_out, rows, stride, where);                \
    run<PX_X4X2>(d_vol, d_out, rows, stride, where);               \
    run<PX_X3X3>(d_vol, d_out, rows, stride, where);               \
    run<PX_X1X6>(d_vol, d_out, rows, stride, where);
    ALL(2, ROWB, "L1")
    ALL(64, ROWB, "L2")
    return 0;
}
