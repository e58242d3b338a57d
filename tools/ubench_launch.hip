// Developer microbenchmark: what makes the ~5.6 us gap between two dependent kernels of one stream?
// A chain of 200 launches of a kernel that spins for 20 us (constant 100 MHz clock); the chain's time per launch
// minus 20 us is gap + launch cost.  Variants of the kernel: plain / dynamic LDS / static LDS / big by-value struct /
// many blocks that exit at once / alternating two different kernels.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_launch tools/ubench_launch.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

struct Big { float f[60]; int i[20]; };

__device__ __forceinline__ void spin(int ticks)
{
    const unsigned long long t0 = wall_clock64();
    while ((long long)(wall_clock64() - t0) < ticks) { }
}

__global__ void k_plain(int ticks, int* out) { if (blockIdx.x < 256) spin(ticks); if (out && threadIdx.x == 999) out[0] = 1; }
__global__ void k_plain2(int ticks, int* out) { if (blockIdx.x < 256) spin(ticks); if (out && threadIdx.x == 998) out[1] = 1; }
__global__ void k_dyn(int ticks, int* out) { extern __shared__ float sh[]; if (blockIdx.x < 256) spin(ticks); if (out && threadIdx.x == 999) out[0] = (int)sh[threadIdx.x]; }
__global__ void k_static(int ticks, int* out) { __shared__ float sh[1536]; sh[threadIdx.x] = ticks; __syncthreads(); if (blockIdx.x < 256) spin(ticks); if (out && threadIdx.x == 999) out[0] = (int)sh[threadIdx.x ^ 1]; }
__global__ void k_big(Big b, int ticks, int* out) { if (blockIdx.x < 256) spin(ticks); if (out && threadIdx.x == 999) out[0] = (int)b.f[ticks & 31] + b.i[3]; }
__global__ void k_write(int ticks, int* out, int n) { if (blockIdx.x < 256) spin(ticks); for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = i; }

// ~150 KB of code that never runs (ticks is never negative)
__global__ void k_fat(int ticks, int* out, float x)
{
    if (blockIdx.x < 256) spin(ticks);
    if (ticks < 0) {
        float a = x, b = x * 2.0f;
#pragma unroll
        for (int i = 0; i < 6000; i++) { a = a * b + (float)i; b = b * a - 1.5f; a = a - b * 0.25f; }
        out[threadIdx.x] = (int)(a + b);
    }
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_wpe(int ticks, int* out) { if (blockIdx.x < 256) spin(ticks); if (out && threadIdx.x == 999) out[0] = 1; }

template <typename F>
static double chain(F launch, int n, hipStream_t st)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 20; i++) launch(i);
    hipStreamSynchronize(st);
    hipEventRecord(a, st);
    for (int i = 0; i < n; i++) launch(i);
    hipEventRecord(b, st);
    hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return ms * 1000.0 / n;
}

int main()
{
    hipStream_t st; hipStreamCreate(&st);
    int* out; hipMalloc(&out, 64 << 20);
    const int T = 2000, N = 200;   // 20 us
    Big big = {};
    printf("per launch (us), 20 us of spinning included\n");
    printf("plain, 256 blocks            %7.2f\n", chain([&](int) { hipLaunchKernelGGL(k_plain, dim3(256), dim3(256), 0, st, T, out); }, N, st));
    printf("plain, 1024 blocks           %7.2f\n", chain([&](int) { hipLaunchKernelGGL(k_plain, dim3(1024), dim3(256), 0, st, T, out); }, N, st));
    printf("plain, 16896 blocks          %7.2f\n", chain([&](int) { hipLaunchKernelGGL(k_plain, dim3(16896), dim3(256), 0, st, T, out); }, N, st));
    printf("dynamic LDS 25 KB            %7.2f\n", chain([&](int) { hipLaunchKernelGGL(k_dyn, dim3(256), dim3(256), 25600, st, T, out); }, N, st));
    printf("static LDS 6 KB              %7.2f\n", chain([&](int) { hipLaunchKernelGGL(k_static, dim3(256), dim3(256), 0, st, T, out); }, N, st));
    printf("320-byte struct argument     %7.2f\n", chain([&](int) { hipLaunchKernelGGL(k_big, dim3(256), dim3(256), 0, st, big, T, out); }, N, st));
    printf("two kernels alternating      %7.2f\n", chain([&](int i) { if (i & 1) hipLaunchKernelGGL(k_plain, dim3(256), dim3(256), 0, st, T, out); else hipLaunchKernelGGL(k_plain2, dim3(256), dim3(256), 0, st, T, out); }, N, st));
    printf("plain <-> dynamic LDS        %7.2f\n", chain([&](int i) { if (i & 1) hipLaunchKernelGGL(k_plain, dim3(256), dim3(256), 0, st, T, out); else hipLaunchKernelGGL(k_dyn, dim3(256), dim3(256), 25600, st, T, out); }, N, st));
    printf("plain <-> static LDS         %7.2f\n", chain([&](int i) { if (i & 1) hipLaunchKernelGGL(k_plain, dim3(256), dim3(256), 0, st, T, out); else hipLaunchKernelGGL(k_static, dim3(256), dim3(256), 0, st, T, out); }, N, st));
    printf("writes 32 MB then plain      %7.2f\n", chain([&](int i) { if (i & 1) hipLaunchKernelGGL(k_plain, dim3(256), dim3(256), 0, st, T, out); else hipLaunchKernelGGL(k_write, dim3(1024), dim3(256), 0, st, T, out, 8 << 20); }, N, st));
    printf("plain <-> fat code           %7.2f\n", chain([&](int i) { if (i & 1) hipLaunchKernelGGL(k_plain, dim3(256), dim3(256), 0, st, T, out); else hipLaunchKernelGGL(k_fat, dim3(256), dim3(256), 0, st, T, out, 1.0f); }, N, st));
    printf("fat code                     %7.2f\n", chain([&](int i) { hipLaunchKernelGGL(k_fat, dim3(256), dim3(256), 0, st, T, out, 1.0f); }, N, st));
    printf("plain <-> waves_per_eu(2,2)  %7.2f\n", chain([&](int i) { if (i & 1) hipLaunchKernelGGL(k_plain, dim3(256), dim3(256), 0, st, T, out); else hipLaunchKernelGGL(k_wpe, dim3(256), dim3(256), 0, st, T, out); }, N, st));
    printf("plain 0 us (launch rate)     %7.2f\n", chain([&](int) { hipLaunchKernelGGL(k_plain, dim3(256), dim3(256), 0, st, 0, out); }, N, st));
    printf("dynamic LDS 0 us             %7.2f\n", chain([&](int) { hipLaunchKernelGGL(k_dyn, dim3(256), dim3(256), 25600, st, 0, out); }, N, st));
    return 0;
}
