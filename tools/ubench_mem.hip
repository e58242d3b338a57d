// Micro-benchmark: what does ONE re-gathered sample of the streaming scan kernel cost the CU's vector-memory
// path, by access shape?  The streaming kernel (k2_scan_stream) is bound by its gathers, not by VALU
// (profiles/r02_c5_stream_pmc.json), so the shape of the load decides its speed.
//
// Access pattern = the real one: row s of an EPI (rows `rowb` bytes apart, pixels 12 bytes: RGB interleaved),
// a wave reads the 64 (+1) consecutive pixels starting at u0 + floor((s_hat - s) * d), for every view s, ten
// passes per hypothesis d.  Four waves per workgroup = four hypothesis ranges of one tile, as in the kernel.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_mem tools/ubench_mem.hip && tools/ubench_mem
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define LDS_PTR(p) ((void __attribute__((address_space(3)))*)(p))
#define GLB_PTR(p) ((const void __attribute__((address_space(1)))*)(p))

constexpr int S = 201, SHAT = 100, PITCH = 4160, C = 3, NPASS = 10;
typedef float f3 __attribute__((ext_vector_type(3)));
constexpr int G = 4;   // samples in flight per wave

enum Mode {
    GATHER_X4_X2 = 0,   // per lane: 16 B + 8 B at its own pixel (what the kernel does now)
    COAL_X3,            // per lane: its pixel only, 12 B (right tap would come from lane + 1)
    COAL_X3_MASKED,     // COAL_X3 + a second 12-B load with only lane 63 active
    DMA_X4,             // wave-wide contiguous 16 B per lane straight into LDS (50 lanes = 66 pixels), no read-back
    DMA_X4_READ2,       // DMA_X4 + both taps read back per lane with 3 ds_read2_b32
    DMA_X4_READ64,      // DMA_X4 + both taps read back per lane with 3 ds_read_b64 (4-byte aligned addresses)
    COAL_X4,            // wave-wide contiguous 16 B per lane into registers
    GATHER_X2,          // per lane: 8 B only
    GATHER_X4,          // per lane: 16 B only
    NMODES
};
static const char* kNames[NMODES] = {"gather x4+x2 (now)", "coalesced x3", "coalesced x3 + 1-lane x3", "LDS-DMA x4, no read",
                                     "LDS-DMA x4 + 3 ds_read2_b32", "LDS-DMA x4 + 3 ds_read_b64", "coalesced x4 -> regs",
                                     "gather x2 only", "gather x4 only"};

template <int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ vol, float* __restrict__ out, int nV, int hyps, float dstep)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* stage = lds + wave * (2 * G * 256);   // 2 x G slots of 1 KiB per wave
    const int tile = blockIdx.x % 40, v = (blockIdx.x / 40) % nV;
    const int u0 = 700 + tile * 64;
    const char* epi = (const char*)(vol + (size_t)v * S * PITCH * C);
    const unsigned rowb = PITCH * C * 4;
    float acc = 0.0f;
    for (int h = 0; h < hyps; h++) {
        const float d = -2.0f + dstep * (float)(wave * hyps + h);
        for (int pass = 0; pass < NPASS; pass++) {
            for (int s0 = 0; s0 < S - G + 1; s0 += G) {
                float e[G][6];
                int fo[G];
#pragma unroll
                for (int j = 0; j < G; j++)
                    fo[j] = __builtin_amdgcn_readfirstlane((int)floorf((float)(SHAT - (s0 + j)) * d));
                if (MODE == GATHER_X4_X2 || MODE == GATHER_X2 || MODE == GATHER_X4) {
#pragma unroll
                    for (int j = 0; j < G; j++) {
                        const float* p = (const float*)(epi + (unsigned)(s0 + j) * rowb + (unsigned)(u0 + lane + fo[j]) * 12u);
                        if (MODE != GATHER_X2) {
                            const float4 a = *(const float4*)p;
                            e[j][0] = a.x, e[j][1] = a.y, e[j][2] = a.z, e[j][3] = a.w;
                        } else {
                            e[j][0] = e[j][1] = e[j][2] = e[j][3] = 0.0f;
                        }
                        if (MODE != GATHER_X4) {
                            const float2 b = *(const float2*)(p + 4);
                            e[j][4] = b.x, e[j][5] = b.y;
                        } else {
                            e[j][4] = e[j][5] = 0.0f;
                        }
                    }
                } else if (MODE == COAL_X3 || MODE == COAL_X3_MASKED) {
#pragma unroll
                    for (int j = 0; j < G; j++) {
                        const float* p = (const float*)(epi + (unsigned)(s0 + j) * rowb + (unsigned)(u0 + lane + fo[j]) * 12u);
                        const f3 a = *(const f3*)p;
                        e[j][0] = a.x, e[j][1] = a.y, e[j][2] = a.z;
                        e[j][3] = e[j][4] = e[j][5] = 0.0f;
                        if (MODE == COAL_X3_MASKED && lane == 63) {
                            const f3 b = *(const f3*)(p + 3);
                            e[j][3] = b.x, e[j][4] = b.y, e[j][5] = b.z;
                        }
                    }
                } else if (MODE == COAL_X4) {
#pragma unroll
                    for (int j = 0; j < G; j++) {
                        const float* p = (const float*)(epi + (unsigned)(s0 + j) * rowb + (unsigned)(u0 + fo[j]) * 12u + lane * 16);
                        const float4 a = *(const float4*)p;
                        e[j][0] = a.x, e[j][1] = a.y, e[j][2] = a.z, e[j][3] = a.w;
                        e[j][4] = e[j][5] = 0.0f;
                    }
                } else {
                    float* slot = stage + ((s0 / G) & 1) * (G * 256);
#pragma unroll
                    for (int j = 0; j < G; j++) {
                        const char* p = epi + (unsigned)(s0 + j) * rowb + (unsigned)(u0 + fo[j]) * 12u + lane * 16;
                        if (lane < 50)
                            __builtin_amdgcn_global_load_lds(GLB_PTR(p), LDS_PTR(slot + j * 256), 16, 0, 0);
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                    for (int j = 0; j < G; j++) {
                        const float* q = slot + j * 256 + lane * 3;
                        if (MODE == DMA_X4_READ2) {
#pragma unroll
                            for (int i = 0; i < 6; i++)
                                e[j][i] = q[i];
                        } else if (MODE == DMA_X4_READ64) {
                            float2 a, b, c;
                            asm volatile("ds_read_b64 %0, %3\n\tds_read_b64 %1, %3 offset:8\n\tds_read_b64 %2, %3 offset:16\n\ts_waitcnt lgkmcnt(0)"
                                         : "=v"(a), "=v"(b), "=v"(c) : "v"((unsigned)(size_t)LDS_PTR(q)) : "memory");
                            e[j][0] = a.x, e[j][1] = a.y, e[j][2] = b.x, e[j][3] = b.y, e[j][4] = c.x, e[j][5] = c.y;
                        } else {
#pragma unroll
                            for (int i = 0; i < 6; i++)
                                e[j][i] = 0.0f;
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < G; j++)
#pragma unroll
                    for (int i = 0; i < 6; i++)
                        acc += e[j][i];
            }
        }
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
}

// correctness of the LDS-DMA forms at 4-byte-aligned global addresses and of 4-byte-aligned ds_read_b64
__global__ void check(const float* __restrict__ g, int* bad)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    int nb = 0;
    for (int shift = 0; shift < 8; shift++) {
        const float* src = g + 3 * shift + 1;   // 4-byte aligned, every residue mod 16
        if (lane < 50)
            __builtin_amdgcn_global_load_lds(GLB_PTR((const char*)src + lane * 16), LDS_PTR(lds), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        float2 a, b, c;
        asm volatile("ds_read_b64 %0, %3\n\tds_read_b64 %1, %3 offset:8\n\tds_read_b64 %2, %3 offset:16\n\ts_waitcnt lgkmcnt(0)"
                     : "=v"(a), "=v"(b), "=v"(c) : "v"((unsigned)(size_t)LDS_PTR(lds + lane * 3)) : "memory");
        const float got[6] = {a.x, a.y, b.x, b.y, c.x, c.y};
        for (int i = 0; i < 6; i++) {
            if (got[i] != src[lane * 3 + i])
                nb++;
            if (lds[lane * 3 + i] != src[lane * 3 + i])
                nb += 1000;
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (nb)
        atomicAdd(bad, nb);
}

template <int MODE>
void run(const float* d_vol, float* d_out, int nV, int blocks_per_cu)
{
    const int hyps = 8;
    const int blocks = 256 * blocks_per_cu;
    const size_t lds = 4 * 2 * G * 1024;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), lds, 0, d_vol, d_out, nV, 1, 1.0f / 64);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), lds, 0, d_vol, d_out, nV, hyps, 1.0f / 64);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double wave_samples_per_cu = (double)blocks / 256.0 * 4 * hyps * NPASS * (S / G * G);
    printf("%-30s blocks/CU=%d  %8.3f ms  %7.1f clk of CU time per (wave, sample) @2.4GHz\n", kNames[MODE], blocks_per_cu, ms,
           ms * 1e-3 * 2.4e9 / wave_samples_per_cu);
    fflush(stdout);
}

int main()
{
    const int nV = 24;   // 24 EPIs x 10 MB
    const size_t n = (size_t)nV * S * PITCH * C;
    std::vector<float> h(n);
    for (size_t i = 0; i < n; i++)
        h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) * (1.0f / 65536);
    float *d_vol, *d_out;
    int* d_bad;
    hipMalloc(&d_vol, n * 4);
    hipMalloc(&d_out, sizeof(float) * 256 * 256 * 8);
    hipMalloc(&d_bad, 4);
    hipMemcpy(d_vol, h.data(), n * 4, hipMemcpyHostToDevice);
    hipMemset(d_bad, 0, 4);
    hipLaunchKernelGGL(check, dim3(1), dim3(64), 4096, 0, d_vol, d_bad);
    int bad = -1;
    hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost);
    printf("LDS-DMA at 4-byte-aligned source + 4-byte-aligned ds_read_b64: %s (%d)\n", bad == 0 ? "values correct" : "MISMATCH", bad);
    for (int b : {1, 2}) {
        run<GATHER_X4_X2>(d_vol, d_out, nV, b);
        run<GATHER_X2>(d_vol, d_out, nV, b);
        run<GATHER_X4>(d_vol, d_out, nV, b);
        run<COAL_X3>(d_vol, d_out, nV, b);
        run<COAL_X3_MASKED>(d_vol, d_out, nV, b);
        run<COAL_X4>(d_vol, d_out, nV, b);
        run<DMA_X4>(d_vol, d_out, nV, b);
        run<DMA_X4_READ2>(d_vol, d_out, nV, b);
        run<DMA_X4_READ64>(d_vol, d_out, nV, b);
    }
    return 0;
}
