// Sustained v_mfma_f32_32x32x2_f32 / 16x16x4_f32 / 32x32x16_bf16 issue rate on the whole chip: the practical ceiling
// the GEMM roofline fraction should be read against (clocks under matrix load are below the 2.4 GHz peak).
// build: hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_peak.hip -o ab/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ __launch_bounds__(256) void k_f32_32(float* out, int iters, float a0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float a = a0 + threadIdx.x, b = a0 * 2 + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k_f32_16(float* out, int iters, float a0) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 4; ++e) acc[i][e] = 0.f;
    float a = a0 + threadIdx.x, b = a0 * 2 + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 4; ++e) s += acc[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k_bf16_32(float* out, int iters, float a0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(a0 + e); b[e] = (__bf16)(a0 - e); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename F>
static void run(const char* name, F launch, double flops_per_wave_iter, int blocks, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(iters / 10);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0); launch(iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    double fl = flops_per_wave_iter * iters * 4.0 * blocks;
    printf("%-28s blocks=%5d  %8.3f ms  %7.1f TFLOP/s\n", name, blocks, best, fl / best / 1e9);
}

__global__ void k_clock(long long* o, int iters) {        // s_memtime ticks per s_memrealtime tick (100 MHz) under MFMA load
    f32x16 acc; for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    long long m0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int it = 0; it < iters; ++it) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(1.f, 2.f, acc, 0, 0, 0);
    long long m1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { o[0] = m1 - m0; o[1] = r1 - r0; o[2] = (long long)acc[0]; }
}

int main() {
    float* out; hipMalloc(&out, 8192 * 256 * 4);
    {
        long long* d; hipMalloc(&d, 64); long long h[3];
        k_clock<<<1024, 256>>>(d, 200000); hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        printf("s_memtime under full-chip fp32 MFMA load: %.0f MHz (wall_clock64 = 100 MHz)\n", 100.0 * h[0] / h[1]);
        k_clock<<<1, 64>>>(d, 200000); hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        printf("s_memtime with one wave running:           %.0f MHz\n", 100.0 * h[0] / h[1]);
    }
    const int iters = 20000;
    for (int bpc : {1, 2, 4}) {
        int blocks = 256 * bpc;
        run("f32 32x32x2  4 acc", [&](int n) { k_f32_32<4><<<blocks, 256>>>(out, n, 1.f); }, 8 * 4 * 4096.0, blocks, iters);
        run("f32 32x32x2  3 acc", [&](int n) { k_f32_32<3><<<blocks, 256>>>(out, n, 1.f); }, 8 * 3 * 4096.0, blocks, iters);
        run("f32 32x32x2  1 acc", [&](int n) { k_f32_32<1><<<blocks, 256>>>(out, n, 1.f); }, 8 * 1 * 4096.0, blocks, iters);
        run("f32 16x16x4  4 acc", [&](int n) { k_f32_16<4><<<blocks, 256>>>(out, n, 1.f); }, 8 * 4 * 2048.0, blocks, iters);
        run("bf16 32x32x16 4 acc", [&](int n) { k_bf16_32<4><<<blocks, 256>>>(out, n, 1.f); }, 8 * 4 * 32768.0, blocks, iters);
    }
    return 0;
}
