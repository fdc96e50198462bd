// Issue-rate microbenchmark for v_mfma_f32_16x16x32_bf16 in the wave geometry of the pipelined GEMM: one workgroup per CU,
// NW waves, each wave a 4 x 7 accumulator tile (28 independent accumulators), operands in registers (variant 0), or
// re-read from LDS with ds_read_b128 in the GEMM's pattern (variant 1: 11 reads per 28 MFMAs).  Prints cycles per MFMA
// per SIMD (16 = the matrix pipe's rate).
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/mfma_bf16_rate.hip -o scripts/micro/mfma_bf16_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)

template <int VAR, int NTHREADS>
__global__ __launch_bounds__(NTHREADS, NTHREADS / 256) void k(const bf16x8* __restrict__ in, float* __restrict__ out, unsigned long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 65536 / 16; i += NTHREADS) reinterpret_cast<bf16x8*>(lds)[i] = in[i & 1023];
    __syncthreads();
    f32x4 acc[4][7];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 7; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a[4], b[7];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = in[lane + 64 * i];
#pragma unroll
    for (int j = 0; j < 7; ++j) b[j] = in[lane + 64 * (4 + j)];
    const unsigned roff = (lane & 15) * 128 + ((((lane >> 4)) ^ ((lane & 15) >> 1)) << 4) + (tid >> 6) * 2048;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (VAR == 2 || VAR == 3) {
        // VAR 2: the GEMM's k-tile: barrier, then the tile's reads, then its 56 MFMAs (two k-steps: b re-read per step)
        // VAR 3: the next tile's first fragments are read BEFORE the barrier (as a 3-stage ring would allow)
        bf16x8 an[4], bn[7];
        if (VAR == 3) {
#pragma unroll
            for (int j = 0; j < 7; ++j) bn[j] = *reinterpret_cast<const bf16x8*>(lds + ((roff + 2048 * j) & 65535));
#pragma unroll
            for (int i = 0; i < 4; ++i) an[i] = *reinterpret_cast<const bf16x8*>(lds + ((roff + 16384 + 2048 * i) & 65535));
        }
        for (int it = 0; it < iters / 2; ++it) {
            __builtin_amdgcn_s_barrier();
            if (VAR == 2) {
#pragma unroll
                for (int j = 0; j < 7; ++j) b[j] = *reinterpret_cast<const bf16x8*>(lds + ((roff + 2048 * j + 64 * (it & 1)) & 65535));
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const bf16x8*>(lds + ((roff + 16384 + 2048 * i) & 65535));
            } else {
#pragma unroll
                for (int j = 0; j < 7; ++j) b[j] = bn[j];
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = an[i];
            }
            bf16x8 a1[4], b1[7];
#pragma unroll
            for (int j = 0; j < 7; ++j) b1[j] = *reinterpret_cast<const bf16x8*>(lds + ((roff + 32768 + 2048 * j + 64 * (it & 1)) & 65535));
#pragma unroll
            for (int i = 0; i < 4; ++i) a1[i] = *reinterpret_cast<const bf16x8*>(lds + ((roff + 49152 + 2048 * i) & 65535));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 7; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
            if (VAR == 3) {
#pragma unroll
                for (int j = 0; j < 7; ++j) bn[j] = *reinterpret_cast<const bf16x8*>(lds + ((roff + 2048 * j + 64 * (it & 1)) & 65535));
#pragma unroll
                for (int i = 0; i < 4; ++i) an[i] = *reinterpret_cast<const bf16x8*>(lds + ((roff + 16384 + 2048 * i + 64 * (it & 1)) & 65535));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 7; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[j], a1[i], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else
    for (int it = 0; it < iters; ++it) {
        if (VAR == 1) {
#pragma unroll
            for (int j = 0; j < 7; ++j) b[j] = *reinterpret_cast<const bf16x8*>(lds + ((roff + 2048 * j + 64 * (it & 1)) & 65535));
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const bf16x8*>(lds + ((roff + 16384 + 2048 * i) & 65535));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 7; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 7; ++j) s += acc[i][j][0] + acc[i][j][3];
    out[blockIdx.x * NTHREADS + tid] = s;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    bf16x8* in; float* out; unsigned long long* cyc;
    CK(hipMalloc(&in, 1 << 20)); CK(hipMalloc(&out, 256 * 1024 * 4)); CK(hipMalloc(&cyc, 256 * 8));
    unsigned short* h = (unsigned short*)malloc(1 << 20);
    for (int i = 0; i < (1 << 19); ++i) h[i] = 0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15);     // random-ish bf16 around +-1
    CK(hipMemcpy(in, h, 1 << 20, hipMemcpyHostToDevice));
    const int iters = 2000;
    unsigned long long hc[256];
    auto report = [&](const char* name, int waves) {
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(hc, cyc, sizeof hc, hipMemcpyDeviceToHost));
        double s = 0; for (int i = 0; i < 256; ++i) s += hc[i];
        const double per_wave = s / 256 / iters / 28.0;                  // cycles per MFMA of one wave
        printf("%-34s %d waves/CU: %6.1f cycles per MFMA per wave = %5.1f per SIMD-MFMA (pipe rate 16)\n", name, waves, per_wave,
               per_wave / (waves / 4.0));
        return 0;
    };
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<0, 256>), dim3(256), dim3(256), 0, 0, in, out, cyc, iters); report("registers only", 4);
        hipLaunchKernelGGL((k<0, 512>), dim3(256), dim3(512), 0, 0, in, out, cyc, iters); report("registers only", 8);
        hipLaunchKernelGGL((k<1, 256>), dim3(256), dim3(256), 0, 0, in, out, cyc, iters); report("11 ds_read_b128 per 28 MFMA", 4);
        hipLaunchKernelGGL((k<1, 512>), dim3(256), dim3(512), 0, 0, in, out, cyc, iters); report("11 ds_read_b128 per 28 MFMA", 8);
        hipLaunchKernelGGL((k<2, 512>), dim3(256), dim3(512), 0, 0, in, out, cyc, iters); report("k-tile: barrier, reads, 56 MFMA", 8);
        hipLaunchKernelGGL((k<3, 512>), dim3(256), dim3(512), 0, 0, in, out, cyc, iters); report("k-tile, first reads before barrier", 8);
    }
    return 0;
}
