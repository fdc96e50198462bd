// Shared device helpers for the gfx950 kernels of libcalmvit_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/calm_vit.h"

#define CALM_WAVE 64

#define CALM_LAUNCH_CHECK()                              \
    do {                                                 \
        hipError_t e__ = hipGetLastError();              \
        if (e__ != hipSuccess) return (int)e__;          \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// erf-GELU with ONE exponential: erf by Abramowitz-Stegun 7.1.26 (|abs error| <= 1.5e-7, the level of
// fp32 erff itself), and the same e^{-x^2/2} serves the Gaussian term of the derivative.
//   gelu(x)  = 0.5 x (1 + erf(x/sqrt2)),   gelu'(x) = 0.5 (1 + erf(x/sqrt2)) + x e^{-x^2/2} / sqrt(2 pi)
__device__ __forceinline__ void gelu_parts_f(float x, float& cdf, float& gauss) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float e = __expf(-z * z);                       // e^{-x^2/2}
    // v_rcp_f32 (1 ulp) — __frcp_rn expands to the 10-instruction IEEE division sequence; the argument is >= 1 and
    // the A-S polynomial's own error (1.5e-7) is above what one ulp of t contributes
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f),
                                        -0.284496736f), 0.254829592f);
    const float erf_abs = fmaf(-poly, e, 1.0f);           // erf(|x|/sqrt2)
    cdf = 0.5f * (1.0f + copysignf(erf_abs, x));
    gauss = e;
}
// Value only (the forward GELUs: two per pixel and channel in the CNN tail, the MLP / mask-MLP epilogues): ONE
// transcendental and no division.  erf(z) = 1 - 2^{P(z)} on z = |x|/sqrt2 in [0, 5] with P(z) = z (c0 + c1 z + ... + c6 z^6),
// a weighted minimax fit of log2(erfc(z)) (absolute error of erf <= 1.6e-7 in fp32 — the Abramowitz-Stegun form above has
// 1.5e-7 — and erfc(5) = 1.5e-12, so the clamp costs nothing); 13 VALU operations against 16, one quarter-rate
// instruction (v_exp_f32) against two (v_exp_f32 + v_rcp_f32): the CNN kernels are bound by exactly this issue stream.
__device__ __forceinline__ float gelu_erf_f(float x) {
    const float z = fminf(fabsf(x) * 0.70710678118654752440f, 5.0f);
    float p = fmaf(1.002031474778603e-4f, z, -4.6146623459936717e-4f);
    p = fmaf(p, z, -2.3024777777127534e-3f);
    p = fmaf(p, z, 2.9452769646990513e-2f);
    p = fmaf(p, z, -1.4896380039388554e-1f);
    p = fmaf(p, z, -9.18328614377605e-1f);
    p = fmaf(p, z, -1.6279137340146728f);
    const float e = __builtin_amdgcn_exp2f(p * z);           // erfc(z)
    const float h = fmaf(-0.5f, e, 0.5f);                    // erf(z) / 2
    return x * (0.5f + copysignf(h, x));
}
__device__ __forceinline__ float gelu_erf_grad_f(float x) {
    float cdf, g;
    gelu_parts_f(x, cdf, g);
    return fmaf(x * 0.39894228040143267794f, g, cdf);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Block-wide sum for blockDim.x == 256 (4 waves); `red` is >= 4 floats of LDS.  All threads get the result.
__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float block_max_256(float v, float* red) {
    v = wave_max(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- fixed-order cross-workgroup reductions (ABI v7) -------------------------------------------------------------------
// Sums that span workgroups (LayerNorm dw, RoPE d_inv_freq, bias column sums, the latent KL sum, the CNN tail's weight
// gradients) are two-stage: workgroup g writes its partial row to partials[g * stride .. + n) (caller-provided scratch)
// and this second launch adds the G rows in a fixed tree over g — sixteen row lanes (g mod 16), each with four running sums
// in increasing g, combined pairwise in a fixed order — and ADDS the result to the output.  No atomics: the result
// repeats bit for bit.  (Rounds 1-3 left every workgroup with one fp32 atomic per output element, which made the
// backward differ in the last bits from run to run.)  Up to six output tensors side by side in one partial row:
// columns [begin[k], begin[k + 1]) go to out[k].
struct CalmReduceDst {
    float* out[6];
    int begin[7];
    int nseg;
};
#define CALM_RED_THREADS 1024
static __global__ __launch_bounds__(CALM_RED_THREADS) void calm_reduce_partials_kernel(const float* __restrict__ part, int G,
                                                                                      int n, int stride,
                                                                                      const CalmReduceDst dst) {
    // 16 row lanes (waves) x 64 columns; a row lane owns g = rg, rg + 16, ... with FOUR running sums (g mod 64 picks the
    // sum), i.e. four loads in flight per thread: the first form of this kernel (4 row lanes x 2 sums, 256 threads) took
    // 19.5 us per call on average — G / 8 dependent L2 round trips — and, at 173 calls per step, 3.4 ms of the step
    __shared__ float red[CALM_RED_THREADS / 64][64];
    const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (c < n) {
        const float* p = part + c;
        int g = rg;
        for (; g + 48 < G; g += 64) {
            a0 += p[(long)g * stride];
            a1 += p[(long)(g + 16) * stride];
            a2 += p[(long)(g + 32) * stride];
            a3 += p[(long)(g + 48) * stride];
        }
        if (g < G) a0 += p[(long)g * stride];
        if (g + 16 < G) a1 += p[(long)(g + 16) * stride];
        if (g + 32 < G) a2 += p[(long)(g + 32) * stride];
    }
    red[rg][lane] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (rg == 0 && c < n) {
        float t[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            t[q] = (red[4 * q][lane] + red[4 * q + 1][lane]) + (red[4 * q + 2][lane] + red[4 * q + 3][lane]);
        int k = 0;
        while (k + 1 < dst.nseg && c >= dst.begin[k + 1]) ++k;
        dst.out[k][c - dst.begin[k]] += (t[0] + t[1]) + (t[2] + t[3]);
    }
}
static inline void calm_reduce_partials(const float* part, int G, int n, float* out, hipStream_t s) {
    CalmReduceDst d{};
    d.out[0] = out; d.begin[0] = 0; d.begin[1] = n; d.nseg = 1;
    hipLaunchKernelGGL(calm_reduce_partials_kernel, dim3((n + 63) / 64), dim3(CALM_RED_THREADS), 0, s, part, G, n, n, d);
}
