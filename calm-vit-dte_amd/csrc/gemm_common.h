// Shared pieces of the calm_gemm kernel families (gemm_f32.hip: exact fp32 MFMA; gemm_bf16.hip: bf16-operand MFMA):
// launch parameters, operand addressing of batched / grouped launches, the fused epilogue, and the launchers the
// dispatcher (gemm.hip) calls.  One translation unit per family keeps the build parallel.
#pragma once
#include "common.h"
#include <type_traits>

namespace calm_gemm_detail {

#ifndef CALM_GEMM_BK
#define CALM_GEMM_BK 16          // k-tile of the fp32 family (A/B'd: 16 vs 32)
#endif
constexpr int BM = 128, BK = CALM_GEMM_BK, NTHREADS = 256;
constexpr int CK = 32;                                  // k-tile of the bf16-operand family
constexpr int WTHREADS = 512, WBM = 256, WBN = 128;     // its wide tile
// min 4 waves/SIMD: keeps the accumulators in arch VGPRs (<=128 registers in total) instead of VGPR+AGPR (152-182),
// i.e. 4 resident workgroups per CU instead of 3 (BN=96) / 2 (BN=128); A/B'd +3% over the shape mix, +10..15% on
// short-K and per-head batched shapes
#ifndef CALM_GEMM_WAVES
#define CALM_GEMM_WAVES 4
#endif
#ifndef CALM_GEMM_WAVES96
#define CALM_GEMM_WAVES96 5      // 128x96 tile: B image at its own row stride (29.7 KB LDS) and <=96 VGPRs -> 5 workgroups per CU (A/B -1.3% time)
#endif
#ifndef CALM_GEMM_BF16_WAVES
#define CALM_GEMM_BF16_WAVES 3      // bf16-operand family: 3 (A/B: bf16 -4% time; 4 spills; bf16x3 is LDS-limited to 2 either way)
#endif
#ifndef CALM_GEMM_WIDE_MIN_TILES
#define CALM_GEMM_WIDE_MIN_TILES 512
#endif

struct GemmP {
    const void* A; const void* B; void* C;      // element type per a_type / b_type / c_type (strides in elements)
    int M, N, K;
    int batch1;
    long a_rs, a_cs, a_b0, a_b1;
    long b_rs, b_cs, b_b0, b_b1;
    long c_rs, c_b0, c_b1;
    float alpha;
    const float* inv_scale; const float* bias; const float* col_scale;
    const void* residual; long r_rs, r_b0, r_b1;
    void* C_pre; const void* aux;               // C_pre has C's type
    int act, accumulate;
    int kpb;        // k-blocks per batch entry
    int kb_total;   // k-blocks in the whole reduction space walked by grid.y
    int kb_per_z;   // k-blocks per grid.y slice
    int atomic;     // partial results combined with fp32 atomics (split-K / batch-reduce)
    int tiles_m, tiles_n;
    // grouped form: the b0 entries are separate allocations with their own spectral-norm scale
    int slices_per_batch;          // batched split-K: k-slices per batch entry (0: off)
    int n_group, reduce_group;     // reduce_group: the groups are summed into one C
    const void* Ag[4]; const void* Bg[4]; void* Cg[4]; const float* Sg[4];
    float* ws; long ws_slice;      // split launches with a workspace: slice blockIdx.y stores its partial tile at ws + y * ws_slice
    int a_type, b_type, c_type, aux_type, r_type;      // CALM_ST_*: only the bf16-operand family takes bf16 tensors
    const float* dq_a; const float* dq_b;              // fp8 operands: device dequantisation factors (amax / FP8_MAX)
};

// operand base of batch entry (b0, b1); T = the operand's storage type
template <typename T = float>
__device__ __forceinline__ const T* operand_base(const void* base, const void* const (&tab)[4], int n_group,
                                                 long s0, long s1, int b0, int b1) {
    if (n_group && tab[0]) return reinterpret_cast<const T*>(tab[b0]) + b1 * s1;
    return reinterpret_cast<const T*>(base) + b0 * s0 + b1 * s1;
}
// element access of the epilogue operands: TYPED = false (fp32 family) compiles to plain fp32 accesses
template <bool TYPED>
__device__ __forceinline__ float ld_elem(const void* base, long i, int type) {
    if (TYPED && type == CALM_ST_BF16) return (float)reinterpret_cast<const __bf16*>(base)[i];
    return reinterpret_cast<const float*>(base)[i];
}
template <bool TYPED>
__device__ __forceinline__ void st_elem(void* base, long i, float v, int type) {
    if (TYPED && type == CALM_ST_BF16) reinterpret_cast<__bf16*>(base)[i] = (__bf16)v;
    else reinterpret_cast<float*>(base)[i] = v;
}
__device__ __forceinline__ float group_sigma(const GemmP& p, int g) { return p.Sg[g] ? p.Sg[g][0] : 1.f; }

// grouped reduction (C = sum_g A_g B_g^T / sigma_g): the accumulators are kept in units of the CURRENT group's sigma
// — on entering group g they are multiplied by sigma_g / sigma_{g-1} — and the epilogue divides by the last one.
template <int MT, int NT>
__device__ __forceinline__ void group_rescale(const GemmP& p, f32x16 (&acc)[MT][NT], int g) {
    const float ratio = group_sigma(p, g) / group_sigma(p, g - 1);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] *= ratio;
}

// compile-time walk over the MT x NT accumulator tiles of a wave (indices stay constants: with 2x3 tiles the
// optimiser no longer unrolls a runtime double loop of this size and the accumulators would end up in scratch)
template <int IDX, int MT, int NT, class F>
__device__ __forceinline__ void for_each_subtile(F&& f, f32x16 (&acc)[MT][NT]) {
    if constexpr (IDX < MT * NT) {
        f(IDX / NT, IDX % NT, acc[IDX / NT][IDX % NT]);
        for_each_subtile<IDX + 1, MT, NT>(f, acc);
    }
}

// Shared epilogue: acc (32x32 MFMA C layout: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)) ->
// scale, bias, optional pre-activation store, GELU / GELU', LayerScale, residual, accumulate or atomics.
template <int MT, int NT, bool TYPED = false>
__device__ __forceinline__ void gemm_epilogue(const GemmP& p, f32x16 (&acc)[MT][NT], int m0, int n0, int wm, int wn,
                                              int r, int h, int z, int sgroup) {
    float scale = p.alpha;
    if (p.inv_scale) scale = scale / p.inv_scale[0];
    if (TYPED && p.dq_a) scale *= p.dq_a[0] * p.dq_b[0];
    const int zc = (p.atomic && !p.slices_per_batch) ? 0 : z;
    const int cb0 = zc / p.batch1, cb1 = zc - cb0 * p.batch1;
    const long coff = cb0 * p.c_b0 + cb1 * p.c_b1;
    // bases as byte pointers + element offsets (the element size depends on the tensor's storage type)
    const int csz = (TYPED && p.c_type == CALM_ST_BF16) ? 2 : 4;
    char* __restrict__ Cb = reinterpret_cast<char*>(p.C) + coff * csz;
    if (p.n_group) {
        // independent groups: group cb0's sigma and output; grouped reduction: `sgroup` = last group of this k-range
        scale = scale / group_sigma(p, p.reduce_group ? sgroup : cb0);
        if (!p.reduce_group && p.Cg[0]) Cb = reinterpret_cast<char*>(p.Cg[cb0]) + cb1 * p.c_b1 * csz;
    }
    char* __restrict__ Pb = p.C_pre ? reinterpret_cast<char*>(p.C_pre) + coff * csz : nullptr;
    const char* __restrict__ Xb =
        p.aux ? reinterpret_cast<const char*>(p.aux) + coff * ((TYPED && p.aux_type == CALM_ST_BF16) ? 2 : 4) : nullptr;
    const char* __restrict__ Rb = p.residual ? reinterpret_cast<const char*>(p.residual) +
                                                   (cb0 * p.r_b0 + cb1 * p.r_b1) * ((TYPED && p.r_type == CALM_ST_BF16) ? 2 : 4)
                                             : nullptr;

    for_each_subtile<0, MT, NT>([&](int i, int j, const f32x16& a) {
        {
            const int col = n0 + wn * (32 * NT) + 32 * j + r;
            if (col >= p.N) return;
            const float bj = p.bias ? p.bias[col] : 0.f;
            const float sj = p.col_scale ? p.col_scale[col] : 1.f;
            const int row0 = m0 + wm * (32 * MT) + 32 * i + 4 * h;
            if (p.atomic) {
                if (p.ws) {                      // dense [M][N] partial of this k-slice; splitk_reduce sums the slices
                    float* __restrict__ Wb = p.ws + (long)blockIdx.y * p.ws_slice;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int row = row0 + (e & 3) + 8 * (e >> 2);
                        if (row < p.M) Wb[(long)row * p.N + col] = a[e] * scale;
                    }
                    return;
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = row0 + (e & 3) + 8 * (e >> 2);
                    if (row < p.M) atomicAdd(reinterpret_cast<float*>(Cb) + (long)row * p.c_rs + col, a[e] * scale);
                }
                return;
            }
            // Each extra operand (aux for GELU', residual, old C) is fetched as 16 independent loads into one
            // temporary (rows past M clamped to row 0) and folded into the accumulator in place, one operand at a
            // time: loads stay in flight together without holding three 16-register arrays live.
            float v[16], t[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = a[e] * scale + bj;
            if (Pb) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = row0 + (e & 3) + 8 * (e >> 2);
                    if (row < p.M) st_elem<TYPED>(Pb, (long)row * p.c_rs + col, v[e], p.c_type);
                }
            }
            if (p.act == CALM_ACT_GELU) {
#pragma unroll
                for (int e = 0; e < 16; ++e) v[e] = gelu_erf_f(v[e]);
            } else if (p.act == CALM_ACT_GELU_BWD) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = row0 + (e & 3) + 8 * (e >> 2);
                    t[e] = ld_elem<TYPED>(Xb, (long)(row < p.M ? row : 0) * p.c_rs + col, p.aux_type);
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) v[e] *= gelu_erf_grad_f(t[e]);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] *= sj;
            if (Rb) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = row0 + (e & 3) + 8 * (e >> 2);
                    t[e] = ld_elem<TYPED>(Rb, (long)(row < p.M ? row : 0) * p.r_rs + col, p.r_type);
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) v[e] += t[e];
            }
            if (p.accumulate) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = row0 + (e & 3) + 8 * (e >> 2);
                    t[e] = ld_elem<TYPED>(Cb, (long)(row < p.M ? row : 0) * p.c_rs + col, p.c_type);
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) v[e] += t[e];
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = row0 + (e & 3) + 8 * (e >> 2);
                if (row < p.M) st_elem<TYPED>(Cb, (long)row * p.c_rs + col, v[e], p.c_type);
            }
        }
    }, acc);
}

// launchers of the kernel families (defined in gemm_f32.hip / gemm_bf16.hip)
int launch_f32(const GemmP& p, dim3 grid, int bn, bool akc, bool bkc, bool vec, hipStream_t s);
int launch_bf16(const GemmP& p, dim3 grid, int bn, bool akc, bool bkc, int npass, hipStream_t s);
int launch_bf16_wide(const GemmP& p, dim3 grid, bool akc, bool bkc, hipStream_t s);
int launch_fp8(const GemmP& p, dim3 grid, hipStream_t s);           // gemm_fp8.hip

}  // namespace calm_gemm_detail
