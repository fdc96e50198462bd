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
#ifndef CALM_GEMM_VEC_EPILOGUE
#define CALM_GEMM_VEC_EPILOGUE 1     // 0: always the one-element-per-access epilogue (A/B runs)
#endif
#ifndef CALM_GEMM_KC_SWIZZLE
#define CALM_GEMM_KC_SWIZZLE 1     // bf16 k-contiguous staging: transposed row slots per wave (conflict-free ds_write_b128); 0 for A/B
#endif
#ifndef CALM_GEMM_BF16_DEEP
#define CALM_GEMM_BF16_DEEP 1       // bf16-tensor kernels keep two k-tiles of global loads in flight (0: one, for A/B runs)
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
    int epi_vec;    // every epilogue tensor is addressable in aligned groups of 4 columns (dispatcher): vector epilogue
    int nz;         // pipelined family (gemm_bf16p.h): batch entries / k-slices per tile = work items per tile
    int stagger;    // experiment: start delay step of the persistent workgroups
    int epi_unit;   // pipelined family: columns per lane in the row-layout epilogue (8: every epilogue tensor is bf16)
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
// the epilogue's LDS scratch is passed as an LDS-address-space pointer: through a generic pointer the accesses
// would compile to FLAT instructions
typedef __attribute__((address_space(3))) float lds_float;
typedef __attribute__((address_space(3))) f32x4 lds_f32x4;
// four consecutive elements (i a multiple of 4, base 16-byte aligned): one 16-byte (fp32) or 8-byte (bf16) access
template <bool TYPED>
__device__ __forceinline__ f32x4 ld_elem4(const void* base, long i, int type) {
    if (TYPED && type == CALM_ST_BF16) {
        typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
        const bf16x4_t b = *reinterpret_cast<const bf16x4_t*>(reinterpret_cast<const __bf16*>(base) + i);
        return f32x4{(float)b[0], (float)b[1], (float)b[2], (float)b[3]};
    }
    return *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + i);
}
template <bool TYPED>
__device__ __forceinline__ void st_elem4(void* base, long i, const f32x4& v, int type) {
    if (TYPED && type == CALM_ST_BF16) {
        typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
        *reinterpret_cast<bf16x4_t*>(reinterpret_cast<__bf16*>(base) + i) =
            bf16x4_t{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    } else {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(base) + i) = v;
    }
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
//
// In the accumulator layout a lane owns ONE column of 16 rows, so every epilogue tensor would be touched by 16
// single-element accesses per 32x32 tile (measured: a third of the time of a K=672 product, and GELU' with its
// 2-byte gathers doubled it).  Plain (non-atomic) launches whose tensors allow it (p.epi_vec) therefore turn each
// tile through `scratch` — 32x32 fp32 of LDS private to the wave, carved out of the operand stages after the k-loop —
// into the row layout: lane = (row l>>3 of 8, columns 4(l&7)..+3), four passes per tile, and bias / LayerScale /
// aux / residual / old C / C itself move as 4-element vectors.
template <int MT, int NT, bool TYPED = false>
__device__ __forceinline__ void gemm_epilogue(const GemmP& p, f32x16 (&acc)[MT][NT], int m0, int n0, int wm, int wn,
                                              int r, int h, int z, int sgroup, lds_float* __restrict__ scratch) {
#ifdef CALM_GEMM_NO_EPILOGUE            // timing experiment: everything but the epilogue (alpha is never this value)
    if (p.alpha != 12345.f) return;
#endif
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

    // every GemmP field the tile loops use, read once: the kernel-argument struct stays in constant memory only while
    // the optimiser can see all of its uses (hundreds of `p.` reads in the unrolled tile code made it fall back to a
    // private copy of the struct in scratch)
    const int pM = p.M, pN = p.N, p_act = p.act, p_accumulate = p.accumulate, p_atomic = p.atomic;
    const int c_type = p.c_type, aux_type = p.aux_type, r_type = p.r_type;
    const long c_rs = p.c_rs, r_rs = p.r_rs, ws_slice = p.ws_slice;
    const float* __restrict__ p_bias = p.bias;
    const float* __restrict__ p_col_scale = p.col_scale;
    float* __restrict__ p_ws = p.ws;
    if (p.epi_vec && !p_atomic) {
        const int lane = r + 32 * h, rl = lane >> 3, c4 = 4 * (lane & 7);
        for_each_subtile<0, MT, NT>([&](int i, int j, const f32x16& a) __attribute__((always_inline)) {
            const int col0 = n0 + wn * (32 * NT) + 32 * j, row0 = m0 + wm * (32 * MT) + 32 * i;
            if (col0 >= pN || row0 >= pM) return;                    // wave-uniform
#pragma unroll
            for (int e = 0; e < 16; ++e) scratch[(4 * h + (e & 3) + 8 * (e >> 2)) * 32 + r] = a[e];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int col = col0 + c4;
            const bool col_ok = col < pN;                              // N % 4 == 0: a group is inside or outside
            const int colc = col_ok ? col : col0;
            const f32x4 bj = p_bias ? *reinterpret_cast<const f32x4*>(p_bias + colc) : f32x4{0.f, 0.f, 0.f, 0.f};
            const f32x4 sj = p_col_scale ? *reinterpret_cast<const f32x4*>(p_col_scale + colc) : f32x4{1.f, 1.f, 1.f, 1.f};
            // two passes (16 rows) at a time: their loads are in flight together, and the live set stays small enough
            // for the 96-register kernels (accumulators + 2 x (value, operand) vectors)
#pragma unroll
            for (int q0 = 0; q0 < 4; q0 += 2) {
                f32x4 v[2], t[2];
                int ro[2];                                              // row for loads (rows past M clamped)
                bool ok[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int row = row0 + rl + 8 * (q0 + q);
                    ok[q] = col_ok && row < pM;
                    ro[q] = row < pM ? row : row0;
                    v[q] = *(const lds_f32x4*)(scratch + (rl + 8 * (q0 + q)) * 32 + c4);
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[q][c] = v[q][c] * scale + bj[c];
                }
                if (Pb) {
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        if (ok[q]) st_elem4<TYPED>(Pb, (long)ro[q] * c_rs + colc, v[q], c_type);
                }
                if (p_act == CALM_ACT_GELU) {
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int c = 0; c < 4; ++c) v[q][c] = gelu_erf_f(v[q][c]);
                } else if (p_act == CALM_ACT_GELU_BWD) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) t[q] = ld_elem4<TYPED>(Xb, (long)ro[q] * c_rs + colc, aux_type);
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int c = 0; c < 4; ++c) v[q][c] *= gelu_erf_grad_f(t[q][c]);
                }
#pragma unroll
                for (int q = 0; q < 2; ++q) v[q] *= sj;
                if (Rb) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) t[q] = ld_elem4<TYPED>(Rb, (long)ro[q] * r_rs + colc, r_type);
#pragma unroll
                    for (int q = 0; q < 2; ++q) v[q] += t[q];
                }
                if (p_accumulate) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) t[q] = ld_elem4<TYPED>(Cb, (long)ro[q] * c_rs + colc, c_type);
#pragma unroll
                    for (int q = 0; q < 2; ++q) v[q] += t[q];
                }
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    if (ok[q]) st_elem4<TYPED>(Cb, (long)ro[q] * c_rs + colc, v[q], c_type);
            }
            // the next tile's scratch writes are issued after these reads: the LDS executes a wave's accesses in order
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }, acc);
        return;
    }
    for_each_subtile<0, MT, NT>([&](int i, int j, const f32x16& a) __attribute__((always_inline)) {
        {
            const int col = n0 + wn * (32 * NT) + 32 * j + r;
            if (col >= pN) return;
            const float bj = p_bias ? p_bias[col] : 0.f;
            const float sj = p_col_scale ? p_col_scale[col] : 1.f;
            const int row0 = m0 + wm * (32 * MT) + 32 * i + 4 * h;
            if (p_atomic) {
                if (p_ws) {                      // dense [M][N] partial of this k-slice; splitk_reduce sums the slices
                    float* __restrict__ Wb = p_ws + (long)blockIdx.y * ws_slice;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int row = row0 + (e & 3) + 8 * (e >> 2);
                        if (row < pM) Wb[(long)row * pN + col] = a[e] * scale;
                    }
                    return;
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = row0 + (e & 3) + 8 * (e >> 2);
                    if (row < pM) atomicAdd(reinterpret_cast<float*>(Cb) + (long)row * c_rs + col, a[e] * scale);
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
                    if (row < pM) st_elem<TYPED>(Pb, (long)row * c_rs + col, v[e], c_type);
                }
            }
            if (p_act == CALM_ACT_GELU) {
#pragma unroll
                for (int e = 0; e < 16; ++e) v[e] = gelu_erf_f(v[e]);
            } else if (p_act == CALM_ACT_GELU_BWD) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = row0 + (e & 3) + 8 * (e >> 2);
                    t[e] = ld_elem<TYPED>(Xb, (long)(row < pM ? row : 0) * c_rs + col, aux_type);
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
                    t[e] = ld_elem<TYPED>(Rb, (long)(row < pM ? row : 0) * r_rs + col, r_type);
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) v[e] += t[e];
            }
            if (p_accumulate) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = row0 + (e & 3) + 8 * (e >> 2);
                    t[e] = ld_elem<TYPED>(Cb, (long)(row < pM ? row : 0) * c_rs + col, c_type);
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) v[e] += t[e];
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = row0 + (e & 3) + 8 * (e >> 2);
                if (row < pM) st_elem<TYPED>(Cb, (long)row * c_rs + col, v[e], c_type);
            }
        }
    }, acc);
}

// launchers of the kernel families (defined in gemm_f32.hip / gemm_bf16.hip)
int launch_f32(const GemmP& p, dim3 grid, int bn, bool akc, bool bkc, bool vec, hipStream_t s);
int launch_bf16(const GemmP& p, dim3 grid, int bn, bool akc, bool bkc, int npass, hipStream_t s);
int launch_bf16_wide(const GemmP& p, dim3 grid, bool akc, bool bkc, hipStream_t s);
int launch_fp8(const GemmP& p, dim3 grid, hipStream_t s);           // gemm_fp8.hip
// pipelined persistent bf16-tensor family (gemm_bf16p.h), one translation unit per operand-layout pair:
// kk = A, B k-contiguous; km = A k-contiguous, B row-contiguous; mm = both row-contiguous.  tile (64 mt) x (32 nt)
int launch_pipe_kk(const GemmP& p, int mt, int nt, int grid, hipStream_t s);
int launch_pipe_km(const GemmP& p, int mt, int nt, int grid, hipStream_t s);
int launch_pipe_mm(const GemmP& p, int mt, int nt, int grid, hipStream_t s);
// the same family on fp32 tensors (gemm_f32p_kernel, v_mfma_f32_16x16x4_f32, k-tiles of 32)
int launch_pipe32_kk(const GemmP& p, int mt, int nt, int grid, hipStream_t s);
int launch_pipe32_km(const GemmP& p, int mt, int nt, int grid, hipStream_t s);
int launch_pipe32_mm(const GemmP& p, int mt, int nt, int grid, hipStream_t s);

}  // namespace calm_gemm_detail
