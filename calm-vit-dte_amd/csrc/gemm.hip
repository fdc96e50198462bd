// calm_gemm: strided / batched / split-K GEMM with fused epilogue for gfx950 (fp32 tensors).
// Two kernel families share tiling, remap, split logic and epilogue: the exact fp32 MFMA family below and the
// bf16-operand family (gemm_bf16c_kernel) further down.
//
// Tile 128x128x16 (4 waves as 2x2, each 64x64 = 2x2 v_mfma_f32_32x32x2_f32 accumulators) or 128x96x16 (4x1 waves,
// 1x3 accumulators) per 256-thread workgroup.  Both operands are staged K-MAJOR in LDS ([k][row], row stride 132 floats): fragment
// reads are then one conflict-free ds_read_b32 per MFMA operand for every source layout, and the
// four source layouts (k- or row-contiguous A and B) only differ in the global->register->LDS
// staging.  fp32 MFMA issues every 64 cycles per SIMD, so LDS/VALU work hides under it; the loop is
// a register-prefetch double buffer (global loads of tile t+1 issued before the MFMAs of tile t).
// Workgroup ids are remapped so that consecutive tiles (n fastest: they share an A panel) land on
// the same XCD / L2.
#include "common.h"
#include <type_traits>

namespace {

#ifndef CALM_GEMM_BK
#define CALM_GEMM_BK 16          // k-tile of the fp32 family (A/B'd: 16 vs 32)
#endif
constexpr int BM = 128, BK = CALM_GEMM_BK, LDT = 132, NTHREADS = 256;
constexpr int NREG = BK / 2;     // staging floats per thread per operand (128 rows x BK / 256 threads)

struct GemmP {
    const float* A; const float* B; float* C;
    int M, N, K;
    int batch1;
    long a_rs, a_cs, a_b0, a_b1;
    long b_rs, b_cs, b_b0, b_b1;
    long c_rs, c_b0, c_b1;
    float alpha;
    const float* inv_scale; const float* bias; const float* col_scale;
    const float* residual; long r_rs, r_b0, r_b1;
    float* C_pre; const float* aux;
    int act, accumulate;
    int kpb;        // k-blocks per batch entry
    int kb_total;   // k-blocks in the whole reduction space walked by grid.y
    int kb_per_z;   // k-blocks per grid.y slice
    int atomic;     // partial results combined with fp32 atomics (split-K / batch-reduce)
    int tiles_m, tiles_n;
    // grouped form: the b0 entries are separate allocations with their own spectral-norm scale
    int slices_per_batch;          // batched split-K: k-slices per batch entry (0: off)
    int n_group, reduce_group;     // reduce_group: the groups are summed into one C
    const float* Ag[4]; const float* Bg[4]; float* Cg[4]; const float* Sg[4];
    float* ws; long ws_slice;      // split launches with a workspace: slice blockIdx.y stores its partial tile at ws + y * ws_slice
};

// operand base of batch entry (b0, b1)
__device__ __forceinline__ const float* operand_base(const float* base, const float* const (&tab)[4], int n_group,
                                                     long s0, long s1, int b0, int b1) {
    if (n_group && tab[0]) return tab[b0] + b1 * s1;
    return base + b0 * s0 + b1 * s1;
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

// 16-byte staging with a per-thread cursor: the row part of every address is computed once per (tile, batch entry),
// the k-loop only advances the pointers.  Rows past the tile (edge tiles, and rows 96..127 of the 128-row thread map
// on a 96-row B tile) are CLAMPED to the tile's last row instead of masked: they only feed output rows / columns
// that are never stored, so the loads stay unconditional (no exec masking, no zero fill).  Only a partial last
// k-block (K % BK != 0) takes the masked form.
template <bool KC, int ROWS>
struct OperandCursor {
    const float* base;               // uniform (SGPR pair): tile origin at the current k-block
    unsigned off[NREG / 4];          // per-thread byte offsets from it (constant over the k-loop)
    long step;
    __device__ __forceinline__ void init(const float* origin, long rs, long cs, int row0, int nrows_all, int k0) {
        const int tid = threadIdx.x;
        const int last = min(nrows_all - row0, ROWS) - 1;        // last live row of the tile, tile-local
        if constexpr (KC) {
            constexpr int KQ = BK / 4, RPP = NTHREADS / KQ;
            base = origin + (long)row0 * rs + k0;
#pragma unroll
            for (int i = 0; i < NREG / 4; ++i)
                off[i] = (unsigned)(min((tid / KQ) + RPP * i, last) * rs + 4 * (tid % KQ)) * 4u;
            step = BK;
        } else {
            base = origin + (long)k0 * cs + row0;
            const int row = min(4 * (tid & 31), last & ~3);      // rows come in aligned groups of 4 (M % 4 == 0)
#pragma unroll
            for (int i = 0; i < NREG / 4; ++i) off[i] = (unsigned)(((tid >> 5) + 8 * i) * cs + row) * 4u;
            step = BK * cs;
        }
    }
    // k_left = K - k0 (> 0).  FULL: K is a whole number of k-blocks — the loads are unconditional (the kernel holds
    // one copy of its k-loop per case: with a run-time choice in one loop the compiler folds both forms into the
    // masked one, 16 zero fills and 4 exec-mask branches per iteration)
    template <bool FULL>
    __device__ __forceinline__ void load(int k_left, float (&reg)[NREG]) {
        const int tid = threadIdx.x;
        const char* b = reinterpret_cast<const char*>(base);
#pragma unroll
        for (int i = 0; i < NREG / 4; ++i) {
            f32x4 v;
            if constexpr (FULL) {
                v = *reinterpret_cast<const f32x4*>(b + off[i]);
            } else {
                const int k = KC ? 4 * (tid % (BK / 4)) : (tid >> 5) + 8 * i;
                v = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (k < k_left) v = *reinterpret_cast<const f32x4*>(b + off[i]);
            }
            reg[4 * i + 0] = v[0]; reg[4 * i + 1] = v[1]; reg[4 * i + 2] = v[2]; reg[4 * i + 3] = v[3];
        }
        base += step;
    }
};

template <bool KC, int VEC, int ROWS>
__device__ __forceinline__ void load_operand(const float* __restrict__ base, long rs, long cs, int row0,
                                             int nrows_all, int k0, int K, float (&reg)[NREG]) {
    const int tid = threadIdx.x;
    const int nrows = min(nrows_all, row0 + ROWS);       // rows of THIS tile only
    if constexpr (VEC == 4) {
        if constexpr (KC) {
            constexpr int KQ = BK / 4, RPP = NTHREADS / KQ;       // float4 per row, rows per pass
            const int k = k0 + 4 * (tid % KQ);
#pragma unroll
            for (int i = 0; i < NREG / 4; ++i) {
                const int row = row0 + (tid / KQ) + RPP * i;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (row < nrows && k < K) v = *reinterpret_cast<const f32x4*>(base + (long)row * rs + k);
                reg[4 * i + 0] = v[0]; reg[4 * i + 1] = v[1]; reg[4 * i + 2] = v[2]; reg[4 * i + 3] = v[3];
            }
        } else {
            const int row = row0 + 4 * (tid & 31);
#pragma unroll
            for (int i = 0; i < NREG / 4; ++i) {
                const int k = k0 + (tid >> 5) + 8 * i;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (row < nrows && k < K) v = *reinterpret_cast<const f32x4*>(base + (long)k * cs + row);
                reg[4 * i + 0] = v[0]; reg[4 * i + 1] = v[1]; reg[4 * i + 2] = v[2]; reg[4 * i + 3] = v[3];
            }
        }
    } else {
        if constexpr (KC) {
            const int k = k0 + (tid % BK);
#pragma unroll
            for (int i = 0; i < NREG; ++i) {
                const int row = row0 + (tid / BK) + (NTHREADS / BK) * i;
                reg[i] = (row < nrows && k < K) ? base[(long)row * rs + k] : 0.f;
            }
        } else {
            const int row = row0 + (tid & 127);
#pragma unroll
            for (int i = 0; i < NREG; ++i) {
                const int k = k0 + (tid >> 7) + 2 * i;
                reg[i] = (row < nrows && k < K) ? base[(long)k * cs + row] : 0.f;
            }
        }
    }
}

// ROWS = rows of the tile this operand stages (128, or 96 for the B side of the 128x96 tile): the 128-row thread
// mapping is shared, rows past ROWS are simply not stored (their image row stride LD may be too short for them).
template <bool KC, int VEC, int LD, int ROWS>
__device__ __forceinline__ void store_operand(float (*T)[LD], const float (&reg)[NREG]) {
    const int tid = threadIdx.x;
    if constexpr (VEC == 4) {
        if constexpr (KC) {
            constexpr int KQ = BK / 4, RPP = NTHREADS / KQ;
            const int kq = 4 * (tid % KQ);
            // a wave stages 64 / KQ consecutive rows per pass: whether they lie past ROWS is wave-uniform (scalar branch)
            const int wave_row = __builtin_amdgcn_readfirstlane(tid >> 6) * (64 / KQ);
#pragma unroll
            for (int i = 0; i < NREG / 4; ++i) {
                const int row = (tid / KQ) + RPP * i;
                if (ROWS < 128 && ROWS % (64 / KQ) == 0 && wave_row + RPP * i >= ROWS) continue;
                if (ROWS < 128 && ROWS % (64 / KQ) != 0 && row >= ROWS) continue;
#pragma unroll
                for (int j = 0; j < 4; ++j) T[kq + j][row] = reg[4 * i + j];
            }
        } else {
            const int row = 4 * (tid & 31);
            if (ROWS < 128 && row >= ROWS) return;
#pragma unroll
            for (int i = 0; i < NREG / 4; ++i) {
                const int k = (tid >> 5) + 8 * i;
                f32x4 v = {reg[4 * i + 0], reg[4 * i + 1], reg[4 * i + 2], reg[4 * i + 3]};
                *reinterpret_cast<f32x4*>(&T[k][row]) = v;
            }
        }
    } else {
        if constexpr (KC) {
            const int k = tid % BK;
#pragma unroll
            for (int i = 0; i < NREG; ++i) {
                const int row = (tid / BK) + (NTHREADS / BK) * i;
                if (ROWS < 128 && row >= ROWS) continue;
                T[k][row] = reg[i];
            }
        } else {
            const int row = tid & 127;
            if (ROWS < 128 && row >= ROWS) return;
#pragma unroll
            for (int i = 0; i < NREG; ++i) T[(tid >> 7) + 2 * i][row] = reg[i];
        }
    }
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
template <int MT, int NT>
__device__ __forceinline__ void gemm_epilogue(const GemmP& p, f32x16 (&acc)[MT][NT], int m0, int n0, int wm, int wn,
                                              int r, int h, int z, int sgroup) {
    float scale = p.alpha;
    if (p.inv_scale) scale = scale / p.inv_scale[0];
    const int zc = (p.atomic && !p.slices_per_batch) ? 0 : z;
    const int cb0 = zc / p.batch1, cb1 = zc - cb0 * p.batch1;
    const long coff = cb0 * p.c_b0 + cb1 * p.c_b1;
    float* __restrict__ Cb = p.C + coff;
    if (p.n_group) {
        // independent groups: group cb0's sigma and output; grouped reduction: `sgroup` = last group of this k-range
        scale = scale / group_sigma(p, p.reduce_group ? sgroup : cb0);
        if (!p.reduce_group && p.Cg[0]) Cb = p.Cg[cb0] + cb1 * p.c_b1;
    }
    float* __restrict__ Pb = p.C_pre ? p.C_pre + coff : nullptr;
    const float* __restrict__ Xb = p.aux ? p.aux + coff : nullptr;
    const float* __restrict__ Rb = p.residual ? p.residual + cb0 * p.r_b0 + cb1 * p.r_b1 : nullptr;

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
                    if (row < p.M) atomicAdd(Cb + (long)row * p.c_rs + col, a[e] * scale);
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
                    if (row < p.M) Pb[(long)row * p.c_rs + col] = v[e];
                }
            }
            if (p.act == CALM_ACT_GELU) {
#pragma unroll
                for (int e = 0; e < 16; ++e) v[e] = gelu_erf_f(v[e]);
            } else if (p.act == CALM_ACT_GELU_BWD) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = row0 + (e & 3) + 8 * (e >> 2);
                    t[e] = Xb[(long)(row < p.M ? row : 0) * p.c_rs + col];
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
                    t[e] = Rb[(long)(row < p.M ? row : 0) * p.r_rs + col];
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) v[e] += t[e];
            }
            if (p.accumulate) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = row0 + (e & 3) + 8 * (e >> 2);
                    t[e] = Cb[(long)(row < p.M ? row : 0) * p.c_rs + col];
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) v[e] += t[e];
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = row0 + (e & 3) + 8 * (e >> 2);
                if (row < p.M) Cb[(long)row * p.c_rs + col] = v[e];
            }
        }
    }, acc);
}

template <bool AKC, bool BKC, int VEC, int BN_>
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
__global__ __launch_bounds__(NTHREADS, BN_ == 96 ? CALM_GEMM_WAVES96 : CALM_GEMM_WAVES) void gemm_f32_kernel(const GemmP p) {
    constexpr int WN = BN_ == 128 ? 2 : 1;        // wave grid: 2x2 (128x128 tile) or 4x1 (128x96 tile)
    constexpr int MT = BN_ == 128 ? 2 : 1;        // 32x32 MFMA tiles per wave along M
    constexpr int NT = BN_ / (WN * 32);           // ... along N (2 or 3)
    constexpr int LDB = BN_ == 128 ? LDT : 100;   // B image row stride: 96 columns need no more (100 = 4 mod 32 banks too)
    __shared__ __attribute__((aligned(16))) float As[2][BK][LDT];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][LDB];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave id in an SGPR
    const int wm = WN == 2 ? wave >> 1 : wave, wn = WN == 2 ? wave & 1 : 0;
    const int r = lane & 31, h = lane >> 5;

    // XCD-aware, bijective tile remap (blocks b and b+8 share an XCD).
    const int tiles = p.tiles_m * p.tiles_n;
    int lin = blockIdx.x;
    if (tiles >= 8) {
        const int q = tiles >> 3, rem = tiles & 7, x = lin & 7, idx = lin >> 3;
        lin = (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + idx;
    }
    const int tn = lin % p.tiles_n, tm = lin / p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN_;
    // grid.y: batch entry (plain), k-slice of the concatenated reduction (split-K / reduce_batch), or — batched
    // split-K, slices_per_batch > 0 — k-slice `z % spb` of batch entry `z / spb` (entry-local reduction range)
    int z = blockIdx.y;
    int kb_begin = z * p.kb_per_z;
    int kb_end = min(kb_begin + p.kb_per_z, p.kb_total);
    if (p.slices_per_batch) {
        const int b = z / p.slices_per_batch, sl = z - b * p.slices_per_batch;
        kb_begin = b * p.kpb + sl * p.kb_per_z;
        kb_end = min(kb_begin + p.kb_per_z, (b + 1) * p.kpb);
        z = b;                                   // the epilogue's batch index
    }
    if (kb_begin >= kb_end && p.atomic) return;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    float ra[NREG], rb[NREG];

    OperandCursor<AKC, BM> ca;
    OperandCursor<BKC, BN_> cb;
    // the 8 k-pairs of one staged k-block: 1 (2) A and 3 (2) B fragments per 3 (4) MFMAs.  A wave whose rows all lie
    // past M (edge tile of a short M: the 40- and 176-row sequence-axis products) stages and synchronises but
    // issues no MFMAs: its accumulators are never stored.
    const bool wave_live = m0 + wm * (32 * MT) < p.M;
    auto multiply = [&](int buf) {
        if (!wave_live) return;
#pragma unroll
        for (int s = 0; s < BK / 2; ++s) {
            const int kk = 2 * s + h;
            float af[MT], bf[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) af[i] = As[buf][kk][wm * (32 * MT) + 32 * i + r];
#pragma unroll
            for (int j = 0; j < NT; ++j) bf[j] = Bs[buf][kk][wn * (32 * NT) + 32 * j + r];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    };
    auto stash = [&](int buf) {
        store_operand<AKC, VEC, LDT, BM>(As[buf], ra);
        store_operand<BKC, VEC, LDB, BN_>(Bs[buf], rb);
    };

    auto k_loop = [&](auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        int cur_b = -1;                       // batch entry the cursors point into
        auto fetch = [&](int kb) {
            const int b = p.kb_total == p.kpb ? 0 : kb / p.kpb;          // single-entry launches skip the divisions
            const int k0 = (kb - b * p.kpb) * BK;
            if constexpr (VEC == 4) {
                if (b != cur_b) {                                         // wave-uniform: first fetch, or a batch boundary
                    const int b0 = b / p.batch1, b1 = b - b0 * p.batch1;
                    ca.init(operand_base(p.A, p.Ag, p.n_group, p.a_b0, p.a_b1, b0, b1), p.a_rs, p.a_cs, m0, p.M, k0);
                    cb.init(operand_base(p.B, p.Bg, p.n_group, p.b_b0, p.b_b1, b0, b1), p.b_rs, p.b_cs, n0, p.N, k0);
                    cur_b = b;
                }
                ca.template load<FULL>(p.K - k0, ra);
                cb.template load<FULL>(p.K - k0, rb);
            } else {
                const int b0 = b / p.batch1, b1 = b - b0 * p.batch1;
                load_operand<AKC, VEC, BM>(operand_base(p.A, p.Ag, p.n_group, p.a_b0, p.a_b1, b0, b1), p.a_rs, p.a_cs,
                                           m0, p.M, k0, p.K, ra);
                load_operand<BKC, VEC, BN_>(operand_base(p.B, p.Bg, p.n_group, p.b_b0, p.b_b1, b0, b1), p.b_rs,
                                            p.b_cs, n0, p.N, k0, p.K, rb);
            }
        };
        if (kb_begin < kb_end) {
            fetch(kb_begin);
            stash(0);
        }
        __syncthreads();
        // two k-blocks per trip: the LDS stage of each half is a compile-time constant (no per-iteration address math)
        auto step = [&](int kb, auto stage_tag) {
            constexpr int ST = decltype(stage_tag)::value;
            const bool more = kb + 1 < kb_end;
            if (p.reduce_group && kb != kb_begin && kb % p.kpb == 0) group_rescale<MT, NT>(p, acc, kb / p.kpb);
            if (more) fetch(kb + 1);
            multiply(ST);
            if (more) stash(ST ^ 1);
            __syncthreads();
        };
        for (int kb = kb_begin; kb < kb_end; kb += 2) {
            step(kb, std::integral_constant<int, 0>{});
            if (kb + 1 < kb_end) step(kb + 1, std::integral_constant<int, 1>{});
        }
    };
    if constexpr (VEC == 4) {
        if (p.K % BK == 0) k_loop(std::true_type{});
        else k_loop(std::false_type{});
    } else {
        k_loop(std::false_type{});
    }
    // (a peeled loop without the per-iteration decisions for single-entry, whole-k-block launches measured +2% on
    // forward / input-gradient shapes, -2% on weight gradients and -0.6% on the training step: not kept)

    gemm_epilogue<MT, NT>(p, acc, m0, n0, wm, wn, r, h, z, (kb_end - 1) / p.kpb);
}


// =====================================================================================================
// bf16-operand GEMM family: tensors stay fp32 in HBM; operands are converted to bf16 while being staged
// into LDS and multiplied on v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate) with fp32 accumulation.
//   NPASS == 1: plain bf16 operands                  (what autocast(bfloat16) computes for Linear/matmul)
//   NPASS == 3: split a = hi + lo (both bf16); acc += hi*hi + hi*lo + lo*hi   ("bf16x3": products accurate to
//               ~2^-17 relative, i.e. fp32-level results at a fraction of the fp32-MFMA time, because the
//               kernel is bound by staging fp32 bytes, not by the matrix pipe).
// k-contiguous operands use a [row][k] LDS image (80-byte rows: conflict-free ds_read_b128 fragments);
// row-contiguous ("transposed") operands keep their natural [k][row] image (320-byte rows), written with
// contiguous ds_write_b64 and read as k-contiguous MFMA fragments by ds_read_b64_tr_b16.
// =====================================================================================================
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int CK = 32;                 // k-tile
constexpr int KC_LD = 40;              // bf16 per row of a [row][k] image
constexpr int MC_LD = 160;             // bf16 per k-row of a [k][row] image
constexpr int PLANE = 128 * KC_LD;     // == 32 * MC_LD bf16 = 10240 B

template <bool KC, int ROWS>
__device__ __forceinline__ void c_load(const float* __restrict__ base, long rs, long cs, int row0, int nrows_all,
                                       int k0, int K, f32x4 (&reg)[4]) {
    const int tid = threadIdx.x;
    const int nrows = min(nrows_all, row0 + ROWS);
    if constexpr (KC) {
        const int k = k0 + 4 * (tid & 7);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = row0 + (tid >> 3) + 32 * i;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (row < nrows && k < K) v = *reinterpret_cast<const f32x4*>(base + (long)row * rs + k);
            reg[i] = v;
        }
    } else {
        const int row = row0 + 4 * (tid & 31);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = k0 + 4 * (tid >> 5) + j;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (row < nrows && k < K) v = *reinterpret_cast<const f32x4*>(base + (long)k * cs + row);
            reg[j] = v;
        }
    }
}

// cursor form of c_load (see OperandCursor): row addresses once per (tile, batch entry), clamped edge rows,
// unconditional 16-byte loads for full k-blocks
template <bool KC, int ROWS>
struct CCursor {
    const float* base;
    unsigned off[4];
    long step;
    __device__ __forceinline__ void init(const float* origin, long rs, long cs, int row0, int nrows_all, int k0) {
        const int tid = threadIdx.x;
        const int last = min(nrows_all - row0, ROWS) - 1;
        if constexpr (KC) {
            base = origin + (long)row0 * rs + k0;
#pragma unroll
            for (int i = 0; i < 4; ++i) off[i] = (unsigned)(min((tid >> 3) + 32 * i, last) * rs + 4 * (tid & 7)) * 4u;
            step = CK;
        } else {
            base = origin + (long)k0 * cs + row0;
            const int row = min(4 * (tid & 31), last & ~3);
#pragma unroll
            for (int j = 0; j < 4; ++j) off[j] = (unsigned)((4 * (tid >> 5) + j) * cs + row) * 4u;
            step = CK * cs;
        }
    }
    template <bool FULL>
    __device__ __forceinline__ void load(int k_left, f32x4 (&reg)[4]) {
        const int tid = threadIdx.x;
        const char* b = reinterpret_cast<const char*>(base);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 v;
            if constexpr (FULL) {
                v = *reinterpret_cast<const f32x4*>(b + off[i]);
            } else {
                const int k = KC ? 4 * (tid & 7) : 4 * (tid >> 5) + i;
                v = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (k < k_left) v = *reinterpret_cast<const f32x4*>(b + off[i]);
            }
            reg[i] = v;
        }
        base += step;
    }
};

template <int NPASS>
__device__ __forceinline__ void split4(const f32x4& v, bf16x4& hi, bf16x4& lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        hi[e] = (__bf16)v[e];
        if constexpr (NPASS == 3) lo[e] = (__bf16)(v[e] - (float)hi[e]);
    }
}

template <bool KC, int NPASS>
__device__ __forceinline__ void c_store(__bf16* __restrict__ hi_plane, __bf16* __restrict__ lo_plane,
                                        const f32x4 (&reg)[4]) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        bf16x4 hi, lo;
        split4<NPASS>(reg[i], hi, lo);
        int off;
        if constexpr (KC) off = ((tid >> 3) + 32 * i) * KC_LD + 4 * (tid & 7);          // [row][k]
        else off = (4 * (tid >> 5) + i) * MC_LD + 4 * (tid & 31);                       // [k][row]
        *reinterpret_cast<bf16x4*>(hi_plane + off) = hi;
        if constexpr (NPASS == 3) *reinterpret_cast<bf16x4*>(lo_plane + off) = lo;
    }
}

// MFMA A/B fragment (8 consecutive k for row `rowbase + (lane&31)`, k = 16*s + 8*(lane>>5) + 0..7)
template <bool KC, int LD = MC_LD>
__device__ __forceinline__ bf16x8 c_frag(const __bf16* __restrict__ plane, int rowbase, int s, int lane) {
    if constexpr (KC) {
        return *reinterpret_cast<const bf16x8*>(plane + (rowbase + (lane & 31)) * KC_LD + 16 * s + 8 * (lane >> 5));
    } else {
        // hardware transpose read: each 16-lane group fetches a 4(k) x 16(row) block and gets it column-major
        const int q = (lane & 15) >> 2, pp = lane & 3, gi = lane >> 4;
        const __bf16* a0 = plane + (16 * s + 8 * (gi >> 1) + q) * LD + rowbase + 16 * (gi & 1) + 4 * pp;
        const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(a0));
        const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(a0 + 4 * LD));
        s16x8 v = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
        return __builtin_bit_cast(bf16x8, v);
    }
}

template <bool AKC, bool BKC, int NPASS, int BN_>
__global__ __launch_bounds__(NTHREADS, NPASS == 3 ? 2 : CALM_GEMM_BF16_WAVES) void gemm_bf16c_kernel(const GemmP p) {
    constexpr int WN = BN_ == 128 ? 2 : 1;
    constexpr int MT = BN_ == 128 ? 2 : 1;
    constexpr int NT = BN_ / (WN * 32);
    constexpr int NPL = NPASS == 3 ? 2 : 1;                     // planes per operand (hi [, lo])
    __shared__ __attribute__((aligned(16))) __bf16 lds[2][2][NPL][PLANE];   // [stage][A|B][plane]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = WN == 2 ? wave >> 1 : wave, wn = WN == 2 ? wave & 1 : 0;
    const int r = lane & 31, h = lane >> 5;

    const int tiles = p.tiles_m * p.tiles_n;
    int lin = blockIdx.x;
    if (tiles >= 8) {
        const int q = tiles >> 3, rem = tiles & 7, x = lin & 7, idx = lin >> 3;
        lin = (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + idx;
    }
    const int tn = lin % p.tiles_n, tm = lin / p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN_;
    // grid.y: batch entry (plain), k-slice of the concatenated reduction (split-K / reduce_batch), or — batched
    // split-K, slices_per_batch > 0 — k-slice `z % spb` of batch entry `z / spb` (entry-local reduction range)
    int z = blockIdx.y;
    int kb_begin = z * p.kb_per_z;
    int kb_end = min(kb_begin + p.kb_per_z, p.kb_total);
    if (p.slices_per_batch) {
        const int b = z / p.slices_per_batch, sl = z - b * p.slices_per_batch;
        kb_begin = b * p.kpb + sl * p.kb_per_z;
        kb_end = min(kb_begin + p.kb_per_z, (b + 1) * p.kpb);
        z = b;                                   // the epilogue's batch index
    }
    if (kb_begin >= kb_end && p.atomic) return;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    f32x4 ra[4], rb[4];
    CCursor<AKC, BM> ca;
    CCursor<BKC, BN_> cb;
    // one copy of the k-loop per case (whole k-tiles / tailed K): see OperandCursor::load
    auto k_loop = [&](auto full_tag) {
    constexpr bool FULL = decltype(full_tag)::value;
    int cur_b = -1;
    auto fetch = [&](int kb) {
        const int b = p.kb_total == p.kpb ? 0 : kb / p.kpb;
        const int k0 = (kb - b * p.kpb) * CK;
        if (b != cur_b) {
            const int b0 = b / p.batch1, b1 = b - b0 * p.batch1;
            ca.init(operand_base(p.A, p.Ag, p.n_group, p.a_b0, p.a_b1, b0, b1), p.a_rs, p.a_cs, m0, p.M, k0);
            cb.init(operand_base(p.B, p.Bg, p.n_group, p.b_b0, p.b_b1, b0, b1), p.b_rs, p.b_cs, n0, p.N, k0);
            cur_b = b;
        }
        ca.template load<FULL>(p.K - k0, ra);
        cb.template load<FULL>(p.K - k0, rb);
    };
    auto stash = [&](int st) {
        c_store<AKC, NPASS>(lds[st][0][0], lds[st][0][NPL - 1], ra);
        c_store<BKC, NPASS>(lds[st][1][0], lds[st][1][NPL - 1], rb);
    };

    int buf = 0;
    if (kb_begin < kb_end) {
        fetch(kb_begin);
        stash(0);
    }
    __syncthreads();

    for (int kb = kb_begin; kb < kb_end; ++kb) {
        const bool more = kb + 1 < kb_end;
        if (p.reduce_group && kb != kb_begin && kb % p.kpb == 0) group_rescale<MT, NT>(p, acc, kb / p.kpb);
        if (more) fetch(kb + 1);
#pragma unroll
        for (int s = 0; s < CK / 16; ++s) {
            bf16x8 ah[MT], bh[NT], al[MT], bl[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                ah[i] = c_frag<AKC>(lds[buf][0][0], wm * (32 * MT) + 32 * i, s, lane);
                if constexpr (NPASS == 3) al[i] = c_frag<AKC>(lds[buf][0][NPL - 1], wm * (32 * MT) + 32 * i, s, lane);
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                bh[j] = c_frag<BKC>(lds[buf][1][0], wn * (32 * NT) + 32 * j, s, lane);
                if constexpr (NPASS == 3) bl[j] = c_frag<BKC>(lds[buf][1][NPL - 1], wn * (32 * NT) + 32 * j, s, lane);
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    if constexpr (NPASS == 3) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
        if (more) stash(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    };
    if (p.K % CK == 0) k_loop(std::true_type{});
    else k_loop(std::false_type{});
    gemm_epilogue<MT, NT>(p, acc, m0, n0, wm, wn, r, h, z, (kb_end - 1) / p.kpb);
}

// ---- wide tile of the bf16-operand family: 256x128x32 per 512-thread workgroup (8 waves as 4x2, each 64x64) ----
// The 128-row tiles above are bound by re-reading the fp32 operand panels from L2 / Infinity Cache (32 flop per
// byte staged); 256 rows raise that to 42.7.  60 KB LDS, <=128 VGPRs: two workgroups (16 waves) per CU.  Used for
// the data-parallel launches (forward, data gradients) with enough tiles to fill the chip; NPASS == 1 only.
constexpr int WTHREADS = 512, WBM = 256, WBN = 128;
#ifndef CALM_GEMM_WIDE_MIN_TILES
#define CALM_GEMM_WIDE_MIN_TILES 512
#endif
constexpr int MC_LDW = 288;                  // bf16 per k-row of a 256-row [k][row] image (same bank residue as 160)
constexpr int WPLANE_A = WBM * KC_LD;        // 20480 B (>= 32 * MC_LDW)

template <bool KC, int ROWS>
struct WCursor {
    static constexpr int NV = ROWS * CK / (4 * WTHREADS);     // 16-byte vectors per thread per k-tile (4 or 2)
    static constexpr int LD = ROWS == WBM ? MC_LDW : MC_LD;
    const float* base;
    unsigned off[NV];
    long step;
    __device__ __forceinline__ void init(const float* origin, long rs, long cs, int row0, int nrows_all, int k0) {
        const int tid = threadIdx.x;
        const int last = min(nrows_all - row0, ROWS) - 1;
        if constexpr (KC) {
            base = origin + (long)row0 * rs + k0;
#pragma unroll
            for (int i = 0; i < NV; ++i) off[i] = (unsigned)(min((tid >> 3) + 64 * i, last) * rs + 4 * (tid & 7)) * 4u;
            step = CK;
        } else {
            constexpr int LPR = ROWS / 4;                     // threads across the rows
            base = origin + (long)k0 * cs + row0;
            const int row = min(4 * (tid % LPR), last & ~3);
#pragma unroll
            for (int j = 0; j < NV; ++j) off[j] = (unsigned)((NV * (tid / LPR) + j) * cs + row) * 4u;
            step = CK * cs;
        }
    }
    template <bool FULL>
    __device__ __forceinline__ void load(int k_left, f32x4 (&reg)[NV]) {
        const int tid = threadIdx.x;
        const char* b = reinterpret_cast<const char*>(base);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            f32x4 v;
            if constexpr (FULL) {
                v = *reinterpret_cast<const f32x4*>(b + off[i]);
            } else {
                const int k = KC ? 4 * (tid & 7) : NV * (tid / (ROWS / 4)) + i;
                v = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (k < k_left) v = *reinterpret_cast<const f32x4*>(b + off[i]);
            }
            reg[i] = v;
        }
        base += step;
    }
    __device__ __forceinline__ void store(__bf16* __restrict__ plane, const f32x4 (&reg)[NV]) const {
        const int tid = threadIdx.x;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            bf16x4 hi, lo;
            split4<1>(reg[i], hi, lo);
            int off;
            if constexpr (KC) off = ((tid >> 3) + 64 * i) * KC_LD + 4 * (tid & 7);
            else off = (NV * (tid / (ROWS / 4)) + i) * LD + 4 * (tid % (ROWS / 4));
            *reinterpret_cast<bf16x4*>(plane + off) = hi;
        }
    }
};

template <bool AKC, bool BKC>
__global__ __launch_bounds__(WTHREADS, 4) void gemm_bf16w_kernel(const GemmP p) {
    constexpr int MT = 2, NT = 2;
    __shared__ __attribute__((aligned(16))) __bf16 lds_a[2][WPLANE_A];
    __shared__ __attribute__((aligned(16))) __bf16 lds_b[2][PLANE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;

    const int tiles = p.tiles_m * p.tiles_n;
    int lin = blockIdx.x;
    if (tiles >= 8) {
        const int q = tiles >> 3, rem = tiles & 7, x = lin & 7, idx = lin >> 3;
        lin = (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + idx;
    }
    const int tn = lin % p.tiles_n, tm = lin / p.tiles_n;
    const int m0 = tm * WBM, n0 = tn * WBN;
    int z = blockIdx.y;
    int kb_begin = z * p.kb_per_z;
    int kb_end = min(kb_begin + p.kb_per_z, p.kb_total);
    if (p.slices_per_batch) {                        // batched split-K (grouped weight gradients), as in the 128-row kernels
        const int b = z / p.slices_per_batch, sl = z - b * p.slices_per_batch;
        kb_begin = b * p.kpb + sl * p.kb_per_z;
        kb_end = min(kb_begin + p.kb_per_z, (b + 1) * p.kpb);
        z = b;
    }
    if (kb_begin >= kb_end && p.atomic) return;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    WCursor<AKC, WBM> ca;
    WCursor<BKC, WBN> cb;
    f32x4 ra[WCursor<AKC, WBM>::NV], rb[WCursor<BKC, WBN>::NV];
    auto k_loop = [&](auto full_tag) {
    constexpr bool FULL = decltype(full_tag)::value;
    int cur_b = -1;
    auto fetch = [&](int kb) {
        const int b = p.kb_total == p.kpb ? 0 : kb / p.kpb;
        const int k0 = (kb - b * p.kpb) * CK;
        if (b != cur_b) {
            const int b0 = b / p.batch1, b1 = b - b0 * p.batch1;
            ca.init(operand_base(p.A, p.Ag, p.n_group, p.a_b0, p.a_b1, b0, b1), p.a_rs, p.a_cs, m0, p.M, k0);
            cb.init(operand_base(p.B, p.Bg, p.n_group, p.b_b0, p.b_b1, b0, b1), p.b_rs, p.b_cs, n0, p.N, k0);
            cur_b = b;
        }
        ca.template load<FULL>(p.K - k0, ra);
        cb.template load<FULL>(p.K - k0, rb);
    };

    int buf = 0;
    if (kb_begin < kb_end) {
        fetch(kb_begin);
        ca.store(lds_a[0], ra);
        cb.store(lds_b[0], rb);
    }
    __syncthreads();

    for (int kb = kb_begin; kb < kb_end; ++kb) {
        const bool more = kb + 1 < kb_end;
        if (p.reduce_group && kb != kb_begin && kb % p.kpb == 0) group_rescale<MT, NT>(p, acc, kb / p.kpb);
        if (more) fetch(kb + 1);
#pragma unroll
        for (int s = 0; s < CK / 16; ++s) {
            bf16x8 af[MT], bf[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) af[i] = c_frag<AKC, MC_LDW>(lds_a[buf], wm * 64 + 32 * i, s, lane);
#pragma unroll
            for (int j = 0; j < NT; ++j) bf[j] = c_frag<BKC, MC_LD>(lds_b[buf], wn * 64 + 32 * j, s, lane);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (more) {
            ca.store(lds_a[buf ^ 1], ra);
            cb.store(lds_b[buf ^ 1], rb);
        }
        __syncthreads();
        buf ^= 1;
    }
    };
    if (p.K % CK == 0) k_loop(std::true_type{});
    else k_loop(std::false_type{});
    gemm_epilogue<MT, NT>(p, acc, m0, n0, wm, wn, r, h, z, (kb_end - 1) / p.kpb);
}

int launch_wide(const GemmP& p, dim3 grid, bool akc, bool bkc, hipStream_t s) {
    if (akc && bkc) hipLaunchKernelGGL((gemm_bf16w_kernel<true, true>), grid, dim3(WTHREADS), 0, s, p);
    else if (akc && !bkc) hipLaunchKernelGGL((gemm_bf16w_kernel<true, false>), grid, dim3(WTHREADS), 0, s, p);
    else if (!akc && bkc) hipLaunchKernelGGL((gemm_bf16w_kernel<false, true>), grid, dim3(WTHREADS), 0, s, p);
    else hipLaunchKernelGGL((gemm_bf16w_kernel<false, false>), grid, dim3(WTHREADS), 0, s, p);
    CALM_LAUNCH_CHECK();
    return 0;
}

template <bool AKC, bool BKC, int NPASS>
int launch_c(const GemmP& p, dim3 grid, int bn, hipStream_t s) {
    if (bn == 128) hipLaunchKernelGGL((gemm_bf16c_kernel<AKC, BKC, NPASS, 128>), grid, dim3(NTHREADS), 0, s, p);
    else hipLaunchKernelGGL((gemm_bf16c_kernel<AKC, BKC, NPASS, 96>), grid, dim3(NTHREADS), 0, s, p);
    CALM_LAUNCH_CHECK();
    return 0;
}

template <int NPASS>
int launch_c_layout(const GemmP& p, dim3 grid, int bn, bool akc, bool bkc, hipStream_t s) {
    if (akc && bkc) return launch_c<true, true, NPASS>(p, grid, bn, s);
    if (akc && !bkc) return launch_c<true, false, NPASS>(p, grid, bn, s);
    if (!akc && bkc) return launch_c<false, true, NPASS>(p, grid, bn, s);
    return launch_c<false, false, NPASS>(p, grid, bn, s);
}

template <bool AKC, bool BKC, int VEC>
int launch(const GemmP& p, dim3 grid, int bn, hipStream_t s) {
    if (bn == 128) hipLaunchKernelGGL((gemm_f32_kernel<AKC, BKC, VEC, 128>), grid, dim3(NTHREADS), 0, s, p);
    else hipLaunchKernelGGL((gemm_f32_kernel<AKC, BKC, VEC, 96>), grid, dim3(NTHREADS), 0, s, p);
    CALM_LAUNCH_CHECK();
    return 0;
}

// Second pass of a split launch with a workspace: C[m][n] (+= if accumulate) sum over the slices' partial tiles.
// grid.y = output (group); outputs are Cg[y] when given, else C + y * c_b0.
struct ReduceP {
    const float* ws; long ws_slice; int nslices;
    float* C; float* Cg[4]; long c_b0, c_rs;
    int M, N, accumulate;
};
__global__ __launch_bounds__(256) void splitk_reduce(const ReduceP q) {
    const long total = (long)q.M * q.N;
    const float* w = q.ws + (long)blockIdx.y * q.nslices * q.ws_slice;
    float* out = q.Cg[0] ? q.Cg[blockIdx.y] : q.C + blockIdx.y * q.c_b0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        float s0 = 0.f, s1 = 0.f;
        int sl = 0;
        for (; sl + 1 < q.nslices; sl += 2) {
            s0 += w[(long)sl * q.ws_slice + i];
            s1 += w[(long)(sl + 1) * q.ws_slice + i];
        }
        if (sl < q.nslices) s0 += w[(long)sl * q.ws_slice + i];
        const unsigned m = (unsigned)i / (unsigned)q.N, n = (unsigned)i - m * (unsigned)q.N;
        float* dst = out + (long)m * q.c_rs + n;
        *dst = (q.accumulate ? *dst : 0.f) + s0 + s1;
    }
}

inline bool mult4(int64_t x) { return (x & 3) == 0; }

}  // namespace

#ifndef CALM_GEMM_WS_MIN_SLICES
#define CALM_GEMM_WS_MIN_SLICES 48     // per-slice partial tiles + one reduction instead of atomics from this many k-slices
#endif                                 // per output (A/B with plain stores in place of the atomics: outputs of 528 rows and
                                       // more, <= 42 slices, do not change; 384x768 ... 240x240, 53-106 slices, -15..-30%)

// query != nullptr: plan only and report the workspace size of the launch (calm_gemm_workspace_bytes)
static int gemm_run(const calm_gemm_args* a, void* stream, int64_t* query) {
    if (!a || !a->A || !a->B || !a->C) return CALM_E_INVAL;
    if (a->M <= 0 || a->N <= 0 || a->K <= 0 || a->batch0 <= 0 || a->batch1 <= 0) return CALM_E_INVAL;
    if (a->dtype != CALM_F32 && a->dtype != CALM_BF16 && a->dtype != CALM_BF16X3) return CALM_E_UNSUPP;
    if (a->a_rs != 1 && a->a_cs != 1) return CALM_E_LAYOUT;
    if (a->b_rs != 1 && a->b_cs != 1) return CALM_E_LAYOUT;
    // the staging cursors address a tile with 32-bit byte offsets from a per-tile base: 255 rows x stride x 4 B < 2^32
    if (a->a_rs >= (1 << 21) || a->a_cs >= (1 << 21) || a->b_rs >= (1 << 21) || a->b_cs >= (1 << 21)) return CALM_E_UNSUPP;
    if (a->act == CALM_ACT_GELU_BWD && !a->aux) return CALM_E_INVAL;
    if (a->act < 0 || a->act > CALM_ACT_GELU_BWD) return CALM_E_INVAL;
    if (a->n_group < 0 || a->n_group > CALM_GEMM_MAX_GROUP) return CALM_E_INVAL;
    if (a->n_group) {
        if (a->batch0 != a->n_group || a->batch1 != 1) return CALM_E_INVAL;
        if (a->C_pre || a->aux || a->residual || a->inv_scale) return CALM_E_UNSUPP;
        for (int g = 0; g < a->n_group; ++g) {
            if ((a->A_group[0] && !a->A_group[g]) || (a->B_group[0] && !a->B_group[g])) return CALM_E_INVAL;
            if (!a->reduce_batch && a->C_group[0] && !a->C_group[g]) return CALM_E_INVAL;
        }
    }
    hipStream_t s = as_stream(stream);

    GemmP p;
    p.A = (const float*)a->A; p.B = (const float*)a->B; p.C = (float*)a->C;
    p.M = a->M; p.N = a->N; p.K = a->K;
    p.batch1 = a->batch1;
    p.a_rs = a->a_rs; p.a_cs = a->a_cs; p.a_b0 = a->a_b0; p.a_b1 = a->a_b1;
    p.b_rs = a->b_rs; p.b_cs = a->b_cs; p.b_b0 = a->b_b0; p.b_b1 = a->b_b1;
    p.c_rs = a->c_rs; p.c_b0 = a->c_b0; p.c_b1 = a->c_b1;
    p.alpha = a->alpha; p.inv_scale = a->inv_scale; p.bias = a->bias; p.col_scale = a->col_scale;
    p.residual = (const float*)a->residual; p.r_rs = a->r_rs; p.r_b0 = a->r_b0; p.r_b1 = a->r_b1;
    p.C_pre = (float*)a->C_pre; p.aux = (const float*)a->aux;
    p.act = a->act; p.accumulate = a->accumulate;
    p.n_group = a->n_group;
    p.reduce_group = a->n_group && a->reduce_batch;
    for (int g = 0; g < 4; ++g) {
        const bool on = g < a->n_group;
        p.Ag[g] = on ? (const float*)a->A_group[g] : nullptr;
        p.Bg[g] = on ? (const float*)a->B_group[g] : nullptr;
        p.Cg[g] = on ? (float*)a->C_group[g] : nullptr;
        p.Sg[g] = on ? a->inv_scale_group[g] : nullptr;
    }
    const int batch = a->batch0 * a->batch1;
    const bool akc = a->a_cs == 1;
    const bool bkc = a->b_cs == 1;
    bool vec = aligned16(a->A) && aligned16(a->B) && mult4(a->a_b0) && mult4(a->a_b1) && mult4(a->b_b0) &&
               mult4(a->b_b1);
    for (int g = 0; g < a->n_group; ++g) vec = vec && aligned16(a->A_group[g]) && aligned16(a->B_group[g]);
    vec = vec && (akc ? (mult4(a->K) && mult4(a->a_rs)) : (mult4(a->M) && mult4(a->a_cs)));
    vec = vec && (bkc ? (mult4(a->K) && mult4(a->b_rs)) : (mult4(a->N) && mult4(a->b_cs)));
    // bf16-operand kernels need the 16-byte staging path; anything else runs on the exact fp32 kernels
    const int family = vec ? a->dtype : CALM_F32;
    const int bk = family == CALM_F32 ? BK : CK;
    p.kpb = (a->K + bk - 1) / bk;
    const bool trivial_epi = !a->bias && !a->col_scale && !a->residual && !a->C_pre && a->act == CALM_ACT_NONE;

    // N tile: 128 (2x2 waves) or 96 (4x1 waves).  k-split launches (work spread evenly whatever the tile count) take
    // the width that pads N less (672, 528, 1344, 1056, 480 ... are multiples of 96 or nearly so), ties to 128 for
    // the better A-panel reuse.  Data-parallel launches take the width with the lower estimated time: the
    // workgroups resident on a CU share its matrix pipe, so a launch lasts about ceil(work items / CUs) x width
    // (A/B over the model's shapes: N=240 and N=352 are 12-20% faster on 96 although 128 pads less).
    p.tiles_m = (a->M + BM - 1) / BM;
    const int pad128 = (a->N + 127) / 128 * 128, pad96 = (a->N + 95) / 96 * 96;
    int bn = pad96 < pad128 ? 96 : 128;
    const bool grouped_reduce_unsplit = a->n_group && a->reduce_batch && a->split_k <= 1;
    // grouped weight gradients (dW_g = dY_g^T X of the projections that share X): every group gets its own k-slices
    const bool group_split = a->n_group && !a->reduce_batch && a->split_k == 0 && trivial_epi && a->C_group[0] &&
                             p.tiles_m * ((a->N + bn - 1) / bn) * batch < 256 && p.kpb >= 64;
    const bool k_split = group_split || (a->reduce_batch && !grouped_reduce_unsplit) || a->split_k > 1 ||
                         (a->split_k == 0 && batch == 1 && trivial_epi && p.tiles_m * ((a->N + bn - 1) / bn) < 256 &&
                          p.kpb >= 64);
    if (!k_split) {
        const long cus = 256;
        const long nb = grouped_reduce_unsplit ? 1 : batch;     // launches of the y grid dimension
        const long items96 = (long)p.tiles_m * (pad96 / 96) * nb, items128 = (long)p.tiles_m * (pad128 / 128) * nb;
        const long cost96 = (items96 + cus - 1) / cus * 96, cost128 = (items128 + cus - 1) / cus * 128;
        bn = cost96 < cost128 ? 96 : 128;
    }
    // bf16 operands, data-parallel launch with at least a full resident round (2 per CU) of 256x128 tiles: wide kernel
    bool wide = false;
    if (family == CALM_BF16 && !k_split) {
        const long nb = grouped_reduce_unsplit ? 1 : batch;
        const long rows_w = (long)(a->M + WBM - 1) / WBM * WBM;             // at most 1/8 of the rows padded
        wide = 8 * rows_w <= 9 * (long)a->M &&
               rows_w / WBM * ((a->N + WBN - 1) / WBN) * nb >= CALM_GEMM_WIDE_MIN_TILES;
        if (wide) {
            bn = WBN;
            p.tiles_m = (a->M + WBM - 1) / WBM;
        }
    }
    // ... and the split-K weight gradients (plain or grouped) whose output pads by at most 1/4 in 256-row tiles (672,
    // 768, 1056, 1344 rows): with bf16 operands they are bound by re-reading the fp32 panels from L2, not by padded MFMAs
    const bool wide_split = family == CALM_BF16 && k_split && !a->reduce_batch && (group_split || batch == 1) &&
                            4 * ((long)(a->M + WBM - 1) / WBM * WBM) <= 5 * (long)a->M;
    if (wide_split) {
        wide = true;
        bn = WBN;
        p.tiles_m = (a->M + WBM - 1) / WBM;
    }
    p.tiles_n = (a->N + bn - 1) / bn;
    const int tiles = p.tiles_m * p.tiles_n;

    // k-slices of a split launch: one resident round of workgroups (256 CUs x 5 for the 96-wide tile, x 4 for the
    // 128-wide one; A/B over the weight-gradient shapes: -3% time against 768, -12% on 1344x672)
    const int split_slots = wide ? 256 * 2 : bn == 96 ? 256 * CALM_GEMM_WAVES96 : 256 * CALM_GEMM_WAVES;
    int nsplit = 1;
    p.atomic = 0;
    p.slices_per_batch = 0;
    if (group_split) {
        nsplit = split_slots / (tiles * batch);
        const int max_split = (p.kpb + 15) / 16;
        if (nsplit > max_split) nsplit = max_split;
        if (nsplit < 1) nsplit = 1;
        p.kb_total = batch * p.kpb;
        if (nsplit > 1) {
            p.atomic = 1;
            p.slices_per_batch = nsplit;
        }
    } else if (grouped_reduce_unsplit) {
        p.kb_total = batch * p.kpb;                         // one k-range over all groups, plain epilogue
    } else if (a->reduce_batch) {
        p.atomic = 1;
        p.kb_total = batch * p.kpb;
        nsplit = a->split_k > 1 ? a->split_k : split_slots / tiles;
        const int max_split = (p.kb_total + 7) / 8;
        if (nsplit > max_split) nsplit = max_split;
        if (nsplit < 1) nsplit = 1;
    } else if (k_split) {
        if (batch != 1) return CALM_E_UNSUPP;
        nsplit = a->split_k > 1 ? a->split_k : split_slots / tiles;
        const int max_split = (p.kpb + 15) / 16;
        if (nsplit > max_split) nsplit = max_split;
        if (nsplit < 1) nsplit = 1;
        p.kb_total = p.kpb;
        p.atomic = nsplit > 1;
    } else {
        p.kb_total = batch * p.kpb;
    }
    if (p.slices_per_batch) {
        p.kb_per_z = (p.kpb + nsplit - 1) / nsplit;
        p.slices_per_batch = (p.kpb + p.kb_per_z - 1) / p.kb_per_z;      // no empty trailing slices
    } else if (p.atomic) {
        if (!trivial_epi) return CALM_E_UNSUPP;
        p.kb_per_z = (p.kb_total + nsplit - 1) / nsplit;
    } else if (grouped_reduce_unsplit) {
        p.kb_per_z = p.kb_total;                            // grid.y == 1
    } else {
        p.kb_per_z = p.kpb;   // grid.y == batch
    }
    const int gy = p.slices_per_batch ? batch * p.slices_per_batch : (p.kb_total + p.kb_per_z - 1) / p.kb_per_z;
    if (gy > 65535) return CALM_E_UNSUPP;
    dim3 grid(tiles, gy);

    // how the k-slices are combined: fp32 atomics onto a zeroed (or accumulated-into) C, or — many slices per output
    // and a caller-provided workspace — one dense partial tile per slice and a reduction pass
    const int n_out = p.slices_per_batch ? batch : 1;
    const int slices_per_out = gy / n_out;
    p.ws = nullptr;
    p.ws_slice = (long)a->M * a->N;
    int64_t ws_need = 0;
    if (p.atomic && slices_per_out >= CALM_GEMM_WS_MIN_SLICES && p.ws_slice >= 100000)     // tiny outputs: the second launch costs more than their atomics
        ws_need = (int64_t)sizeof(float) * gy * p.ws_slice;
    if (query) {
        *query = ws_need;
        return 0;
    }
    const bool use_ws = ws_need > 0 && a->workspace && a->workspace_bytes >= ws_need && aligned16(a->workspace);
    if (use_ws) {
        p.ws = (float*)a->workspace;
    } else if (p.atomic && !a->accumulate) {
        for (int g = 0; g < n_out; ++g) {
            float* out = p.slices_per_batch ? p.Cg[g] : p.C;
            hipError_t e;
            if (a->c_rs == a->N) e = hipMemsetAsync(out, 0, sizeof(float) * (size_t)a->M * a->N, s);
            else e = hipMemset2DAsync(out, sizeof(float) * a->c_rs, 0, sizeof(float) * a->N, a->M, s);
            if (e != hipSuccess) return (int)e;
        }
    }

    auto launch_main = [&]() -> int {
        if (wide) return launch_wide(p, grid, akc, bkc, s);
        if (family == CALM_BF16) return launch_c_layout<1>(p, grid, bn, akc, bkc, s);
        if (family == CALM_BF16X3) return launch_c_layout<3>(p, grid, bn, akc, bkc, s);
        if (vec) {
            if (akc && bkc) return launch<true, true, 4>(p, grid, bn, s);
            if (akc && !bkc) return launch<true, false, 4>(p, grid, bn, s);
            if (!akc && bkc) return launch<false, true, 4>(p, grid, bn, s);
            return launch<false, false, 4>(p, grid, bn, s);
        }
        if (akc && bkc) return launch<true, true, 1>(p, grid, bn, s);
        if (akc && !bkc) return launch<true, false, 1>(p, grid, bn, s);
        if (!akc && bkc) return launch<false, true, 1>(p, grid, bn, s);
        return launch<false, false, 1>(p, grid, bn, s);
    };
    const int rc = launch_main();
    if (rc || !use_ws) return rc;

    ReduceP q;
    q.ws = p.ws; q.ws_slice = p.ws_slice; q.nslices = slices_per_out;
    q.C = p.C; q.c_b0 = a->c_b0; q.c_rs = a->c_rs;
    for (int g = 0; g < 4; ++g) q.Cg[g] = p.slices_per_batch ? p.Cg[g] : nullptr;
    q.M = a->M; q.N = a->N; q.accumulate = a->accumulate;
    const long total = (long)a->M * a->N;
    const int gx = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(splitk_reduce, dim3(gx, n_out), dim3(256), 0, s, q);
    CALM_LAUNCH_CHECK();
    return 0;
}

extern "C" int calm_gemm(const calm_gemm_args* a, void* stream) { return gemm_run(a, stream, nullptr); }

extern "C" int64_t calm_gemm_workspace_bytes(const calm_gemm_args* a) {
    int64_t bytes = 0;
    return gemm_run(a, nullptr, &bytes) == 0 ? bytes : 0;
}
