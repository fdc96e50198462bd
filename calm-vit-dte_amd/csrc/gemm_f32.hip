// Exact fp32 MFMA kernel family of calm_gemm (dispatcher: gemm.hip; shared pieces: gemm_common.h).
//
// Tile 128x128x16 (4 waves as 2x2, each 64x64 = 2x2 v_mfma_f32_32x32x2_f32 accumulators) or 128x96x16 (4x1 waves,
// 1x3 accumulators) per 256-thread workgroup.  Both operands are staged K-MAJOR in LDS ([k][row], row stride 132 floats): fragment
// reads are then one conflict-free ds_read_b32 per MFMA operand for every source layout, and the
// four source layouts (k- or row-contiguous A and B) only differ in the global->register->LDS
// staging.  fp32 MFMA issues every 64 cycles per SIMD, so LDS/VALU work hides under it; the loop is
// a register-prefetch double buffer (global loads of tile t+1 issued before the MFMAs of tile t).
// Workgroup ids are remapped so that consecutive tiles (n fastest: they share an A panel) land on
// the same XCD / L2.
#include "gemm_common.h"

namespace calm_gemm_detail {

constexpr int LDT = 132;
constexpr int NREG = BK / 2;     // staging floats per thread per operand (128 rows x BK / 256 threads)

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

template <bool AKC, bool BKC, int VEC, int BN_>
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

    static_assert(sizeof(As) >= 4096 * (NTHREADS / 64), "the epilogue's per-wave scratch lives in the A stages");
    gemm_epilogue<MT, NT>(p, acc, m0, n0, wm, wn, r, h, z, (kb_end - 1) / p.kpb, (lds_float*)(&As[0][0][0]) + 1024 * wave);
}

template <bool AKC, bool BKC, int VEC>
int launch(const GemmP& p, dim3 grid, int bn, hipStream_t s) {
    if (bn == 128) hipLaunchKernelGGL((gemm_f32_kernel<AKC, BKC, VEC, 128>), grid, dim3(NTHREADS), 0, s, p);
    else hipLaunchKernelGGL((gemm_f32_kernel<AKC, BKC, VEC, 96>), grid, dim3(NTHREADS), 0, s, p);
    CALM_LAUNCH_CHECK();
    return 0;
}

int launch_f32(const GemmP& p, dim3 grid, int bn, bool akc, bool bkc, bool vec, hipStream_t s) {
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
}

}  // namespace calm_gemm_detail
