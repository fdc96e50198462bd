// Pipelined, persistent bf16-tensor GEMM of calm_gemm (dispatcher: gemm.hip) — the kernel body.  One translation unit
// per operand-layout pair (gemm_bf16p_{kk,km,mm}.hip) instantiates it so that the build stays parallel.
//
// Why a second bf16 family: the 256x128x32 register-staged kernels of gemm_bf16.hip spend 40 % of a K = 672 launch
// outside the matrix loop (first-load latency and epilogue of every tile with two workgroups per CU to overlap them) and
// are LDS-bound inside it (ds_write staging + 64x64 wave tiles).  This family is built the CDNA4 way instead:
//   * operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction, no VGPRs, no ds_write);
//     the LDS image is lane-linear, so the bank swizzle is applied to each lane's SOURCE address and undone by the
//     fragment reads (k-contiguous operands: 128-byte rows, 16-byte chunk c of row r stored at c ^ ((r >> 1) & 7) ->
//     conflict-free ds_read_b128; row-contiguous operands: [k][row] image, chunk ^ (f(k) << 1) -> conflict-free
//     ds_read_b64_tr_b16; both checked with scripts/micro/lds_bank_sim.py);
//   * tile (64 MT) x (32 NT) x 64 per 512-thread workgroup (8 waves as 4 x 2, wave tile 16 MT x 16 NT on
//     v_mfma_f32_16x16x32_bf16), MT in {2,3,4}, NT in {4..8}: N = 672 / 1344 / 448 / 224 (NT 7), 384 / 528 (NT 6),
//     480 (NT 5), 768 / 240 (NT 8) tile without padded MFMAs; 128 KiB of LDS (two 64 KiB stages), one workgroup per CU;
//   * PERSISTENT workgroups walk the (tile, batch / k-slice) items of the launch: while the last k-tile of an item is
//     multiplied, the first k-tile of the next item is already in flight, and it keeps landing during the epilogue —
//     the first-load latency of every tile but the first is hidden;
//   * the products are issued with the operands swapped (D^T = B A^T), which leaves each lane with 4 CONSECUTIVE
//     columns of one output row: the epilogue reads / writes 8- or 16-byte vectors straight from the accumulators,
//     without the LDS turn of the 32x32 kernels.
// The LDS-DMA instructions are inline asm on purpose: issued through the builtin, hipcc (ROCm 7.2) drains them with
// s_waitcnt vmcnt(0) before the first ds_read that follows (it cannot tell the two stages apart), which serialises the
// pipeline.  The k-loop contains no other vector-memory instruction, so the hand-placed vmcnt(0) per k-tile is exact.
#pragma once
#include "gemm_common.h"
#include "lds_dma.h"

#ifndef CALM_PIPE_PRIO_FLIP
#define CALM_PIPE_PRIO_FLIP 0      // trading priority between the two waves of a SIMD once per k-step: measured +-0 (36.8k vs 36.9k cycles)
#endif

namespace calm_gemm_detail {

using namespace calm_lds_dma;

typedef __bf16 pbf16x8 __attribute__((ext_vector_type(8)));
typedef short ps16x4 __attribute__((ext_vector_type(4)));
typedef short ps16x8 __attribute__((ext_vector_type(8)));

constexpr int PTHREADS = 512;
constexpr unsigned PSTAGE = 65536, PB_OFF = 32768;

// k-row swizzle of the [k][row] images: the 8 k-rows one 32-lane half of a transposed read touches get 8 different
// 32-byte granules of the 256-byte bank row
__device__ __forceinline__ int mc_swz(int kk) { return (kk & 3) | (((kk >> 3) & 1) << 2); }

// Staging of one operand: T rows (64 MT or 32 NT) x 128 bytes of k per k-tile (KT = 64 bf16 or 32 fp32 values; ES =
// bytes per element).
//   KC: memory [row][k] (k contiguous).  Image [row][128 B]; instruction q covers rows 8q..8q+7.
//   MC: memory [k][row] (rows contiguous).  Image [KT k][W B], W = 128 ES (<= 128 rows) or 256 ES; instruction q covers
//       1024 / W k-rows.  Chunk swizzle by k-row: bf16 mc_swz(k) << 1 (transposed reads), fp32 ((k >> 2) & 3) << 2 (the
//       four lane groups of a ds_read_b32 fragment read — k = 4 g + j, 16 rows x 4 B each — take four different 64-byte
//       quarters of the 256-byte bank row).
template <bool KC, int ROWS, int ES>
struct PStage {
    static constexpr int KT = 128 / ES;                               // k per k-tile
    static constexpr int RPC = 16 / ES;                               // rows (MC) or k (KC) per 16-byte chunk
    static constexpr int W = (ROWS <= 128 ? 128 : 256) * ES;          // MC: bytes per k-row
    static constexpr int TOTAL = KC ? ROWS / 8 : KT * W / 1024;       // LDS-DMA instructions per k-tile (all waves)
    static constexpr int PER_WAVE = (TOTAL + 7) / 8;
    unsigned off[PER_WAVE];            // per-lane byte offset from the tile's uniform base
    const char* base;                  // uniform: operand base of the batch entry + k offset of the current tile
    long step;                         // bytes per k-tile

    static __device__ __forceinline__ int mc_chunk_swz(int kk) { return ES == 2 ? (mc_swz(kk) << 1) : (((kk >> 2) & 3) << 2); }

    __device__ __forceinline__ void init(const void* origin, long rs, long cs, int row0, int nrows_all, int k0,
                                         int wave, int lane) {
        if constexpr (KC) {
            const int g = (lane & 7) ^ (((wave & 1) << 2) | (lane >> 4));
#pragma unroll
            for (int i = 0; i < PER_WAVE; ++i) {
                const int q = wave + 8 * i;
                const int row = min(row0 + 8 * q + (lane >> 3), nrows_all - 1);
                off[i] = (unsigned)(row * rs * ES + g * 16);
            }
            base = reinterpret_cast<const char*>(origin) + (long)k0 * ES;
            step = 128;
        } else {
            constexpr int CPR = W / 16, KPI = 1024 / W;
#pragma unroll
            for (int i = 0; i < PER_WAVE; ++i) {
                const int q = wave + 8 * i;
                const int kk = q * KPI + lane / CPR;
                const int c = (lane % CPR) ^ mc_chunk_swz(kk);
                int row = row0 + RPC * c;
                if (RPC * c >= ROWS || row >= nrows_all) row = row0;        // never used by a live output: any valid address
                off[i] = (unsigned)(kk * cs * ES + row * ES);
            }
            base = reinterpret_cast<const char*>(origin) + (long)k0 * cs * ES;
            step = KT * cs * ES;
        }
    }
    // first k (within the k-tile) of this lane's chunk in instruction q — recomputed where needed, not kept across the k-loop
    static __device__ __forceinline__ int kfirst_of(int q, int wave, int lane) {
        return KC ? RPC * ((lane & 7) ^ (((wave & 1) << 2) | (lane >> 4))) : q * (1024 / W) + lane / (W / 16);
    }
    // piece i (one LDS-DMA instruction of this wave) of the k-tile at `base` into the image at LDS byte `dst`;
    // k_left < KT: last k-tile of a reduction whose length is not a multiple of KT — chunks past the end read zeros
    template <int I, bool TAIL>
    __device__ __forceinline__ void issue_piece(unsigned dst, int wave, int lane, int k_left, unsigned long long ubase) {
        const int q = wave + 8 * I;
        if (!(TOTAL % 8 == 0 || q < TOTAL)) return;
        if constexpr (!TAIL) {
            glds16_u(ubase, off[I], dst + 1024u * q);
        } else {
            const char* a = kfirst_of(q, wave, lane) < k_left ? base + off[I] : reinterpret_cast<const char*>(calm_zero_block);
            glds16_addr(a, dst + 1024u * q);
        }
    }
    __device__ __forceinline__ void advance() { base += step; }
    // issue the k-tile at `base` into the image at LDS byte `dst`, then advance
    __device__ __forceinline__ void issue(unsigned dst, int wave) {
#pragma unroll
        for (int i = 0; i < PER_WAVE; ++i) {
            const int q = wave + 8 * i;
            if (TOTAL % 8 == 0 || q < TOTAL) glds16(base, off[i], dst + 1024u * q);
        }
        base += step;
    }
    // last k-tile of a reduction with k_left (< KT) valid k: chunks past the end read zeros
    __device__ __forceinline__ void issue_tail(unsigned dst, int wave, int lane, int k_left) {
#pragma unroll
        for (int i = 0; i < PER_WAVE; ++i) {
            const int q = wave + 8 * i;
            const char* a = kfirst_of(q, wave, lane) < k_left ? base + off[i] : reinterpret_cast<const char*>(calm_zero_block);
            if (TOTAL % 8 == 0 || q < TOTAL) glds16_addr(a, dst + 1024u * q);
        }
        base += step;
    }
};

// MFMA operand fragment (16 bytes per lane) of the 16 rows starting at tile index `t` (units of 16 rows) for k-step ks
// (half a k-tile).  bf16: lane l gets row (l & 15), k = 32 ks + 8 (l >> 4) + 0..7 — one v_mfma_f32_16x16x32_bf16 operand.
// fp32: row (l & 15), k = 16 ks + 4 (l >> 4) + 0..3 — element j is the operand of the j-th of four
// v_mfma_f32_16x16x4_f32 (which then multiplies k in {j, 4 + j, 8 + j, 12 + j}: a fixed permutation of the reduction
// order, the same for A and B).
template <int ES> struct pfrag_type { typedef pbf16x8 type; };
template <> struct pfrag_type<4> { typedef f32x4 type; };

template <bool KC, int ROWS, int ES>
struct PFrag {
    typedef typename pfrag_type<ES>::type frag_t;
    static constexpr int W = PStage<KC, ROWS, ES>::W;
    unsigned rd[KC ? 1 : ROWS / 16];       // KC: one base (tiles by immediate); MC: one per 16-row tile of the WAVE (filled up to n)
    template <int N>
    __device__ __forceinline__ void init(int tile0, int lane) {        // tile0: first 16-row tile of this wave, N tiles
        if constexpr (KC) {
            rd[0] = (unsigned)((tile0 * 16 + (lane & 15)) * 128 + (((lane >> 4) ^ ((lane & 15) >> 1)) << 4));
        } else if constexpr (ES == 2) {
            const int f = ((lane >> 2) & 3) | (((lane >> 4) & 1) << 2);
            const unsigned bl = (unsigned)((8 * (lane >> 4) + ((lane & 15) >> 2)) * W + 8 * (lane & 1));
#pragma unroll
            for (int t = 0; t < N; ++t) {
                const int c = 2 * (tile0 + t) + ((lane >> 1) & 1);
                rd[t] = bl + (unsigned)((c ^ (f << 1)) << 4);
            }
        } else {
            const int r = lane & 15, g = lane >> 4;
#pragma unroll
            for (int t = 0; t < N; ++t)
                rd[t] = (unsigned)(4 * g * W + (((4 * (tile0 + t) + (r >> 2)) ^ (g << 2)) << 4) + 4 * (r & 3));
        }
    }
    __device__ __forceinline__ frag_t load(const char* __restrict__ image, int t, int ks) const {
        if constexpr (KC) {
            const unsigned a = (rd[0] ^ (ks ? 64u : 0u)) + 2048u * t;
            return *reinterpret_cast<const frag_t*>(image + a);
        } else if constexpr (ES == 2) {
            const char* a0 = image + rd[t] + 32 * ks * W;
            const ps16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ps16x4*)(a0));
            const ps16x4 hi4 =
                __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ps16x4*)(a0 + 4 * W));
            ps16x8 v = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
            return __builtin_bit_cast(pbf16x8, v);
        } else {
            const float* a0 = reinterpret_cast<const float*>(image + rd[t] + 16 * ks * W);
            return f32x4{a0[0], a0[W / 4], a0[2 * (W / 4)], a0[3 * (W / 4)]};
        }
    }
};

// one k-step of one 16 x 16 output tile in the swapped orientation (D^T += B A^T)
__device__ __forceinline__ void pipe_mma(const pbf16x8& b, const pbf16x8& a, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, acc, 0, 0, 0);
}

// Epilogue of the swapped-operand accumulators: acc[mt][nt][e] = C(row0 + 16 mt + (lane & 15), col0 + 16 nt + 4 (lane >> 4) + e).
// Same arithmetic as gemm_epilogue (gemm_common.h).
//
// What decides the cost of an epilogue on this chip is the number of row segments a store instruction touches, not its
// bytes (scripts/micro/store_rate.hip, one workgroup per CU writing a 256 x 224 bf16 tile into 1344-byte rows: the
// accumulator layout itself — 8 bytes per lane, 16 rows x 32 bytes per instruction — 26 GB/s per CU = 4.5 us per tile;
// 16 bytes per lane over whole 448-byte row segments 102 GB/s = 1.1 us).  So every 16-row strip of a wave is turned
// through `scratch` (8 KiB of LDS per wave, fp32, 16 rows x 128 floats, 16-byte chunk c of row r at c ^ (r & 7) so that
// the 8 rows of a ds_write_b128 lane group spread over 8 chunks) and continues in the ROW layout: a lane owns UNIT = 4 or
// 8 consecutive columns of one row, and bias / LayerScale / aux / residual / old C / C / C_pre all move as 16-byte (or
// 8-byte: bf16 x 4) pieces of whole row segments.  UNIT = 8 when every epilogue tensor is bf16 (p.epi_unit).
//
// Plain launches store through a buffer descriptor whose range check drops the lanes past M / N: every wave ISSUES the
// same number of store instructions per output tensor whatever the tile's position (pipe_store_count), so the k-loop of
// the next item can wait for its prefetched operands with a counted vmcnt(stores) instead of draining the stores as
// well (vmcnt counts loads and stores together, in issue order).  k-split launches add their strips with float atomics
// of 64 consecutive floats of one row per wave-instruction (atomics run at full rate only on 256 contiguous bytes) or
// store them to the workspace slice of the item.
typedef unsigned pu32x4 __attribute__((ext_vector_type(4)));
typedef unsigned pu32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 pbf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t pipe_rsrc(const void* base, long bytes) { return make_rsrc(base, bytes); }
// U consecutive elements (fp32 or bf16 per `type`) at element index i of `base` -> v[0 .. U/4)
template <int U>
__device__ __forceinline__ void pipe_load(const char* __restrict__ base, long i, int type, f32x4 (&v)[U / 4]) {
    if (type == CALM_ST_BF16) {
        if constexpr (U == 8) {
            const pbf16x8 h = *reinterpret_cast<const pbf16x8*>(base + i * 2);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e >> 2][e & 3] = (float)h[e];
        } else {
            const pbf16x4 h = *reinterpret_cast<const pbf16x4*>(base + i * 2);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[0][e] = (float)h[e];
        }
    } else {
#pragma unroll
        for (int q = 0; q < U / 4; ++q) v[q] = *reinterpret_cast<const f32x4*>(base + i * 4 + 16 * q);
    }
}
// store through the descriptor at byte offset `off` (0xFFFFFFFF: dropped by the range check); exactly ONE instruction
// per call for bf16 (U = 4: 8 bytes, U = 8: 16 bytes) and U / 4 for fp32
template <int U>
__device__ __forceinline__ void pipe_store(__amdgpu_buffer_rsrc_t rs, unsigned off, const f32x4 (&v)[U / 4], int type) {
    if (type == CALM_ST_BF16) {
        if constexpr (U == 8) {
            pbf16x8 h;
#pragma unroll
            for (int e = 0; e < 8; ++e) h[e] = (__bf16)v[e >> 2][e & 3];
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pu32x4, h), rs, off, 0, 0);
        } else {
            const pbf16x4 h = {(__bf16)v[0][0], (__bf16)v[0][1], (__bf16)v[0][2], (__bf16)v[0][3]};
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(pu32x2, h), rs, off, 0, 0);
        }
    } else {
#pragma unroll
        for (int q = 0; q < U / 4; ++q)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pu32x4, v[q]), rs, off == 0xFFFFFFFFu ? off : off + 16 * q, 0, 0);
    }
}
// store instructions one wave issues per output tensor and item (UNIT 8 implies bf16 tensors: one instruction per unit)
template <int MT, int NT>
__host__ __device__ constexpr int pipe_store_count(int unit) {
    return unit == 8 ? MT * ((16 * (16 * NT / 8) + 63) / 64) : MT * NT;
}

// TENSORS: the epilogue reads C-shaped tensors (aux, residual, old C); without them the only vector-memory
// instructions of an item's epilogue are its stores.  The per-column operands (bias, LayerScale) of the tile are in LDS
// (`colv`: bias[256] then scale[256], written by the kernel before the epilogue's barrier).  The unit loop is a rolled
// loop on purpose: unrolled, its address arithmetic for every unit is hoisted above the first strip, where all the
// accumulators are still live, and spills — and a scratch reload is a vector-memory load whose wait also waits for
// every older store.
template <int MT, int NT, int UNIT, bool TENSORS>
__device__ __forceinline__ void pipe_epilogue_rows(const GemmP& p, f32x4 (&acc)[MT][NT], int row0, int col0, int tile_col0,
                                                   int lane, float scale, char* __restrict__ Cb, char* __restrict__ Pb,
                                                   const char* __restrict__ Xb, const char* __restrict__ Rb,
                                                   float* __restrict__ sc, const float* __restrict__ colv) {
    constexpr int Q = UNIT / 4;
    constexpr int UPR = 16 * NT / UNIT;                 // units per strip row
    constexpr int TOTAL = 16 * UPR, NI = (TOTAL + 63) / 64;
    const int pM = p.M, pN = p.N, p_act = p.act;
    const bool p_accumulate = TENSORS && p.accumulate;
    const bool has_colv = p.bias || p.col_scale;
    const int c_type = p.c_type, aux_type = p.aux_type, r_type = p.r_type;
    const int csz = c_type == CALM_ST_BF16 ? 2 : 4;
    const long c_rs = p.c_rs, r_rs = p.r_rs;
    const int rl = lane & 15, g = lane >> 4;
    const long span = ((long)(pM - 1) * c_rs + pN) * csz;
    const __amdgpu_buffer_rsrc_t rs_c = pipe_rsrc(Cb, span);
    const __amdgpu_buffer_rsrc_t rs_p = pipe_rsrc(Pb ? Pb : Cb, Pb ? span : 0);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            *reinterpret_cast<f32x4*>(sc + rl * 128 + (((4 * nt + g) ^ (rl & 7)) << 2)) = acc[mt][nt] * scale;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int srow = row0 + 16 * mt;
        auto unit_of = [&](int i, int& r, int& col, bool& live) __attribute__((always_inline)) {
            const int idx_raw = 64 * i + lane;
            const int idx = min(idx_raw, TOTAL - 1);
            r = idx / UPR;
            col = col0 + UNIT * (idx - r * UPR);
            live = (TOTAL % 64 == 0 || idx_raw < TOTAL) && srow + r < pM && col < pN;
        };
        // t: the unit's C-shaped operand — at most ONE of aux / residual / old C per launch (dispatcher)
        auto finish = [&](int r, int col, bool live, f32x4 (&v)[Q], const f32x4 (&t)[Q]) __attribute__((always_inline)) {
            const unsigned so = live ? (unsigned)(((long)(srow + r) * c_rs + col) * csz) : 0xFFFFFFFFu;
            if (has_colv) {
#pragma unroll
                for (int q = 0; q < Q; ++q) v[q] += *reinterpret_cast<const f32x4*>(colv + (col - tile_col0) + 4 * q);
            }
            if (Pb) pipe_store<UNIT>(rs_p, so, v, c_type);
            if (p_act == CALM_ACT_GELU) {
#pragma unroll
                for (int q = 0; q < Q; ++q)
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[q][c] = gelu_erf_f(v[q][c]);
            } else if (TENSORS && p_act == CALM_ACT_GELU_BWD) {
#pragma unroll
                for (int q = 0; q < Q; ++q)
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[q][c] *= gelu_erf_grad_f(t[q][c]);
            }
            if (has_colv) {
#pragma unroll
                for (int q = 0; q < Q; ++q) v[q] *= *reinterpret_cast<const f32x4*>(colv + 256 + (col - tile_col0) + 4 * q);
            }
            if (TENSORS && (Rb || p_accumulate)) {
#pragma unroll
                for (int q = 0; q < Q; ++q) v[q] += t[q];
            }
            pipe_store<UNIT>(rs_c, so, v, c_type);
        };
        if constexpr (TENSORS) {
            // every C-shaped operand of the strip is requested before the first one is used: one memory round trip per
            // strip instead of one per unit (the wait for a load also waits for every older store, so loads issued
            // between the stores would each pay the store latency)
            // The storage-type branch encloses the whole batch of loads: with the branch inside pipe_load (round 3) the
            // bf16 side compiled to load / s_waitcnt vmcnt(0) / convert PER UNIT (ISA of <true,false,4,7>: 16 exposed
            // round trips per tile, each of which also drained every older store).
            f32x4 t[NI][Q];
            const char* __restrict__ Tb = p_act == CALM_ACT_GELU_BWD ? Xb : Rb ? Rb : Cb;
            const long t_rs = Rb && p_act != CALM_ACT_GELU_BWD ? r_rs : c_rs;
            const int t_type = p_act == CALM_ACT_GELU_BWD ? aux_type : Rb ? r_type : c_type;
            long toff[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                int r, col;
                bool live;
                unit_of(i, r, col, live);
                const long ro = min(srow + r, pM - 1);
                toff[i] = ro * t_rs + (col < pN ? col : 0);
            }
            if (t_type == CALM_ST_BF16) {
                typedef __bf16 praw __attribute__((ext_vector_type(UNIT)));
                praw h[NI];
#pragma unroll
                for (int i = 0; i < NI; ++i) h[i] = *reinterpret_cast<const praw*>(Tb + toff[i] * 2);
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int e = 0; e < UNIT; ++e) t[i][e >> 2][e & 3] = (float)h[i][e];
            } else {
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int q = 0; q < Q; ++q) t[i][q] = *reinterpret_cast<const f32x4*>(Tb + toff[i] * 4 + 16 * q);
            }
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                int r, col;
                bool live;
                unit_of(i, r, col, live);
                const int j = (col - col0) / UNIT;
                f32x4 v[Q];
#pragma unroll
                for (int q = 0; q < Q; ++q) v[q] = *reinterpret_cast<const f32x4*>(sc + r * 128 + (((Q * j + q) ^ (r & 7)) << 2));
                finish(r, col, live, v, t[i]);
            }
        } else {
#pragma unroll 1
            for (int i = 0; i < NI; ++i) {
                int r, col;
                bool live;
                unit_of(i, r, col, live);
                const int j = (col - col0) / UNIT;
                f32x4 v[Q];
#pragma unroll
                for (int q = 0; q < Q; ++q) v[q] = *reinterpret_cast<const f32x4*>(sc + r * 128 + (((Q * j + q) ^ (r & 7)) << 2));
                finish(r, col, live, v, v);
            }
        }
        // the next strip's scratch writes are issued after these reads: the LDS executes a wave's accesses in order
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

template <int MT, int NT>
__device__ __forceinline__ void pipe_epilogue(const GemmP& p, f32x4 (&acc)[MT][NT], int row0, int col0, int z,
                                              int yslice, char* __restrict__ scratch, const float* __restrict__ colv,
                                              int tile_col0) {
    // the lane id is re-derived here (v_mbcnt): taken from threadIdx at kernel entry it — and everything computed from
    // it — would be kept alive across the k-loop, i.e. spilled, and a scratch reload is a vector-memory load whose
    // wait also waits for every older store
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    float scale = p.alpha;
    if (p.inv_scale) scale = scale / p.inv_scale[0];
    const int zc = (p.atomic && !p.slices_per_batch) ? 0 : z;
    const int cb0 = zc / p.batch1, cb1 = zc - cb0 * p.batch1;
    const long coff = cb0 * p.c_b0 + cb1 * p.c_b1;
    const int csz = p.c_type == CALM_ST_BF16 ? 2 : 4;
    char* __restrict__ Cb = reinterpret_cast<char*>(p.C) + coff * csz;
    if (p.n_group) {
        scale = scale / group_sigma(p, cb0);
        if (p.Cg[0]) Cb = reinterpret_cast<char*>(p.Cg[cb0]) + cb1 * p.c_b1 * csz;
    }
    float* __restrict__ sc = reinterpret_cast<float*>(scratch);
    if (p.atomic) {
        const int pM = p.M, pN = p.N;
        const long c_rs = p.c_rs;
        const int rl = lane & 15, g = lane >> 4;
        float* __restrict__ Cf = p.ws ? p.ws + (long)yslice * p.ws_slice : reinterpret_cast<float*>(Cb);
        const long ld = p.ws ? (long)pN : c_rs;
        const bool plain_store = p.ws != nullptr;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                *reinterpret_cast<f32x4*>(sc + rl * 128 + (((4 * nt + g) ^ (rl & 7)) << 2)) = acc[mt][nt] * scale;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + 16 * mt + r;
#pragma unroll
                for (int c0 = 0; c0 < 16 * NT; c0 += 64) {
                    const int cl = c0 + lane, col = col0 + cl;
                    if (row < pM && cl < 16 * NT && col < pN) {
                        const float x = sc[r * 128 + (cl ^ ((r & 7) << 2))];
                        if (plain_store) Cf[(long)row * ld + col] = x;
                        else atomicAdd(Cf + (long)row * ld + col, x);
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        return;
    }
    char* __restrict__ Pb = p.C_pre ? reinterpret_cast<char*>(p.C_pre) + coff * csz : nullptr;
    const char* __restrict__ Xb =
        p.aux ? reinterpret_cast<const char*>(p.aux) + coff * (p.aux_type == CALM_ST_BF16 ? 2 : 4) : nullptr;
    const char* __restrict__ Rb = p.residual ? reinterpret_cast<const char*>(p.residual) +
                                                   (cb0 * p.r_b0 + cb1 * p.r_b1) * (p.r_type == CALM_ST_BF16 ? 2 : 4)
                                             : nullptr;
    const bool tensors = Xb || Rb || p.accumulate;
    if (p.epi_unit == 8) {
        if (tensors) pipe_epilogue_rows<MT, NT, 8, true>(p, acc, row0, col0, tile_col0, lane, scale, Cb, Pb, Xb, Rb, sc, colv);
        else pipe_epilogue_rows<MT, NT, 8, false>(p, acc, row0, col0, tile_col0, lane, scale, Cb, Pb, Xb, Rb, sc, colv);
    } else {
        if (tensors) pipe_epilogue_rows<MT, NT, 4, true>(p, acc, row0, col0, tile_col0, lane, scale, Cb, Pb, Xb, Rb, sc, colv);
        else pipe_epilogue_rows<MT, NT, 4, false>(p, acc, row0, col0, tile_col0, lane, scale, Cb, Pb, Xb, Rb, sc, colv);
    }
}

template <typename E, bool AKC, bool BKC, int MT, int NT>
__device__ __forceinline__ void pipe_body(const GemmP& p) {
    constexpr int BM_ = 64 * MT, BN_ = 32 * NT;
    constexpr int ES = sizeof(E), PBK = 128 / ES;              // k per k-tile: 64 bf16 / 32 fp32
    typedef PStage<AKC, BM_, ES> StA;
    typedef PStage<BKC, BN_, ES> StB;
    typedef typename pfrag_type<ES>::type frag_t;
    __shared__ __attribute__((aligned(1024))) char lds[2 * PSTAGE + 2048];     // two operand stages + the tile's bias / LayerScale

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    PFrag<AKC, BM_, ES> fa;
    PFrag<BKC, BN_, ES> fb;
    fa.template init<MT>(wm * MT, lane);
    fb.template init<NT>(wn * NT, lane);

    const int tiles = p.tiles_m * p.tiles_n;
    const int n_items = tiles * p.nz;
    const bool ktail = (p.K % PBK) != 0;

    StA sa;
    StB sb;
    // decode item -> (tile, z); XCD-aware: the items dealt to one XCD (equal index mod 8) are consecutive tiles
    auto decode = [&](int item, int& m0, int& n0, int& z, int& zy, int& kb0, int& kb1) {
        int lin = item;
        if (n_items >= 8) {
            const int q = n_items >> 3, rem = n_items & 7, x = item & 7, idx = item >> 3;
            lin = (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + idx;
        }
        const int t = lin % tiles;
        zy = lin / tiles;
        m0 = (t / p.tiles_n) * BM_;
        n0 = (t % p.tiles_n) * BN_;
        if (p.slices_per_batch) {
            z = zy / p.slices_per_batch;
            const int sl = zy - z * p.slices_per_batch;
            kb0 = sl * p.kb_per_z;
            kb1 = min(kb0 + p.kb_per_z, p.kpb);
        } else if (p.atomic) {
            z = 0;
            kb0 = zy * p.kb_per_z;
            kb1 = min(kb0 + p.kb_per_z, p.kpb);
        } else {
            z = zy;
            kb0 = 0;
            kb1 = p.kpb;
        }
    };
    auto setup = [&](int m0, int n0, int z, int kb0) {
        const int b0 = z / p.batch1, b1 = z - b0 * p.batch1;
        sa.init(operand_base<E>(p.A, p.Ag, p.n_group, p.a_b0, p.a_b1, b0, b1), p.a_rs, p.a_cs, m0, p.M, kb0 * PBK, wave,
                lane);
        sb.init(operand_base<E>(p.B, p.Bg, p.n_group, p.b_b0, p.b_b1, b0, b1), p.b_rs, p.b_cs, n0, p.N, kb0 * PBK, wave,
                lane);
    };
    // k-tile kb of the item being staged -> stage st
    auto stage_tile = [&](int kb, int kb_last_of_entry, unsigned st) {
        const unsigned dst = st * PSTAGE;
        if (ktail && kb == kb_last_of_entry) {
            sa.issue_tail(dst, wave, lane, p.K - kb * PBK);
            sb.issue_tail(dst + PB_OFF, wave, lane, p.K - kb * PBK);
        } else {
            sa.issue(dst, wave);
            sb.issue(dst + PB_OFF, wave);
        }
    };

    int item = blockIdx.x;
    if (item >= n_items) return;
    // optional start stagger (p.stagger x 127 x 64 cycles per phase step): workgroups that start together stay in
    // lockstep and hit HBM with their epilogues at the same time
    for (int i = ((blockIdx.x >> 3) & 3) * (p.stagger < 50 ? p.stagger : 0); i > 0; --i) __builtin_amdgcn_s_sleep(127);
    int m0, n0, z, zy, kb0, kb1;
    decode(item, m0, n0, z, zy, kb0, kb1);
    setup(m0, n0, z, kb0);
    unsigned st = 0;
    stage_tile(kb0, p.kpb - 1, st);
    int stores_behind = 0;             // store instructions this wave has issued since its last operand request

#ifdef CALM_PIPE_STAMP
    int stamp_item = 0;
#endif
    while (true) {
#ifdef CALM_PIPE_STAMP
        const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
        unsigned long long t_wait = 0;
#endif
        f32x4 acc[MT][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        const int next_item = item + gridDim.x;
        int nm0 = 0, nn0 = 0, nz_ = 0, nzy = 0, nkb0 = 0, nkb1 = 0;
        const bool has_next = next_item < n_items;
        if (has_next) decode(next_item, nm0, nn0, nz_, nzy, nkb0, nkb1);

        for (int kb = kb0; kb < kb1; ++kb) {
            // this wave's share of k-tile kb has landed; after the barrier everybody's has, and nobody reads the
            // other stage any more.  First k-tile of an item after a plain epilogue: the operands were requested BEFORE
            // the epilogue's stores, so waiting until only the stores are outstanding is enough
#ifdef CALM_PIPE_STAMP
            const unsigned long long tw0 = __builtin_amdgcn_s_memtime();
#endif
            constexpr int S4 = pipe_store_count<MT, NT>(4), S8 = pipe_store_count<MT, NT>(8);
            if (kb != kb0 || stores_behind == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (stores_behind == S4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(S4 > 63 ? 63 : S4) : "memory");
            else if (stores_behind == 2 * S4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * S4 > 63 ? 63 : 2 * S4) : "memory");
            else if (stores_behind == S8) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(S8 > 63 ? 63 : S8) : "memory");
            else if (stores_behind == 2 * S8) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * S8 > 63 ? 63 : 2 * S8) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
#ifdef CALM_PIPE_STAMP
            t_wait += __builtin_amdgcn_s_memtime() - tw0;
#endif
            // the next k-tile (of this item, or the first one of the next item: it lands during the epilogue) is
            // requested piece by piece between the MFMA groups of k-step 0: an LDS-DMA instruction costs its wave 60-180
            // cycles of issue, which the matrix pipe spends on the group issued just before
            int k_left = 0;                                   // 0: nothing to stage
            if (kb + 1 < kb1) {
                k_left = (ktail && kb + 1 == p.kpb - 1) ? p.K - (kb + 1) * PBK : PBK;
            } else if (has_next) {
                setup(nm0, nn0, nz_, nkb0);
                k_left = (ktail && nkb0 == p.kpb - 1) ? p.K - nkb0 * PBK : PBK;
            }
#ifdef CALM_PIPE_STAMP
            if (p.stagger >= 77) k_left = 0;                 // diagnostic: k-loop without its operand traffic (wrong results)
#endif
            const unsigned dst = (st ^ 1) * PSTAGE;
            const char* __restrict__ ia = lds + st * PSTAGE;
            const char* __restrict__ ib = ia + PB_OFF;
            constexpr int NPA = StA::PER_WAVE, NPB = StB::PER_WAVE, NPT = NPA + NPB, PPG = (NPT + MT - 1) / MT;
            const unsigned long long ua = pipe_uniform64(sa.base), ub = pipe_uniform64(sb.base);
            auto stage_group = [&](auto gtag, auto tail_tag) __attribute__((always_inline)) {
                constexpr int G = decltype(gtag)::value;
                constexpr bool TAIL = decltype(tail_tag)::value;
                {
                    if constexpr (G * PPG + 0 < NPT) {
                        constexpr int P0 = G * PPG + 0;
                        if constexpr (P0 < NPA) sa.template issue_piece<P0, TAIL>(dst, wave, lane, k_left, ua);
                        else sb.template issue_piece<P0 - NPA, TAIL>(dst + PB_OFF, wave, lane, k_left, ub);
                    }
                    if constexpr (PPG > 1 && G * PPG + 1 < NPT && 1 < PPG) {
                        constexpr int P1 = G * PPG + 1;
                        if constexpr (P1 < NPA) sa.template issue_piece<P1, TAIL>(dst, wave, lane, k_left, ua);
                        else sb.template issue_piece<P1 - NPA, TAIL>(dst + PB_OFF, wave, lane, k_left, ub);
                    }
                    if constexpr (PPG > 2 && G * PPG + 2 < NPT) {
                        constexpr int P2 = G * PPG + 2;
                        if constexpr (P2 < NPA) sa.template issue_piece<P2, TAIL>(dst, wave, lane, k_left, ua);
                        else sb.template issue_piece<P2 - NPA, TAIL>(dst + PB_OFF, wave, lane, k_left, ub);
                    }
                    if constexpr (PPG > 3 && G * PPG + 3 < NPT) {
                        constexpr int P3 = G * PPG + 3;
                        if constexpr (P3 < NPA) sa.template issue_piece<P3, TAIL>(dst, wave, lane, k_left, ua);
                        else sb.template issue_piece<P3 - NPA, TAIL>(dst + PB_OFF, wave, lane, k_left, ub);
                    }
                }
            };
            auto stage_pieces = [&](auto gtag) __attribute__((always_inline)) {
                if (k_left >= PBK) stage_group(gtag, std::false_type{});
                else if (k_left > 0) stage_group(gtag, std::true_type{});
            };
            static_assert(PPG <= 4, "at most four staging pieces per MFMA group");
            // Fragment reads run AHEAD of the MFMAs that use them; sched_barrier pins the order [staging pieces, prefetch
            // reads | MFMA group]: left alone, the scheduler sinks every prefetch behind the MFMA group in front of it to
            // save registers, and every group then starts with an exposed LDS round trip.
#ifndef CALM_PIPE_DEEP
#define CALM_PIPE_DEEP 0      // measured on 57344 x 672 x 672: 36.9k vs 37.0k cycles per k-loop, at +40 VGPRs — off
#endif
            constexpr bool DEEP = CALM_PIPE_DEEP && ES == 2 && MT * NT <= 28;    // registers for two full fragment sets beside the accumulators
            if constexpr (DEEP) {
                // k-step 1's fragments (MT + NT reads) are requested while k-step 0's MT NT products issue — a whole
                // k-step of lookahead (an LDS read under load takes ~300 cycles, an MFMA group NT x 16): one exposed LDS
                // latency per k-tile, the one right after the barrier
                frag_t a0[MT], b0[NT], a1[MT], b1[NT];
#pragma unroll
                for (int j = 0; j < NT; ++j) b0[j] = fb.load(ib, j, 0);
#pragma unroll
                for (int i = 0; i < MT; ++i) a0[i] = fa.load(ia, i, 0);
                constexpr int RPG = (MT + NT + MT - 1) / MT;          // k-step 1 reads issued per k-step 0 group
                auto group0 = [&](auto itag) __attribute__((always_inline)) {
                    constexpr int i = decltype(itag)::value;
                    stage_pieces(std::integral_constant<int, i>{});
#pragma unroll
                    for (int r = 0; r < RPG; ++r) {
                        constexpr int dummy = 0;
                        const int idx = i * RPG + r;                  // compile-time after unrolling
                        if (idx < NT) b1[idx] = fb.load(ib, idx, 1);
                        else if (idx < NT + MT) a1[idx - NT] = fa.load(ia, idx - NT, 1);
                        (void)dummy;
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < NT; ++j) pipe_mma(b0[j], a0[i], acc[i][j]);
                    __builtin_amdgcn_sched_barrier(0);
                };
                group0(std::integral_constant<int, 0>{});
                group0(std::integral_constant<int, 1>{});
                if constexpr (MT > 2) group0(std::integral_constant<int, 2>{});
                if constexpr (MT > 3) group0(std::integral_constant<int, 3>{});
#pragma unroll
                for (int i = 0; i < MT; ++i) {
#pragma unroll
                    for (int j = 0; j < NT; ++j) pipe_mma(b1[j], a1[i], acc[i][j]);
                }
                __builtin_amdgcn_sched_barrier(0);
            } else {
            // (one group of lookahead: the A fragment of the next group, and the B fragments of k-step 1 replace those of
            // k-step 0 one by one behind their last use)
            frag_t bf[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) bf[j] = fb.load(ib, j, 0);
            frag_t a_cur = fa.load(ia, 0, 0);
            auto group = [&](auto kstag, auto itag) __attribute__((always_inline)) {
                constexpr int ks = decltype(kstag)::value, i = decltype(itag)::value;
#if CALM_PIPE_PRIO_FLIP
                // the two waves of a SIMD trade priority once per k-step: with age-based arbitration alone the older wave
                // runs ahead, reaches the barrier early and leaves the younger one to finish the k-tile alone
                if constexpr (i == 0) {
                    if ((wave < 4) == (ks == 0)) __builtin_amdgcn_s_setprio(1);
                    else __builtin_amdgcn_s_setprio(0);
                }
#endif
                if constexpr (ks == 0) stage_pieces(std::integral_constant<int, i>{});
                frag_t a_next = a_cur;
                if constexpr (i + 1 < MT) a_next = fa.load(ia, i + 1, ks);
                else if constexpr (ks == 0) a_next = fa.load(ia, 0, 1);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (ES == 2) {
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        pipe_mma(bf[j], a_cur, acc[i][j]);
                        if constexpr (i == MT - 1 && ks == 0) bf[j] = fb.load(ib, j, 1);
                    }
                } else {
                    // four 16x16x4 products per tile and k-step, element by element across the NT tiles: consecutive
                    // MFMAs never share an accumulator
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int j = 0; j < NT; ++j) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j][e], a_cur[e], acc[i][j], 0, 0, 0);
                            if constexpr (i == MT - 1 && ks == 0) {
                                if (e == 3) bf[j] = fb.load(ib, j, 1);
                            }
                        }
                }
                __builtin_amdgcn_sched_barrier(0);
                a_cur = a_next;
            };
            group(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
            group(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
            if constexpr (MT > 2) group(std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
            if constexpr (MT > 3) group(std::integral_constant<int, 0>{}, std::integral_constant<int, 3>{});
            group(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
            group(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
            if constexpr (MT > 2) group(std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{});
            if constexpr (MT > 3) group(std::integral_constant<int, 1>{}, std::integral_constant<int, 3>{});
            }
            if (k_left) {
                sa.advance();
                sb.advance();
            }
            st ^= 1;
        }
#ifdef CALM_PIPE_STAMP
        const unsigned long long ts1 = __builtin_amdgcn_s_memtime();
#endif
        // the epilogue turns its strips through the stage that was read last: every wave must be done with it (the
        // stage the next item's first k-tile is landing in is the other one; the k-loop's first barrier keeps the next
        // staging out of the scratch until every wave has left its epilogue)
        float* colv = reinterpret_cast<float*>(lds + 2 * PSTAGE);
        if ((p.bias || p.col_scale) && tid < 256) {              // per-column operands of the tile (no k-split here)
            const int col = n0 + tid;
            colv[tid] = (p.bias && col < p.N) ? p.bias[col] : 0.f;
            colv[256 + tid] = (p.col_scale && col < p.N) ? p.col_scale[col] : 1.f;
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        pipe_epilogue<MT, NT>(p, acc, m0 + wm * (16 * MT), n0 + wn * (16 * NT), z, zy,
                              lds + (st ^ 1) * PSTAGE + wave * 8192, colv, n0);
        // fp32 tensors with UNIT 4 issue one 16-byte store per unit as well (Q = 1)
        stores_behind = p.atomic ? 0 : pipe_store_count<MT, NT>(p.epi_unit) * (p.C_pre ? 2 : 1);
#ifdef CALM_PIPE_STAMP
        if (!p.atomic && p.ws && lane == 0 && (wave == 0 || wave == 7) && stamp_item < 8) {
            const unsigned long long ts2 = __builtin_amdgcn_s_memtime();
            unsigned long long* d = reinterpret_cast<unsigned long long*>(p.ws) + ((blockIdx.x * 2 + (wave == 7)) * 8 + stamp_item) * 4;
            d[0] = ts1 - ts0; d[1] = t_wait; d[2] = ts2 - ts1; d[3] = ts0;
        }
        ++stamp_item;
#endif
        if (!has_next) break;
        item = next_item;
        m0 = nm0; n0 = nn0; z = nz_; zy = nzy; kb0 = nkb0; kb1 = nkb1;
    }
}

template <bool AKC, bool BKC, int MT, int NT>
__global__ __launch_bounds__(PTHREADS, 2) void gemm_bf16p_kernel(const GemmP p) {
    pipe_body<__bf16, AKC, BKC, MT, NT>(p);
}
// the same pipeline on fp32 tensors and v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulate): k-tiles of 32
template <bool AKC, bool BKC, int MT, int NT>
__global__ __launch_bounds__(PTHREADS, 2) void gemm_f32p_kernel(const GemmP p) {
    pipe_body<float, AKC, BKC, MT, NT>(p);
}

#define CALM_PIPE_LAUNCH(KERNEL, NTV) \
    case NTV: hipLaunchKernelGGL((KERNEL<AKC, BKC, MT, NTV>), dim3(grid), dim3(PTHREADS), 0, s, p); break;

template <bool F32, bool AKC, bool BKC, int MT>
int launch_pipe_nt(const GemmP& p, int nt, int grid, hipStream_t s) {
    if constexpr (F32) {
        switch (nt) {
            CALM_PIPE_LAUNCH(gemm_f32p_kernel, 4) CALM_PIPE_LAUNCH(gemm_f32p_kernel, 5) CALM_PIPE_LAUNCH(gemm_f32p_kernel, 6)
            CALM_PIPE_LAUNCH(gemm_f32p_kernel, 7) CALM_PIPE_LAUNCH(gemm_f32p_kernel, 8)
        default: return CALM_E_UNSUPP;
        }
    } else {
        switch (nt) {
            CALM_PIPE_LAUNCH(gemm_bf16p_kernel, 4) CALM_PIPE_LAUNCH(gemm_bf16p_kernel, 5) CALM_PIPE_LAUNCH(gemm_bf16p_kernel, 6)
            CALM_PIPE_LAUNCH(gemm_bf16p_kernel, 7) CALM_PIPE_LAUNCH(gemm_bf16p_kernel, 8)
        default: return CALM_E_UNSUPP;
        }
    }
    CALM_LAUNCH_CHECK();
    return 0;
}
template <bool F32, bool AKC, bool BKC>
int launch_pipe_layout(const GemmP& p, int mt, int nt, int grid, hipStream_t s) {
    if (mt == 2) return launch_pipe_nt<F32, AKC, BKC, 2>(p, nt, grid, s);
    if (mt == 3) return launch_pipe_nt<F32, AKC, BKC, 3>(p, nt, grid, s);
    if (mt == 4) return launch_pipe_nt<F32, AKC, BKC, 4>(p, nt, grid, s);
    return CALM_E_UNSUPP;
}

}  // namespace calm_gemm_detail
