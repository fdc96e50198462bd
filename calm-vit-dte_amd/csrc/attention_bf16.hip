// Fused cross-axial latent-mask attention on the bf16 matrix pipe (Vi_Tools_CNN_less_V2.py:288-299 under
// autocast(bfloat16), distributed_trainer_cls.py:84) — the kernels of the bf16 pipeline.  All tensors of the
// attention are bf16 in HBM (q, k, v, out, the saved R / hidden / mask), accumulation and softmax are fp32.
//
//   R      = Q_all K_all^T                      (all heads concatenated, raw, un-scaled)     [S,S]
//   M      = W2 gelu(W1 R^T + b1) + b2          (2-layer MLP along the KEY axis, W/sigma)    [S,S]
//   O_h    = softmax_j(Q_h K_h^T / sqrt(hd) + M) V_h                                         per head
//
// Same dataflow as the fp32 kernel (attention_fused.hip): one wave = 16 queries on the MFMA lane/column index, the
// keys on the accumulator rows ("transposed" orientation), so every accumulator tile is directly the B operand of the
// next product and the softmax row reduction is in-lane + two shuffles.  With v_mfma_f32_16x16x32_bf16 a k-step is 32
// deep = TWO 16-row accumulator tiles: lane (c = lane & 15, g = lane >> 4) holds rows 4g..4g+3 of both tiles, which
// defines the k order inside a step,
//       element e of lane group g  <->  k = 16 (e >> 2) + 4 g + (e & 3)          ("paired-tile order"),
// and the A operand of that product is fetched in the same order: two 8-byte reads per lane from a k-contiguous
// image (W1 / W2 chunks), or two ds_read_b64_tr_b16 from a row-major [key][d] image (V_h for P·V, K_h for dS·K).
// Products whose k axis is a feature axis (Q K^T, V dO^T) read ordinary 16-byte fragments (k = 8 g + e).
// Keys are padded to a multiple of 32 (pad keys get mask -inf), head dims to a multiple of 32 (zero columns).
// The probabilities never exist in HBM: the forward saves the row log-sum-exp, the backward recomputes P.
//
// LDS images (row strides from scripts/micro/lds_bank_sim.py): [row][k] images read by ds_read_b128 AND by
// ds_read_b64_tr_b16 use a stride = 32 (mod 64) bytes — both conflict-free; images read in paired-tile order
// (ds_read_b64) use a stride = 16 (mod 32) bytes.
#include "common.h"
#include "lds_dma.h"
#include <stdlib.h>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ bf16x4 ld4(const __bf16* p) { return *reinterpret_cast<const bf16x4*>(p); }
__device__ __forceinline__ bf16x8 cat8(bf16x4 a, bf16x4 b) {
    return (bf16x8){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}
__device__ __forceinline__ bf16x4 pack4(const f32x4v& v) {
    return (bf16x4){(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
}
__device__ __forceinline__ f32x4v unpack4(bf16x4 v) { return (f32x4v){(float)v[0], (float)v[1], (float)v[2], (float)v[3]}; }
// transposed read: 16-lane group G reads a 4(row) x 16(col) block of a row-major 16-bit image and every lane gets its
// column's 4 rows.  `p` = address of (row 0 of the block, this lane's 4-column chunk)
__device__ __forceinline__ bf16x4 tr4(const __bf16* p) {
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
    return __builtin_bit_cast(bf16x4, v);
}

struct Attn16P {
    const __bf16* q; const __bf16* k; const __bf16* v;
    const __bf16* w1; const float* b1; const float* s1;
    const __bf16* w2; const float* b2; const float* s2;
    __bf16* out;
    __bf16* R; __bf16* hp; __bf16* hg; __bf16* Mk; __bf16* MkT; float* lse;      // saved for backward
    int B, S, H, hd;
    float scale;
    int kv_shared;      // K_h and V_h do not both fit in LDS: one image, V_h staged after the Q K^T products
    int groups;         // query groups (workgroups) per image — attn16_fwd2_kernel's 1-D grid
};

// 1-D grids of (groups x B) workgroups.  Workgroup ids go round-robin over the 8 XCDs; the row groups of ONE image are
// dealt to the same XCD next to each other in time, so that the second group's K / V / Q / dO requests hit the L2 the
// first one filled (PMC, round 3: the query-side backward read 1.3 GB per launch at 4.7 TB/s — every image's operands
// fetched once per group from HBM).  Needs B % 8 == 0; other batches keep the plain order.
__device__ __forceinline__ void image_and_group(int id, int groups, int B, int& b, int& grp) {
    if ((B & 7) == 0) {
        const int xcd = id & 7, idx = id >> 3;                           // idx-th workgroup of this XCD
        grp = idx % groups;
        b = (idx / groups) * 8 + xcd;
    } else {
        grp = id % groups;
        b = id / groups;
    }
}

constexpr size_t LDS_BUDGET = 80 * 1024;       // two workgroups per CU when a kernel stays below this

// stride (in bf16 elements) of a [row][cols] image that is read by 16-byte fragments and by transposed reads
__host__ __device__ constexpr int ld_rt(int cols) { return cols + 16; }        // cols % 32 == 0 -> bytes = 32 (mod 64)
// ... of an image read in paired-tile order (two 8-byte reads per lane)
__host__ __device__ constexpr int ld_pt(int cols) { return ((cols + 15) / 16 * 16) + 8; }   // bytes = 16 (mod 32)

// copy a [rows x cols] block (global row stride gstride, all in elements; cols % 4 == 0, 8-byte aligned rows) into an
// LDS image with row stride ld; rows >= rows_valid and columns >= cols_valid are zero-filled up to (rows_img, cols_img)
__device__ __forceinline__ void stage_block(__bf16* __restrict__ dst, int ld, const __bf16* __restrict__ src, long gstride,
                                            int rows_valid, int cols_valid, int rows_img, int cols_img) {
    const int per_row = cols_img >> 2;
    const int total = rows_img * per_row;
    for (int f = threadIdx.x; f < total; f += blockDim.x) {
        const int row = f / per_row, c4 = 4 * (f - row * per_row);
        bf16x4 v = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
        if (row < rows_valid && c4 < cols_valid) v = ld4(src + (long)row * gstride + c4);
        *reinterpret_cast<bf16x4*>(dst + row * ld + c4) = v;
    }
}

// B fragment straight from global memory: 8 consecutive features [c0 + 8 g, +8) of this lane's row (zero past cmax)
__device__ __forceinline__ bf16x8 row_frag(const __bf16* __restrict__ row, int c0, int g, int cmax) {
    const int c = c0 + 8 * g;
    const bf16x4 z = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
    const bf16x4 a = c < cmax ? ld4(row + c) : z;
    const bf16x4 b = c + 4 < cmax ? ld4(row + c + 4) : z;
    return cat8(a, b);
}

// waves per workgroup as a function of the key-tile pairs: the 2 NP query tiles of one image split evenly over at most
// 8-wave workgroups (compile-time, so that the per-thread share of every staged block is a constant).  Measured at
// S = 224 (14 tiles): 2 x 7 waves 601 us forward, 3 x 5 waves (two workgroups per CU, but 1.5 rounds of them) 867 us.
__host__ __device__ constexpr int waves_for(int np) {
    return (2 * np + (2 * np + 7) / 8 - 1) / ((2 * np + 7) / 8);
}

// Register-staged copy of a [rows_img x cols_img] block: `load` issues the global reads into NV 8-byte vectors per
// thread (zero-fill outside the valid region), `store` writes them to the LDS image — a chunk's loads are issued
// before the MFMAs of the previous chunk and only waited for when the next image is built.
template <int NV, int NTH>
struct BlockStage {
    bf16x4 v[NV];
    __device__ __forceinline__ void load(const __bf16* __restrict__ src, long gstride, int rows_valid, int cols_valid,
                                         int rows_img, int cols_img) {
        const int per_row = cols_img >> 2;
        const int total = rows_img * per_row;
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int f = threadIdx.x + u * NTH;
            const int row = f / per_row, c4 = 4 * (f - row * per_row);
            bf16x4 x = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
            if (f < total && row < rows_valid && c4 < cols_valid) x = ld4(src + (long)row * gstride + c4);
            v[u] = x;
        }
    }
    __device__ __forceinline__ void store(__bf16* __restrict__ dst, int ld, int rows_img, int cols_img) const {
        const int per_row = cols_img >> 2;
        const int total = rows_img * per_row;
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int f = threadIdx.x + u * NTH;
            const int row = f / per_row, c4 = 4 * (f - row * per_row);
            if (f < total) *reinterpret_cast<bf16x4*>(dst + row * ld + c4) = v[u];
        }
    }
};

// NP = pairs of 16-key tiles (keys padded to 32 NP), HDP = head dim padded to 32
// two workgroups share a CU (one's load / barrier stalls under the other's MFMAs) when both fit at 3 waves per SIMD
#ifndef ATT16_BWDQ_PRE_MAX
#define ATT16_BWDQ_PRE_MAX 448     // query-side backward: next head's K_h / V_h prefetched into registers up to this NP x HDP
#endif
#ifndef ATT16_BWDKV_PRE_MAX
#define ATT16_BWDKV_PRE_MAX 448
#endif
#ifndef ATT16_FWD_OCC3
#define ATT16_FWD_OCC3 0     // A/B'd: constraining the registers for a second workgroup per CU spills (S=176: 367 vs 240 us)
#endif
__host__ __device__ constexpr int fwd_waves_per_simd(int np) {
    return (ATT16_FWD_OCC3 && 2 * waves_for(np) <= 12 && np <= 6) ? 3 : 2;
}

template <int NP, int HDP>
__global__ __launch_bounds__(64 * waves_for(NP), fwd_waves_per_simd(NP)) void attn16_fwd_kernel(const Attn16P p) {
    constexpr int NJ = 2 * NP, SP = 32 * NP, NW = waves_for(NP), NTH = 64 * NW;
    constexpr bool KEEP_MASK = NP <= 8;
    extern __shared__ __attribute__((aligned(16))) __bf16 smem16[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c16 = lane & 15, g = lane >> 4;
    int b, qg;
    image_and_group(blockIdx.x, p.groups, p.B, b, qg);
    const int S = p.S, D = p.H * p.hd, hd = p.hd;
    const int q_lane = qg * (16 * NW) + 16 * wave + c16;                            // this lane's query
    const bool q_ok = q_lane < S;
    const int q_ld = q_ok ? q_lane : S - 1;                                         // clamped: loads stay in range
    const __bf16* qrow = p.q + ((long)b * S + q_ld) * D;
    const __bf16* kb = p.k + (long)b * S * D;
    const __bf16* vb = p.v + (long)b * S * D;

    f32x4v acc[NJ];
    bf16x8 Rf[NP];                       // R^T (then the mask) of this wave's queries as packed B fragments
#ifdef ATT16_STAMP
    const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
#endif

    // ================= phase 1: R^T[j,i] = sum_c K_all[j,c] Q_all[i,c], 64 columns per step =================
    // (round 3: 64 instead of 32 columns per step — every step ends in a barrier behind which the next chunk's global
    // loads are waited for, and its 14 MFMAs per wave are far shorter than that latency: in-kernel stamps gave 3.7k cycles
    // per 32-column step at S = 224, i.e. the phase is a chain of exposed load latencies; half as many links)
    {
#pragma unroll
        for (int t = 0; t < NJ; ++t) acc[t] = (f32x4v){0.f, 0.f, 0.f, 0.f};
        constexpr int CW = 64;
        constexpr int LDK = ld_rt(CW);
        constexpr int NV = (SP * (CW / 4) + NTH - 1) / NTH;
        __bf16* img = smem16;
        const int nch = (D + CW - 1) / CW;
        BlockStage<NV, NTH> st;
        st.load(kb, D, S, D, SP, CW);
        bf16x8 bq_next[CW / 32];
#pragma unroll
        for (int ks = 0; ks < CW / 32; ++ks) bq_next[ks] = row_frag(qrow, 32 * ks, g, D);
#pragma unroll 1
        for (int c = 0; c < nch; ++c) {
            __syncthreads();                                         // the previous chunk's reads are done
            st.store(img, LDK, SP, CW);
            bf16x8 bq[CW / 32];
#pragma unroll
            for (int ks = 0; ks < CW / 32; ++ks) bq[ks] = bq_next[ks];
            __syncthreads();
            if (c + 1 < nch) {                                       // next chunk's loads fly under this chunk's MFMAs
                st.load(kb + CW * (c + 1), D, S, D - CW * (c + 1), SP, CW);
#pragma unroll
                for (int ks = 0; ks < CW / 32; ++ks) bq_next[ks] = row_frag(qrow, CW * (c + 1) + 32 * ks, g, D);
            }
#pragma unroll
            for (int ks = 0; ks < CW / 32; ++ks)
#pragma unroll
                for (int t = 0; t < NJ; ++t) {
                    const bf16x8 a = *reinterpret_cast<const bf16x8*>(img + (16 * t + c16) * LDK + 32 * ks + 8 * g);
                    acc[t] = MFMA_BF16(a, bq[ks], acc[t]);
                }
        }
        __bf16* Rrow = p.R + ((long)b * S + q_ld) * S;
#pragma unroll
        for (int t = 0; t < NJ; ++t) {
            const int j = 16 * t + 4 * g;
            const bf16x4 r4 = pack4(acc[t]);
            if (q_ok && j < S) *reinterpret_cast<bf16x4*>(Rrow + j) = r4;
            if (t & 1) Rf[t >> 1] = cat8(pack4(acc[t - 1]), r4);
        }
    }

#ifdef ATT16_STAMP
    const unsigned long long ts1 = __builtin_amdgcn_s_memtime();
#endif
    // ================= phase 2: M^T = W2 gelu(W1 R^T + b1) + b2, 32 hidden units per step =================
    {
        const float inv1 = 1.0f / p.s1[0], inv2 = 1.0f / p.s2[0];
        const int NH = 2 * S;
        constexpr int LD1 = ld_pt(SP), LD2 = ld_pt(32);
        constexpr int NV = (SP * 8 + NTH - 1) / NTH;                 // both chunk images are SP x 32 elements
        __bf16* img1 = smem16;                                       // W1 chunk  [32 hidden][SP keys]
        __bf16* img2 = smem16 + 32 * LD1;                            // W2 chunk  [SP keys][32 hidden]
#pragma unroll
        for (int t = 0; t < NJ; ++t) acc[t] = (f32x4v){0.f, 0.f, 0.f, 0.f};
        const int nch = (NH + 31) / 32;
        BlockStage<NV, NTH> s1, s2;
        s1.load(p.w1, S, NH, S, 32, SP);
        s2.load(p.w2, NH, S, NH, SP, 32);
#pragma unroll 1
        for (int c = 0; c < nch; ++c) {
            const int n0 = 32 * c;
            __syncthreads();
            s1.store(img1, LD1, 32, SP);
            s2.store(img2, LD2, SP, 32);
            __syncthreads();
            if (c + 1 < nch) {
                s1.load(p.w1 + (long)(n0 + 32) * S, S, NH - n0 - 32, S, 32, SP);
                s2.load(p.w2 + n0 + 32, NH, S, NH - n0 - 32, SP, 32);
            }
            f32x4v h0 = {0.f, 0.f, 0.f, 0.f}, h1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int pr = 0; pr < NP; ++pr) {
                const __bf16* r0 = img1 + c16 * LD1 + 32 * pr + 4 * g;
                const __bf16* r1 = r0 + 16 * LD1;
                h0 = MFMA_BF16(cat8(ld4(r0), ld4(r0 + 16)), Rf[pr], h0);
                h1 = MFMA_BF16(cat8(ld4(r1), ld4(r1 + 16)), Rf[pr], h1);
            }
            // hidden units n0 + 4g + r (tile 0) and n0 + 16 + 4g + r (tile 1) of query c16
            const int na = n0 + 4 * g, nb = na + 16;
            const f32x4v ba = na < NH ? *reinterpret_cast<const f32x4v*>(p.b1 + na) : (f32x4v){0.f, 0.f, 0.f, 0.f};
            const f32x4v bb = nb < NH ? *reinterpret_cast<const f32x4v*>(p.b1 + nb) : (f32x4v){0.f, 0.f, 0.f, 0.f};
            f32x4v pa, pb, ga, gb;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pa[r] = h0[r] * inv1 + ba[r]; ga[r] = gelu_erf_f(pa[r]);
                pb[r] = h1[r] * inv1 + bb[r]; gb[r] = gelu_erf_f(pb[r]);
            }
            const bf16x4 ga4 = pack4(ga), gb4 = pack4(gb);
            if (q_ok) {
                const long ho = ((long)b * S + q_lane) * NH;
                if (na < NH) {
                    *reinterpret_cast<bf16x4*>(p.hp + ho + na) = pack4(pa);
                    *reinterpret_cast<bf16x4*>(p.hg + ho + na) = ga4;
                }
                if (nb < NH) {
                    *reinterpret_cast<bf16x4*>(p.hp + ho + nb) = pack4(pb);
                    *reinterpret_cast<bf16x4*>(p.hg + ho + nb) = gb4;
                }
            }
            const bf16x8 hf = cat8(ga4, gb4);        // hidden columns >= NH meet zero columns of the W2 image
#pragma unroll
            for (int t = 0; t < NJ; ++t) {
                const __bf16* r2 = img2 + (16 * t + c16) * LD2 + 4 * g;
                acc[t] = MFMA_BF16(cat8(ld4(r2), ld4(r2 + 16)), hf, acc[t]);
            }
        }
        // the mask as the backward will read it: rounded to bf16, -inf on the pad keys
        __bf16* Mrow = p.Mk + ((long)b * S + q_ld) * S;
        // (the bias chunks are requested four at a time, clamped instead of branched: one at a time the compiler put
        // `s_waitcnt vmcnt(0)` behind each of the NJ loads — ISA, round 4)
#pragma unroll
        for (int t0 = 0; t0 < NJ; t0 += 4) {
            f32x4v b2v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = 16 * (t0 + u) + 4 * g;
                if (t0 + u < NJ) b2v[u] = *reinterpret_cast<const f32x4v*>(p.b2 + (j < S ? j : 0));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = t0 + u;
                if (t < NJ) {
                    const int j = 16 * t + 4 * g;
                    f32x4v m;
#pragma unroll
                    for (int r = 0; r < 4; ++r) m[r] = j < S ? acc[t][r] * inv2 + b2v[u][r] : -INFINITY;
                    const bf16x4 m4 = pack4(m);
                    if (q_ok && j < S) *reinterpret_cast<bf16x4*>(Mrow + j) = m4;
                    acc[t] = unpack4(m4);
                    if (t & 1) Rf[t >> 1] = cat8(pack4(acc[t - 1]), m4);
                }
            }
        }
        if constexpr (!KEEP_MASK) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // own mask stores before the re-reads
    }

#ifdef ATT16_STAMP
    const unsigned long long ts2 = __builtin_amdgcn_s_memtime();
#endif
    // ================= phase 3: per head  softmax(scale K_h Q_h^T + M^T),  O^T = V_h^T P^T =================
    constexpr int hdp = HDP, LDH = ld_rt(HDP), nks = HDP / 32, ndt = HDP / 16;
    // K_h / V_h of the NEXT head are fetched into registers while this head computes — where both images are resident
    // and the staging registers fit beside the accumulators without spilling (NP x HDP <= 640)
    constexpr bool KV_SHARED = (size_t)2 * SP * LDH * sizeof(__bf16) > LDS_BUDGET;
    constexpr bool PRE3 = !KV_SHARED && NP * HDP <= 640;
    constexpr int NVH = PRE3 ? (SP * HDP / 4 + NTH - 1) / NTH : 1;   // 8-byte vectors per thread per head image
    __bf16* imgK = smem16;
    __bf16* imgV = KV_SHARED ? smem16 : smem16 + SP * LDH;
    const int q4 = c16 >> 2, p4 = c16 & 3;
    BlockStage<NVH, NTH> sk, sv;
    if (PRE3) {
        sk.load(kb, D, S, hd, SP, hdp);
        sv.load(vb, D, S, hd, SP, hdp);
    }
    bf16x8 bq_pre[nks];                  // this lane's q fragments of the next head: their global latency under the current head
#pragma unroll
    for (int ks = 0; ks < nks; ++ks) bq_pre[ks] = row_frag(qrow, 32 * ks, g, hd);
#pragma unroll 1
    for (int h = 0; h < p.H; ++h) {
        __syncthreads();                                             // previous head's (or phase 2's) reads are done
        if (PRE3) {
            sk.store(imgK, LDH, SP, hdp);
            sv.store(imgV, LDH, SP, hdp);
        } else {
            stage_block(imgK, LDH, kb + h * hd, D, S, hd, SP, hdp);
            if (!KV_SHARED) stage_block(imgV, LDH, vb + h * hd, D, S, hd, SP, hdp);
        }
#pragma unroll
        for (int t = 0; t < NJ; ++t) acc[t] = (f32x4v){0.f, 0.f, 0.f, 0.f};
        bf16x8 bq[nks];
#pragma unroll
        for (int ks = 0; ks < nks; ++ks) bq[ks] = bq_pre[ks];        // requested one head ago (round 3)
        __syncthreads();
        if (PRE3 && h + 1 < p.H) {
            sk.load(kb + (h + 1) * hd, D, S, hd, SP, hdp);
            sv.load(vb + (h + 1) * hd, D, S, hd, SP, hdp);
        }
        if (h + 1 < p.H) {
#pragma unroll
            for (int ks = 0; ks < nks; ++ks) bq_pre[ks] = row_frag(qrow + (h + 1) * hd, 32 * ks, g, hd);
        }
#pragma unroll
        for (int ks = 0; ks < nks; ++ks) {
#pragma unroll
            for (int t = 0; t < NJ; ++t) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(imgK + (16 * t + c16) * LDH + 32 * ks + 8 * g);
                acc[t] = MFMA_BF16(a, bq[ks], acc[t]);
            }
        }
        if (KV_SHARED) {                                             // V_h takes the image's place
            __syncthreads();
            stage_block(imgV, LDH, vb + h * hd, D, S, hd, SP, hdp);
            __syncthreads();
        }
        // softmax over the keys: 4*NJ in-lane values, then the 4 lane groups (xor 16, xor 32)
        float mx = -INFINITY;
        if constexpr (KEEP_MASK) {
#pragma unroll
            for (int t = 0; t < NJ; ++t) {
                const bf16x8 mf = Rf[t >> 1];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc[t][r] = fmaf(acc[t][r], p.scale, (float)mf[4 * (t & 1) + r]);
                    mx = fmaxf(mx, acc[t][r]);
                }
            }
        } else {
            // long rows: the mask row (this lane's own stores of phase 2, L2-resident) is re-read per head instead of
            // occupying 4 NP registers through the head loop
            const __bf16* Mrow = p.Mk + ((long)b * S + q_ld) * S;
#pragma unroll
            for (int t = 0; t < NJ; ++t) {
                const int j = 16 * t + 4 * g;
                const bf16x4 m4 = j < S ? ld4(Mrow + j) : (bf16x4){(__bf16)-INFINITY, (__bf16)-INFINITY, (__bf16)-INFINITY, (__bf16)-INFINITY};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc[t][r] = fmaf(acc[t][r], p.scale, (float)m4[r]);
                    mx = fmaxf(mx, acc[t][r]);
                }
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < NJ; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc[t][r] = __expf(acc[t][r] - mx);
                sum += acc[t][r];
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
        if (q_ok && g == 0) p.lse[((long)b * p.H + h) * S + q_lane] = mx + __logf(sum);
        // O^T[d,i] = sum_j V_h[j,d] P^T[j,i]: V^T fragments by transposed reads of the row-major [key][d] image.
        // One output tile at a time (rolled loop): unrolled over the tiles the compiler hoists all NP x ndt transposed
        // reads ahead of the products (+100 VGPRs, scratch).
        bf16x8 Pf[NP];
#pragma unroll
        for (int pr = 0; pr < NP; ++pr) Pf[pr] = cat8(pack4(acc[2 * pr] * inv), pack4(acc[2 * pr + 1] * inv));
        __bf16* orow = p.out + ((long)b * S + q_ld) * D + h * hd;
#pragma unroll 1
        for (int dt = 0; dt < ndt; ++dt) {
            f32x4v o = {0.f, 0.f, 0.f, 0.f};
            const __bf16* vbase = imgV + (4 * g + q4) * LDH + 16 * dt + 4 * p4;
#pragma unroll
            for (int pr = 0; pr < NP; ++pr) {
                const bf16x4 v0 = tr4(vbase + (32 * pr) * LDH);
                const bf16x4 v1 = tr4(vbase + (32 * pr + 16) * LDH);
                o = MFMA_BF16(cat8(v0, v1), Pf[pr], o);
            }
            const int d = 16 * dt + 4 * g;
            if (q_ok && d < hd) *reinterpret_cast<bf16x4*>(orow + d) = pack4(o);
        }
    }
#ifdef ATT16_STAMP
    __syncthreads();
    if (tid == 0 && qg == 0) {       // diagnostic build only: phase cycles over the lse of the first queries of head 0
        const unsigned long long ts3 = __builtin_amdgcn_s_memtime();
        float* d = p.lse + (long)b * p.H * S;
        d[0] = (float)(ts1 - ts0); d[1] = (float)(ts2 - ts1); d[2] = (float)(ts3 - ts2);
    }
#endif
}

// MkT[b][key][query] = Mk[b][query][key] for the key-side backward pass (keys on the lanes there).  A pass of its own:
// written from the forward kernel's accumulators it is 8 NP scattered 2-byte stores per lane whose addresses cost the
// kernel ~70 VGPRs at their peak.
__global__ __launch_bounds__(256) void mask_transpose_kernel(const __bf16* __restrict__ in, __bf16* __restrict__ out, int S) {
    __shared__ __bf16 tile[32][34];
    const long base = (long)blockIdx.z * S * S;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;            // 32 x 8 threads, 32 x 32 tile
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + 8 * k, c = c0 + tx;
        if (r < S && c < S) tile[ty + 8 * k][tx] = in[base + (long)r * S + c];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = c0 + ty + 8 * k, c = r0 + tx;                    // out row = in column
        if (r < S && c < S) out[base + (long)r * S + c] = tile[tx][ty + 8 * k];
    }
}

#include "attention_bf16_fwd2.h"
#include "attention_bf16_fwd3.h"

// CALM_ATTN16_V2=0 in the environment: the register-staged forward for every shape (A/B runs)
inline bool fwd2_enabled() {
    static const int on = [] { const char* e = getenv("CALM_ATTN16_V2"); return (e && e[0] == '0') ? 0 : 1; }();
    return on != 0;
}
// CALM_ATTN16_V3=1: EXPERIMENTAL (round 4, off by default) — phases 1-2 as attn16_fwd2_kernel<.., MASK_ONLY> and the head
// loop as the persistent per-(image, head) kernel of attention_bf16_fwd3.h.  Parity-tested (tests/test_attention16_gpu.py
// runs it in a child process); at bs = 256 it is 3 % faster than the fused kernel at S = 224 and 3-20 % slower at the smaller
// stages — DESIGN.md section 7 has the measurements that led there and what they say bounds the head loop.
inline bool fwd3_enabled() {
    static const int on = [] { const char* e = getenv("CALM_ATTN16_V3"); return (e && e[0] == '1') ? 1 : 0; }();
    return on != 0;
}

template <int NP, int HDP>
int launch_fwd16_t(const Attn16P& p, size_t lds, hipStream_t s) {
    if constexpr (Fwd2Geo<NP, HDP>::OK) {
        if (fwd2_enabled()) {
            constexpr int lds2 = Fwd2Geo<NP, HDP>::LDS;
            hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn16_fwd2_kernel<NP, HDP>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, lds2);
            if (e2 != hipSuccess) return (int)e2;
            constexpr int nw2 = waves_for(NP);
            const int tiles2 = (p.S + 15) / 16;
            Attn16P p2 = p;
            p2.groups = (tiles2 + nw2 - 1) / nw2;
            if constexpr (Fwd3Geo<NP, HDP>::OK) {
                if (fwd3_enabled()) {
                    // phases 1-2 (R, mask MLP -> Mk), then one workgroup per (image, head) for the head loop
                    e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn16_fwd2_kernel<NP, HDP, true>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, lds2);
                    if (e2 != hipSuccess) return (int)e2;
                    constexpr int lds3 = Fwd3Geo<NP, HDP>::LDS;
                    e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn16_fwd3_core_kernel<NP, HDP, true>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, lds3);
                    if (e2 != hipSuccess) return (int)e2;
                    e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn16_fwd3_core_kernel<NP, HDP, false>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, lds3);
                    if (e2 != hipSuccess) return (int)e2;
                    hipLaunchKernelGGL((attn16_fwd2_kernel<NP, HDP, true>), dim3(p2.groups * p.B), dim3(64 * nw2), lds2, s, p2);
                    CALM_LAUNCH_CHECK();
                    // persistent: as many workgroups as the chip holds at once (a multiple of 8: one share per XCD)
                    constexpr int nw3 = Fwd3Geo<NP, HDP>::NW;
                    const int slots3 = 256 * Fwd3Geo<NP, HDP>::WG_PER_CU;
                    const int grid3 = p.B < slots3 ? p.B : slots3;        // one image (all its heads) at a time per workgroup
                    // start offset of the second wave of every SIMD, x 64 cycles (default: about half an item at S = 224)
                    static const int stagger = [] { const char* e = getenv("CALM_ATTN16_STAGGER"); return e ? atoi(e) : 60; }();
                    p2.kv_shared = stagger;
                    if (p.S == 32 * NP)
                        hipLaunchKernelGGL((attn16_fwd3_core_kernel<NP, HDP, true>), dim3(grid3), dim3(64 * nw3), lds3, s, p2);
                    else
                        hipLaunchKernelGGL((attn16_fwd3_core_kernel<NP, HDP, false>), dim3(grid3), dim3(64 * nw3), lds3, s, p2);
                    CALM_LAUNCH_CHECK();
                    const int t32c = (p.S + 31) / 32;
                    hipLaunchKernelGGL(mask_transpose_kernel, dim3(t32c, t32c, p.B), dim3(256), 0, s, (const __bf16*)p.Mk, p.MkT, p.S);
                    CALM_LAUNCH_CHECK();
                    return 0;
                }
            }
            hipLaunchKernelGGL((attn16_fwd2_kernel<NP, HDP>), dim3(p2.groups * p.B), dim3(64 * nw2), lds2, s, p2);
            CALM_LAUNCH_CHECK();
            const int t32b = (p.S + 31) / 32;
            hipLaunchKernelGGL(mask_transpose_kernel, dim3(t32b, t32b, p.B), dim3(256), 0, s, (const __bf16*)p.Mk, p.MkT, p.S);
            CALM_LAUNCH_CHECK();
            return 0;
        }
    }
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn16_fwd_kernel<NP, HDP>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    constexpr int nw = waves_for(NP);
    const int tiles = (p.S + 15) / 16;
    Attn16P p1 = p;
    p1.groups = (tiles + nw - 1) / nw;
    hipLaunchKernelGGL((attn16_fwd_kernel<NP, HDP>), dim3(p1.groups * p.B), dim3(64 * nw), lds, s, p1);
    CALM_LAUNCH_CHECK();
    const int t32 = (p.S + 31) / 32;
    hipLaunchKernelGGL(mask_transpose_kernel, dim3(t32, t32, p.B), dim3(256), 0, s, (const __bf16*)p.Mk, p.MkT, p.S);
    CALM_LAUNCH_CHECK();
    return 0;
}
template <int NP>
int launch_fwd16(const Attn16P& p, int, size_t lds, hipStream_t s) {
    switch ((p.hd + 31) / 32) {
        case 1: return launch_fwd16_t<NP, 32>(p, lds, s);
        case 2: return launch_fwd16_t<NP, 64>(p, lds, s);
        case 3: return launch_fwd16_t<NP, 96>(p, lds, s);
        case 4: return launch_fwd16_t<NP, 128>(p, lds, s);
    }
    return CALM_E_UNSUPP;
}

// waves per workgroup of the backward kernels
inline int pick_waves16(int S) { return waves_for((S + 31) / 32); }

// K_h and V_h images side by side when that leaves room for two workgroups per CU, else one shared image
inline bool fwd_kv_shared(int S, int hd) {
    const int SP = (S + 31) / 32 * 32, hdp = (hd + 31) / 32 * 32;
    return (size_t)2 * SP * ld_rt(hdp) * sizeof(__bf16) > LDS_BUDGET;
}
inline size_t fwd_lds_bytes(int S, int hd) {
    const int NP = (S + 31) / 32, SP = 32 * NP, hdp = (hd + 31) / 32 * 32;
    const size_t ph1 = (size_t)SP * ld_rt(64);
    const size_t ph2 = (size_t)32 * ld_pt(SP) + (size_t)SP * ld_pt(32);
    const size_t ph3 = (size_t)(fwd_kv_shared(S, hd) ? 1 : 2) * SP * ld_rt(hdp);
    size_t m = ph1 > ph2 ? ph1 : ph2;
    if (ph3 > m) m = ph3;
    return m * sizeof(__bf16);
}


// =====================================================================================================
// Backward of the attention core with P recomputed from q, k, the saved mask and the row log-sum-exp.
//   delta_i = sum_d dO[i,d] O[i,d]           (= rowsum(P o dP); from the saved bf16 output)
//   P       = exp(scale q k^T + M - lse)     dP = dO V^T      dS = P o (dP - delta)
//   dM = sum_h dS      dQ_h = scale dS K_h      dV_h = P^T dO_h      dK_h = scale dS^T Q_h
// Two kernels, no cross-wave reduction, both streaming over pairs of 16-row tiles of the opposite axis:
//   query side  one wave = 16 queries on the lanes; per key-tile pair: S^T, dP^T (16-byte fragments of the K_h / V_h
//               images), dS^T, dM^T += dS^T (registers, all heads), dQ^T += K_h^T dS^T (transposed reads of the K_h image)
//   key side    one wave = 16 keys on the lanes; per query-tile pair: S, dP (fragments of the Q_h / dO_h images), P, dS,
//               dV^T += dO_h^T P, dK^T += Q_h^T dS (transposed reads of the same images)
// The images hold `ch` tile pairs at a time (so that two workgroups fit per CU); accumulators carry across chunks.
// =====================================================================================================
struct Attn16BP {
    const __bf16* q; const __bf16* k; const __bf16* v; const __bf16* out; const __bf16* dout;
    const __bf16* Mk; const __bf16* MkT; const float* lse;
    float* delta;
    __bf16* dq; __bf16* dk; __bf16* dv; __bf16* dM;
    int B, S, H, hd;
    float scale;
    int ch;             // tile pairs per LDS chunk
    int groups;         // row groups (workgroups) per image
};

// tile pairs whose two images fit the LDS budget of a workgroup (two workgroups per CU)
__host__ __device__ constexpr int bwd_chunk_pairs_c(int np, int hdp) {
    const int per_pair = 2 * 32 * ld_rt(hdp) * 2;
    const int ch = (80 * 1024) / per_pair;
    return ch < 1 ? 1 : ch > np ? np : ch;
}

// Both kernels are compiled per (NP, HDP = head dim padded to 32): with run-time trip counts the fully unrolled
// tile loops (register-array indices must be constants) cost > 256 VGPRs and kilobytes of scratch.

// the two [32 ch][hdp] images of a chunk: rows row0.. of the head slices a / b (zero past rows_valid / hd)
__device__ __forceinline__ void stage_pair_images(__bf16* imgA, __bf16* imgB, int LDH, const __bf16* a, const __bf16* b,
                                                  long gstride, int row0, int rows_valid, int hd, int rows_img, int hdp) {
    stage_block(imgA, LDH, a + (long)row0 * gstride, gstride, rows_valid - row0, hd, rows_img, hdp);
    stage_block(imgB, LDH, b + (long)row0 * gstride, gstride, rows_valid - row0, hd, rows_img, hdp);
}

template <int NP, int HDP>
__global__ __launch_bounds__(64 * waves_for(NP)) void attn16_bwd_q_kernel(const Attn16BP p) {
    constexpr int NJ = 2 * NP;
    extern __shared__ __attribute__((aligned(16))) __bf16 smem16[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c16 = lane & 15, g = lane >> 4, q4 = c16 >> 2, p4 = c16 & 3;
    int b, qg;
    image_and_group(blockIdx.x, p.groups, p.B, b, qg);
    const int S = p.S, D = p.H * p.hd, hd = p.hd;
    const int q_lane = qg * (16 * (blockDim.x >> 6)) + 16 * wave + c16;
    const bool q_ok = q_lane < S;
    const int q_ld = q_ok ? q_lane : S - 1;
    const long qoff = ((long)b * S + q_ld) * D;
    const __bf16* kb = p.k + (long)b * S * D;
    const __bf16* vb = p.v + (long)b * S * D;
    constexpr int hdp = HDP, LDH = ld_rt(HDP), nks = HDP / 32, ndt = HDP / 16, MAXKS = nks, MAXDT = ndt;
    const int rows_img = 32 * p.ch;
    __bf16* imgK = smem16;
    __bf16* imgV = smem16 + rows_img * LDH;

    // the mask row of this lane's query is re-read (L2) pair by pair in every head instead of living in NP x 4 registers
    const __bf16* Mrow = p.Mk + ((long)b * S + q_ld) * S;
    const bf16x4 ninf = {(__bf16)-INFINITY, (__bf16)-INFINITY, (__bf16)-INFINITY, (__bf16)-INFINITY};
    f32x4v accM[NJ];
#pragma unroll
    for (int t = 0; t < NJ; ++t) accM[t] = (f32x4v){0.f, 0.f, 0.f, 0.f};
    // one chunk per head (all keys resident): the NEXT head's K_h / V_h are fetched into registers under this head's math
    // (up to NP x HDP = 448 — the stages of Base-224: beyond 256 the prefetch registers spill (156-184 bytes of scratch per
    // lane), which still pays: the spilled vectors sit in L1 / L2 when the next head needs them, S=176 -6.6 %, S=224 -3.8 %)
    constexpr int NTHB = 64 * waves_for(NP);
    constexpr bool PRE = bwd_chunk_pairs_c(NP, HDP) == NP && NP * HDP <= ATT16_BWDQ_PRE_MAX;
    constexpr int NVB = PRE ? (32 * NP * HDP / 4 + NTHB - 1) / NTHB : 1;
    constexpr bool pre = PRE;
    BlockStage<NVB, NTHB> sa, sb;
    if (pre) {
        sa.load(kb, D, S, hd, 32 * NP, hdp);
        sb.load(vb, D, S, hd, 32 * NP, hdp);
    }

#pragma unroll 1
    for (int h = 0; h < p.H; ++h) {
        bf16x8 qf[MAXKS], dof[MAXKS];
        float part = 0.f;
#pragma unroll
        for (int ks = 0; ks < MAXKS; ++ks) {
            if (ks < nks) {
                qf[ks] = row_frag(p.q + qoff + h * hd, 32 * ks, g, hd);
                dof[ks] = row_frag(p.dout + qoff + h * hd, 32 * ks, g, hd);
                const bf16x8 of = row_frag(p.out + qoff + h * hd, 32 * ks, g, hd);
#pragma unroll
                for (int e = 0; e < 8; ++e) part = fmaf((float)dof[ks][e], (float)of[e], part);
            }
        }
        part += __shfl_xor(part, 16, 64);
        part += __shfl_xor(part, 32, 64);
        const float delta = part;
        const float lse = p.lse[((long)b * p.H + h) * S + q_ld];
        if (q_ok && g == 0) p.delta[((long)b * p.H + h) * S + q_lane] = delta;
        f32x4v dqa[MAXDT];
#pragma unroll
        for (int dt = 0; dt < MAXDT; ++dt) dqa[dt] = (f32x4v){0.f, 0.f, 0.f, 0.f};
        // the mask of a pair is requested one pair ahead (round 3: it used to be loaded and waited for inside the pair —
        // one exposed memory round trip per head and key pair)
        auto mask_of = [&](int pr) __attribute__((always_inline)) {
            const int j0 = 32 * pr + 4 * g, j1 = j0 + 16;             // pad keys: mask -inf -> P = 0
            return cat8(j0 < S ? ld4(Mrow + j0) : ninf, j1 < S ? ld4(Mrow + j1) : ninf);
        };
        bf16x8 mf_next = mask_of(0);
#pragma unroll
        for (int pr = 0; pr < NP; ++pr) {
            __builtin_amdgcn_sched_barrier(0);             // one pair at a time: no hoisting of the next pair's reads
            const bf16x8 mf = mf_next;
            if (pr + 1 < NP) mf_next = mask_of(pr + 1);
            const int lp = pr % p.ch;                      // uniform
            if (lp == 0) {
                __syncthreads();
                if (pre) {
                    sa.store(imgK, LDH, 32 * NP, hdp);
                    sb.store(imgV, LDH, 32 * NP, hdp);
                } else {
                    stage_pair_images(imgK, imgV, LDH, kb + h * hd, vb + h * hd, D, 32 * pr, S, hd, rows_img, hdp);
                }
                __syncthreads();
                if (pre && h + 1 < p.H) {
                    sa.load(kb + (h + 1) * hd, D, S, hd, 32 * NP, hdp);
                    sb.load(vb + (h + 1) * hd, D, S, hd, 32 * NP, hdp);
                }
            }
            f32x4v s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, d0 = s0, d1 = s0;
#pragma unroll
            for (int ks = 0; ks < MAXKS; ++ks) {
                if (ks < nks) {
                    const int o0 = (32 * lp + c16) * LDH + 32 * ks + 8 * g, o1 = o0 + 16 * LDH;
                    s0 = MFMA_BF16(*reinterpret_cast<const bf16x8*>(imgK + o0), qf[ks], s0);
                    s1 = MFMA_BF16(*reinterpret_cast<const bf16x8*>(imgK + o1), qf[ks], s1);
                    d0 = MFMA_BF16(*reinterpret_cast<const bf16x8*>(imgV + o0), dof[ks], d0);
                    d1 = MFMA_BF16(*reinterpret_cast<const bf16x8*>(imgV + o1), dof[ks], d1);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pa = __expf(fmaf(s0[r], p.scale, (float)mf[r]) - lse);
                const float pb = __expf(fmaf(s1[r], p.scale, (float)mf[4 + r]) - lse);
                s0[r] = pa * (d0[r] - delta);
                s1[r] = pb * (d1[r] - delta);
            }
            accM[2 * pr] = accM[2 * pr] + s0;
            accM[2 * pr + 1] = accM[2 * pr + 1] + s1;
            const bf16x8 dsf = cat8(pack4(s0), pack4(s1));
            const __bf16* kt = imgK + (32 * lp + 4 * g + q4) * LDH + 4 * p4;
#pragma unroll
            for (int dt = 0; dt < MAXDT; ++dt) {
                if (dt < ndt) dqa[dt] = MFMA_BF16(cat8(tr4(kt + 16 * dt), tr4(kt + 16 * dt + 16 * LDH)), dsf, dqa[dt]);
            }
        }
        __bf16* dqrow = p.dq + qoff + h * hd;
#pragma unroll
        for (int dt = 0; dt < MAXDT; ++dt) {
            const int d = 16 * dt + 4 * g;
            if (dt < ndt && q_ok && d < hd) *reinterpret_cast<bf16x4*>(dqrow + d) = pack4(dqa[dt] * p.scale);
        }
    }
    if (q_ok) {
        __bf16* mrow = p.dM + ((long)b * S + q_lane) * S;
#pragma unroll
        for (int t = 0; t < NJ; ++t) {
            const int j = 16 * t + 4 * g;
            if (j < S) *reinterpret_cast<bf16x4*>(mrow + j) = pack4(accM[t]);
        }
    }
}

template <int NP, int HDP>
__global__ __launch_bounds__(64 * waves_for(NP)) void attn16_bwd_kv_kernel(const Attn16BP p) {
    extern __shared__ __attribute__((aligned(16))) __bf16 smem16[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c16 = lane & 15, g = lane >> 4, q4 = c16 >> 2, p4 = c16 & 3;
    int b, kg;
    image_and_group(blockIdx.x, p.groups, p.B, b, kg);
    const int S = p.S, D = p.H * p.hd, hd = p.hd;
    const int k_lane = kg * (16 * (blockDim.x >> 6)) + 16 * wave + c16;               // this lane's key
    const bool k_ok = k_lane < S;
    const int k_ld = k_ok ? k_lane : S - 1;
    const long koff = ((long)b * S + k_ld) * D;
    const __bf16* qb = p.q + (long)b * S * D;
    const __bf16* dob = p.dout + (long)b * S * D;
    constexpr int hdp = HDP, LDH = ld_rt(HDP), nks = HDP / 32, ndt = HDP / 16, MAXKS = nks, MAXDT = ndt;
    const int rows_img = 32 * p.ch;
    __bf16* imgQ = smem16;
    __bf16* imgO = smem16 + rows_img * LDH;

    // the mask column of this lane's key (MkT[b][key][query]) is re-read (L2) pair by pair in every head
    const __bf16* Mcol = p.MkT + ((long)b * S + k_ld) * S;
    const bf16x4 zero4 = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
    constexpr int NTHB = 64 * waves_for(NP);
    constexpr bool PRE = bwd_chunk_pairs_c(NP, HDP) == NP && NP * HDP <= ATT16_BWDKV_PRE_MAX;      // see the query-side kernel
    constexpr int NVB = PRE ? (32 * NP * HDP / 4 + NTHB - 1) / NTHB : 1;
    constexpr bool pre = PRE;
    BlockStage<NVB, NTHB> sa, sb;
    if (pre) {
        sa.load(qb, D, S, hd, 32 * NP, hdp);
        sb.load(dob, D, S, hd, 32 * NP, hdp);
    }
#pragma unroll 1
    for (int h = 0; h < p.H; ++h) {
        bf16x8 kf[MAXKS], vf[MAXKS];
#pragma unroll
        for (int ks = 0; ks < MAXKS; ++ks) {
            if (ks < nks) {
                kf[ks] = row_frag(p.k + koff + h * hd, 32 * ks, g, hd);
                vf[ks] = row_frag(p.v + koff + h * hd, 32 * ks, g, hd);
            }
        }
        const float* lse_h = p.lse + ((long)b * p.H + h) * S;
        const float* del_h = p.delta + ((long)b * p.H + h) * S;
        f32x4v dva[MAXDT], dka[MAXDT];
#pragma unroll
        for (int dt = 0; dt < MAXDT; ++dt) { dva[dt] = (f32x4v){0.f, 0.f, 0.f, 0.f}; dka[dt] = dva[dt]; }
        // mask column, lse and delta of a query pair are requested one pair ahead (round 3: were loaded and waited for
        // inside the pair)
        struct PairIn { bf16x8 mf; f32x4v l0, l1, e0, e1; };
        auto pair_in = [&](int pr) __attribute__((always_inline)) {
            const int i0 = 32 * pr + 4 * g, i1 = i0 + 16;
            const f32x4v z4 = {0.f, 0.f, 0.f, 0.f};
            PairIn o;
            o.l0 = i0 < S ? *reinterpret_cast<const f32x4v*>(lse_h + i0) : z4;
            o.l1 = i1 < S ? *reinterpret_cast<const f32x4v*>(lse_h + i1) : z4;
            o.e0 = i0 < S ? *reinterpret_cast<const f32x4v*>(del_h + i0) : z4;
            o.e1 = i1 < S ? *reinterpret_cast<const f32x4v*>(del_h + i1) : z4;
            o.mf = cat8(i0 < S ? ld4(Mcol + i0) : zero4, i1 < S ? ld4(Mcol + i1) : zero4);
            return o;
        };
        PairIn nxt = pair_in(0);
#pragma unroll
        for (int pr = 0; pr < NP; ++pr) {
            __builtin_amdgcn_sched_barrier(0);
            const PairIn cur = nxt;
            if (pr + 1 < NP) nxt = pair_in(pr + 1);
            const int lp = pr % p.ch;
            if (lp == 0) {
                __syncthreads();
                if (pre) {
                    sa.store(imgQ, LDH, 32 * NP, hdp);
                    sb.store(imgO, LDH, 32 * NP, hdp);
                } else {
                    stage_pair_images(imgQ, imgO, LDH, qb + h * hd, dob + h * hd, D, 32 * pr, S, hd, rows_img, hdp);
                }
                __syncthreads();
                if (pre && h + 1 < p.H) {
                    sa.load(qb + (h + 1) * hd, D, S, hd, 32 * NP, hdp);
                    sb.load(dob + (h + 1) * hd, D, S, hd, 32 * NP, hdp);
                }
            }
            f32x4v s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, d0 = s0, d1 = s0;
#pragma unroll
            for (int ks = 0; ks < MAXKS; ++ks) {
                if (ks < nks) {
                    const int o0 = (32 * lp + c16) * LDH + 32 * ks + 8 * g, o1 = o0 + 16 * LDH;
                    s0 = MFMA_BF16(*reinterpret_cast<const bf16x8*>(imgQ + o0), kf[ks], s0);
                    s1 = MFMA_BF16(*reinterpret_cast<const bf16x8*>(imgQ + o1), kf[ks], s1);
                    d0 = MFMA_BF16(*reinterpret_cast<const bf16x8*>(imgO + o0), vf[ks], d0);
                    d1 = MFMA_BF16(*reinterpret_cast<const bf16x8*>(imgO + o1), vf[ks], d1);
                }
            }
            // queries 32 pr + 4 g + r (tile 0) and + 16 (tile 1); pad queries: Q / dO rows are zero -> no contribution
            const f32x4v l0 = cur.l0, l1 = cur.l1, e0 = cur.e0, e1 = cur.e1;
            const bf16x8 mf = cur.mf;
            f32x4v pa, pb;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pa[r] = __expf(fmaf(s0[r], p.scale, (float)mf[r]) - l0[r]);
                pb[r] = __expf(fmaf(s1[r], p.scale, (float)mf[4 + r]) - l1[r]);
                s0[r] = pa[r] * (d0[r] - e0[r]);
                s1[r] = pb[r] * (d1[r] - e1[r]);
            }
            const bf16x8 pf = cat8(pack4(pa), pack4(pb)), dsf = cat8(pack4(s0), pack4(s1));
            const int to = (32 * lp + 4 * g + q4) * LDH + 4 * p4;
#pragma unroll
            for (int dt = 0; dt < MAXDT; ++dt) {
                if (dt < ndt) {
                    dva[dt] = MFMA_BF16(cat8(tr4(imgO + to + 16 * dt), tr4(imgO + to + 16 * dt + 16 * LDH)), pf, dva[dt]);
                    dka[dt] = MFMA_BF16(cat8(tr4(imgQ + to + 16 * dt), tr4(imgQ + to + 16 * dt + 16 * LDH)), dsf, dka[dt]);
                }
            }
        }
        __bf16* dvrow = p.dv + koff + h * hd;
        __bf16* dkrow = p.dk + koff + h * hd;
#pragma unroll
        for (int dt = 0; dt < MAXDT; ++dt) {
            const int d = 16 * dt + 4 * g;
            if (dt < ndt && k_ok && d < hd) {
                *reinterpret_cast<bf16x4*>(dvrow + d) = pack4(dva[dt]);
                *reinterpret_cast<bf16x4*>(dkrow + d) = pack4(dka[dt] * p.scale);
            }
        }
    }
}

inline int bwd_chunk_pairs(int S, int hd) { return bwd_chunk_pairs_c((S + 31) / 32, (hd + 31) / 32 * 32); }

#include "attention_bf16_bwd2.h"

// CALM_ATTN16_BWD2=0 in the environment: the register-staged backward kernels for every shape (A/B runs)
inline bool bwd2_enabled() {
    static const int on = [] { const char* e = getenv("CALM_ATTN16_BWD2"); return (e && e[0] == '0') ? 0 : 1; }();
    return on != 0;
}

template <int NP, int HDP>
int launch_bwd16_t(const Attn16BP& p, int nw, hipStream_t s) {
    if constexpr (Bwd2Geo<NP, HDP, false>::OK && Bwd2Geo<NP, HDP, true>::OK) {
        if (bwd2_enabled()) {
            Attn16BP p2 = p;
            const int tiles2 = (p.S + 15) / 16;
            p2.groups = (tiles2 + nw - 1) / nw;
            const dim3 grid2(p2.groups * p.B);
            constexpr int ldq = Bwd2Geo<NP, HDP, false>::LDS, ldk = Bwd2Geo<NP, HDP, true>::LDS;
            hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn16_bwd2_kernel<NP, HDP, false>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, ldq);
            if (e2 != hipSuccess) return (int)e2;
            e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn16_bwd2_kernel<NP, HDP, true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, ldk);
            if (e2 != hipSuccess) return (int)e2;
            hipLaunchKernelGGL((attn16_bwd2_kernel<NP, HDP, false>), grid2, dim3(64 * nw), ldq, s, p2);
            CALM_LAUNCH_CHECK();
            hipLaunchKernelGGL((attn16_bwd2_kernel<NP, HDP, true>), grid2, dim3(64 * nw), ldk, s, p2);
            CALM_LAUNCH_CHECK();
            return 0;
        }
    }
    const size_t lds = (size_t)2 * 32 * p.ch * ld_rt(HDP) * sizeof(__bf16);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn16_bwd_q_kernel<NP, HDP>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn16_bwd_kv_kernel<NP, HDP>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    const int tiles = (p.S + 15) / 16;
    Attn16BP p1 = p;
    p1.groups = (tiles + nw - 1) / nw;
    const dim3 grid(p1.groups * p.B);
    hipLaunchKernelGGL((attn16_bwd_q_kernel<NP, HDP>), grid, dim3(64 * nw), lds, s, p1);
    CALM_LAUNCH_CHECK();
    hipLaunchKernelGGL((attn16_bwd_kv_kernel<NP, HDP>), grid, dim3(64 * nw), lds, s, p1);
    CALM_LAUNCH_CHECK();
    return 0;
}
template <int NP>
int launch_bwd16(const Attn16BP& p, int nw, hipStream_t s) {
    switch ((p.hd + 31) / 32) {
        case 1: return launch_bwd16_t<NP, 32>(p, nw, s);
        case 2: return launch_bwd16_t<NP, 64>(p, nw, s);
        case 3: return launch_bwd16_t<NP, 96>(p, nw, s);
        case 4: return launch_bwd16_t<NP, 128>(p, nw, s);
    }
    return CALM_E_UNSUPP;
}

}  // namespace

extern "C" {

int calm_attention16_supported(int32_t S, int32_t H, int32_t hd) {
    if (S <= 0 || H <= 0 || hd <= 0) return 0;
    if ((S & 7) || (hd & 3) || hd > 128 || S > 384) return 0;      // bf16 W1 rows / head slices as 8-byte vectors
    if (fwd_lds_bytes(S, hd) > 160 * 1024) return 0;
    return 1;
}

int calm_attention16_fwd(const void* q, const void* k, const void* v, const void* w1, const float* b1, const float* s1,
                         const void* w2, const float* b2, const float* s2, void* out, void* R, void* hp, void* hg,
                         void* Mk, void* MkT, float* lse, int32_t B, int32_t S, int32_t H, int32_t hd, void* stream) {
    if (!q || !k || !v || !w1 || !b1 || !s1 || !w2 || !b2 || !s2 || !out || !R || !hp || !hg || !Mk || !MkT || !lse ||
        B <= 0)
        return CALM_E_INVAL;
    if (!calm_attention16_supported(S, H, hd)) return CALM_E_UNSUPP;
    if (B > 65535) return CALM_E_UNSUPP;                    // mask_transpose_kernel's grid.z
    Attn16P p{(const __bf16*)q, (const __bf16*)k, (const __bf16*)v, (const __bf16*)w1, b1, s1, (const __bf16*)w2, b2, s2,
              (__bf16*)out, (__bf16*)R, (__bf16*)hp, (__bf16*)hg, (__bf16*)Mk, (__bf16*)MkT, lse, B, S, H, hd,
              1.0f / sqrtf((float)hd),
              fwd_kv_shared(S, hd) ? 1 : 0, 0};
    hipStream_t s = as_stream(stream);
    const int nw = pick_waves16(S);
    const size_t lds = fwd_lds_bytes(S, hd);
    switch ((S + 31) / 32) {
        case 1: return launch_fwd16<1>(p, nw, lds, s);
        case 2: return launch_fwd16<2>(p, nw, lds, s);
        case 3: return launch_fwd16<3>(p, nw, lds, s);
        case 4: return launch_fwd16<4>(p, nw, lds, s);
        case 5: return launch_fwd16<5>(p, nw, lds, s);
        case 6: return launch_fwd16<6>(p, nw, lds, s);
        case 7: return launch_fwd16<7>(p, nw, lds, s);
        case 8: return launch_fwd16<8>(p, nw, lds, s);
        case 9: return launch_fwd16<9>(p, nw, lds, s);
        case 10: return launch_fwd16<10>(p, nw, lds, s);
        case 11: return launch_fwd16<11>(p, nw, lds, s);
        case 12: return launch_fwd16<12>(p, nw, lds, s);
    }
    return CALM_E_UNSUPP;
}

int calm_attention16_bwd(const void* q, const void* k, const void* v, const void* out, const void* dout, const void* Mk,
                         const void* MkT, const float* lse, float* delta, void* dq, void* dk, void* dv, void* dM,
                         int32_t B, int32_t S, int32_t H, int32_t hd, void* stream) {
    if (!q || !k || !v || !out || !dout || !Mk || !MkT || !lse || !delta || !dq || !dk || !dv || !dM || B <= 0)
        return CALM_E_INVAL;
    if (!calm_attention16_supported(S, H, hd)) return CALM_E_UNSUPP;

    Attn16BP p{(const __bf16*)q, (const __bf16*)k, (const __bf16*)v, (const __bf16*)out, (const __bf16*)dout,
               (const __bf16*)Mk, (const __bf16*)MkT, lse, delta, (__bf16*)dq, (__bf16*)dk, (__bf16*)dv, (__bf16*)dM,
               B, S, H, hd, 1.0f / sqrtf((float)hd), bwd_chunk_pairs(S, hd), 0};
    hipStream_t s = as_stream(stream);
    const int nw = pick_waves16(S);
    switch ((S + 31) / 32) {
        case 1: return launch_bwd16<1>(p, nw, s);
        case 2: return launch_bwd16<2>(p, nw, s);
        case 3: return launch_bwd16<3>(p, nw, s);
        case 4: return launch_bwd16<4>(p, nw, s);
        case 5: return launch_bwd16<5>(p, nw, s);
        case 6: return launch_bwd16<6>(p, nw, s);
        case 7: return launch_bwd16<7>(p, nw, s);
        case 8: return launch_bwd16<8>(p, nw, s);
        case 9: return launch_bwd16<9>(p, nw, s);
        case 10: return launch_bwd16<10>(p, nw, s);
        case 11: return launch_bwd16<11>(p, nw, s);
        case 12: return launch_bwd16<12>(p, nw, s);
    }
    return CALM_E_UNSUPP;
}

}  // extern "C"
