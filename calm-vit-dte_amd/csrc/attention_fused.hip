// Fused cross-axial latent-mask attention, forward (Vi_Tools_CNN_less_V2.py:288-299).
//
//   R      = Q_all K_all^T                      (all heads concatenated, raw, un-scaled)     [Sq,Skv]
//   M      = W2 gelu(W1 R^T + b1) + b2          (2-layer MLP along the KEY axis, W/sigma)    [Sq,Skv]
//   O_h    = softmax_j(Q_h K_h^T / sqrt(hd) + M) V_h                                         per head
//
// One workgroup = NW waves = NW 16-query tiles of one batch element; the wave keeps its 16 queries on
// the MFMA *lane/column* index and the keys on the accumulator rows ("transposed" orientation):
//   R^T[j,i], M^T[j,i], S^T[j,i], P^T[j,i], O^T[d,i]   with i = lane&15.
// With v_mfma_f32_16x16x4_f32 an accumulator tile (rows in registers, column on the lane) is directly the
// B operand of the next product that sums over its ROW index, so the whole chain
//   R^T -> (W1 . R^T) -> gelu -> (W2 . hid) -> + scale K_h Q_h^T -> softmax -> (V_h^T . P^T)
// runs register-to-register: nothing but the streamed operands (K/Q column chunks, W1/W2 row chunks, V
// key chunks) goes through LDS, and the [H,Sq,Skv] probabilities never have to exist in HBM.
// The softmax reduction over keys is 4*NJ in-lane values + two wave shuffles (xor 16, xor 32).
// LDS images are K-major ([k][row]) with row strides chosen so that the 4 lane groups of a b32 read
// land 16 banks apart (conflict-free); global->LDS staging is register-prefetched one chunk ahead.
#include "common.h"

#ifndef ATT_TR_ROWFAST
#define ATT_TR_ROWFAST 1   // W1-chunk staging: hidden rows fastest over lanes (16-way LDS write conflict otherwise)
#endif

namespace {

typedef float f32x4v __attribute__((ext_vector_type(4)));

struct AttnFwdP {
    const float* q; const float* k; const float* v;
    const float* w1; const float* b1; const float* s1;
    const float* w2; const float* b2; const float* s2;
    float* out;
    float* R; float* hp; float* hg; float* Mk; float* P;   // saved for backward (P optional)
    int B, Sq, Skv, H, hd;
    float scale;
};

constexpr int NV_Q = 1;   // ... for a [16*NW x 16] query chunk

// ---- staging helpers (all threads of the block cooperate) -------------------------------------
// block of `rows` rows x 16 columns (columns c0..c0+15 of a row-major matrix, row stride `stride`),
// staged K-MAJOR: dst[c][row].  Columns >= cmax are zero-filled.
template <int NV> struct Regs { f32x4v v[NV]; };

template <int NT, int NV>
__device__ __forceinline__ void km_load(Regs<NV>& rg, const float* __restrict__ src, long stride, int rows, int c0,
                                        int cmax) {
    constexpr int nt = NT;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int f = threadIdx.x + u * nt;
        const int row = f >> 2, c = c0 + 4 * (f & 3);
        f32x4v val = {0.f, 0.f, 0.f, 0.f};
        if (row < rows && c < cmax) val = *reinterpret_cast<const f32x4v*>(src + (long)row * stride + c);
        rg.v[u] = val;
    }
}
template <int NT, int NV>
__device__ __forceinline__ void km_store(const Regs<NV>& rg, float* __restrict__ dst, int ld, int rows) {
    constexpr int nt = NT;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int f = threadIdx.x + u * nt;
        const int row = f >> 2, cq = 4 * (f & 3);
        if (row < rows) {
#pragma unroll
            for (int e = 0; e < 4; ++e) dst[(cq + e) * ld + row] = rg.v[u][e];
        }
    }
}
// block of 16 rows x `cols` contiguous columns staged TRANSPOSED: dst[col][r] (r = 0..15), ld = row stride of dst
template <int NT, int NV>
__device__ __forceinline__ void tr_load(Regs<NV>& rg, const float* __restrict__ src, long stride, int cols) {
    constexpr int nt = NT;
    const int per_row = cols >> 2;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int f = threadIdx.x + u * nt;
#if ATT_TR_ROWFAST
        const int r = f & 15, cq = f >> 4;            // 16 rows fastest: LDS writes hit 16 consecutive banks
        f32x4v val = {0.f, 0.f, 0.f, 0.f};
        if (cq < per_row) val = *reinterpret_cast<const f32x4v*>(src + (long)r * stride + 4 * cq);
#else
        const int r = f / per_row, cq = f - r * per_row;
        f32x4v val = {0.f, 0.f, 0.f, 0.f};
        if (r < 16) val = *reinterpret_cast<const f32x4v*>(src + (long)r * stride + 4 * cq);
#endif
        rg.v[u] = val;
    }
}
template <int NT, int NV>
__device__ __forceinline__ void tr_store(const Regs<NV>& rg, float* __restrict__ dst, int ld, int cols) {
    constexpr int nt = NT;
    const int per_row = cols >> 2;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int f = threadIdx.x + u * nt;
#if ATT_TR_ROWFAST
        const int r = f & 15, cq = f >> 4;
        if (cq < per_row) {
#else
        const int r = f / per_row, cq = f - r * per_row;
        if (r < 16) {
#endif
#pragma unroll
            for (int e = 0; e < 4; ++e) dst[(4 * cq + e) * ld + r] = rg.v[u][e];
        }
    }
}
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

template <int NJ, int NW>
__global__ __launch_bounds__(64 * NW) void attn_fwd_kernel(const AttnFwdP p) {
    constexpr int NTH = 64 * NW;
    constexpr int NV_K = (NJ + NW - 1) / NW;   // float4 per thread for a [16*NJ x 16] chunk
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int SKV = 16 * NJ;
    constexpr int LDJ = SKV + ((SKV % 32 == 16) ? 0 : 16);    // [c][j] images: 4 lane groups 16 banks apart
    constexpr int LDN1 = 20;                                   // W1 chunk image [j][nn]
    constexpr int LDJ2 = SKV + 4;                              // W2 chunk image [nn][j]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int TQ = 16 * NW;
    constexpr int LDQ = TQ + ((TQ % 32 == 16) ? 0 : 16);
    const int r16 = lane & 15, g = lane >> 4;
    const int b = blockIdx.y;
    const int q0 = blockIdx.x * TQ;                            // first query of the workgroup
    const int nq = min(TQ, p.Sq - q0);                         // queries of this workgroup (multiple of 16)
    const bool active = 16 * wave < nq;
    const int iq = q0 + 16 * wave + r16;                       // this lane's query (valid if active)
    const int D = p.H * p.hd;

    const float* qb = p.q + ((long)b * p.Sq + q0) * D;
    const float* kb = p.k + (long)b * p.Skv * D;
    const float* vb = p.v + (long)b * p.Skv * D;

    auto bufK = [&](int i) { return smem + i * 16 * LDJ; };
    auto bufQ = [&](int i) { return smem + 32 * LDJ + i * 16 * LDQ; };

    f32x4v accR[NJ];
#pragma unroll
    for (int t = 0; t < NJ; ++t) accR[t] = (f32x4v){0.f, 0.f, 0.f, 0.f};

    // ================= phase 1: R^T[j,i] = sum_c K_all[j,c] Q_all[i,c] =================
    {
        Regs<NV_K> rk; Regs<NV_Q> rq;
        const int nch = (D + 15) / 16;
        km_load<NTH>(rk, kb, D, SKV, 0, D);
        km_load<NTH>(rq, qb, D, nq, 0, D);
        km_store<NTH>(rk, bufK(0), LDJ, SKV);
        km_store<NTH>(rq, bufQ(0), LDQ, nq);
        __syncthreads();
#pragma unroll 1
        for (int c = 0; c < nch; ++c) {
            const int cur = c & 1;
            if (c + 1 < nch) {
                km_load<NTH>(rk, kb, D, SKV, 16 * (c + 1), D);
                km_load<NTH>(rq, qb, D, nq, 16 * (c + 1), D);
            }
            if (active) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const float bq = bufQ(cur)[(4 * s + g) * LDQ + 16 * wave + r16];
#pragma unroll
                    for (int t = 0; t < NJ; ++t)
                        accR[t] = MFMA16(bufK(cur)[(4 * s + g) * LDJ + 16 * t + r16], bq, accR[t]);
                    __builtin_amdgcn_sched_barrier(0);       // keep at most NJ operand loads in flight
                }
            }
            if (c + 1 < nch) {
                km_store<NTH>(rk, bufK(cur ^ 1), LDJ, SKV);
                km_store<NTH>(rq, bufQ(cur ^ 1), LDQ, nq);
            }
            __syncthreads();
        }
    }
    if (active) {   // save R[b,i,j] (4 consecutive keys per register group)
        float* Rrow = p.R + ((long)b * p.Sq + iq) * p.Skv;
#pragma unroll
        for (int t = 0; t < NJ; ++t) *reinterpret_cast<f32x4v*>(Rrow + 16 * t + 4 * g) = accR[t];
    }

    // ================= phase 2: M^T = W2 gelu(W1 R^T + b1) + b2 =================
    f32x4v accM[NJ];
#pragma unroll
    for (int t = 0; t < NJ; ++t) accM[t] = (f32x4v){0.f, 0.f, 0.f, 0.f};
    {
        const float inv1 = 1.0f / p.s1[0], inv2 = 1.0f / p.s2[0];
        auto bufW1 = [&](int i) { return smem + i * SKV * LDN1; };
        auto bufW2 = [&](int i) { return smem + 2 * SKV * LDN1 + i * 16 * LDJ2; };
        const int nch = 2 * NJ;                                 // 2*Skv hidden units, 16 per chunk
        Regs<NV_K> r1, r2;
        tr_load<NTH>(r1, p.w1, SKV, SKV);                            // rows n0..n0+15 of W1 [2Skv, Skv]
        km_load<NTH>(r2, p.w2, 2 * SKV, SKV, 0, 2 * SKV);            // columns n0..n0+15 of W2 [Skv, 2Skv]
        tr_store<NTH>(r1, bufW1(0), LDN1, SKV);
        km_store<NTH>(r2, bufW2(0), LDJ2, SKV);
        __syncthreads();
#pragma unroll 1
        for (int c = 0; c < nch; ++c) {
            const int cur = c & 1, n0 = 16 * c;
            if (c + 1 < nch) {
                tr_load<NTH>(r1, p.w1 + (long)(n0 + 16) * SKV, SKV, SKV);
                km_load<NTH>(r2, p.w2, 2 * SKV, SKV, n0 + 16, 2 * SKV);
            }
            if (active) {
                // two partial accumulators: a single 16x16x4 chain would stall on its 40-cycle dependent latency
                f32x4v hid = {0.f, 0.f, 0.f, 0.f}, hid2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < NJ; ++t) {
#pragma unroll
                    for (int r = 0; r < 4; r += 2) {
                        hid = MFMA16(bufW1(cur)[(16 * t + 4 * g + r) * LDN1 + r16], accR[t][r], hid);
                        hid2 = MFMA16(bufW1(cur)[(16 * t + 4 * g + r + 1) * LDN1 + r16], accR[t][r + 1], hid2);
                    }
                }
                hid = hid + hid2;
                const f32x4v bb = *reinterpret_cast<const f32x4v*>(p.b1 + n0 + 4 * g);
                f32x4v pre, act;
#pragma unroll
                for (int r = 0; r < 4; ++r) { pre[r] = hid[r] * inv1 + bb[r]; act[r] = gelu_erf_f(pre[r]); }
                const long ho = ((long)b * p.Sq + iq) * (2 * SKV) + n0 + 4 * g;
                *reinterpret_cast<f32x4v*>(p.hp + ho) = pre;
                *reinterpret_cast<f32x4v*>(p.hg + ho) = act;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#pragma unroll
                    for (int t = 0; t < NJ; ++t)
                        accM[t] = MFMA16(bufW2(cur)[(4 * g + r) * LDJ2 + 16 * t + r16], act[r], accM[t]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (c + 1 < nch) {
                tr_store<NTH>(r1, bufW1(cur ^ 1), LDN1, SKV);
                km_store<NTH>(r2, bufW2(cur ^ 1), LDJ2, SKV);
            }
            __syncthreads();
        }
        // the mask tile leaves the registers here (it would otherwise stay live across the head loop):
        // Mk[b,i,j] is written once and re-read per head from L2 (NJ float4 per lane)
        if (active) {
            float* Mrow = p.Mk + ((long)b * p.Sq + iq) * p.Skv;
            constexpr int BG = 4;                              // bias chunks requested four at a time (see attn_bwd_q_kernel)
#pragma unroll
            for (int t0 = 0; t0 < NJ; t0 += BG) {
                f32x4v bb[BG];
#pragma unroll
                for (int u = 0; u < BG; ++u)
                    if (t0 + u < NJ) bb[u] = *reinterpret_cast<const f32x4v*>(p.b2 + 16 * (t0 + u) + 4 * g);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < BG; ++u) {
                    if (t0 + u < NJ) {
                        f32x4v m;
#pragma unroll
                        for (int r = 0; r < 4; ++r) m[r] = accM[t0 + u][r] * inv2 + bb[u][r];
                        *reinterpret_cast<f32x4v*>(Mrow + 16 * (t0 + u) + 4 * g) = m;
                    }
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }

    // ================= phase 3: per head  softmax(scale K_h Q_h^T + M^T) , O^T = V_h^T P^T =================
    const int hd = p.hd;
    const int DT = (hd + 15) / 16;                              // output d-tiles (<= 8)
    const int hdp = 16 * DT;
    const int LDV = hdp + 4;                                    // 4*LDV % 32 == 16
    float* bufV = smem + 32 * LDJ + 32 * LDQ;                   // the WHOLE V_h [Skv][LDV], filled during the QK loop
    const int v_per_row = hdp >> 2;                             // float4 per V row
    const int v_total = SKV * v_per_row;
    const int nchq = (hd + 15) / 16;                            // QK^T column chunks
    const int v_share = (v_total + nchq - 1) / nchq;            // float4 of V staged per QK chunk (<= NV_K * NTH)
#pragma unroll 1
    for (int h = 0; h < p.H; ++h) {
        const float* qh = qb + h * hd;
        const float* kh = kb + h * hd;
        const float* vh = vb + h * hd;
        f32x4v accS[NJ];
#pragma unroll
        for (int t = 0; t < NJ; ++t) accS[t] = (f32x4v){0.f, 0.f, 0.f, 0.f};
        {
            Regs<NV_K> rk; Regs<NV_Q> rq; Regs<NV_K> rv;
            auto v_load = [&](int c) {
#pragma unroll
                for (int u = 0; u < NV_K; ++u) {
                    const int f = c * v_share + tid + u * NTH;
                    const int row = f / v_per_row, cq = f - row * v_per_row;
                    f32x4v val = {0.f, 0.f, 0.f, 0.f};
                    if (tid + u * NTH < v_share && f < v_total && 4 * cq < hd)
                        val = *reinterpret_cast<const f32x4v*>(vh + (long)row * D + 4 * cq);
                    rv.v[u] = val;
                }
            };
            auto v_store = [&](int c) {
#pragma unroll
                for (int u = 0; u < NV_K; ++u) {
                    const int f = c * v_share + tid + u * NTH;
                    const int row = f / v_per_row, cq = f - row * v_per_row;
                    if (tid + u * NTH < v_share && f < v_total)
                        *reinterpret_cast<f32x4v*>(bufV + row * LDV + 4 * cq) = rv.v[u];
                }
            };
            km_load<NTH>(rk, kh, D, SKV, 0, hd);
            km_load<NTH>(rq, qh, D, nq, 0, hd);
            km_store<NTH>(rk, bufK(0), LDJ, SKV);
            km_store<NTH>(rq, bufQ(0), LDQ, nq);
            __syncthreads();
#pragma unroll 1
            for (int c = 0; c < nchq; ++c) {
                const int cur = c & 1;
                v_load(c);
                if (c + 1 < nchq) {
                    km_load<NTH>(rk, kh, D, SKV, 16 * (c + 1), hd);
                    km_load<NTH>(rq, qh, D, nq, 16 * (c + 1), hd);
                }
                if (active) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const float bq = bufQ(cur)[(4 * s + g) * LDQ + 16 * wave + r16];
#pragma unroll
                        for (int t = 0; t < NJ; ++t)
                            accS[t] = MFMA16(bufK(cur)[(4 * s + g) * LDJ + 16 * t + r16], bq, accS[t]);
                    }
                }
                v_store(c);
                if (c + 1 < nchq) {
                    km_store<NTH>(rk, bufK(cur ^ 1), LDJ, SKV);
                    km_store<NTH>(rq, bufQ(cur ^ 1), LDQ, nq);
                }
                __syncthreads();
            }
        }
        // softmax over the keys: 4*NJ in-lane values, then the 4 lane groups (xor 16, xor 32)
        float mx = -INFINITY;
        {
            // own writes of this thread; the pointer is laundered so that the compiler re-loads the tile per
            // head instead of forwarding the stored values (which would keep 4*NJ registers live)
            const float* Mrow = p.Mk + ((long)b * p.Sq + (active ? iq : q0)) * p.Skv;
            asm volatile("" : "+v"(Mrow));
#pragma unroll
            for (int t = 0; t < NJ; ++t) {
                const f32x4v m = *reinterpret_cast<const f32x4v*>(Mrow + 16 * t + 4 * g);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    accS[t][r] = accS[t][r] * p.scale + m[r];
                    mx = fmaxf(mx, accS[t][r]);
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
                accS[t][r] = __expf(accS[t][r] - mx);        // v_exp_f32 form: arguments <= 0, error ~1e-6 relative
                sum += accS[t][r];
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int t = 0; t < NJ; ++t) accS[t] = accS[t] * inv;
        if (active && p.P) {
            float* Prow = p.P + (((long)b * p.H + h) * p.Sq + iq) * p.Skv;
#pragma unroll
            for (int t = 0; t < NJ; ++t) *reinterpret_cast<f32x4v*>(Prow + 16 * t + 4 * g) = accS[t];
        }
        // O^T[d,i] = sum_j V_h[j,d] P^T[j,i]   (V_h complete in LDS since the last barrier of the QK loop)
        if (active) {
            float* orow = p.out + ((long)b * p.Sq + iq) * D + h * hd;
#pragma unroll 1
            for (int d = 0; d < DT; d += 2) {                   // two d-tiles = two independent MFMA chains
                f32x4v accO = {0.f, 0.f, 0.f, 0.f}, accO2 = {0.f, 0.f, 0.f, 0.f};
                const float* vcol = bufV + 16 * d + r16;
                const int d2 = (d + 1 < DT) ? 16 : 0;           // odd DT: the second chain redoes tile d (discarded)
#pragma unroll
                for (int t = 0; t < NJ; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        accO = MFMA16(vcol[(16 * t + 4 * g + r) * LDV], accS[t][r], accO);
                        accO2 = MFMA16(vcol[(16 * t + 4 * g + r) * LDV + d2], accS[t][r], accO2);
                    }
                if (16 * d + 4 * g < hd) *reinterpret_cast<f32x4v*>(orow + 16 * d + 4 * g) = accO;
                if (d + 1 < DT && 16 * (d + 1) + 4 * g < hd)
                    *reinterpret_cast<f32x4v*>(orow + 16 * (d + 1) + 4 * g) = accO2;
            }
        }
        __syncthreads();                                        // bufV / bufK / bufQ are rewritten by the next head
    }
}

// =====================================================================================================
// Backward of the attention core, two kernels with the same tiling/staging as the forward and NO cross-wave
// reductions (the mask-MLP backward and the dR terms stay GEMMs):
//   Q side  (queries on the lanes, one wave = 16 queries, all keys on the accumulator rows)
//       dP^T[j,i] = sum_d V_h[j,d] dO_h[i,d]          (same loop as QK^T: V in the K role, dO in the Q role)
//       delta_i   = sum_j P[i,j] dP[i,j]               (in-lane + 2 shuffles)
//       dS        = P o (dP - delta)                   -> written once to HBM for the K/V side
//       dM       += dS                                 (summed over heads in registers)
//       dQ^T[d,i] = scale sum_j K_h[j,d] dS^T[j,i]     (same block as PV: K_h stripe in the V role)
//   KV side (keys on the lanes, one wave = 16 keys, all queries on the accumulator rows)
//       dV^T[d,j] = sum_i dO_h[i,d] P[i,j]             (PV block: dO_h stripe, P tiles as B operand)
//       dK^T[d,j] = scale sum_i Q_h[i,d] dS[i,j]       (PV block: Q_h stripe, dS tiles as B operand)
// =====================================================================================================
struct AttnBwdP {
    const float* q; const float* k; const float* v; const float* dout;
    const float* P;              // [B,H,Sq,Skv] probabilities saved by the forward
    float* dS;                   // [B,H,Sq,Skv] written by the Q side, read by the KV side
    float* dq; float* dk; float* dv;
    float* dM;                   // [B,Sq,Skv]
    int B, Sq, Skv, H, hd;
    float scale;
};

template <int NJ, int NW>
__global__ __launch_bounds__(64 * NW) void attn_bwd_q_kernel(const AttnBwdP p) {
    constexpr int NTH = 64 * NW;
    constexpr int NV_K = (NJ + NW - 1) / NW;   // float4 per thread for a [16*NJ x 16] chunk
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int SKV = 16 * NJ;
    constexpr int LDJ = SKV + ((SKV % 32 == 16) ? 0 : 16);
    constexpr int TQ = 16 * NW;
    constexpr int LDQ = TQ + ((TQ % 32 == 16) ? 0 : 16);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    const int b = blockIdx.y;
    const int q0 = blockIdx.x * TQ;
    const int nq = min(TQ, p.Sq - q0);
    const bool active = 16 * wave < nq;
    const int iq = q0 + 16 * wave + r16;
    const int D = p.H * p.hd;
    const int hd = p.hd;
    const int DT = (hd + 15) / 16, hdp = 16 * DT, LDV = hdp + 4;
    auto bufK = [&](int i) { return smem + i * 16 * LDJ; };
    auto bufQ = [&](int i) { return smem + 32 * LDJ + i * 16 * LDQ; };
    float* bufV = smem + 32 * LDJ + 32 * LDQ;                   // whole K_h stripe for the dQ block
    const int v_per_row = hdp >> 2, v_total = SKV * v_per_row;
    const int nchq = (hd + 15) / 16;
    const int v_share = (v_total + nchq - 1) / nchq;

    const float* dob = p.dout + ((long)b * p.Sq + q0) * D;
    const float* kb = p.k + (long)b * p.Skv * D;
    const float* vb = p.v + (long)b * p.Skv * D;

    f32x4v accM[NJ];
#pragma unroll
    for (int t = 0; t < NJ; ++t) accM[t] = (f32x4v){0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
    for (int h = 0; h < p.H; ++h) {
        const float* doh = dob + h * hd;
        const float* kh = kb + h * hd;
        const float* vh = vb + h * hd;
        f32x4v accD[NJ];
#pragma unroll
        for (int t = 0; t < NJ; ++t) accD[t] = (f32x4v){0.f, 0.f, 0.f, 0.f};
        {
            Regs<NV_K> rk; Regs<NV_Q> rq; Regs<NV_K> rv;
            auto s_load = [&](int c) {                         // K_h stripe, a share per chunk iteration
#pragma unroll
                for (int u = 0; u < NV_K; ++u) {
                    const int f = c * v_share + tid + u * NTH;
                    const int row = f / v_per_row, cq = f - row * v_per_row;
                    f32x4v val = {0.f, 0.f, 0.f, 0.f};
                    if (tid + u * NTH < v_share && f < v_total && 4 * cq < hd)
                        val = *reinterpret_cast<const f32x4v*>(kh + (long)row * D + 4 * cq);
                    rv.v[u] = val;
                }
            };
            auto s_store = [&](int c) {
#pragma unroll
                for (int u = 0; u < NV_K; ++u) {
                    const int f = c * v_share + tid + u * NTH;
                    const int row = f / v_per_row, cq = f - row * v_per_row;
                    if (tid + u * NTH < v_share && f < v_total)
                        *reinterpret_cast<f32x4v*>(bufV + row * LDV + 4 * cq) = rv.v[u];
                }
            };
            km_load<NTH>(rk, vh, D, SKV, 0, hd);
            km_load<NTH>(rq, doh, D, nq, 0, hd);
            km_store<NTH>(rk, bufK(0), LDJ, SKV);
            km_store<NTH>(rq, bufQ(0), LDQ, nq);
            __syncthreads();
#pragma unroll 1
            for (int c = 0; c < nchq; ++c) {
                const int cur = c & 1;
                s_load(c);
                if (c + 1 < nchq) {
                    km_load<NTH>(rk, vh, D, SKV, 16 * (c + 1), hd);
                    km_load<NTH>(rq, doh, D, nq, 16 * (c + 1), hd);
                }
                if (active) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const float bq = bufQ(cur)[(4 * s + g) * LDQ + 16 * wave + r16];
#pragma unroll
                        for (int t = 0; t < NJ; ++t)
                            accD[t] = MFMA16(bufK(cur)[(4 * s + g) * LDJ + 16 * t + r16], bq, accD[t]);
                    }
                }
                s_store(c);
                if (c + 1 < nchq) {
                    km_store<NTH>(rk, bufK(cur ^ 1), LDJ, SKV);
                    km_store<NTH>(rq, bufQ(cur ^ 1), LDQ, nq);
                }
                __syncthreads();
            }
        }
        // delta, dS = P o (dP - delta); P is read twice (second time from L2) instead of being held in registers.
        // P and dS through __restrict__ locals: with the struct's plain pointers the compiler had to assume that the dS
        // store of tile t may alias the P load of tile t + 1 and put `s_waitcnt vmcnt(0)` between them (ISA, round 4:
        // 18-24 of the kernel's 26-35 global loads were followed by a full drain) — one exposed L2 round trip per key tile
        // and head.
        const long prow = (((long)b * p.H + h) * p.Sq + (active ? iq : q0)) * p.Skv;
        const float* __restrict__ Prow = p.P + prow + 4 * g;
        float* __restrict__ dSrow = p.dS + prow + 4 * g;
        // ... and in GROUPS of PG tiles requested together: left to itself the compiler reuses one register quad for all
        // NJ loads of a loop — load, s_waitcnt vmcnt(0), use, eleven times over — i.e. 2 NJ exposed round trips per head
        constexpr int PG = 4;
        float part = 0.f;
#pragma unroll
        for (int t0 = 0; t0 < NJ; t0 += PG) {
            f32x4v pv[PG];
#pragma unroll
            for (int u = 0; u < PG; ++u)
                if (t0 + u < NJ) pv[u] = *reinterpret_cast<const f32x4v*>(Prow + 16 * (t0 + u));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < PG; ++u)
                if (t0 + u < NJ)
#pragma unroll
                    for (int r = 0; r < 4; ++r) part += pv[u][r] * accD[t0 + u][r];
        }
        part += __shfl_xor(part, 16, 64);
        part += __shfl_xor(part, 32, 64);
#pragma unroll
        for (int t0 = 0; t0 < NJ; t0 += PG) {
            f32x4v pv[PG];
#pragma unroll
            for (int u = 0; u < PG; ++u)
                if (t0 + u < NJ) pv[u] = *reinterpret_cast<const f32x4v*>(Prow + 16 * (t0 + u));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < PG; ++u) {
                if (t0 + u < NJ) {
                    const int t = t0 + u;
#pragma unroll
                    for (int r = 0; r < 4; ++r) accD[t][r] = pv[u][r] * (accD[t][r] - part);
                    accM[t] = accM[t] + accD[t];
                    if (active) *reinterpret_cast<f32x4v*>(dSrow + 16 * t) = accD[t];
                }
            }
        }
        // dQ^T[d,i] = scale * sum_j K_h[j,d] dS^T[j,i]
        if (active) {
            float* qrow = p.dq + ((long)b * p.Sq + iq) * D + h * hd;
#pragma unroll 1
            for (int d = 0; d < DT; d += 2) {
                f32x4v accO = {0.f, 0.f, 0.f, 0.f}, accO2 = {0.f, 0.f, 0.f, 0.f};
                const float* kcol = bufV + 16 * d + r16;
                const int d2 = (d + 1 < DT) ? 16 : 0;
#pragma unroll
                for (int t = 0; t < NJ; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        accO = MFMA16(kcol[(16 * t + 4 * g + r) * LDV], accD[t][r], accO);
                        accO2 = MFMA16(kcol[(16 * t + 4 * g + r) * LDV + d2], accD[t][r], accO2);
                    }
                if (16 * d + 4 * g < hd) *reinterpret_cast<f32x4v*>(qrow + 16 * d + 4 * g) = accO * p.scale;
                if (d + 1 < DT && 16 * (d + 1) + 4 * g < hd)
                    *reinterpret_cast<f32x4v*>(qrow + 16 * (d + 1) + 4 * g) = accO2 * p.scale;
            }
        }
        __syncthreads();
    }
    if (active) {
        float* mrow = p.dM + ((long)b * p.Sq + iq) * p.Skv;
#pragma unroll
        for (int t = 0; t < NJ; ++t) *reinterpret_cast<f32x4v*>(mrow + 16 * t + 4 * g) = accM[t];
    }
}

// NI = query tiles (accumulator rows), one wave = 16 keys on the lanes
template <int NI, int NW>
__global__ __launch_bounds__(64 * NW) void attn_bwd_kv_kernel(const AttnBwdP p) {
    constexpr int NTH = 64 * NW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int SQ = 16 * NI;
    constexpr int TK = 16 * NW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    const int b = blockIdx.y;
    const int k0 = blockIdx.x * TK;
    const int nk = min(TK, p.Skv - k0);
    const bool active = 16 * wave < nk;
    const int jk = k0 + 16 * wave + r16;                        // this lane's key
    const int D = p.H * p.hd;
    const int hd = p.hd;
    const int DT = (hd + 15) / 16, hdp = 16 * DT, LDV = hdp + 4;
    float* bufV = smem;                                         // whole dO_h / Q_h stripe [Sq][LDV]
    const int v_per_row = hdp >> 2, v_total = SQ * v_per_row;
    const float* qb = p.q + (long)b * p.Sq * D;
    const float* dob = p.dout + (long)b * p.Sq * D;

    // [Sq x hd] stripe (row stride D) -> bufV, four 16-byte loads in flight per thread and trip: as a rolled
    // load -> store loop every trip waited out a full memory round trip (6-8 per stripe, two stripes per head)
    constexpr int SB = 4;
    auto stage = [&](const float* src) {
#pragma unroll 1
        for (int f0 = tid; f0 < v_total; f0 += SB * NTH) {
            f32x4v val[SB];
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const int f = f0 + u * NTH;
                const int row = f / v_per_row, cq = f - row * v_per_row;
                val[u] = (f32x4v){0.f, 0.f, 0.f, 0.f};
                if (f < v_total && 4 * cq < hd) val[u] = *reinterpret_cast<const f32x4v*>(src + (long)row * D + 4 * cq);
            }
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const int f = f0 + u * NTH;
                const int row = f / v_per_row, cq = f - row * v_per_row;
                if (f < v_total) *reinterpret_cast<f32x4v*>(bufV + row * LDV + 4 * cq) = val[u];
            }
        }
    };
    auto contract = [&](const f32x4v (&X)[NI], float* out_row, float scale) {   // out^T[d,j] = sum_i stripe[i,d] X[i,j]
#pragma unroll 1
        for (int d = 0; d < DT; d += 2) {
            f32x4v accO = {0.f, 0.f, 0.f, 0.f}, accO2 = {0.f, 0.f, 0.f, 0.f};
            const float* col = bufV + 16 * d + r16;
            const int d2 = (d + 1 < DT) ? 16 : 0;
#pragma unroll
            for (int t = 0; t < NI; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    accO = MFMA16(col[(16 * t + 4 * g + r) * LDV], X[t][r], accO);
                    accO2 = MFMA16(col[(16 * t + 4 * g + r) * LDV + d2], X[t][r], accO2);
                }
            if (16 * d + 4 * g < hd) *reinterpret_cast<f32x4v*>(out_row + 16 * d + 4 * g) = accO * scale;
            if (d + 1 < DT && 16 * (d + 1) + 4 * g < hd)
                *reinterpret_cast<f32x4v*>(out_row + 16 * (d + 1) + 4 * g) = accO2 * scale;
        }
    };

#pragma unroll 1
    for (int h = 0; h < p.H; ++h) {
        const long base = ((long)b * p.H + h) * p.Sq * p.Skv + (active ? jk : k0);
        f32x4v X[NI];
        // P tiles [i rows, key lanes]: 16 consecutive keys = one 64-byte segment per row
#pragma unroll
        for (int t = 0; t < NI; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) X[t][r] = p.P[base + (long)(16 * t + 4 * g + r) * p.Skv];
        stage(dob + h * hd);
        __syncthreads();
        if (active) contract(X, p.dv + ((long)b * p.Skv + jk) * D + h * hd, 1.0f);
        __syncthreads();
#pragma unroll
        for (int t = 0; t < NI; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) X[t][r] = p.dS[base + (long)(16 * t + 4 * g + r) * p.Skv];
        stage(qb + h * hd);
        __syncthreads();
        if (active) contract(X, p.dk + ((long)b * p.Skv + jk) * D + h * hd, p.scale);
        __syncthreads();
    }
}

template <int NJ, int NW>
int launch_bwd(const AttnBwdP& p, hipStream_t s) {
    const int tiles = p.Sq / 16;
    const int TQ = 16 * NW, SKV = 16 * NJ;
    const int LDJ = SKV + ((SKV % 32 == 16) ? 0 : 16);
    const int LDQ = TQ + ((TQ % 32 == 16) ? 0 : 16);
    const int hdp = (p.hd + 15) / 16 * 16, LDV = hdp + 4;
    const size_t lds_q = sizeof(float) * (size_t)(32 * LDJ + 32 * LDQ + SKV * LDV);
    const size_t lds_kv = sizeof(float) * (size_t)(SKV * LDV);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_q_kernel<NJ, NW>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_kv_kernel<NJ, NW>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_kv);
    if (e != hipSuccess) return (int)e;
    dim3 grid((tiles + NW - 1) / NW, p.B);
    hipLaunchKernelGGL((attn_bwd_q_kernel<NJ, NW>), grid, dim3(64 * NW), lds_q, s, p);
    CALM_LAUNCH_CHECK();
    hipLaunchKernelGGL((attn_bwd_kv_kernel<NJ, NW>), grid, dim3(64 * NW), lds_kv, s, p);
    CALM_LAUNCH_CHECK();
    return 0;
}

#ifndef ATT_NW11
#define ATT_NW11 11    // waves per workgroup for the 11-tile (S=176) instantiation: one wave per query tile, a single
#endif                 // pass (A/B: 684 -> 473 us against 6 waves in two passes; 4 waves: slower still)
#ifndef ATT_NW14
#define ATT_NW14 7     // 14-tile (S=224) instantiation
#endif
inline int pick_waves(int tiles) {
    if (tiles == 11) return ATT_NW11;
    if (tiles == 14) return ATT_NW14;
    const int groups = (tiles + 7) / 8;
    return (tiles + groups - 1) / groups;
}

template <int NJ, int NW>
int launch_fwd(const AttnFwdP& p, hipStream_t s) {
    const int tiles = p.Sq / 16;
    const int TQ = 16 * NW;
    const int SKV = 16 * NJ;
    const int LDJ = SKV + ((SKV % 32 == 16) ? 0 : 16);
    const int LDQ = TQ + ((TQ % 32 == 16) ? 0 : 16);
    const int hdp = (p.hd + 15) / 16 * 16;
    const int LDV = hdp + 4;
    const int ph13 = 32 * LDJ + 32 * LDQ + SKV * LDV;       // K/Q chunk double buffers + the whole V_h
    const int ph2 = 2 * SKV * 20 + 32 * (SKV + 4);
    const size_t lds = sizeof(float) * (size_t)(ph13 > ph2 ? ph13 : ph2);
    if (lds > 160 * 1024) return CALM_E_UNSUPP;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_kernel<NJ, NW>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    dim3 grid((tiles + NW - 1) / NW, p.B);
    hipLaunchKernelGGL((attn_fwd_kernel<NJ, NW>), grid, dim3(64 * NW), lds, s, p);
    CALM_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" {

int calm_attention_fwd_supported(int32_t Sq, int32_t Skv, int32_t H, int32_t hd) {
    if (Sq <= 0 || Skv <= 0 || H <= 0 || hd <= 0) return 0;
    if (Sq != Skv || (Sq & 15) || (hd & 3) || hd > 128) return 0;   // every attention of the model has Sq == Skv
    const int nj = Skv / 16;
    if (!(nj == 2 || nj == 3 || nj == 5 || nj == 8 || nj == 11 || nj == 14)) return 0;
    // staging registers: a [Skv x 16] chunk is 4*Skv float4 (NV_K per thread), a [16 x hd_pad] V chunk 4*hd_pad
    const int nt = 64 * pick_waves(Sq / 16);
    const int hdp = (hd + 15) / 16 * 16, ldj = Skv + ((Skv % 32 == 16) ? 0 : 16), tq = nt / 4;
    const int ldq = tq + ((tq % 32 == 16) ? 0 : 16);
    if ((32 * ldj + 32 * ldq + Skv * (hdp + 4)) * 4 > 160 * 1024) return 0;   // whole V_h must fit in LDS
    return 1;
}

int calm_attention_fwd(const float* q, const float* k, const float* v, const float* w1, const float* b1,
                       const float* s1, const float* w2, const float* b2, const float* s2, float* out, float* R,
                       float* hp, float* hg, float* Mk, float* P, int32_t B, int32_t Sq, int32_t Skv, int32_t H, int32_t hd,
                       void* stream) {
    if (!q || !k || !v || !w1 || !b1 || !s1 || !w2 || !b2 || !s2 || !out || !R || !hp || !hg || !Mk || B <= 0)
        return CALM_E_INVAL;
    if (!calm_attention_fwd_supported(Sq, Skv, H, hd)) return CALM_E_UNSUPP;
    if (B > 65535) return CALM_E_UNSUPP;
    AttnFwdP p{q, k, v, w1, b1, s1, w2, b2, s2, out, R, hp, hg, Mk, P, B, Sq, Skv, H, hd, 1.0f / sqrtf((float)hd)};
    hipStream_t s = as_stream(stream);
    switch (Skv / 16) {
        // <key tiles, waves per workgroup = pick_waves(Sq/16)>
        case 2: return launch_fwd<2, 2>(p, s);
        case 3: return launch_fwd<3, 3>(p, s);
        case 5: return launch_fwd<5, 5>(p, s);
        case 8: return launch_fwd<8, 8>(p, s);
        case 11: return launch_fwd<11, ATT_NW11>(p, s);
        case 14: return launch_fwd<14, ATT_NW14>(p, s);
    }
    return CALM_E_UNSUPP;
}

// Measured on MI355X (scripts/ab_attn_bwd.py, same process): the two fused launches beat the GEMM composition
// for head dims <= 64 (S=128/80 stages of Small-224, every stage of Base-224: 1.1-1.7x) and tie or lose above
// (hd 112: 1.00x, hd 88: 0.76x), where the per-head GEMMs already fill 128-wide tiles.
int calm_attention_bwd_preferred(int32_t Sq, int32_t Skv, int32_t H, int32_t hd) {
    // measured against the composition of batched GEMMs + softmax_bwd + sum_heads (scripts/ab_attn_bwd.py): faster for
    // head dims <= 64 (1.0-1.6x) and for the 11-tile stage (S=176, hd 88: 0.68 vs 0.77 ms, one wave per key/query
    // tile); at S=224 with hd 112 the composition still wins (1.36 vs 1.43 ms)
    return calm_attention_fwd_supported(Sq, Skv, H, hd) && (hd <= 64 || Skv / 16 == 11);
}

int calm_attention_bwd(const float* q, const float* k, const float* v, const float* dout, const float* P, float* dS,
                       float* dq, float* dk, float* dv, float* dM, int32_t B, int32_t Sq, int32_t Skv, int32_t H,
                       int32_t hd, void* stream) {
    if (!q || !k || !v || !dout || !P || !dS || !dq || !dk || !dv || !dM || B <= 0) return CALM_E_INVAL;
    if (!calm_attention_fwd_supported(Sq, Skv, H, hd)) return CALM_E_UNSUPP;
    if (B > 65535) return CALM_E_UNSUPP;
    AttnBwdP p{q, k, v, dout, P, dS, dq, dk, dv, dM, B, Sq, Skv, H, hd, 1.0f / sqrtf((float)hd)};
    hipStream_t s = as_stream(stream);
    switch (Skv / 16) {
        case 2: return launch_bwd<2, 2>(p, s);
        case 3: return launch_bwd<3, 3>(p, s);
        case 5: return launch_bwd<5, 5>(p, s);
        case 8: return launch_bwd<8, 8>(p, s);
        case 11: return launch_bwd<11, ATT_NW11>(p, s);
        case 14: return launch_bwd<14, ATT_NW14>(p, s);
    }
    return CALM_E_UNSUPP;
}

}  // extern "C"
