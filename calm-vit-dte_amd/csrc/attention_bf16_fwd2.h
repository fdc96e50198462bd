// Pipelined forward of the fused bf16 latent-mask attention (included by attention_bf16.hip, inside its anonymous
// namespace; same arithmetic, fragment orders and LDS image geometries as attn16_fwd_kernel there — see that file's
// header for the dataflow).  What changes is how the operands reach LDS.
//
// In-kernel stamps of attn16_fwd_kernel (S = 224, 12 heads of 56; ATT16_STAMP) gave 78k / 96k / 162k cycles per
// workgroup for the three phases against 9k / 13k / 22k cycles of MFMA work per SIMD: every K / W1 / W2 chunk and every
// head's K_h / V_h went global -> registers -> LDS one step ahead of its use, and a step's matrix work (0.5-1k cycles) is
// far shorter than a memory round trip, so each phase was a chain of exposed load latencies.  Here every operand is
// brought in by LDS-DMA (lds_dma.h: no staging registers, no ds_write) into a ring of stages that runs 2-3 steps ahead:
//   phase 1   R^T = K_all Q_all^T   stage = 64 columns of all keys AND of this workgroup's queries, 128-byte rows with the
//                                   chunk swizzle c ^ ((row >> 1) & 7) of the GEMM's k-contiguous images
//   phase 2   mask MLP              stage = 32 hidden units: W1 rows [32][keys] and W2 columns [keys][32] in the
//                                   paired-tile images of the first kernel (row strides = 16 mod 32 bytes — the pad
//                                   chunk of a row is a DMA lane that fetches the zero block)
//   phase 3   per head              stage = K_h and V_h [keys][hd] (row stride = 32 mod 64 bytes), next head in flight
// A wave waits for ITS OWN share of a stage with a counted s_waitcnt vmcnt (vector-memory operations retire in issue
// order; `vm_seq` is the wave's ledger of them), then the workgroup barrier publishes the stage and, at the same time,
// frees the stage read one step earlier for the next request.  For the ledger to be exact every global store of the
// kernel goes through a buffer descriptor (masked lanes store to an out-of-range offset: the instruction is still
// issued), and the loops contain no compiler-visible global load except the next head's q fragments, whose wait the
// compiler places where the stage wait already is.  The biases of the mask MLP are read from an LDS table.
#pragma once

using namespace calm_lds_dma;

typedef unsigned au32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void buf_store4(__amdgpu_buffer_rsrc_t rs, unsigned off, bf16x4 v) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(au32x2, v), rs, off, 0, 0);
}

// issue this wave's share of an image of NINSTR LDS-DMA instructions (instruction q covers the 64 16-byte chunks
// 64 q .. 64 q + 63 of the image in LDS order); addr_of(L) = global address of image chunk L.  Returns the count issued.
template <int NINSTR, int NW, class F>
__device__ __forceinline__ int dma_issue(unsigned lds_base, int wave, int lane, F addr_of) {
    int n = 0;
#pragma unroll
    for (int i = 0; i < (NINSTR + NW - 1) / NW; ++i) {
        const int q = wave + NW * i;
        if (NINSTR % NW == 0 || q < NINSTR) {
            glds16_addr(addr_of(64 * q + lane), lds_base + 1024u * q);
            ++n;
        }
    }
    return n;
}

// The same from per-lane byte offsets computed ONCE per phase (off[i] for this wave's i-th instruction) and a wave-uniform
// base per instruction: four instructions per request.  (The first version of this kernel evaluated addr_of for every
// request — divisions, 64-bit multiplies, selects against the zero block: 220-350 cycles per LDS-DMA instruction by the
// in-kernel stamps, 72k of a workgroup's 280k cycles at S = 224.)  Lanes whose chunk has no valid source (pad keys, pad
// columns, the pad chunk of a row) re-read a valid chunk instead of the zero block: every such element meets a zero or
// a masked value on the other side of its product, except in the tail steps, which take the addr_of form.
template <int NINSTR, int NW, int PW, class FB>
__device__ __forceinline__ int dma_issue_fast(unsigned lds_base, int wave, const unsigned (&off)[PW], FB base_of) {
    static_assert(PW == (NINSTR + NW - 1) / NW, "offsets per wave");
    int n = 0;
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        const int q = wave + NW * i;
        if (NINSTR % NW == 0 || q < NINSTR) {
            glds16_u(pipe_uniform64(base_of(q)), off[i], lds_base + 1024u * q);
            ++n;
        }
    }
    return n;
}

// one instruction of the above (piece i of this wave), for requests spread between the MFMA groups of a step: an LDS-DMA
// instruction blocks its wave until the texture path takes it (stamps: 140-250 cycles each when all waves of the
// workgroup issue their shares at once right after the barrier, i.e. the waves queue for the path and then all compute
// at once), so each wave issues one or two per group of products, whose matrix time covers the wait.
template <int NINSTR, int NW, int PW, class FB>
__device__ __forceinline__ int dma_piece_fast(int i, unsigned lds_base, int wave, const unsigned (&off)[PW], FB base_of) {
    const int q = wave + NW * i;
    if (i < PW && (NINSTR % NW == 0 || q < NINSTR)) {
        glds16_u(pipe_uniform64(base_of(q)), off[i], lds_base + 1024u * q);
        return 1;
    }
    return 0;
}

// acc[t] += A_t B_ks over t < 2 NP, ks < NKS with the A fragments (16-byte reads at frag(t, ks)) requested one GROUP of GS
// ahead of the products that use them.  sched_barrier pins [reads of the next group | products of this group]: left
// alone, the scheduler serialises read -> wait -> product through one register quad and every product exposes an LDS
// round trip (ISA of the first version of this kernel: 28 x `ds_read2_b64; s_waitcnt lgkmcnt(0); v_mfma` per step).
template <int NP, int NKS, int GS, class FA, class FH>
__device__ __forceinline__ void grouped_products(f32x4v (&acc)[2 * NP], const bf16x8 (&bq)[NKS], FA frag, FH hook) {
    constexpr int NJ = 2 * NP, NF = NJ * NKS, NG = (NF + GS - 1) / GS;     // fragment f = ks NJ + t, groups of GS
    bf16x8 cur[GS], nxt[GS];
#pragma unroll
    for (int i = 0; i < GS; ++i)
        if (i < NF) cur[i] = frag(i % NJ, i / NJ);
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
        if (gi + 1 < NG) {
#pragma unroll
            for (int i = 0; i < GS; ++i) {
                const int f = (gi + 1) * GS + i;
                if (f < NF) nxt[i] = frag(f % NJ, f / NJ);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < GS; ++i) {
            const int f = gi * GS + i;
            if (f < NF) acc[f % NJ] = MFMA_BF16(cur[i], bq[f / NJ], acc[f % NJ]);
        }
        hook(gi);                                    // (staging requests of a later step, behind this group's products)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < GS; ++i) cur[i] = nxt[i];
    }
}

// geometry of the pipelined forward for (NP key-tile pairs, head dim padded to HDP)
template <int NP, int HDP>
struct Fwd2Geo {
    static constexpr int SP = 32 * NP, NW = waves_for(NP), QR = 16 * NW;
    static constexpr int ROWS1 = SP + QR, ST1 = ROWS1 * 128, NI1 = ROWS1 / 8;            // phase 1 stage
    static constexpr int CPR1 = SP / 8 + 1, NI2A = (32 * CPR1 + 63) / 64, NI2B = (SP * 5 + 63) / 64;
    static constexpr int ST2 = (NI2A + NI2B) * 1024;                                      // phase 2 stage (W1 + W2 chunk)
    static constexpr int CPRH = HDP / 8 + 2, NI3 = (SP * CPRH + 63) / 64, ST3 = 2 * NI3 * 1024;   // phase 3 stage (K_h + V_h)
    static constexpr int TBL = 12 * SP;                                                   // b1 [2 SP] + b2 [SP] floats
    static constexpr int AVAIL = 160 * 1024 - TBL;
    static constexpr int ns(int st) { return AVAIL / st > 4 ? 4 : AVAIL / st; }
    static constexpr int NS1 = ns(ST1), NS2 = ns(ST2), NS3 = AVAIL / ST3 >= 2 ? 2 : 1;
    static constexpr int max3(int a, int b, int c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); }
    static constexpr int RING = max3(NS1 * ST1, NS2 * ST2, NS3 * ST3);
    static constexpr int LDS = RING + TBL;
    static constexpr bool OK = NP <= 7 && HDP <= 64 && NS1 >= 2 && NS2 >= 2 && NS3 >= 2;
};

#ifdef ATT16_STAMP
#define F2_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define F2_ADD(acc, a, b) acc += (b) - (a)
#else
#define F2_T(var)
#define F2_ADD(acc, a, b)
#endif

// MASK_ONLY (round 4): phases 1 and 2 alone — R, the hidden states and the mask M are written out and the kernel ends;
// the per-head core then runs as attn16_fwd3_core_kernel (attention_bf16_fwd3.h), one workgroup per (image, head).
template <int NP, int HDP, bool MASK_ONLY = false>
__global__ __launch_bounds__(64 * waves_for(NP), 2) void attn16_fwd2_kernel(const Attn16P p) {
    typedef Fwd2Geo<NP, HDP> G;
    constexpr int NJ = 2 * NP, SP = G::SP, NW = G::NW;
    extern __shared__ __attribute__((aligned(1024))) char smem2[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c16 = lane & 15, g = lane >> 4;
    // 1-D grid of groups x B workgroups.  Workgroup ids go round-robin over the 8 XCDs; the query groups of ONE image are
    // dealt to the same XCD, next to each other in time, so that the second group's K / V / Q requests hit the L2 the
    // first one filled (PMC of the first kernel: 913 MB read per launch at S = 224, 4 TB/s — every image's K fetched four
    // times, V twice: the kernel sat at the HBM roofline of its own redundant traffic)
    int b, qg;
    {
        const int id = blockIdx.x, groups = p.groups;
        if ((p.B & 7) == 0) {
            const int xcd = id & 7, idx = id >> 3;                       // idx-th workgroup of this XCD
            qg = idx % groups;
            b = (idx / groups) * 8 + xcd;
        } else {
            qg = id % groups;
            b = id / groups;
        }
    }
    const int S = p.S, D = p.H * p.hd, hd = p.hd, NH = 2 * S;
    const int q0 = qg * G::QR;                                                            // first query of the workgroup
    const int q_lane = q0 + 16 * wave + c16;
    const bool q_ok = q_lane < S;
    const int q_ld = q_ok ? q_lane : S - 1;
    const __bf16* qb = p.q + (long)b * S * D;
    const __bf16* kb = p.k + (long)b * S * D;
    const __bf16* vb = p.v + (long)b * S * D;
    const char* zero = reinterpret_cast<const char*>(calm_zero_block);
    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)reinterpret_cast<uintptr_t>(smem2));
    float* b1s = reinterpret_cast<float*>(smem2 + G::RING);
    float* b2s = b1s + 2 * SP;
    int vm_seq = 0;                      // vector-memory instructions this wave has issued (ledger, wave-uniform)

    // the bias tables (before any DMA is in flight: the compiler drains these loads with vmcnt(0))
    for (int i = tid; i < 2 * SP; i += 64 * NW) b1s[i] = i < NH ? p.b1[i] : 0.f;
    for (int i = tid; i < SP; i += 64 * NW) b2s[i] = i < S ? p.b2[i] : 0.f;

    f32x4v acc[NJ];
    bf16x8 Rf[NP];
#ifdef ATT16_STAMP
    const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
    unsigned long long tw[3] = {0, 0, 0}, ti[3] = {0, 0, 0}, tc[3] = {0, 0, 0};      // wait+barrier / DMA issue / compute per phase
#endif

    // ================= phase 1: R^T[j,i] = sum_c K_all[j,c] Q_all[i,c], 64 columns per stage =================
    {
        const int nch = (D + 63) / 64;
        constexpr int PW1 = (G::NI1 + NW - 1) / NW;
        unsigned off1[PW1];
#pragma unroll
        for (int i = 0; i < PW1; ++i) {
            const int L = 64 * (wave + NW * i) + lane, r = L >> 3, sc = (L & 7) ^ ((r >> 1) & 7);
            const int src = r < SP ? min(r, S - 1) : min(q0 + r - SP, S - 1);
            off1[i] = (unsigned)((src * D + 8 * sc) * 2);
        }
        const bool tail1 = (D & 63) != 0;
        constexpr int GS1 = NP > 4 ? 4 : NP, NG1 = (2 * NJ + GS1 - 1) / GS1, PPG1 = (PW1 + NG1 - 1) / NG1;
        auto piece1 = [&](int i, int ch) __attribute__((always_inline)) {
            vm_seq += dma_piece_fast<G::NI1, NW, PW1>(i, lds0 + (unsigned)(ch % G::NS1) * G::ST1, wave, off1, [&](int q) {
                return reinterpret_cast<const char*>((q < SP / 8 ? kb : qb) + 64 * ch);
            });
        };
        auto issue1 = [&](int ch) __attribute__((always_inline)) {
            if (!(tail1 && ch == nch - 1)) {
                vm_seq += dma_issue_fast<G::NI1, NW, PW1>(lds0 + (unsigned)(ch % G::NS1) * G::ST1, wave, off1, [&](int q) {
                    return reinterpret_cast<const char*>((q < SP / 8 ? kb : qb) + 64 * ch);
                });
                return;
            }
            const int n = dma_issue<G::NI1, NW>(lds0 + (unsigned)(ch % G::NS1) * G::ST1, wave, lane, [&](int L) {
                const int r = L >> 3, sc = (L & 7) ^ ((r >> 1) & 7);
                const int col = 64 * ch + 8 * sc;
                const __bf16* row = r < SP ? kb + (long)min(r, S - 1) * D : qb + (long)min(q0 + r - SP, S - 1) * D;
                return (col < D && (r >= SP || r < S)) ? reinterpret_cast<const char*>(row + col) : zero;
            });
            vm_seq += n;
        };
        int mark[G::NS1 - 1];            // ledger position right after the request of chunk c, c + 1, ...
#pragma unroll
        for (int s = 0; s < G::NS1 - 1; ++s) {
            if (s < nch) issue1(s);
            mark[s] = vm_seq;
        }
#pragma unroll
        for (int t = 0; t < NJ; ++t) acc[t] = (f32x4v){0.f, 0.f, 0.f, 0.f};
        const unsigned fr = (unsigned)(c16 * 128);
        const unsigned fc0 = (unsigned)((g ^ (c16 >> 1)) << 4), fc1 = (unsigned)(((4 + g) ^ (c16 >> 1)) << 4);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                // the table writes, before the first barrier
#pragma unroll 1
        for (int c = 0; c < nch; ++c) {
            F2_T(t_a);
            vm_wait_le(vm_seq - mark[0]);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            F2_T(t_b);
#pragma unroll
            for (int s = 0; s + 1 < G::NS1 - 1; ++s) mark[s] = mark[s + 1];
            const int cn = c + G::NS1 - 1;                               // the chunk requested during this step
            const bool spread = cn < nch && !(tail1 && cn == nch - 1);   // ... piece by piece between the product groups
            if (cn < nch && !spread) issue1(cn);
            F2_T(t_c);
            F2_ADD(tw[0], t_a, t_b);
            F2_ADD(ti[0], t_b, t_c);
            const char* img = smem2 + (c % G::NS1) * G::ST1;
            const char* qimg = img + (SP + 16 * wave) * 128 + fr;
            const bf16x8 bq[2] = {*reinterpret_cast<const bf16x8*>(qimg + fc0), *reinterpret_cast<const bf16x8*>(qimg + fc1)};
            grouped_products<NP, 2, GS1>(acc, bq, [&](int t, int ks) {
                return *reinterpret_cast<const bf16x8*>(img + 2048 * t + fr + (ks ? fc1 : fc0));
            }, [&](int gi) {
                if (spread) {
#pragma unroll
                    for (int k = 0; k < PPG1; ++k) piece1(gi * PPG1 + k, cn);
                }
            });
            mark[G::NS1 - 2] = vm_seq;
            asm volatile("" ::: "memory");
            F2_T(t_d);
            F2_ADD(tc[0], t_c, t_d);
        }
    }
#ifdef ATT16_STAMP
    const unsigned long long ts1 = __builtin_amdgcn_s_memtime();
#endif

    // ================= phase 2: M^T = W2 gelu(W1 R^T + b1) + b2, 32 hidden units per stage =================
    f32x4v mk[NJ];                       // the mask of this wave's queries, x log2(e), for the head loop
    const float inv1 = 1.0f / p.s1[0], inv2 = 1.0f / p.s2[0];
    {
        constexpr int LD1 = ld_pt(SP), LD2 = ld_pt(32);
        static_assert(LD1 * 2 == G::CPR1 * 16 && LD2 * 2 == 5 * 16, "phase-2 image strides");
        const int nch = (NH + 31) / 32;
        constexpr int PW2 = (G::NI2A + G::NI2B + NW - 1) / NW;
        unsigned off2[PW2];
#pragma unroll
        for (int i = 0; i < PW2; ++i) {
            const int q = wave + NW * i, L = 64 * q + lane;
            if (q < G::NI2A) {
                const int r = L / G::CPR1, c = L - r * G::CPR1;
                off2[i] = (unsigned)((min(r, 31) * S + 8 * min(c, S / 8 - 1)) * 2);
            } else {
                const int L2 = L - 64 * G::NI2A, r = L2 / 5, c = L2 - 5 * r;
                off2[i] = (unsigned)((min(r, S - 1) * NH + 8 * min(c, 3)) * 2);
            }
        }
        const bool tail2 = (NH & 31) != 0;
        auto piece2 = [&](int i, int ch) __attribute__((always_inline)) {
            vm_seq += dma_piece_fast<G::NI2A + G::NI2B, NW, PW2>(i, lds0 + (unsigned)(ch % G::NS2) * G::ST2, wave, off2, [&](int q) {
                return reinterpret_cast<const char*>(q < G::NI2A ? p.w1 + (long)(32 * ch) * S : p.w2 + 32 * ch);
            });
        };
        auto issue2 = [&](int ch) __attribute__((always_inline)) {
            const int n0 = 32 * ch;
            if (!(tail2 && ch == nch - 1)) {
                vm_seq += dma_issue_fast<G::NI2A + G::NI2B, NW, PW2>(lds0 + (unsigned)(ch % G::NS2) * G::ST2, wave, off2, [&](int q) {
                    return reinterpret_cast<const char*>(q < G::NI2A ? p.w1 + (long)n0 * S : p.w2 + n0);
                });
                return;
            }
            const int n = dma_issue<G::NI2A + G::NI2B, NW>(lds0 + (unsigned)(ch % G::NS2) * G::ST2, wave, lane, [&](int L) {
                if (L < 64 * G::NI2A) {                                  // W1 chunk [32 hidden][SP keys + pad chunk]
                    const int r = L / G::CPR1, c = L - r * G::CPR1;
                    const bool ok = r < 32 && n0 + r < NH && 8 * c < S;
                    return ok ? reinterpret_cast<const char*>(p.w1 + (long)(n0 + r) * S + 8 * c) : zero;
                }
                const int L2 = L - 64 * G::NI2A;                         // W2 chunk [SP keys][32 hidden + pad chunk]
                const int r = L2 / 5, c = L2 - 5 * r;
                const bool ok = r < S && c < 4 && n0 + 8 * c < NH;
                return ok ? reinterpret_cast<const char*>(p.w2 + (long)r * NH + n0 + 8 * c) : zero;
            });
            vm_seq += n;
        };
        // every wave has left phase 1's last stage; the first mask-MLP stages are requested before R is written out
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        int mark[G::NS2 - 1];
#pragma unroll
        for (int s = 0; s < G::NS2 - 1; ++s) {
            if (s < nch) issue2(s);
            mark[s] = vm_seq;
        }
        {   // phase 1's result: R (saved for the backward) and the packed B fragments of the mask MLP
            const __amdgpu_buffer_rsrc_t rs_r = make_rsrc(p.R + (long)b * S * S, (long)S * S * 2);
#pragma unroll
            for (int t = 0; t < NJ; ++t) {
                const int j = 16 * t + 4 * g;
                // (pad keys: zero — the W1 image holds re-read valid columns there, not zeros)
                const bf16x4 r4 = j < S ? pack4(acc[t]) : (bf16x4){(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
                buf_store4(rs_r, (q_ok && j < S) ? (unsigned)((q_lane * S + j) * 2) : 0xFFFFFFFFu, r4);
                if (t & 1) {
                    const bf16x4 r3 = j - 16 < S ? pack4(acc[t - 1]) : (bf16x4){(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
                    Rf[t >> 1] = cat8(r3, r4);
                }
            }
            vm_seq += NJ;
        }
#pragma unroll
        for (int t = 0; t < NJ; ++t) acc[t] = (f32x4v){0.f, 0.f, 0.f, 0.f};
        const __amdgpu_buffer_rsrc_t rs_hp = make_rsrc(p.hp + (long)b * S * NH, (long)S * NH * 2);
        const __amdgpu_buffer_rsrc_t rs_hg = make_rsrc(p.hg + (long)b * S * NH, (long)S * NH * 2);
#pragma unroll 1
        for (int c = 0; c < nch; ++c) {
            const int n0 = 32 * c;
            F2_T(t_a);
            vm_wait_le(vm_seq - mark[0]);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            F2_T(t_b);
#pragma unroll
            for (int s = 0; s + 1 < G::NS2 - 1; ++s) mark[s] = mark[s + 1];
            const int cn = c + G::NS2 - 1;
            const bool spread = cn < nch && !(tail2 && cn == nch - 1);
            if (cn < nch && !spread) issue2(cn);
            constexpr int PQ2 = (PW2 + 3) / 4;                           // requests per slot: four slots in the step
            auto slot2 = [&](int k) __attribute__((always_inline)) {
                if (spread) {
#pragma unroll
                    for (int j = 0; j < PQ2; ++j) piece2(k * PQ2 + j, cn);
                }
            };
            F2_T(t_c);
            F2_ADD(tw[1], t_a, t_b);
            F2_ADD(ti[1], t_b, t_c);
            const __bf16* img1 = reinterpret_cast<const __bf16*>(smem2 + (c % G::NS2) * G::ST2);
            const __bf16* img2 = img1 + G::NI2A * 512;
            f32x4v h0 = {0.f, 0.f, 0.f, 0.f}, h1 = {0.f, 0.f, 0.f, 0.f};
            {   // every W1 fragment of the step is requested before the first product (see grouped_products)
                bf16x8 wa[NP], wb[NP];
#pragma unroll
                for (int pr = 0; pr < NP; ++pr) {
                    const __bf16* r0 = img1 + c16 * LD1 + 32 * pr + 4 * g;
                    const __bf16* r1 = r0 + 16 * LD1;
                    wa[pr] = cat8(ld4(r0), ld4(r0 + 16));
                    wb[pr] = cat8(ld4(r1), ld4(r1 + 16));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int pr = 0; pr < NP; ++pr) {
                    h0 = MFMA_BF16(wa[pr], Rf[pr], h0);
                    h1 = MFMA_BF16(wb[pr], Rf[pr], h1);
                }
                slot2(0);
            }
            // ... and the W2 fragments while the hidden units go through bias + GELU
            bf16x8 w2f[NJ];
#pragma unroll
            for (int t = 0; t < NJ; ++t) {
                const __bf16* r2 = img2 + (16 * t + c16) * LD2 + 4 * g;
                w2f[t] = cat8(ld4(r2), ld4(r2 + 16));
            }
            slot2(1);
            __builtin_amdgcn_sched_barrier(0);
            const int na = n0 + 4 * g, nb = na + 16;
            const f32x4v ba = *reinterpret_cast<const f32x4v*>(b1s + na);        // zero past NH (table of 2 SP entries)
            const f32x4v bb = *reinterpret_cast<const f32x4v*>(b1s + nb);
            f32x4v pa, pb, ga, gb;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pa[r] = h0[r] * inv1 + ba[r]; ga[r] = gelu_erf_f(pa[r]);
                pb[r] = h1[r] * inv1 + bb[r]; gb[r] = gelu_erf_f(pb[r]);
            }
            const bf16x4 ga4 = pack4(ga), gb4 = pack4(gb);
            __builtin_amdgcn_sched_barrier(0);
            slot2(2);
            const unsigned oa = (q_ok && na < NH) ? (unsigned)((q_lane * NH + na) * 2) : 0xFFFFFFFFu;
            const unsigned ob = (q_ok && nb < NH) ? (unsigned)((q_lane * NH + nb) * 2) : 0xFFFFFFFFu;
            buf_store4(rs_hp, oa, pack4(pa));
            buf_store4(rs_hg, oa, ga4);
            buf_store4(rs_hp, ob, pack4(pb));
            buf_store4(rs_hg, ob, gb4);
            vm_seq += 4;
            const bf16x8 hf = cat8(ga4, gb4);        // hidden columns >= NH meet zero columns of the W2 image
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < NJ; ++t) acc[t] = MFMA_BF16(w2f[t], hf, acc[t]);
            slot2(3);
            mark[G::NS2 - 2] = vm_seq;
            asm volatile("" ::: "memory");
            F2_T(t_d);
            F2_ADD(tc[1], t_c, t_d);
        }
        // (the mask is finished below, after the first head's K_h / V_h have been requested)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }

#ifdef ATT16_STAMP
    const unsigned long long ts2 = __builtin_amdgcn_s_memtime();
#endif
    if constexpr (MASK_ONLY) {
        // the mask as every later reader sees it: rounded to bf16 (pad keys are not stored)
        const __amdgpu_buffer_rsrc_t rs_m = make_rsrc(p.Mk + (long)b * S * S, (long)S * S * 2);
#pragma unroll
        for (int t = 0; t < NJ; ++t) {
            const int j = 16 * t + 4 * g;
            f32x4v m = {0.f, 0.f, 0.f, 0.f};
            if (j < S) {
                const f32x4v b2v = *reinterpret_cast<const f32x4v*>(b2s + j);
#pragma unroll
                for (int r = 0; r < 4; ++r) m[r] = acc[t][r] * inv2 + b2v[r];
            }
            buf_store4(rs_m, (q_ok && j < S) ? (unsigned)((q_lane * S + j) * 2) : 0xFFFFFFFFu, pack4(m));
        }
        return;
    }
    // ================= phase 3: per head  softmax(scale K_h Q_h^T + M^T),  O^T = V_h^T P^T =================
    constexpr int LDH = ld_rt(HDP), nks = HDP / 32, ndt = HDP / 16;
    static_assert(LDH * 2 == G::CPRH * 16, "phase-3 image stride");
    const int q4 = c16 >> 2, p4 = c16 & 3;
    const bool straddle = (hd & 4) != 0;            // hd = 44, 20: the last 16-byte chunk of a head slice is half valid
    constexpr int PW3 = (2 * G::NI3 + NW - 1) / NW;
    unsigned off3[PW3];
#pragma unroll
    for (int i = 0; i < PW3; ++i) {
        const int q = wave + NW * i, L = 64 * q + lane;
        const int L1 = q >= G::NI3 ? L - 64 * G::NI3 : L;
        const int r = L1 / G::CPRH, c = L1 - r * G::CPRH;
        off3[i] = (unsigned)((min(r, S - 1) * D + 8 * min(c, (hd - 1) >> 3)) * 2);
    }
    constexpr int GS3 = NP > 4 ? 4 : NP, NG3 = (nks * NJ + GS3 - 1) / GS3, PPG3 = (PW3 + NG3 - 1) / NG3;
    auto piece3 = [&](int i, int h) __attribute__((always_inline)) {
        vm_seq += dma_piece_fast<2 * G::NI3, NW, PW3>(i, lds0 + (unsigned)(h & 1) * G::ST3, wave, off3, [&](int q) {
            return reinterpret_cast<const char*>((q < G::NI3 ? kb : vb) + h * hd);
        });
    };
    auto issue3 = [&](int h) __attribute__((always_inline)) {
        const bool last = straddle && h == p.H - 1 && b == p.B - 1;     // its last row would read 8 bytes past the tensor
        if (!last) {
            vm_seq += dma_issue_fast<2 * G::NI3, NW, PW3>(lds0 + (unsigned)(h & 1) * G::ST3, wave, off3, [&](int q) {
                return reinterpret_cast<const char*>((q < G::NI3 ? kb : vb) + h * hd);
            });
            return;
        }
        const int n = dma_issue<2 * G::NI3, NW>(lds0 + (unsigned)(h & 1) * G::ST3, wave, lane, [&](int L) {
            const bool isv = L >= 64 * G::NI3;
            const int L1 = isv ? L - 64 * G::NI3 : L;
            const int r = L1 / G::CPRH, c = L1 - r * G::CPRH;
            bool ok = r < S && 8 * c < hd;
            if (last && r == S - 1 && 8 * c + 8 > hd) ok = false;
            return ok ? reinterpret_cast<const char*>((isv ? vb : kb) + (long)r * D + h * hd + 8 * c) : zero;
        });
        vm_seq += n;
    };
    issue3(0);
    int mark3 = vm_seq;
    const __bf16* qrow = qb + (long)q_ld * D;
    bf16x8 bq_pre[nks];
#pragma unroll
    for (int ks = 0; ks < nks; ++ks) bq_pre[ks] = row_frag(qrow, 32 * ks, g, hd);
    {   // the mask as the backward will read it: rounded to bf16, -inf on the pad keys; kept x log2(e) in fp32
        const __amdgpu_buffer_rsrc_t rs_m = make_rsrc(p.Mk + (long)b * S * S, (long)S * S * 2);
#pragma unroll
        for (int t = 0; t < NJ; ++t) {
            const int j = 16 * t + 4 * g;
            f32x4v m;
            if (j < S) {
                const f32x4v b2v = *reinterpret_cast<const f32x4v*>(b2s + j);
#pragma unroll
                for (int r = 0; r < 4; ++r) m[r] = acc[t][r] * inv2 + b2v[r];
            } else {
                m = (f32x4v){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            }
            const bf16x4 m4 = pack4(m);
            buf_store4(rs_m, (q_ok && j < S) ? (unsigned)((q_lane * S + j) * 2) : 0xFFFFFFFFu, m4);
            mk[t] = unpack4(m4) * 1.4426950408889634f;
        }
        vm_seq += NJ;
    }
    const float sc2 = p.scale * 1.4426950408889634f;
    const __amdgpu_buffer_rsrc_t rs_o = make_rsrc(p.out + (long)b * S * D, (long)S * D * 2);
    const __amdgpu_buffer_rsrc_t rs_l = make_rsrc(p.lse + (long)b * p.H * S, (long)p.H * S * 4);
#pragma unroll 1
    for (int h = 0; h < p.H; ++h) {
        F2_T(t_a);
        vm_wait_le(vm_seq - mark3);
        if (straddle && h == p.H - 1 && b == p.B - 1) {
            // the half chunk left out above (last key of the last head of the last image): its 8 valid bytes by hand,
            // by the lane whose DMA wrote the zero chunk (this wave has just waited for it)
#pragma unroll
            for (int i = 0; i < (2 * G::NI3 + NW - 1) / NW; ++i) {
                const int qi = wave + NW * i;
                if (qi < 2 * G::NI3) {
                    const int L = 64 * qi + lane;
                    const bool isv = L >= 64 * G::NI3;
                    const int L1 = isv ? L - 64 * G::NI3 : L;
                    const int r = L1 / G::CPRH, c = L1 - r * G::CPRH;
                    if (r == S - 1 && 8 * c < hd && 8 * c + 8 > hd) {
                        const bf16x4 x = ld4((isv ? vb : kb) + (long)r * D + h * hd + 8 * c);
                        *reinterpret_cast<bf16x4*>(smem2 + (h & 1) * G::ST3 + L * 16) = x;
                    }
                }
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        F2_T(t_b);
        const __bf16* imgK = reinterpret_cast<const __bf16*>(smem2 + (h & 1) * G::ST3);
        const __bf16* imgV = imgK + G::NI3 * 512;
        bf16x8 bq[nks];
#pragma unroll
        for (int ks = 0; ks < nks; ++ks) bq[ks] = bq_pre[ks];
        const bool more = h + 1 < p.H;
        const bool spread = more && !(straddle && h + 1 == p.H - 1 && b == p.B - 1);
        if (more) {
#pragma unroll
            for (int ks = 0; ks < nks; ++ks) bq_pre[ks] = row_frag(qrow + (h + 1) * hd, 32 * ks, g, hd);
            if (!spread) issue3(h + 1);
        }
        F2_T(t_c);
        F2_ADD(tw[2], t_a, t_b);
        F2_ADD(ti[2], t_b, t_c);
#pragma unroll
        for (int t = 0; t < NJ; ++t) acc[t] = (f32x4v){0.f, 0.f, 0.f, 0.f};
        grouped_products<NP, nks, GS3>(acc, bq, [&](int t, int ks) {
            return *reinterpret_cast<const bf16x8*>(imgK + (16 * t + c16) * LDH + 32 * ks + 8 * g);
        }, [&](int gi) {
            if (spread) {
#pragma unroll
                for (int k = 0; k < PPG3; ++k) piece3(gi * PPG3 + k, h + 1);
            }
        });
        mark3 = more ? vm_seq : mark3;
        // softmax over the keys in the log2 domain: x = s scale log2e + M log2e; 4 NJ in-lane values, then the 4 lane groups
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < NJ; ++t) {
            acc[t] = acc[t] * sc2 + mk[t];
            mx = fmaxf(mx, fmaxf(fmaxf(acc[t][0], acc[t][1]), fmaxf(acc[t][2], acc[t][3])));
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        f32x4v sum4 = {0.f, 0.f, 0.f, 0.f};
        const f32x4v nmx = {-mx, -mx, -mx, -mx};     // (a vector operand: v_pk_add_f32 instead of four v_sub_f32)
#pragma unroll
        for (int t = 0; t < NJ; ++t) {
            const f32x4v d = acc[t] + nmx;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[t][r] = __builtin_amdgcn_exp2f(d[r]);
            sum4 += acc[t];
        }
        float sum = (sum4[0] + sum4[1]) + (sum4[2] + sum4[3]);
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (mx + __builtin_amdgcn_logf(sum)) * 0.6931471805599453f),
                                              rs_l, (q_ok && g == 0) ? (unsigned)((h * S + q_lane) * 4) : 0xFFFFFFFFu, 0, 0);
        bf16x8 Pf[NP];
#pragma unroll
        for (int pr = 0; pr < NP; ++pr) Pf[pr] = cat8(pack4(acc[2 * pr] * inv), pack4(acc[2 * pr + 1] * inv));
        {   // one output tile's V^T fragments (transposed reads) ahead of the products, as in grouped_products
            const __bf16* vbase = imgV + (4 * g + q4) * LDH + 4 * p4;
            bf16x8 vc[NP], vn[NP];
#pragma unroll
            for (int pr = 0; pr < NP; ++pr) vc[pr] = cat8(tr4(vbase + (32 * pr) * LDH), tr4(vbase + (32 * pr + 16) * LDH));
#pragma unroll
            for (int dt = 0; dt < ndt; ++dt) {
                if (dt + 1 < ndt) {
#pragma unroll
                    for (int pr = 0; pr < NP; ++pr)
                        vn[pr] = cat8(tr4(vbase + (32 * pr) * LDH + 16 * (dt + 1)), tr4(vbase + (32 * pr + 16) * LDH + 16 * (dt + 1)));
                }
                __builtin_amdgcn_sched_barrier(0);
                f32x4v o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int pr = 0; pr < NP; ++pr) o = MFMA_BF16(vc[pr], Pf[pr], o);
                __builtin_amdgcn_sched_barrier(0);
                const int d = 16 * dt + 4 * g;
                buf_store4(rs_o, (q_ok && d < hd) ? (unsigned)((q_lane * D + h * hd + d) * 2) : 0xFFFFFFFFu, pack4(o));
#pragma unroll
                for (int pr = 0; pr < NP; ++pr) vc[pr] = vn[pr];
            }
        }
        vm_seq += ndt + 1;
        asm volatile("" ::: "memory");
        F2_T(t_d);
        F2_ADD(tc[2], t_c, t_d);
    }
#ifdef ATT16_STAMP
    __builtin_amdgcn_s_barrier();
    if (tid == 0 && qg == 0) {
        const unsigned long long ts3 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        float* d = p.lse + (long)b * p.H * S;
        d[0] = (float)(ts1 - ts0); d[1] = (float)(ts2 - ts1); d[2] = (float)(ts3 - ts2);
        for (int i = 0; i < 3; ++i) { d[3 + i] = (float)tw[i]; d[6 + i] = (float)ti[i]; d[9 + i] = (float)tc[i]; }
    }
#endif
}
