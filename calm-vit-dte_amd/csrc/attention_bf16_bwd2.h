// Pipelined backward of the fused bf16 latent-mask attention (included by attention_bf16.hip after
// attention_bf16_fwd2.h; same arithmetic and fragment orders as attn16_bwd_q_kernel / attn16_bwd_kv_kernel there).
//
// What the first pair of kernels lost their time on (ISA + PMC, round 3): the other axis' head slices went global ->
// registers -> LDS one head ahead, which at S = 224 / 176 cost 32 staging registers beside 56 dM accumulators — 156-184
// bytes of scratch per lane, i.e. vector-memory reloads inside the loop; and the mask (query side: row, key side: column of
// the transposed copy), the row log-sum-exp and delta were fetched from global memory pair by pair in every head with
// `s_waitcnt vmcnt(0)` right behind the load: 84 exposed memory round trips per workgroup.  Here
//   * the two [S][hd] images of a head (query side: K_h, V_h; key side: Q_h, dO_h) are brought in by LDS-DMA, the next
//     head's while this head computes, spread between the pair steps (lds_dma.h, as in attn16_fwd2_kernel);
//   * the mask row / column of the wave's 16 queries / keys stays in registers for all heads (packed bf16, 4 NP
//     registers, read once);
//   * key side: the head's lse and delta vectors ride in the same stage (one DMA instruction each) and are read from LDS;
//   * the own-row fragments (q, dO [, O] resp. k, v) of the NEXT head are requested one head ahead;
//   * the products of pair pr + 1 are issued before the exp / dS arithmetic of pair pr, their fragment reads grouped
//     ahead of them.
#pragma once

template <int NP, int HDP, bool KEYSIDE>
struct Bwd2Geo {
    typedef Fwd2Geo<NP, HDP> F;
    static constexpr int NI3 = F::NI3, CPRH = F::CPRH, NW = F::NW, SP = F::SP;
    static constexpr int NIX = KEYSIDE ? 2 : 0;                        // lse_h / delta_h vectors (SP floats each, <= 1 KiB)
    static constexpr int NINSTR = 2 * NI3 + NIX, ST = NINSTR * 1024;
    static constexpr int LDS = 2 * ST;
    static constexpr bool OK = NP <= 7 && HDP <= 64 && LDS <= 160 * 1024 && SP * 4 <= 1024;
};

template <int NP, int HDP, bool KEYSIDE>
__global__ __launch_bounds__(64 * waves_for(NP), 2) void attn16_bwd2_kernel(const Attn16BP p) {
    typedef Bwd2Geo<NP, HDP, KEYSIDE> G;
    constexpr int NJ = 2 * NP, NW = G::NW, LDH = ld_rt(HDP), nks = HDP / 32, ndt = HDP / 16;
    constexpr int NOUT = KEYSIDE ? 2 * ndt : ndt;
    extern __shared__ __attribute__((aligned(1024))) char smemb[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c16 = lane & 15, g = lane >> 4, q4 = c16 >> 2, p4 = c16 & 3;
    int b, grp;
    image_and_group(blockIdx.x, p.groups, p.B, b, grp);
    const int S = p.S, D = p.H * p.hd, hd = p.hd;
    const int r_lane = grp * (16 * NW) + 16 * wave + c16;              // this lane's own row (query resp. key)
    const bool r_ok = r_lane < S;
    const int r_ld = r_ok ? r_lane : S - 1;
    const long roff = ((long)b * S + r_ld) * D;
    // images: the OTHER axis' head slices
    const __bf16* ab = (KEYSIDE ? p.q : p.k) + (long)b * S * D;
    const __bf16* bb = (KEYSIDE ? p.dout : p.v) + (long)b * S * D;
    // own-row fragments
    const __bf16* f1 = (KEYSIDE ? p.k : p.q) + roff;
    const __bf16* f2 = (KEYSIDE ? p.v : p.dout) + roff;
    const char* zero = reinterpret_cast<const char*>(calm_zero_block);
    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)reinterpret_cast<uintptr_t>(smemb));
    int vm_seq = 0;

    // ---- staging of a head: instruction q < NI3: image A rows, < 2 NI3: image B, then (key side) lse_h, delta_h
    constexpr int PW = (G::NINSTR + NW - 1) / NW;
    unsigned off[PW];
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        const int q = wave + NW * i, L = 64 * q + lane;
        if (q < 2 * G::NI3) {
            const int L1 = q >= G::NI3 ? L - 64 * G::NI3 : L;
            const int r = L1 / G::CPRH, c = L1 - r * G::CPRH;
            off[i] = (unsigned)((min(r, S - 1) * D + 8 * min(c, (hd - 1) >> 3)) * 2);
        } else {
            off[i] = (unsigned)(min(lane, S / 4 - 1) * 16);
        }
    }
    const bool straddle = (hd & 4) != 0;
    auto base_of = [&](int q, int h) __attribute__((always_inline)) {
        if (q < G::NI3) return reinterpret_cast<const char*>(ab + h * hd);
        if (q < 2 * G::NI3) return reinterpret_cast<const char*>(bb + h * hd);
        const float* v = (q == 2 * G::NI3 ? p.lse : p.delta) + ((long)b * p.H + h) * S;
        return reinterpret_cast<const char*>(v);
    };
    auto piece = [&](int i, int h) __attribute__((always_inline)) {
        vm_seq += dma_piece_fast<G::NINSTR, NW, PW>(i, lds0 + (unsigned)(h & 1) * G::ST, wave, off, [&](int q) { return base_of(q, h); });
    };
    auto edge_head = [&](int h) { return straddle && h == p.H - 1 && b == p.B - 1; };   // its last row would read past the tensor
    auto issue_all = [&](int h) __attribute__((always_inline)) {
        if (!edge_head(h)) {
            vm_seq += dma_issue_fast<G::NINSTR, NW, PW>(lds0 + (unsigned)(h & 1) * G::ST, wave, off, [&](int q) { return base_of(q, h); });
            return;
        }
        const int n = dma_issue<G::NINSTR, NW>(lds0 + (unsigned)(h & 1) * G::ST, wave, lane, [&](int L) {
            const int q = L >> 6;
            if (q >= 2 * G::NI3) return base_of(q, h) + min(L & 63, S / 4 - 1) * 16;
            const bool isb = q >= G::NI3;
            const int L1 = isb ? L - 64 * G::NI3 : L;
            const int r = L1 / G::CPRH, c = L1 - r * G::CPRH;
            const int rr = min(r, S - 1), cc = min(c, (hd - 1) >> 3);
            if (rr == S - 1 && 8 * cc + 8 > hd) return zero;
            return reinterpret_cast<const char*>((isb ? bb : ab) + (long)rr * D + h * hd + 8 * cc);
        });
        vm_seq += n;
    };
    // the half chunk the edge head left out (last row of the last head of the last image), by the lanes that own it
    auto edge_fixup = [&](int h) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const int q = wave + NW * i;
            if (q < 2 * G::NI3) {
                const int L = 64 * q + lane;
                const bool isb = q >= G::NI3;
                const int L1 = isb ? L - 64 * G::NI3 : L;
                const int r = L1 / G::CPRH, c = L1 - r * G::CPRH;
                if (min(r, S - 1) == S - 1 && min(c, (hd - 1) >> 3) == (hd >> 3)) {
                    const bf16x4 x = ld4((isb ? bb : ab) + (long)(S - 1) * D + h * hd + 8 * (hd >> 3));
                    *reinterpret_cast<bf16x4*>(smemb + (h & 1) * G::ST + L * 16) = x;
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    };

    // ---- the mask row (query side) / column (key side: the transposed copy) of this lane's row, all heads
    bf16x8 mrow[NP];
    {
        const __bf16* M = (KEYSIDE ? p.MkT : p.Mk) + ((long)b * S + r_ld) * S;
        const bf16x4 ninf = {(__bf16)-INFINITY, (__bf16)-INFINITY, (__bf16)-INFINITY, (__bf16)-INFINITY};
#pragma unroll
        for (int pr = 0; pr < NP; ++pr) {
            const int j0 = 32 * pr + 4 * g, j1 = j0 + 16;              // pad keys / pad queries: -inf -> P = 0
            mrow[pr] = cat8(j0 < S ? ld4(M + j0) : ninf, j1 < S ? ld4(M + j1) : ninf);
        }
    }
    f32x4v accM[KEYSIDE ? 1 : NJ];
    if constexpr (!KEYSIDE) {
#pragma unroll
        for (int t = 0; t < NJ; ++t) accM[t] = (f32x4v){0.f, 0.f, 0.f, 0.f};
    }

    // own-row fragments of head 0, and (query side) its lse and the O row for delta
    bf16x8 fa_n[nks], fb_n[nks], fo_n[KEYSIDE ? 1 : nks];
    float lse_n = 0.f;
    auto prefetch_own = [&](int h) __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < nks; ++ks) {
            fa_n[ks] = row_frag(f1 + h * hd, 32 * ks, g, hd);
            fb_n[ks] = row_frag(f2 + h * hd, 32 * ks, g, hd);
            if constexpr (!KEYSIDE) fo_n[ks] = row_frag(p.out + roff + h * hd, 32 * ks, g, hd);
        }
        if constexpr (!KEYSIDE) lse_n = p.lse[((long)b * p.H + h) * S + r_ld];
    };
    issue_all(0);
    int mark = vm_seq;
    if constexpr (KEYSIDE) prefetch_own(0);

    const float LOG2E = 1.4426950408889634f;
    const float sc2 = p.scale * LOG2E;
    const __amdgpu_buffer_rsrc_t rs_1 = make_rsrc((KEYSIDE ? p.dv : p.dq) + (long)b * S * D, (long)S * D * 2);
    const __amdgpu_buffer_rsrc_t rs_2 = KEYSIDE ? make_rsrc(p.dk + (long)b * S * D, (long)S * D * 2)
                                                : make_rsrc(p.delta + (long)b * p.H * S, (long)p.H * S * 4);
    constexpr int PPS = (PW + NP - 1) / NP;                            // staging requests per pair step

#pragma unroll 1
    for (int h = 0; h < p.H; ++h) {
        // this head's own-row operands: key side requested one head ago; the query side (56 dM accumulators: no registers
        // for a second set) requests them here, ahead of the stage wait and the barrier
        if constexpr (!KEYSIDE) prefetch_own(h);
        bf16x8 fa[nks], fb[nks];
#pragma unroll
        for (int ks = 0; ks < nks; ++ks) { fa[ks] = fa_n[ks]; fb[ks] = fb_n[ks]; }
        float delta = 0.f, lse2 = 0.f;
        if constexpr (!KEYSIDE) {
            float part = 0.f;
#pragma unroll
            for (int ks = 0; ks < nks; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e) part = fmaf((float)fb[ks][e], (float)fo_n[ks][e], part);
            part += __shfl_xor(part, 16, 64);
            part += __shfl_xor(part, 32, 64);
            delta = part;
            lse2 = lse_n * LOG2E;
            // pinned here: left to the scheduler the multiply (and with it the wait for the lse load) sinks to its first
            // use inside pair 0 — behind the first LDS-DMA requests of the next head, which the compiler's `vmcnt(0)` then
            // drains as well (ISA, round 4: one exposed DMA round trip per head in the query-side kernel)
            asm volatile("" : "+v"(delta), "+v"(lse2));
        }
        vm_wait_le(vm_seq - mark);
        if (edge_head(h)) edge_fixup(h);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const bool more = h + 1 < p.H;
        const bool spread = more && !edge_head(h + 1);
        if (more) {
            if constexpr (KEYSIDE) prefetch_own(h + 1);
            if (!spread) issue_all(h + 1);
        }
        // one per-lane address per access pattern; everything else is an immediate offset (pair, k-step, image)
        const char* stage = smemb + (h & 1) * G::ST;
        const char* fbase = stage + (c16 * LDH + 8 * g) * 2;             // 16-byte fragments: row c16, chunk g
        const char* tbase = stage + ((4 * g + q4) * LDH + 4 * p4) * 2;   // transposed reads
        const char* lbase = stage + 2 * G::NI3 * 1024 + 16 * g;          // lse_h / delta_h: 4 floats at 4 g
        constexpr int IMGB = G::NI3 * 1024, PAIR = 32 * LDH * 2, HALF = 16 * LDH * 2;

        f32x4v oacc[NOUT];
#pragma unroll
        for (int i = 0; i < NOUT; ++i) oacc[i] = (f32x4v){0.f, 0.f, 0.f, 0.f};

        // S / dP products of a pair: fragment reads first, then the 4 nks products
        f32x4v sd[4];
        auto issue_sd = [&](int pr, f32x4v (&o)[4]) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = (f32x4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < nks; ++ks) {
                const char* f = fbase + pr * PAIR + 64 * ks;
                const bf16x8 r0 = *reinterpret_cast<const bf16x8*>(f), r1 = *reinterpret_cast<const bf16x8*>(f + HALF);
                const bf16x8 r2 = *reinterpret_cast<const bf16x8*>(f + IMGB), r3 = *reinterpret_cast<const bf16x8*>(f + IMGB + HALF);
                __builtin_amdgcn_sched_barrier(0);
                o[0] = MFMA_BF16(r0, fa[ks], o[0]);
                o[1] = MFMA_BF16(r1, fa[ks], o[1]);
                o[2] = MFMA_BF16(r2, fb[ks], o[2]);
                o[3] = MFMA_BF16(r3, fb[ks], o[3]);
            }
        };
#pragma unroll
        for (int pr = 0; pr < NP; ++pr) {
            __builtin_amdgcn_sched_barrier(0);
            issue_sd(pr, sd);
            if (spread) {
#pragma unroll
                for (int k = 0; k < PPS; ++k) piece(pr * PPS + k, h + 1);
            }
            // transposed fragments of the output products of this pair, ahead of the arithmetic
            const char* tp = tbase + pr * PAIR;
            bf16x8 ta[ndt], tb[KEYSIDE ? ndt : 1];
#pragma unroll
            for (int dt = 0; dt < ndt; ++dt) {
                ta[dt] = cat8(tr4(reinterpret_cast<const __bf16*>(tp + 32 * dt)), tr4(reinterpret_cast<const __bf16*>(tp + 32 * dt + HALF)));
                if constexpr (KEYSIDE)
                    tb[dt] = cat8(tr4(reinterpret_cast<const __bf16*>(tp + IMGB + 32 * dt)),
                                  tr4(reinterpret_cast<const __bf16*>(tp + IMGB + 32 * dt + HALF)));
            }
            f32x4v la, lb, ea, eb;                                      // other-axis lse x log2e and delta (key side)
            if constexpr (KEYSIDE) {
                const char* lp = lbase + 128 * pr;                       // queries 32 pr + 4 g .. and + 16
                la = *reinterpret_cast<const f32x4v*>(lp) * LOG2E;
                lb = *reinterpret_cast<const f32x4v*>(lp + 64) * LOG2E;
                ea = *reinterpret_cast<const f32x4v*>(lp + 1024);
                eb = *reinterpret_cast<const f32x4v*>(lp + 1024 + 64);
            } else {
                la = lb = (f32x4v){lse2, lse2, lse2, lse2};
                ea = eb = (f32x4v){delta, delta, delta, delta};
            }
            __builtin_amdgcn_sched_barrier(0);
            f32x4v(&o)[4] = sd;
            const bf16x8 mf = mrow[pr];
            f32x4v pa, pb, s0, s1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pa[r] = __builtin_amdgcn_exp2f(fmaf(o[0][r], sc2, fmaf((float)mf[r], LOG2E, -la[r])));
                pb[r] = __builtin_amdgcn_exp2f(fmaf(o[1][r], sc2, fmaf((float)mf[4 + r], LOG2E, -lb[r])));
            }
            s0 = pa * (o[2] - ea);
            s1 = pb * (o[3] - eb);
            const bf16x8 dsf = cat8(pack4(s0), pack4(s1));
            if constexpr (KEYSIDE) {
                const bf16x8 pf = cat8(pack4(pa), pack4(pb));
#pragma unroll
                for (int dt = 0; dt < ndt; ++dt) {
                    oacc[dt] = MFMA_BF16(tb[dt], pf, oacc[dt]);                      // dV^T += dO_h^T P
                    oacc[ndt + dt] = MFMA_BF16(ta[dt], dsf, oacc[ndt + dt]);        // dK^T += Q_h^T dS
                }
            } else {
                accM[2 * pr] = accM[2 * pr] + s0;
                accM[2 * pr + 1] = accM[2 * pr + 1] + s1;
#pragma unroll
                for (int dt = 0; dt < ndt; ++dt) oacc[dt] = MFMA_BF16(ta[dt], dsf, oacc[dt]);   // dQ^T += K_h^T dS^T
            }
        }
        if (more) mark = vm_seq;
        // results of the head
        const unsigned ro = (unsigned)((r_lane * D + h * hd) * 2);
#pragma unroll
        for (int dt = 0; dt < ndt; ++dt) {
            const int d = 16 * dt + 4 * g;
            const unsigned o1 = (r_ok && d < hd) ? ro + 2 * d : 0xFFFFFFFFu;
            if constexpr (KEYSIDE) {
                buf_store4(rs_1, o1, pack4(oacc[dt]));
                buf_store4(rs_2, o1, pack4(oacc[ndt + dt] * p.scale));
            } else {
                buf_store4(rs_1, o1, pack4(oacc[dt] * p.scale));
            }
        }
        if constexpr (!KEYSIDE) {
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, delta), rs_2,
                                                  (r_ok && g == 0) ? (unsigned)((h * S + r_lane) * 4) : 0xFFFFFFFFu, 0, 0);
            vm_seq += ndt + 1;
        } else {
            vm_seq += 2 * ndt;
        }
        asm volatile("" ::: "memory");
    }
    if constexpr (!KEYSIDE) {
        if (r_ok) {
            __bf16* mr = p.dM + ((long)b * S + r_lane) * S;
#pragma unroll
            for (int t = 0; t < NJ; ++t) {
                const int j = 16 * t + 4 * g;
                if (j < S) *reinterpret_cast<bf16x4*>(mr + j) = pack4(accM[t]);
            }
        }
    }
}
