// Round 4 — the per-head core of the fused bf16 latent-mask attention as a kernel of its own (included by
// attention_bf16.hip inside its anonymous namespace; fragment orders, LDS image geometry and arithmetic of the head loop
// of attn16_fwd2_kernel).
//
//   O_h = softmax_j(Q_h K_h^T / sqrt(hd) + M) V_h       one workgroup per (image, head), the mask M read from HBM / L2
//
// Why split the head loop off.  In attn16_fwd2_kernel a workgroup = (image, 7 query tiles) walks ALL heads: the mask
// stays in registers, but every head costs a workgroup-wide barrier (the K_h / V_h stage is shared), the ring of stages
// and its ledger of vector-memory instructions cost ~400 issue slots per head and wave beside the ~500 the arithmetic
// needs, 7 waves leave one SIMD with a single wave, and 143 KB of LDS leave one workgroup per CU: in-kernel stamps gave
// 129k cycles for the head phase against 22k cycles of MFMA work per SIMD — 25 % of them waiting at the per-head barrier —
// and PMC 14.5 % MFMA busy for the kernel.  Here
//   * a workgroup is 4 waves (one per SIMD) and owns ONE head of one image: K_h and V_h are staged once by LDS-DMA (one
//     wait, one barrier) and then every wave walks its query tiles (16 queries each, tiles wave, wave + 4, ...) with NO
//     further synchronisation — no ring, no ledger, no per-head barrier;
//   * 72 KB of LDS (S = 224, hd <= 64) and <= 256 VGPRs put TWO workgroups on a CU: a SIMD always holds two independent
//     waves, one's softmax (VALU) under the other's products (MFMA) by construction, and the staging of one workgroup
//     runs under the arithmetic of the other;
//   * the mask of a query tile is 14 eight-byte loads per lane (bf16, as the forward's mask MLP stored it — the very
//     values the backward will read), requested one tile ahead together with the tile's q fragments; the heads of an
//     image are dealt to one XCD next to each other in time, so these reads and the K / V staging hit in L2;
//   * P is left un-normalised for the P V product (exp2(x - max) in (0, 1], rounded to bf16) and the 1 / sum goes onto
//     the 16 output values of a lane instead of its 56 probabilities.
//   * a wave pass covers TWO query tiles (32 queries): in this orientation the K_h / V_h fragments are the LDS-fed A
//     operands, and with one tile per pass every 1 KiB fragment read (8 LDS cycles) fed ONE 16-cycle product — four SIMDs
//     then want 256 B / cycle from an LDS that delivers 128 (in-kernel stamps of the one-tile version: 4.5-5.8k cycles
//     for the 28 Q K^T products of a tile, 2.5-3.8k for the 28 P V products, all 8 waves of the CU in the same phase at
//     the same time).  Two tiles per fragment halve the LDS bytes per product.
// Instruction budget per (tile, head) and wave: 56 MFMA, 56 v_exp_f32, ~200 other VALU (mask unpack 56, two packed FMAs
// per pair, max3 chain, packed sums, bf16 packing), 14 + 28 LDS reads, 14 + 2 global loads, 5 stores.
#pragma once

template <int NP, int HDP>
struct Fwd3Geo {
    static constexpr int SP = 32 * NP, NW = 4;
    static constexpr int CPRH = HDP / 8 + 2, NI3 = (SP * CPRH + 63) / 64;        // LDS-DMA instructions per image
    static constexpr int LDS = 2 * NI3 * 1024;                                   // K_h and V_h images
    static constexpr int PW = (2 * NI3 + NW - 1) / NW;                           // instructions per wave
    static constexpr int PWI = (NI3 + NW - 1) / NW;                              // ... per image
    static constexpr bool OK = NP <= 7 && HDP <= 64 && LDS <= 80 * 1024;         // two workgroups per CU
};

// EXACT: S == 32 NP (no pad keys) — a compile-time property so that both variants of the mask request have a fixed number of
// vector-memory instructions (with a run-time branch the compiler cannot count the loads in flight behind the q fragments
// and waits for everything: in-kernel stamps showed the mask latency exposed in front of every pair's first product)
template <int NP, int HDP, bool EXACT>
__global__ __launch_bounds__(256, 2) void attn16_fwd3_core_kernel(const Attn16P p) {
    typedef Fwd3Geo<NP, HDP> G;
    constexpr int NJ = 2 * NP, SP = G::SP, NW = G::NW;
    constexpr int LDH = ld_rt(HDP), nks = HDP / 32, ndt = HDP / 16;
    static_assert(LDH * 2 == G::CPRH * 16, "image stride");
    extern __shared__ __attribute__((aligned(1024))) char smem3[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c16 = lane & 15, g = lane >> 4;
    // (image, head): the heads of one image on one XCD, next to each other in time (workgroup ids go round-robin over
    // the 8 XCDs) — their K / V rows share cache lines (a 128-byte line of a token row holds one head's 56 columns plus
    // a piece of the next) and all of them read the image's mask
    int b, h;
    {
        const int id = blockIdx.x, H = p.H;
        if ((p.B & 7) == 0) {
            const int xcd = id & 7, idx = id >> 3;
            h = idx % H;
            b = (idx / H) * 8 + xcd;
        } else {
            h = id % H;
            b = id / H;
        }
    }
    const int S = p.S, D = p.H * p.hd, hd = p.hd;
#ifdef ATT16_STAMP3
    unsigned long long st3[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
    st3[0] = __builtin_amdgcn_s_memtime();
#define F3_STAMP(i) st3[i] = __builtin_amdgcn_s_memtime()
#else
#define F3_STAMP(i)
#endif
    const __bf16* qb = p.q + (long)b * S * D + h * hd;
    const __bf16* kb = p.k + (long)b * S * D + h * hd;
    const __bf16* vb = p.v + (long)b * S * D + h * hd;
    const __bf16* mb = p.Mk + (long)b * S * S;
    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)reinterpret_cast<uintptr_t>(smem3));

    // ---- stage K_h and V_h: [SP keys][HDP + 16] images; pad keys re-read a valid row (their scores meet a -inf mask,
    //      their V rows a zero probability); pad chunks of a row (columns past the head slice) are not fetched: their
    //      lanes sit out the LDS-DMA instruction (exec mask) and zero their 16 bytes of the image instead — 10 chunks per
    //      row staged, 7 fetched.  K_h is requested first; V_h is requested once K_h has landed and keeps landing under
    //      the products and the softmax of the first tile pair (a second barrier in front of the first P V product).
    const bool straddle = (hd & 4) != 0;            // hd = 44, 20: the last 16-byte chunk of a head slice is half valid
    const bool last = straddle && h == p.H - 1 && b == p.B - 1;           // ... and here it would end 8 bytes past the tensor
    const int cvalid = (hd + 7) >> 3;
    auto stage_image = [&](int img) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < G::PWI; ++i) {
            const int q1 = wave + NW * i;                               // instruction within the image
            if (G::NI3 % NW == 0 || q1 < G::NI3) {
                const int L1 = 64 * q1 + lane;
                const int r = L1 / G::CPRH, c = L1 - r * G::CPRH;
                const unsigned off = (unsigned)((min(r, S - 1) * D + 8 * c) * 2);
                const int q = q1 + img * G::NI3;
                if (c < cvalid) glds16(img ? vb : kb, off, lds0 + 1024u * q);
                else *reinterpret_cast<f32x4v*>(smem3 + (64 * q + lane) * 16) = (f32x4v){0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    if (!last) {
        stage_image(0);
    } else {
        // (one workgroup of the launch) both images through per-lane addresses, the chunk that would cross the end of the
        // tensor fed from the zero block and patched by hand
        const char* zero = reinterpret_cast<const char*>(calm_zero_block);
        dma_issue<2 * G::NI3, NW>(lds0, wave, lane, [&](int L) {
            const bool isv = L >= 64 * G::NI3;
            const int L1 = isv ? L - 64 * G::NI3 : L;
            const int r = L1 / G::CPRH, c = L1 - r * G::CPRH;
            const bool ok = r < S && 8 * c < hd && !(r == S - 1 && 8 * c + 8 > hd);
            return ok ? reinterpret_cast<const char*>((isv ? vb : kb) + (long)r * D + 8 * c) : zero;
        });
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < G::PW; ++i) {
            const int q = wave + NW * i;
            if (q < 2 * G::NI3) {
                const int L = 64 * q + lane;
                const bool isv = L >= 64 * G::NI3;
                const int L1 = isv ? L - 64 * G::NI3 : L;
                const int r = L1 / G::CPRH, c = L1 - r * G::CPRH;
                if (r == S - 1 && 8 * c < hd && 8 * c + 8 > hd)
                    *reinterpret_cast<bf16x4*>(smem3 + L * 16) = ld4((isv ? vb : kb) + (long)r * D + 8 * c);
            }
        }
    }
    const __bf16* imgK = reinterpret_cast<const __bf16*>(smem3);
    const __bf16* imgV = imgK + G::NI3 * 512;
    const int q4 = c16 >> 2, p4 = c16 & 3;
    const int ntiles = (S + 15) >> 4, npairs = (ntiles + 1) >> 1;
    const float scale = p.scale;
    constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
    typedef float f32x2v __attribute__((ext_vector_type(2)));
    typedef unsigned mu32x2 __attribute__((ext_vector_type(2)));
    constexpr int QT = 2;                // query tiles per wave pass: every K / V fragment read from LDS feeds QT products

    // q fragments of a tile pair (requested one pair ahead) and its mask rows (requested at the top of the pair's pass,
    // used after its 4 NJ products).  The mask stays packed (two bf16 per register); keys past S (S % 32 != 0 only) get
    // -inf by a select on a clamped load — no exec-mask branches.
    // (loads are unconditional — columns past the head slice are fetched from a clamped address and zeroed by a select —
    // so that the number of vector-memory instructions in flight is a constant the compiler can count)
    struct QIn { bf16x8 bq[QT][nks]; };
    auto request_q = [&](int pp) __attribute__((always_inline)) {
        QIn t;
        const bf16x4 z4 = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
#pragma unroll
        for (int u = 0; u < QT; ++u) {
            const __bf16* qrow = qb + (long)min(16 * (QT * pp + u) + c16, S - 1) * D;
#pragma unroll
            for (int ks = 0; ks < nks; ++ks) {
                const int c = 32 * ks + 8 * g;
                const bf16x4 a = ld4(qrow + min(c, hd - 4)), bb = ld4(qrow + min(c + 4, hd - 4));
                t.bq[u][ks] = cat8(c < hd ? a : z4, c + 4 < hd ? bb : z4);
            }
        }
        return t;
    };
    auto request_m = [&](int tile, mu32x2 (&m)[NJ]) __attribute__((always_inline)) {
        const __bf16* mrow = mb + (long)min(16 * tile + c16, S - 1) * S + 4 * g;
        if constexpr (EXACT) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) m[j] = *reinterpret_cast<const mu32x2*>(mrow + 16 * j);
        } else {
            const mu32x2 ninf2 = {0xFF80FF80u, 0xFF80FF80u};                     // bf16 -inf x 4
            int jl = (S - 4 * g + 15) / 16 - 1;                                 // last key tile with this lane group's keys in range
            asm volatile("" : "+v"(jl));      // (opaque: otherwise the 2 NP clamped offsets are hoisted out of the pair loop and spilled)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const mu32x2 x = *reinterpret_cast<const mu32x2*>(mrow + 16 * min(j, jl));
                m[j] = j <= jl ? x : ninf2;
            }
        }
    };
    QIn qn = request_q(min(wave, npairs - 1));
    mu32x2 m0n[NJ];
    request_m(QT * min(wave, npairs - 1), m0n);
    F3_STAMP(1);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // this wave's share of K_h (and the first pair's q / mask rows)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (!last) stage_image(1);
    bool v_pending = true;               // V_h not yet published: the first P V product of the workgroup waits and synchronises
    F3_STAMP(2);
    // The two waves that share a SIMD belong to the two co-resident workgroups and would run in lockstep — both in their
    // products, then both in their softmax: the matrix pipe idles while the VALU is contended and vice versa.  The wave
    // in the odd slot of its SIMD starts half a pair late (p.kv_shared x 64 cycles; HW_REG_HW_ID bits 3:0 = wave slot).
    if (p.kv_shared > 0 && (__builtin_amdgcn_s_getreg((3 << 11) | 4) & 1)) {
        for (int i = 0; i < p.kv_shared; i += 10) __builtin_amdgcn_s_sleep(10);
    }

    constexpr int GS = NP > 4 ? 4 : NP, NF = NJ * nks, NG = (NF + GS - 1) / GS;
#pragma unroll 1
    for (int pp = wave; pp < npairs; pp += NW) {
        const QIn qc = qn;
        mu32x2 m0[NJ], m1[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) m0[j] = m0n[j];   // requested under the previous pair's P V products
        f32x4v acc[QT][NJ];
#pragma unroll
        for (int u = 0; u < QT; ++u)
#pragma unroll
            for (int t = 0; t < NJ; ++t) acc[u][t] = (f32x4v){0.f, 0.f, 0.f, 0.f};
        {   // S^T = K_h Q_h^T for both tiles: fragment f = ks NJ + t read one group of GS ahead of its products
            auto frag = [&](int f) __attribute__((always_inline)) {
                return *reinterpret_cast<const bf16x8*>(imgK + (16 * (f % NJ) + c16) * LDH + 32 * (f / NJ) + 8 * g);
            };
            bf16x8 cur[GS], nx[GS];
#pragma unroll
            for (int i = 0; i < GS; ++i)
                if (i < NF) cur[i] = frag(i);
#pragma unroll
            for (int gi = 0; gi < NG; ++gi) {
                if (gi + 1 < NG) {
#pragma unroll
                    for (int i = 0; i < GS; ++i)
                        if ((gi + 1) * GS + i < NF) nx[i] = frag((gi + 1) * GS + i);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < GS; ++i) {
                    const int f = gi * GS + i;
                    if (f < NF) {
                        acc[0][f % NJ] = MFMA_BF16(cur[i], qc.bq[0][f / NJ], acc[0][f % NJ]);
                        acc[1][f % NJ] = MFMA_BF16(cur[i], qc.bq[1][f / NJ], acc[1][f % NJ]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < GS; ++i) cur[i] = nx[i];
            }
        }
#ifdef ATT16_STAMP3
        asm volatile("s_nop 0" :: "v"(acc[0][0][0]), "v"(acc[1][NJ - 1][3]));
        if (pp == wave) F3_STAMP(3);
#endif
        __builtin_amdgcn_sched_barrier(0);
        request_m(QT * pp + 1, m1);                    // ... the second tile's rows land under the first tile's softmax
        // softmax of both tiles: z = s scale + M (natural units), row max over the 4 NJ in-lane values and the 4 lane groups
        bf16x8 Pf[QT][NP];
        float inv[QT];
        const f32x2v sc2v = {scale, scale};
#pragma unroll
        for (int u = 0; u < QT; ++u) {
            const mu32x2 (&mm)[NJ] = u ? m1 : m0;
            float mx = -INFINITY;
#pragma unroll
            for (int t = 0; t < NJ; ++t) {
                // bf16 pair in one register -> two fp32 values (shift / mask), then one packed FMA per pair
                const unsigned u0 = mm[t][0], u1 = mm[t][1];
                const f32x2v m01 = {__builtin_bit_cast(float, u0 << 16), __builtin_bit_cast(float, u0 & 0xFFFF0000u)};
                const f32x2v m23 = {__builtin_bit_cast(float, u1 << 16), __builtin_bit_cast(float, u1 & 0xFFFF0000u)};
                const f32x2v z01 = (f32x2v){acc[u][t][0], acc[u][t][1]} * sc2v + m01;
                const f32x2v z23 = (f32x2v){acc[u][t][2], acc[u][t][3]} * sc2v + m23;
                acc[u][t] = (f32x4v){z01[0], z01[1], z23[0], z23[1]};
                mx = fmaxf(mx, fmaxf(fmaxf(z01[0], z01[1]), fmaxf(z23[0], z23[1])));
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float nb = -mx * LOG2E;
            f32x4v sum4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < NJ; ++t) {
                const f32x4v d = acc[u][t] * LOG2E + nb;          // (z - max) log2(e)
                f32x4v e;
#pragma unroll
                for (int r = 0; r < 4; ++r) e[r] = __builtin_amdgcn_exp2f(d[r]);
                sum4 += e;
                acc[u][t] = e;
            }
#pragma unroll
            for (int pr = 0; pr < NP; ++pr)
                Pf[u][pr] = cat8(pack4(acc[u][2 * pr]), pack4(acc[u][2 * pr + 1]));     // un-normalised, in (0, 1]
            float sum = (sum4[0] + sum4[1]) + (sum4[2] + sum4[3]);
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            inv[u] = __builtin_amdgcn_rcpf(sum);
            const int ql = 16 * (QT * pp + u) + c16;
            if (ql < S && g == 0) p.lse[((long)b * p.H + h) * S + ql] = mx + __builtin_amdgcn_logf(sum) * LN2;
            __builtin_amdgcn_sched_barrier(0);         // (the two tiles' softmax blocks are not interleaved: register pressure)
        }
#ifdef ATT16_STAMP3
        asm volatile("s_nop 0" :: "v"(inv[1]), "v"(Pf[1][NP - 1]));
        if (pp == wave) F3_STAMP(4);
#endif
        if (v_pending) {                               // (first pass of every wave: the waves of a workgroup stay in step)
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            v_pending = false;
        }
        // the next pair's q fragments and first mask rows land under this pair's P V products (requested here, where few
        // registers are live; the last pass re-requests its own pair: the request count stays a constant)
        qn = request_q(min(pp + NW, npairs - 1));
        request_m(QT * min(pp + NW, npairs - 1), m0n);
        {   // O^T = V_h^T P^T for both tiles: one output tile's V^T fragments (transposed reads) ahead of the products
            const __bf16* vbase = imgV + (4 * g + q4) * LDH + 4 * p4;
            bf16x8 vc[NP], vn[NP];
#pragma unroll
            for (int pr = 0; pr < NP; ++pr) vc[pr] = cat8(tr4(vbase + (32 * pr) * LDH), tr4(vbase + (32 * pr + 16) * LDH));
            const int ql0 = 16 * QT * pp + c16, ql1 = ql0 + 16;
            __bf16* orow0 = p.out + ((long)b * S + ql0) * D + h * hd;
            __bf16* orow1 = orow0 + 16 * (long)D;
#pragma unroll
            for (int dt = 0; dt < ndt; ++dt) {
                if (dt + 1 < ndt) {
#pragma unroll
                    for (int pr = 0; pr < NP; ++pr)
                        vn[pr] = cat8(tr4(vbase + (32 * pr) * LDH + 16 * (dt + 1)), tr4(vbase + (32 * pr + 16) * LDH + 16 * (dt + 1)));
                }
                __builtin_amdgcn_sched_barrier(0);
                f32x4v o0 = {0.f, 0.f, 0.f, 0.f}, o1 = o0;
#pragma unroll
                for (int pr = 0; pr < NP; ++pr) {
                    o0 = MFMA_BF16(vc[pr], Pf[0][pr], o0);
                    o1 = MFMA_BF16(vc[pr], Pf[1][pr], o1);
                }
                __builtin_amdgcn_sched_barrier(0);
                const int d = 16 * dt + 4 * g;
                if (ql0 < S && d < hd) *reinterpret_cast<bf16x4*>(orow0 + d) = pack4(o0 * inv[0]);
                if (ql1 < S && d < hd) *reinterpret_cast<bf16x4*>(orow1 + d) = pack4(o1 * inv[1]);
#pragma unroll
                for (int pr = 0; pr < NP; ++pr) vc[pr] = vn[pr];
            }
        }
#ifdef ATT16_STAMP3
        if (pp == wave) F3_STAMP(5);
#endif
    }
    if (v_pending) {                                   // a wave without a tile pair (S <= 96): keep the barrier count
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
#ifdef ATT16_STAMP3
    F3_STAMP(6);
    __builtin_amdgcn_s_barrier();
    if (lane == 0 && wave == 0) {          // timing build: cycles of this workgroup's sections over its lse row (destroys lse)
        const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
        float* d = p.lse + ((long)b * p.H + h) * S;
        for (int i = 1; i < 7; ++i) d[i] = (float)(st3[i] - st3[0]);
        d[0] = (float)(rt1 - rt0);                     // 100 MHz ticks
        d[7] = (float)(st3[0] & 0xFFFFFF);
        d[8] = (float)blockIdx.x;
    }
#endif
}
