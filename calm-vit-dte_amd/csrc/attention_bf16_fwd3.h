// Round 4 — the per-head core of the fused bf16 latent-mask attention as a kernel of its own (included by
// attention_bf16.hip inside its anonymous namespace; fragment orders, LDS image geometry and arithmetic of the head loop
// of attn16_fwd2_kernel).
//
//   O_h = softmax_j(Q_h K_h^T / sqrt(hd) + M) V_h       work item = (image, head); the mask M is read from HBM / L2
//
// Why the head loop left attn16_fwd2_kernel.  There a workgroup = (image, 7 query tiles) walks ALL heads with the mask in
// registers, at the price of a ring of K_h / V_h stages with a hand-kept ledger of vector-memory instructions (~400 issue
// slots per head and wave beside the ~500 of the arithmetic), 7 waves (one SIMD holds a single wave), and ONE query tile
// per wave: in this orientation the K_h / V_h fragments are the LDS-fed A operands, so every 1 KiB fragment read (8 LDS
// cycles) feeds ONE 16-cycle product — four SIMDs want 256 B / cycle from an LDS that delivers 128.  In-kernel stamps: 129k
// cycles for the head phase against 22k cycles of MFMA work per SIMD; PMC 14.5 % MFMA busy.
//
// Organisation here (measured steps in DESIGN.md section 7):
//   * PERSISTENT workgroups walk (image, head) items; an item's K_h and V_h images are brought in by LDS-DMA into the
//     stage the previous item is not using (two stages) while the previous item computes: the HBM / L2 traffic of the
//     launch is spread evenly over its duration instead of arriving in one burst per round of workgroups (a one-shot
//     version — stage, wait, compute — measured 6-8 us of exposed staging per 20 us workgroup, every CU of the chip
//     bursting at the same moment), and there is ONE barrier per item;
//   * the staging is the job of a dedicated LOADER WAVE (the last wave of the workgroup; NP compute waves + 1 = 8 waves at
//     S = 224).  `s_waitcnt vmcnt` retires a wave's vector-memory operations in issue order, LDS-DMA included: when the
//     compute waves issued the next item's requests themselves, every wait for their own mask rows also waited for those
//     requests — timing builds with the products, the P V products or the exponentials REMOVED ran exactly as long as
//     the full kernel, one without the in-loop staging 30 us shorter.  A loader wave keeps the two streams apart;
//   * a wave owns a PAIR of query tiles (32 queries): every K_h / V_h fragment read from LDS feeds two products — half
//     the LDS bytes per product; the mask rows of a pair (bf16, exactly the values the backward will read) are 2 x 2 NP
//     eight-byte loads per lane, requested ahead of their use and kept packed (two bf16 per register) until the packed
//     FMA that adds them; the items of one image are dealt to one XCD next to each other in time, so these reads and
//     the K / V staging share L2 lines;
//   * all loads of a pair are unconditional (clamped addresses + selects) and the pad chunks of an image row are zeroed
//     by the lanes that sit out the LDS-DMA instruction: the number of vector-memory instructions a wave issues per item
//     is a constant, so the compiler's own counted waits stay exact;
//   * P is left un-normalised for the P V product (exp2(x - max) in (0, 1], rounded to bf16) and 1 / sum goes onto the
//     16 output values of a lane instead of its 56 probabilities.
// Instruction budget per (tile, head) and wave: 56 MFMA, 56 v_exp_f32, ~200 other VALU (mask unpack 56, two packed FMAs
// per pair, max3 chain, packed sums, bf16 packing), 14 + 28 LDS reads, 14 + 4 global loads, 5 stores.
#pragma once

__host__ __device__ constexpr int fwd3_waves(int np) { return np + 1; }       // NP compute waves (one tile pair each) + one loader wave

template <int NP, int HDP>
struct Fwd3Geo {
    static constexpr int SP = 32 * NP, NW = fwd3_waves(NP);
    static constexpr int CPRH = HDP / 8 + 2, NI3 = (SP * CPRH + 63) / 64;        // LDS-DMA instructions per image
    static constexpr int STAGE = 2 * NI3 * 1024;                                 // K_h and V_h images of one item
    static constexpr int LDS = 2 * STAGE + 64;                                   // two stages + the FULL / DONE words
    static constexpr int BY_LDS = (160 * 1024) / LDS, BY_WAVES = (NP <= 4 ? 12 : 8) / NW;     // (<= 256 VGPRs: two waves per SIMD; <= 168 below NP = 5: three)
    static constexpr int WG_PER_CU = BY_LDS < BY_WAVES ? BY_LDS : BY_WAVES;
    static constexpr bool OK = NP <= 7 && HDP <= 64 && WG_PER_CU >= 1;
};

#ifndef F3_ABLATE
#define F3_ABLATE 0      // timing-only builds (wrong results): 1 no staging inside the item loop, 2 no mask loads, 3 no P V,
#endif                   // 4 no Q K^T products, 5 no softmax exponentials
#ifdef ATT16_STAMP3
#define F3_STAMP(i) st3[i] = __builtin_amdgcn_s_memtime()
#else
#define F3_STAMP(i)
#endif

// EXACT: S == 32 NP (no pad keys) — a compile-time property so that both variants of the mask request have a fixed number of
// vector-memory instructions.
template <int NP, int HDP, bool EXACT>
__global__ __launch_bounds__(64 * fwd3_waves(NP), 2) void attn16_fwd3_core_kernel(const Attn16P p) {
    typedef Fwd3Geo<NP, HDP> G;
    constexpr int NJ = 2 * NP, NW = G::NW;
    constexpr int LDH = ld_rt(HDP), nks = HDP / 32, ndt = HDP / 16;
    static_assert(LDH * 2 == G::CPRH * 16, "image stride");
    extern __shared__ __attribute__((aligned(1024))) char smem3[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c16 = lane & 15, g = lane >> 4;
    const int S = p.S, D = p.H * p.hd, hd = p.hd, H = p.H;
#ifdef ATT16_STAMP3
    unsigned long long st3[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
    F3_STAMP(0);
#endif
    // items of this workgroup: ALL heads of image blockIdx.x, then of image blockIdx.x + gridDim.x, ...  A compute wave
    // keeps the same tile pair for every head of an image, so the pair's mask rows (2 x 2 NP packed registers) are read
    // once per image and stay in registers — per item a compute wave issues 8 loads and 10 stores.  (Dealing the
    // (image, head) items head-minor over the workgroups instead re-read the mask rows per head: 28 more eight-byte
    // loads per wave and item, ~390 vector-memory instructions per item and CU at >= 16 cycles of address processing
    // each — the items ran at that rate whatever arithmetic was removed from them.)
    const int n_img = (p.B - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int n_it = n_img * H;
    auto item_bh = [&](int it, int& b, int& h) __attribute__((always_inline)) {
        const int im = it / H;
        h = it - im * H;
        b = (int)blockIdx.x + im * (int)gridDim.x;
    };
    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)reinterpret_cast<uintptr_t>(smem3));
    const bool straddle = (hd & 4) != 0;            // hd = 44, 20: the last 16-byte chunk of a head slice is half valid
    const int cvalid = (hd + 7) >> 3;
    const int ntiles = (S + 15) >> 4, npairs = (ntiles + 1) >> 1;
    const bool loader = wave == NW - 1;             // the last wave stages; waves 0 .. NP - 1 own one tile pair each
    const int pp = min(wave, npairs - 1);           // this wave's tile pair (npairs == NP for every S in (32 NP - 32, 32 NP])
    const float scale = p.scale;
    constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
    typedef float f32x2v __attribute__((ext_vector_type(2)));
    typedef unsigned mu32x2 __attribute__((ext_vector_type(2)));
    constexpr int QT = 2;                // query tiles per wave: every K / V fragment read from LDS feeds QT products

    // ---- staging of item (b, h) into stage st: [SP keys][HDP + 16] images; pad keys re-read a valid row (their scores
    //      meet a -inf mask, their V rows a zero probability); pad chunks of a row (columns past the head slice) are not
    //      fetched: their lanes sit out the instruction (exec mask) and zero their 16 bytes of the image ----
    // q fragments and mask rows of this wave's tile pair in item (b, h).  Loads are unconditional — columns past the head
    // slice / keys past S are fetched from a clamped address and replaced by a select.
    // (the request only LOADS — raw 8-byte pieces; the selects that zero the columns past the head slice are applied where
    // the fragments are used, one item later: applied here they made the compiler wait for the loads at once, in front of
    // the P V products — the load latency sat exposed in every item)
    struct QIn { bf16x4 lo[QT][nks], hi[QT][nks]; };
    auto request_q = [&](int b, int h) __attribute__((always_inline)) {
        QIn t;
        const __bf16* qb = p.q + (long)b * S * D + h * hd;
#pragma unroll
        for (int u = 0; u < QT; ++u) {
            const __bf16* qrow = qb + (long)min(16 * (QT * pp + u) + c16, S - 1) * D;
#pragma unroll
            for (int ks = 0; ks < nks; ++ks) {
                const int c = 32 * ks + 8 * g;
                t.lo[u][ks] = ld4(qrow + min(c, hd - 4));
                t.hi[u][ks] = ld4(qrow + min(c + 4, hd - 4));
            }
        }
        return t;
    };
    auto q_frag = [&](const QIn& t, int u, int ks) __attribute__((always_inline)) {
        const bf16x4 z4 = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
        const int c = 32 * ks + 8 * g;
        return cat8(c < hd ? t.lo[u][ks] : z4, c + 4 < hd ? t.hi[u][ks] : z4);
    };
    auto request_m = [&](int b, int u, mu32x2 (&m)[NJ]) __attribute__((always_inline)) {
        const __bf16* mrow = p.Mk + ((long)b * S + min(16 * (QT * pp + u) + c16, S - 1)) * S + 4 * g;
        if (F3_ABLATE == 2) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) m[j] = (mu32x2){(unsigned)b, (unsigned)u};
            return;
        }
        if constexpr (EXACT) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) m[j] = *reinterpret_cast<const mu32x2*>(mrow + 16 * j);
        } else {
            const mu32x2 ninf2 = {0xFF80FF80u, 0xFF80FF80u};                     // bf16 -inf x 4
            int jl = (S - 4 * g + 15) / 16 - 1;                                 // last key tile with this lane group's keys in range
            asm volatile("" : "+v"(jl));      // (opaque: otherwise the 2 NP clamped offsets are hoisted out of the item loop and spilled)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const mu32x2 x = *reinterpret_cast<const mu32x2*>(mrow + 16 * min(j, jl));
                m[j] = j <= jl ? x : ninf2;
            }
        }
    };

    if (n_it <= 0) return;
    // Hand-off between the loader and the compute waves through two words per stage in LDS instead of workgroup barriers:
    // FULL[s] = (item staged in s) + 1, written by the loader once its requests have landed; DONE[s] = number of
    // (compute wave, item) passes that have finished with stage s.  With a barrier per item all compute waves of a
    // workgroup ran their products, their softmax and their P V products at the same time — the matrix pipe idle while
    // the two waves of a SIMD queue for its VALU and vice versa; without one, the start offset given to the second wave
    // of every SIMD persists and one wave's softmax runs under the other's products.
    volatile int* flags = reinterpret_cast<volatile int*>(smem3 + 2 * G::STAGE);
    constexpr int NWC = NW - 1;
    if (tid < 4) flags[tid] = 0;
    int b, h;
    item_bh(0, b, h);
    if (loader) {
        // ---- the loader wave: a loop of its own (its 35 offset registers must not be live in the compute waves' code) ----
        // The loader's per-lane source offsets (instruction q of an image covers image chunks 64 q .. 64 q + 63; K_h and V_h
        // share the geometry) live in its registers for the whole launch — it holds nothing else; pad chunks of a row
        // (columns past the head slice) fetch the 16-byte zero block.
        unsigned soff[G::NI3];
        unsigned long long padmask = 0;                     // bit q: this lane's chunk of instruction q is a pad chunk
        if (loader) {
#pragma unroll
            for (int q = 0; q < G::NI3; ++q) {
                const int L1 = 64 * q + lane;
                const int r = L1 / G::CPRH, c = L1 - r * G::CPRH;
                soff[q] = (unsigned)((min(r, S - 1) * D + 8 * min(c, cvalid - 1)) * 2);
                if (c >= cvalid) padmask |= 1ull << q;
            }
        }
        static_assert(G::NI3 <= 64, "pad mask");
        auto stage_item = [&](int b, int h, int st) __attribute__((always_inline)) {
            const char* kb = reinterpret_cast<const char*>(p.k + (long)b * S * D + h * hd);
            const char* vb = reinterpret_cast<const char*>(p.v + (long)b * S * D + h * hd);
            const char* zero = reinterpret_cast<const char*>(calm_zero_block);
            const unsigned dst0 = lds0 + (unsigned)st * G::STAGE;
            // the last chunk of the last key row of the last head of the last image would end 8 bytes past the tensor: it is
            // fed from the zero block and its 8 valid bytes are patched by hand (one item of the launch, hd % 8 == 4 only)
            const bool last = straddle && h == H - 1 && b == p.B - 1;
            if (!last) {
#pragma unroll
                for (int q = 0; q < G::NI3; ++q) {
                    const bool pad = (padmask >> q) & 1;
                    glds16_addr(pad ? zero : kb + soff[q], dst0 + 1024u * q);
                    glds16_addr(pad ? zero : vb + soff[q], dst0 + 1024u * (G::NI3 + q));
                    __builtin_amdgcn_sched_barrier(0);     // (addresses are formed where they are used, not 70 at once)
                }
            } else {
                // (one item of the launch) everything from the chunk index; the half chunk is fed from the zero block and
                // its 8 valid bytes are patched by hand once the requests have landed
#pragma unroll 1
                for (int q = 0; q < 2 * G::NI3; ++q) {
                    const bool isv = q >= G::NI3;
                    const int L1 = 64 * (isv ? q - G::NI3 : q) + lane;
                    const int r = L1 / G::CPRH, c = L1 - r * G::CPRH;
                    const bool ok = c < cvalid && !(r >= S - 1 && 8 * c + 8 > hd);
                    glds16_addr(ok ? (isv ? vb : kb) + ((long)min(r, S - 1) * D + 8 * c) * 2 : zero, dst0 + 1024u * q);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll 1
                for (int q = 0; q < 2 * G::NI3; ++q) {
                    const bool isv = q >= G::NI3;
                    const int L1 = 64 * (isv ? q - G::NI3 : q) + lane;
                    const int r = L1 / G::CPRH, c = L1 - r * G::CPRH;
                    if (c < cvalid && r >= S - 1 && 8 * c + 8 > hd) {
                        const bf16x4 z = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
                        const __bf16* src = reinterpret_cast<const __bf16*>((isv ? vb : kb) + ((long)(S - 1) * D + 8 * c) * 2);
                        *reinterpret_cast<bf16x8*>(smem3 + st * G::STAGE + (64 * q + lane) * 16) = cat8(ld4(src), z);
                    }
                }
            }
        };
        __builtin_amdgcn_s_barrier();                                  // (the FULL / DONE words are zero)
#pragma unroll 1
        for (int it = 0; it < n_it; ++it) {
            const int sl = it & 1;
            // the stage is free once every compute wave has finished the item that used it last (item it - 2)
            while (flags[2 + sl] < NWC * (it >> 1)) __builtin_amdgcn_s_sleep(4);
            asm volatile("" ::: "memory");
            int bi, hi;
            item_bh(it, bi, hi);
            if (F3_ABLATE != 1 || it < 2) stage_item(bi, hi, sl);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the images have landed ...
            flags[sl] = it + 1;                                            // ... publish: FULL[stage] = item + 1
        }
        return;
    }
    QIn qn;
    mu32x2 m0[NJ], m1[NJ];               // the pair's mask rows, for all heads of the image
    qn = request_q(b, h);
    __builtin_amdgcn_s_barrier();                                      // (the FULL / DONE words are zero)
    asm volatile("" ::: "memory");
    // start offset of the second wave of every SIMD (waves 4 ..): p.kv_shared x 64 cycles, about half an item
    if (wave >= 4)
        for (int i = 0; i < p.kv_shared; i += 10) __builtin_amdgcn_s_sleep(10);
    F3_STAMP(1);

    const int q4 = c16 >> 2, p4 = c16 & 3;
    constexpr int GS = NP > 4 ? 4 : NP, NF = NJ * nks, NG = (NF + GS - 1) / GS;
    int st = 0;
#pragma unroll 1
    for (int it = 0; it < n_it; ++it, st ^= 1) {
        // the next item's images into the other stage (every wave has left it: barrier at the end of the previous item)
        const int itn = it + 1 < n_it ? it + 1 : it;                   // (the last item re-stages itself: constant counts)
        int bn, hn;
        item_bh(itn, bn, hn);
        {
            while (flags[st] < it + 1) __builtin_amdgcn_s_sleep(2);        // the item's images are in stage st
            asm volatile("" ::: "memory");
            const __bf16* imgK = reinterpret_cast<const __bf16*>(smem3 + st * G::STAGE);
            const __bf16* imgV = imgK + G::NI3 * 512;
            bf16x8 bq[QT][nks];
#pragma unroll
            for (int u = 0; u < QT; ++u)
#pragma unroll
                for (int ks = 0; ks < nks; ++ks) bq[u][ks] = q_frag(qn, u, ks);
            if (h == 0) {                                  // a new image: its mask rows for this wave's pair
                request_m(b, 0, m0);
                request_m(b, 1, m1);
            }
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
                for (int gi = 0; gi < (F3_ABLATE == 4 ? 1 : NG); ++gi) {
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
                            acc[0][f % NJ] = MFMA_BF16(cur[i], bq[0][f / NJ], acc[0][f % NJ]);
                            acc[1][f % NJ] = MFMA_BF16(cur[i], bq[1][f / NJ], acc[1][f % NJ]);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < GS; ++i) cur[i] = nx[i];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (it == 0) F3_STAMP(2);
            // softmax of both tiles: z = s scale + M (natural units), row max over the 4 NJ in-lane values and the 4 lane groups
            bf16x8 Pf[QT][NP];
            float inv[QT];
            const f32x2v sc2v = {scale, scale};
            const __amdgpu_buffer_rsrc_t rs_l = make_rsrc(p.lse + ((long)b * H + h) * S, (long)S * 4);
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
                    for (int r = 0; r < 4; ++r) e[r] = F3_ABLATE == 5 ? d[r] : __builtin_amdgcn_exp2f(d[r]);
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
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, mx + __builtin_amdgcn_logf(sum) * LN2), rs_l,
                                                      (ql < S && g == 0) ? (unsigned)(ql * 4) : 0xFFFFFFFFu, 0, 0);
                __builtin_amdgcn_sched_barrier(0);         // (the two tiles' softmax blocks are not interleaved: register pressure)
            }
            if (it == 0) F3_STAMP(3);
            // the next item's q fragments land under this item's P V products
            qn = request_q(bn, hn);
            {   // O^T = V_h^T P^T for both tiles: one output tile's V^T fragments (transposed reads) ahead of the products
                const __bf16* vbase = imgV + (4 * g + q4) * LDH + 4 * p4;
                bf16x8 vc[NP], vn[NP];
#pragma unroll
                for (int pr = 0; pr < NP; ++pr) vc[pr] = cat8(tr4(vbase + (32 * pr) * LDH), tr4(vbase + (32 * pr + 16) * LDH));
                const int ql0 = 16 * QT * pp + c16, ql1 = ql0 + 16;
                const __amdgpu_buffer_rsrc_t rs_o = make_rsrc(p.out + (long)b * S * D + h * hd, ((long)(S - 1) * D + hd) * 2);
#pragma unroll
                for (int dt = 0; dt < (F3_ABLATE == 3 ? 1 : ndt); ++dt) {
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
                    // (stores through the descriptor: a masked lane stores out of range, the instruction count is a constant)
                    buf_store4(rs_o, (ql0 < S && d < hd) ? (unsigned)((ql0 * D + d) * 2) : 0xFFFFFFFFu, pack4(o0 * inv[0]));
                    buf_store4(rs_o, (ql1 < S && d < hd) ? (unsigned)((ql1 * D + d) * 2) : 0xFFFFFFFFu, pack4(o1 * inv[1]));
#pragma unroll
                    for (int pr = 0; pr < NP; ++pr) vc[pr] = vn[pr];
                }
            }
            if (it == 0) F3_STAMP(4);
            // done with the stage: every fragment read has returned (the products that use them have been issued)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_fetch_add(const_cast<int*>(flags) + 2 + st, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (it == 0) F3_STAMP(5);
        b = bn;
        h = hn;
    }
#ifdef ATT16_STAMP3
    F3_STAMP(6);
    if (lane == 0 && wave == 0) {          // timing build: cycles of this workgroup's sections over the lse row of its first item (destroys lse)
        const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
        int b0, h0;
        item_bh(0, b0, h0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        float* d = p.lse + ((long)b0 * H + h0) * S;
        for (int i = 1; i < 7; ++i) d[i] = (float)(st3[i] - st3[0]);
        d[0] = (float)(rt1 - rt0);                     // 100 MHz ticks
        d[7] = (float)n_it;
        d[8] = (float)blockIdx.x;
    }
#endif
}
