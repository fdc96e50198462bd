// bf16-operand kernel family of calm_gemm (dispatcher: gemm.hip; shared pieces: gemm_common.h).
#include "gemm_common.h"

namespace calm_gemm_detail {

// =====================================================================================================
// bf16-operand GEMM family: operands are multiplied on v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate) with fp32
// accumulation.  Each operand tensor is either fp32 in HBM — converted to bf16 while being staged into LDS — or
// already bf16 in HBM (activations and the per-step weight copies of the bf16 pipeline: half the bytes per staged
// element, no conversion; template parameters TA / TB).
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

constexpr int KC_LD = 40;              // bf16 per row of a [row][k] image
constexpr int MC_LD = 160;             // bf16 per k-row of a [k][row] image
constexpr int PLANE = 128 * KC_LD;     // == 32 * MC_LD bf16 = 10240 B

constexpr int MC_LDW = 288;                  // bf16 per k-row of a 256-row [k][row] image (same bank residue as 160)
constexpr int WPLANE_A = WBM * KC_LD;        // 20480 B (>= 32 * MC_LDW)

template <int NPASS>
__device__ __forceinline__ void split4(const f32x4& v, bf16x4& hi, bf16x4& lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        hi[e] = (__bf16)v[e];
        if constexpr (NPASS == 3) lo[e] = (__bf16)(v[e] - (float)hi[e]);
    }
}

// Staging cursor of one operand tile (see OperandCursor in gemm_f32.hip: row addresses once per (tile, batch entry),
// edge rows clamped instead of masked, unconditional 16-byte loads for whole k-blocks), typed by the operand's
// storage: T = float (4 elements per 16-byte vector, converted to bf16 on the way into LDS) or __bf16 (8 per vector,
// stored as loaded).  MAPROWS = rows of the thread map (128 or 256), LIVE = rows of the tile that exist (96 for the B
// side of the 128x96 tile), LDMC = row stride of the [k][row] image.
template <typename T, bool KC, int MAPROWS, int LIVE, int THREADS, int LDMC>
struct TCursor {
    static constexpr int EPV = 16 / (int)sizeof(T);
    static constexpr int NV = MAPROWS * CK / (EPV * THREADS);   // 16-byte vectors per thread per k-tile
    static constexpr int VPR = CK / EPV;                        // KC: vectors per row
    static constexpr int RPP = THREADS / VPR;                   // KC: rows per pass
    static constexpr int LPR = MAPROWS / EPV;                   // MC: threads across the rows of one k
    static_assert(NV >= 1 && NV * EPV * THREADS == MAPROWS * CK, "thread map must tile the operand exactly");
    typedef typename std::conditional<sizeof(T) == 4, f32x4, bf16x8>::type vec_t;
    const T* base;
    unsigned off[NV];
    long step;
    // KC thread map: row slot of a thread within its pass.  bf16 tensors (4 vectors of 16 bytes per 80-byte LDS row): 16
    // consecutive lanes writing rows r..r+3 x 4 chunks collide on 12 banks (row r+3 wraps onto row r); with the slots
    // of each wave's 16 rows transposed (lanes 0-15 -> rows 0,4,8,12, next 16 lanes -> 1,5,9,13 ...) the four rows of a
    // 16-lane group start 16 banks apart and every ds_write_b128 is conflict-free (PMC: 27 % of the LDS cycles of the
    // k-contiguous kernels were bank conflicts).  The same 16 rows per wave-load: global coalescing is unchanged.
    static __device__ __forceinline__ int kc_row(int tid) {
        if constexpr (sizeof(T) == 2 && CALM_GEMM_KC_SWIZZLE) {
            const int q = tid / VPR, s = q & 15;
            return (q & ~15) + 4 * (s & 3) + (s >> 2);
        } else {
            return tid / VPR;
        }
    }
    __device__ __forceinline__ void init(const T* origin, long rs, long cs, int row0, int nrows_all, int k0) {
        const int tid = threadIdx.x;
        const int last = min(nrows_all - row0, LIVE) - 1;
        if constexpr (KC) {
            base = origin + (long)row0 * rs + k0;
#pragma unroll
            for (int i = 0; i < NV; ++i)
                off[i] = (unsigned)(min(kc_row(tid) + RPP * i, last) * rs + EPV * (tid % VPR)) * (unsigned)sizeof(T);
            step = CK;
        } else {
            base = origin + (long)k0 * cs + row0;
            const int row = min(EPV * (tid % LPR), last & ~(EPV - 1));      // rows come in aligned groups of EPV
#pragma unroll
            for (int j = 0; j < NV; ++j) off[j] = (unsigned)((NV * (tid / LPR) + j) * cs + row) * (unsigned)sizeof(T);
            step = CK * cs;
        }
    }
    template <bool FULL>
    __device__ __forceinline__ void load(int k_left, vec_t (&reg)[NV]) {
        const int tid = threadIdx.x;
        const char* b = reinterpret_cast<const char*>(base);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            vec_t v;
            if constexpr (FULL) {
                v = *reinterpret_cast<const vec_t*>(b + off[i]);
            } else {
                const int k = KC ? EPV * (tid % VPR) : NV * (tid / LPR) + i;
                v = vec_t{};
                if (k < k_left) v = *reinterpret_cast<const vec_t*>(b + off[i]);
            }
            reg[i] = v;
        }
        base += step;
    }
    template <int NPASS>
    __device__ __forceinline__ void store(__bf16* __restrict__ hi_plane, __bf16* __restrict__ lo_plane,
                                          const vec_t (&reg)[NV]) const {
        const int tid = threadIdx.x;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            int o;
            if constexpr (KC) o = (kc_row(tid) + RPP * i) * KC_LD + EPV * (tid % VPR);          // [row][k]
            else o = (NV * (tid / LPR) + i) * LDMC + EPV * (tid % LPR);                          // [k][row]
            if constexpr (sizeof(T) == 4) {
                bf16x4 hi, lo;
                split4<NPASS>(reg[i], hi, lo);
                *reinterpret_cast<bf16x4*>(hi_plane + o) = hi;
                if constexpr (NPASS == 3) *reinterpret_cast<bf16x4*>(lo_plane + o) = lo;
            } else {
                static_assert(sizeof(T) == 4 || NPASS == 1, "the hi/lo split needs fp32 operands");
                *reinterpret_cast<bf16x8*>(hi_plane + o) = reg[i];
            }
        }
    }
};

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

template <typename TA, typename TB, bool AKC, bool BKC, int NPASS, int BN_>
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

    typedef TCursor<TA, AKC, BM, BM, NTHREADS, MC_LD> CurA;
    typedef TCursor<TB, BKC, BM, BN_, NTHREADS, MC_LD> CurB;
    // bf16 tensors: two k-tiles of global loads in flight (see gemm_bf16w_kernel) — what the short reductions of the
    // per-image and K <= 256 products are made of is exposed load latency, one per k-tile at distance 1
    constexpr int NSET = (sizeof(TA) == 2 && sizeof(TB) == 2 && NPASS == 1 && CALM_GEMM_BF16_DEEP) ? 2 : 1;
    typename CurA::vec_t ra[NSET][CurA::NV];
    typename CurB::vec_t rb[NSET][CurB::NV];
    CurA ca;
    CurB cb;
    // one copy of the k-loop per case (whole k-tiles / tailed K): see OperandCursor::load
    auto k_loop = [&](auto full_tag) {
    constexpr bool FULL = decltype(full_tag)::value;
    int cur_b = -1;
    auto fetch = [&](int kb, auto set_tag) {
        constexpr int SET = decltype(set_tag)::value;
        const int b = p.kb_total == p.kpb ? 0 : kb / p.kpb;
        const int k0 = (kb - b * p.kpb) * CK;
        if (b != cur_b) {
            const int b0 = b / p.batch1, b1 = b - b0 * p.batch1;
            ca.init(operand_base<TA>(p.A, p.Ag, p.n_group, p.a_b0, p.a_b1, b0, b1), p.a_rs, p.a_cs, m0, p.M, k0);
            cb.init(operand_base<TB>(p.B, p.Bg, p.n_group, p.b_b0, p.b_b1, b0, b1), p.b_rs, p.b_cs, n0, p.N, k0);
            cur_b = b;
        }
        ca.template load<FULL>(p.K - k0, ra[SET]);
        cb.template load<FULL>(p.K - k0, rb[SET]);
    };
    auto stash = [&](int st, auto set_tag) {
        constexpr int SET = decltype(set_tag)::value;
        ca.template store<NPASS>(lds[st][0][0], lds[st][0][NPL - 1], ra[SET]);
        cb.template store<NPASS>(lds[st][1][0], lds[st][1][NPL - 1], rb[SET]);
    };
    typedef std::integral_constant<int, 0> S0;
    typedef std::integral_constant<int, NSET - 1> S1;

    int buf = 0;
    if (kb_begin < kb_end) {
        fetch(kb_begin, S0{});
        stash(0, S0{});
    }
    if constexpr (NSET == 2)
        if (kb_begin + 1 < kb_end) fetch(kb_begin + 1, S1{});
    __syncthreads();

    auto iteration = [&](int kb, auto near_tag, auto far_tag, auto steady_tag) {
        constexpr bool STEADY = decltype(steady_tag)::value;      // both the far fetch and the near stash exist
        if (p.reduce_group && kb != kb_begin && kb % p.kpb == 0) group_rescale<MT, NT>(p, acc, kb / p.kpb);
        if (STEADY || kb + NSET < kb_end) fetch(kb + NSET, far_tag);
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
        if (STEADY || kb + 1 < kb_end) stash(buf ^ 1, near_tag);
        __syncthreads();
        buf ^= 1;
    };
    if constexpr (NSET == 2) {
        // steady state peeled from the tail: with a CONDITIONAL far fetch the compiler must assume the path on which
        // it was skipped, where the near set's loads are the newest, and waits with vmcnt(0) before the stash — which
        // also waits for the far loads just issued and serialises the two tiles again
        int kb = kb_begin;
        for (; kb + 3 < kb_end; kb += 2) {
            iteration(kb, S1{}, S0{}, std::true_type{});
            iteration(kb + 1, S0{}, S1{}, std::true_type{});
        }
        for (; kb < kb_end; kb += 2) {
            iteration(kb, S1{}, S0{}, std::false_type{});
            if (kb + 1 < kb_end) iteration(kb + 1, S0{}, S1{}, std::false_type{});
        }
    } else {
        for (int kb = kb_begin; kb < kb_end; ++kb) iteration(kb, S0{}, S0{}, std::false_type{});
    }
    };
    if (p.K % CK == 0) k_loop(std::true_type{});
    else k_loop(std::false_type{});
    static_assert(sizeof(lds) >= 4096 * (NTHREADS / 64), "the epilogue's per-wave scratch lives in the operand stages");
    gemm_epilogue<MT, NT, true>(p, acc, m0, n0, wm, wn, r, h, z, (kb_end - 1) / p.kpb,
                                (lds_float*)(&lds[0][0][0][0]) + 1024 * wave);
}

// ---- wide tile of the bf16-operand family: 256x128x32 per 512-thread workgroup (8 waves as 4x2, each 64x64) ----
// The 128-row tiles above are bound by re-reading the fp32 operand panels from L2 / Infinity Cache (32 flop per
// byte staged); 256 rows raise that to 42.7.  60 KB LDS, <=128 VGPRs: two workgroups (16 waves) per CU.  Used for
// the data-parallel launches (forward, data gradients) with enough tiles to fill the chip; NPASS == 1 only.
template <typename TA, typename TB, bool AKC, bool BKC>
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

    typedef TCursor<TA, AKC, WBM, WBM, WTHREADS, MC_LDW> CurA;
    typedef TCursor<TB, BKC, WBN, WBN, WTHREADS, MC_LD> CurB;
    // The kernel is bound by the latency of its global loads (bytes in flight per CU), not by the matrix pipe: bf16
    // tensors need only 12 staging registers per k-tile, so TWO k-tiles are kept in flight (register sets 0/1,
    // prefetch distance 2); fp32 tensors (24 registers per k-tile) keep distance 1 to stay within 128 VGPRs.
    // (a row-contiguous A operand needs more address registers: with two sets the 128-VGPR budget spills into the k-loop)
    constexpr int NSET = (sizeof(TA) == 2 && sizeof(TB) == 2 && AKC && CALM_GEMM_BF16_DEEP) ? 2 : 1;
    CurA ca;
    CurB cb;
    typename CurA::vec_t ra[NSET][CurA::NV];
    typename CurB::vec_t rb[NSET][CurB::NV];
    auto k_loop = [&](auto full_tag) {
    constexpr bool FULL = decltype(full_tag)::value;
    int cur_b = -1;
    auto fetch = [&](int kb, auto set_tag) {
        constexpr int SET = decltype(set_tag)::value;
        const int b = p.kb_total == p.kpb ? 0 : kb / p.kpb;
        const int k0 = (kb - b * p.kpb) * CK;
        if (b != cur_b) {
            const int b0 = b / p.batch1, b1 = b - b0 * p.batch1;
            ca.init(operand_base<TA>(p.A, p.Ag, p.n_group, p.a_b0, p.a_b1, b0, b1), p.a_rs, p.a_cs, m0, p.M, k0);
            cb.init(operand_base<TB>(p.B, p.Bg, p.n_group, p.b_b0, p.b_b1, b0, b1), p.b_rs, p.b_cs, n0, p.N, k0);
            cur_b = b;
        }
        ca.template load<FULL>(p.K - k0, ra[SET]);
        cb.template load<FULL>(p.K - k0, rb[SET]);
    };
    auto stash = [&](int st, auto set_tag) {
        constexpr int SET = decltype(set_tag)::value;
        ca.template store<1>(lds_a[st], lds_a[st], ra[SET]);
        cb.template store<1>(lds_b[st], lds_b[st], rb[SET]);
    };
    typedef std::integral_constant<int, 0> S0;
    typedef std::integral_constant<int, NSET - 1> S1;

    int buf = 0;
    if (kb_begin < kb_end) {
        fetch(kb_begin, S0{});
        stash(0, S0{});
    }
    if constexpr (NSET == 2)
        if (kb_begin + 1 < kb_end) fetch(kb_begin + 1, S1{});
    __syncthreads();

    // one k-tile: issue the loads of tile kb+NSET into register set FAR, multiply tile kb out of LDS stage `buf`,
    // move tile kb+1 (register set NEAR: loaded a whole iteration ago when NSET == 2) into the other stage
    auto iteration = [&](int kb, auto near_tag, auto far_tag, auto steady_tag) {
        constexpr bool STEADY = decltype(steady_tag)::value;      // both the far fetch and the near stash exist
        if (p.reduce_group && kb != kb_begin && kb % p.kpb == 0) group_rescale<MT, NT>(p, acc, kb / p.kpb);
        if (STEADY || kb + NSET < kb_end) fetch(kb + NSET, far_tag);
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
        if (STEADY || kb + 1 < kb_end) stash(buf ^ 1, near_tag);
        __syncthreads();
        buf ^= 1;
    };
    if constexpr (NSET == 2) {
        // steady state peeled from the tail: with a CONDITIONAL far fetch the compiler must assume the path on which
        // it was skipped, where the near set's loads are the newest, and waits with vmcnt(0) before the stash — which
        // also waits for the far loads just issued and serialises the two tiles again
        int kb = kb_begin;
        for (; kb + 3 < kb_end; kb += 2) {
            iteration(kb, S1{}, S0{}, std::true_type{});
            iteration(kb + 1, S0{}, S1{}, std::true_type{});
        }
        for (; kb < kb_end; kb += 2) {
            iteration(kb, S1{}, S0{}, std::false_type{});
            if (kb + 1 < kb_end) iteration(kb + 1, S0{}, S1{}, std::false_type{});
        }
    } else {
        for (int kb = kb_begin; kb < kb_end; ++kb) iteration(kb, S0{}, S0{}, std::false_type{});
    }
    };
    if (p.K % CK == 0) k_loop(std::true_type{});
    else k_loop(std::false_type{});
    static_assert(sizeof(lds_a) >= 4096 * (WTHREADS / 64), "the epilogue's per-wave scratch lives in the A stages");
    gemm_epilogue<MT, NT, true>(p, acc, m0, n0, wm, wn, r, h, z, (kb_end - 1) / p.kpb,
                                (lds_float*)(&lds_a[0][0]) + 1024 * wave);
}

// ---- launchers: operand storage types (fp32 / bf16 in HBM) x operand layouts -----------------------------------------
template <typename TA, typename TB>
int launch_wide_t(const GemmP& p, dim3 grid, bool akc, bool bkc, hipStream_t s) {
    if (akc && bkc) hipLaunchKernelGGL((gemm_bf16w_kernel<TA, TB, true, true>), grid, dim3(WTHREADS), 0, s, p);
    else if (akc && !bkc) hipLaunchKernelGGL((gemm_bf16w_kernel<TA, TB, true, false>), grid, dim3(WTHREADS), 0, s, p);
    else if (!akc && bkc) hipLaunchKernelGGL((gemm_bf16w_kernel<TA, TB, false, true>), grid, dim3(WTHREADS), 0, s, p);
    else hipLaunchKernelGGL((gemm_bf16w_kernel<TA, TB, false, false>), grid, dim3(WTHREADS), 0, s, p);
    CALM_LAUNCH_CHECK();
    return 0;
}

template <typename TA, typename TB, bool AKC, bool BKC, int NPASS>
int launch_c(const GemmP& p, dim3 grid, int bn, hipStream_t s) {
    if (bn == 128) hipLaunchKernelGGL((gemm_bf16c_kernel<TA, TB, AKC, BKC, NPASS, 128>), grid, dim3(NTHREADS), 0, s, p);
    else hipLaunchKernelGGL((gemm_bf16c_kernel<TA, TB, AKC, BKC, NPASS, 96>), grid, dim3(NTHREADS), 0, s, p);
    CALM_LAUNCH_CHECK();
    return 0;
}

template <typename TA, typename TB, int NPASS>
int launch_c_layout(const GemmP& p, dim3 grid, int bn, bool akc, bool bkc, hipStream_t s) {
    if (akc && bkc) return launch_c<TA, TB, true, true, NPASS>(p, grid, bn, s);
    if (akc && !bkc) return launch_c<TA, TB, true, false, NPASS>(p, grid, bn, s);
    if (!akc && bkc) return launch_c<TA, TB, false, true, NPASS>(p, grid, bn, s);
    return launch_c<TA, TB, false, false, NPASS>(p, grid, bn, s);
}

int launch_bf16(const GemmP& p, dim3 grid, int bn, bool akc, bool bkc, int npass, hipStream_t s) {
    if (npass == 3) return launch_c_layout<float, float, 3>(p, grid, bn, akc, bkc, s);      // hi/lo split: fp32 tensors only
    if (p.a_type == CALM_ST_BF16 && p.b_type == CALM_ST_BF16) return launch_c_layout<__bf16, __bf16, 1>(p, grid, bn, akc, bkc, s);
    if (p.a_type == CALM_ST_BF16) return launch_c_layout<__bf16, float, 1>(p, grid, bn, akc, bkc, s);
    if (p.b_type == CALM_ST_BF16) return launch_c_layout<float, __bf16, 1>(p, grid, bn, akc, bkc, s);
    return launch_c_layout<float, float, 1>(p, grid, bn, akc, bkc, s);
}
int launch_bf16_wide(const GemmP& p, dim3 grid, bool akc, bool bkc, hipStream_t s) {
    if (p.a_type == CALM_ST_BF16 && p.b_type == CALM_ST_BF16) return launch_wide_t<__bf16, __bf16>(p, grid, akc, bkc, s);
    if (p.a_type == CALM_ST_BF16) return launch_wide_t<__bf16, float>(p, grid, akc, bkc, s);
    if (p.b_type == CALM_ST_BF16) return launch_wide_t<float, __bf16>(p, grid, akc, bkc, s);
    return launch_wide_t<float, float>(p, grid, akc, bkc, s);
}

}  // namespace calm_gemm_detail
