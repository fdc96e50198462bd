// calm_gemm: strided / batched / split-K GEMM with fused epilogue for gfx950 — the dispatcher.
// Two kernel families share tiling, remap, split logic and epilogue (gemm_common.h): the exact fp32 MFMA family
// (gemm_f32.hip) and the bf16-operand family (gemm_bf16.hip).
#include <atomic>
#include "gemm_common.h"
#include <stdlib.h>

using namespace calm_gemm_detail;

namespace {

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

// The same on 16-byte vectors with four slices in flight (N, the output's row stride and the slice size multiples of 4,
// 16-byte aligned bases): the scalar form above moved 3.2 TB/s of mostly cache-resident partial tiles (PMC, round 4:
// 96 calls, 0.78 ms per fp32 step).  Fixed order: slice sl goes to running sum sl % 4, combined (s0 + s1) + (s2 + s3).
__global__ __launch_bounds__(256) void splitk_reduce_vec(const ReduceP q) {
    const long total4 = ((long)q.M * q.N) >> 2;
    const f32x4* w = reinterpret_cast<const f32x4*>(q.ws + (long)blockIdx.y * q.nslices * q.ws_slice);
    const long slice4 = q.ws_slice >> 2;
    float* out = q.Cg[0] ? q.Cg[blockIdx.y] : q.C + blockIdx.y * q.c_b0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
        int sl = 0;
        for (; sl + 3 < q.nslices; sl += 4) {
            s0 += w[(long)sl * slice4 + i];
            s1 += w[(long)(sl + 1) * slice4 + i];
            s2 += w[(long)(sl + 2) * slice4 + i];
            s3 += w[(long)(sl + 3) * slice4 + i];
        }
        if (sl < q.nslices) s0 += w[(long)sl * slice4 + i];
        if (sl + 1 < q.nslices) s1 += w[(long)(sl + 1) * slice4 + i];
        if (sl + 2 < q.nslices) s2 += w[(long)(sl + 2) * slice4 + i];
        const unsigned e = (unsigned)(4 * i);
        const unsigned m = e / (unsigned)q.N, n = e - m * (unsigned)q.N;
        f32x4* dst = reinterpret_cast<f32x4*>(out + (long)m * q.c_rs + n);
        const f32x4 sum = (s0 + s1) + (s2 + s3);
        *dst = q.accumulate ? *dst + sum : sum;
    }
}

static void launch_splitk_reduce(const ReduceP& q, int n_out, hipStream_t s) {
    const long total = (long)q.M * q.N;
    bool vec = (q.N & 3) == 0 && (q.c_rs & 3) == 0 && (q.c_b0 & 3) == 0 && (q.ws_slice & 3) == 0 && aligned16(q.ws) &&
               total < (1l << 31);
    for (int g = 0; g < 4; ++g) vec = vec && (!q.Cg[g] || aligned16(q.Cg[g]));
    vec = vec && (q.Cg[0] || aligned16(q.C));
    if (vec) {
        const long t4 = total >> 2;
        const int gx = (int)((t4 + 255) / 256 < 2048 ? (t4 + 255) / 256 : 2048);
        hipLaunchKernelGGL(splitk_reduce_vec, dim3(gx, n_out), dim3(256), 0, s, q);
    } else {
        const int gx = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
        hipLaunchKernelGGL(splitk_reduce, dim3(gx, n_out), dim3(256), 0, s, q);
    }
}

inline bool mult4(int64_t x) { return (x & 3) == 0; }

}  // namespace

#ifndef CALM_GEMM_WS_MIN_SLICES
#define CALM_GEMM_WS_MIN_SLICES 48     // per-slice partial tiles + one reduction instead of atomics from this many k-slices
#endif                                 // per output (A/B with plain stores in place of the atomics: outputs of 528 rows and
                                       // more, <= 42 slices, do not change; 384x768 ... 240x240, 53-106 slices, -15..-30%)


// ---- pipelined persistent family (gemm_bf16p.h) -----------------------------------------------------------------------
// Takes the launches whose operands are both bf16 tensors: activation x weight (forward), gradient x weight^T (data
// gradient), gradient^T x activation (weight gradient, k-split over the chip), plain batches and independent groups.
// Declines (the 256-thread / 256x128 kernels keep them): batch-reduced and group-reduced sums, scalar epilogues, tiny
// problems.  CALM_GEMM_PIPE=0 in the environment switches the family off (A/B runs).
constexpr int PIPE_DECLINED = -1000;

static std::atomic<int>& pipe_option(int which) {
    static std::atomic<int> opt[3] = {
        [] { const char* e = getenv("CALM_GEMM_PIPE"); return (e && e[0] == '0') ? 0 : 1; }(),
        [] { const char* e = getenv("CALM_GEMM_PIPE32"); return e ? atoi(e) : 0; }(),
        [] { const char* e = getenv("CALM_GEMM_DETERMINISTIC"); return (e && e[0] == '1') ? 1 : 0; }()};
    return opt[which];
}
// every k-split / batch-reduced launch through the workspace + fixed-order reduction (no fp32 atomics)
static bool deterministic() { return pipe_option(CALM_GEMM_OPT_DETERMINISTIC).load(std::memory_order_relaxed) != 0; }
static bool pipe_enabled() { return pipe_option(CALM_GEMM_OPT_PIPE).load(std::memory_order_relaxed) != 0; }
// fp32 instantiation: 0 off (default), 1 every eligible launch, 2 k-contiguous operand pairs only
static int pipe32_mode() { return pipe_enabled() ? pipe_option(CALM_GEMM_OPT_PIPE32).load(std::memory_order_relaxed) : 0; }

// modelled duration of a launch (cycles of one CU, up to a common factor): `rounds` items per persistent workgroup, each
// nk k-tiles of a (64 mt) x (32 nt) tile; a k-tile is bound by its MFMAs (64 mt nt cycles per SIMD at two waves) or by
// staging its (64 mt + 32 nt) x 128 bytes at ~40 B per cycle; the epilogue costs about one extra k-tile per tile row set
// (fp32: eight 32-cycle v_mfma_f32_16x16x4_f32 per tile, k-tile and wave instead of two 16-cycle bf16 ones)
static double pipe_cost(long items, int mt, int nt, int nk, bool split, bool f32) {
    const long rounds = (items + 255) / 256;
    const double mfma = (f32 ? 512.0 : 64.0) * mt * nt, stage = (64.0 * mt + 32.0 * nt) * 128.0 / 40.0;
    const double ktile = (mfma > stage ? mfma : stage) + 120.0;
    const double epi = (split ? 90.0 : f32 ? 70.0 : 45.0) * mt * nt + 600.0;
    return rounds * (nk * ktile + epi);
}

static int pipe_run(const calm_gemm_args* a, GemmP& p, bool akc, bool bkc, bool f32, hipStream_t s, int64_t* query,
                    calm_gemm_plan* plan) {
    if (!akc && bkc) return PIPE_DECLINED;                      // row-contiguous A with k-contiguous B: not instantiated
    if (a->reduce_batch || !p.epi_vec) return PIPE_DECLINED;
    if ((a->act == CALM_ACT_GELU_BWD) + (a->residual != nullptr) + (a->accumulate != 0) > 1) return PIPE_DECLINED;   // one C-shaped epilogue operand
    const int kt = f32 ? 32 : 64;                               // k per k-tile (128 bytes)
    if (a->M < 128 || a->N < 64 || a->K < kt) return PIPE_DECLINED;
    const int batch = a->batch0 * a->batch1;
    // per-lane staging offsets are 32-bit byte offsets from the batch entry's base
    const int64_t span_a = akc ? (int64_t)a->M * a->a_rs : 64 * a->a_cs + a->M;
    const int64_t span_b = bkc ? (int64_t)a->N * a->b_rs : 64 * a->b_cs + a->N;
    const int64_t span_max = f32 ? (1ll << 29) : (1ll << 30);
    if (span_a >= span_max || span_b >= span_max) return PIPE_DECLINED;
    if ((int64_t)a->M * a->c_rs >= (1ll << 29)) return PIPE_DECLINED;        // the epilogue's 32-bit byte offsets into C
    const bool trivial_epi = !a->bias && !a->col_scale && !a->residual && !a->C_pre && a->act == CALM_ACT_NONE;
    const int kpb = (a->K + kt - 1) / kt;

    // k-split: the same decisions as the 256-thread families (weight gradients: one problem, or independent groups,
    // whose tiles alone cannot fill the chip)
    const int tiles_min = ((a->M + 255) / 256) * ((a->N + 255) / 256);
    const bool group_split = a->n_group && a->split_k == 0 && trivial_epi && a->C_group[0] && tiles_min * batch < 128 && kpb >= 32;
    const bool k_split = group_split || a->split_k > 1 ||
                         (a->split_k == 0 && batch == 1 && trivial_epi && tiles_min < 128 && kpb >= 32);
    if (k_split && !group_split && batch != 1) return PIPE_DECLINED;
    // where the 256-thread kernels measured faster inside the training step (Base-224, same box): narrow k-split outputs,
    // small grouped weight gradients, reductions of two or three k-tiles
    if (!f32) {
        if (k_split && (a->M < 200 || a->N < 200)) return PIPE_DECLINED;
        if (group_split && (int64_t)a->M * a->N < 100000) return PIPE_DECLINED;
        if (!k_split && a->K < 160) return PIPE_DECLINED;
    }
    if (k_split && (!trivial_epi || a->c_type != CALM_ST_F32)) return PIPE_DECLINED;

    int best_mt = 0, best_nt = 0, best_split = 1;
    double best = 1e300;
    for (int mt = 2; mt <= 4; ++mt)
        for (int nt = 4; nt <= 8; ++nt) {
            if (f32 && mt == 4 && nt == 8) continue;            // 128 accumulators + fp32 fragments: spills
            const long tiles = (long)((a->M + 64 * mt - 1) / (64 * mt)) * ((a->N + 32 * nt - 1) / (32 * nt));
            if (!k_split) {
                const double c = pipe_cost(tiles * batch, mt, nt, kpb, false, f32);
                if (c < best) { best = c; best_mt = mt; best_nt = nt; best_split = 1; }
                continue;
            }
            const long per = tiles * (group_split ? batch : 1);
            for (int rounds = 1; rounds <= 4; ++rounds) {
                int ns = a->split_k > 1 ? a->split_k : (int)(256L * rounds / per);
                if (ns < 1) ns = 1;
                if (ns > kpb / 4) ns = kpb / 4 > 0 ? kpb / 4 : 1;
                const int nk = (kpb + ns - 1) / ns;
                ns = (kpb + nk - 1) / nk;
                // the slices' partial tiles are combined through atomics / the workspace: M x N x 4 bytes each through a
                // CHIP-WIDE resource at ~2 TB/s (scripts/ab_wgrad_split.py: 240 x 480 x 20480 takes 23.9 us with 16 slices,
                // 31.3 with 64, 38.3 with 128 — 0.13 us per 460 KB slice), in cycles of the launch like pipe_cost
                const double comb = ns > 1 ? (double)a->M * a->N * 4.0 * ns * (group_split ? batch : 1) / 2.0e12 * 2.0e9 : 0.0;
                const double c = pipe_cost(per * ns, mt, nt, nk, ns > 1, f32) + comb;
                if (c < best) { best = c; best_mt = mt; best_nt = nt; best_split = ns; }
            }
        }
    const int mt = best_mt, nt = best_nt, nsplit = best_split;
    p.tiles_m = (a->M + 64 * mt - 1) / (64 * mt);
    p.tiles_n = (a->N + 32 * nt - 1) / (32 * nt);
    p.kpb = kpb;
    p.kb_total = kpb;
    p.reduce_group = 0;
    p.atomic = nsplit > 1;
    p.slices_per_batch = 0;
    if (p.atomic) {
        p.kb_per_z = (kpb + nsplit - 1) / nsplit;
        const int ns = (kpb + p.kb_per_z - 1) / p.kb_per_z;
        if (group_split) {
            p.slices_per_batch = ns;
            p.nz = ns * batch;
        } else {
            p.nz = ns;
        }
    } else {
        p.kb_per_z = kpb;
        p.nz = batch;
    }
    const int n_out = p.slices_per_batch ? batch : 1;
    const int slices_per_out = p.atomic ? p.nz / n_out : 1;
    p.ws = nullptr;
    p.ws_slice = (long)a->M * a->N;
    int64_t ws_need = 0;
    if (p.atomic && (deterministic() || (slices_per_out >= CALM_GEMM_WS_MIN_SLICES && p.ws_slice >= 100000)))
        ws_need = (int64_t)sizeof(float) * p.nz * p.ws_slice;
    if (query) {
        *query = ws_need;
        return 0;
    }
    const bool use_ws = ws_need > 0 && a->workspace && a->workspace_bytes >= ws_need && aligned16(a->workspace);
    if (plan) {
        const long items_ = (long)p.tiles_m * p.tiles_n * p.nz;
        const bool u8_ = !p.atomic && a->c_type == CALM_ST_BF16 && (!a->aux || a->aux_type == CALM_ST_BF16) &&
                         (!a->residual || (a->r_type == CALM_ST_BF16 && !(a->r_rs & 7) && !(a->r_b0 & 7) && !(a->r_b1 & 7))) &&
                         !(a->N & 7) && !(a->c_rs & 7) && !(a->c_b0 & 7) && !(a->c_b1 & 7);
        *plan = calm_gemm_plan{f32 ? 4 : 3, 64 * mt, 32 * nt, f32 ? 32 : 64, p.tiles_m, p.tiles_n, slices_per_out,
                               (int32_t)items_, (int32_t)(items_ < 256 ? items_ : 256), u8_ ? 8 : 4, use_ws ? 1 : 0,
                               512};
        return 0;
    }
    if (use_ws) {
        p.ws = (float*)a->workspace;
    } else if (p.atomic && !a->accumulate) {
        for (int g = 0; g < n_out; ++g) {
            float* out = (float*)(p.slices_per_batch ? p.Cg[g] : p.C);
            hipError_t e;
            if (a->c_rs == a->N) e = hipMemsetAsync(out, 0, sizeof(float) * (size_t)a->M * a->N, s);
            else e = hipMemset2DAsync(out, sizeof(float) * a->c_rs, 0, sizeof(float) * a->N, a->M, s);
            if (e != hipSuccess) return (int)e;
        }
    }
#ifdef CALM_PIPE_STAMP
    if (!p.atomic && a->workspace && a->workspace_bytes >= 256 * 2 * 8 * 4 * 8) p.ws = (float*)a->workspace;   // diagnostic build: stamps
#endif
    const long items = (long)p.tiles_m * p.tiles_n * p.nz;
    const int grid = (int)(items < 256 ? items : 256);
    static const int stagger = [] { const char* e = getenv("CALM_PIPE_STAGGER"); return e ? atoi(e) : 0; }();
    p.stagger = items > 256 ? stagger : 0;
    {   // 8 columns per lane in the epilogue when every tensor it touches is bf16 and addressable in aligned groups of 8
        auto m8 = [](int64_t x) { return (x & 7) == 0; };
        const bool u8 = !p.atomic && a->c_type == CALM_ST_BF16 && (!a->aux || a->aux_type == CALM_ST_BF16) &&
                        (!a->residual || (a->r_type == CALM_ST_BF16 && m8(a->r_rs) && m8(a->r_b0) && m8(a->r_b1))) &&
                        m8(a->N) && m8(a->c_rs) && m8(a->c_b0) && m8(a->c_b1);
        p.epi_unit = u8 ? 8 : 4;
    }
    int rc;
    if (f32) {
        if (akc && bkc) rc = launch_pipe32_kk(p, mt, nt, grid, s);
        else if (akc) rc = launch_pipe32_km(p, mt, nt, grid, s);
        else rc = launch_pipe32_mm(p, mt, nt, grid, s);
    } else if (akc && bkc) rc = launch_pipe_kk(p, mt, nt, grid, s);
    else if (akc) rc = launch_pipe_km(p, mt, nt, grid, s);
    else rc = launch_pipe_mm(p, mt, nt, grid, s);
    if (rc || !use_ws) return rc;
    ReduceP q;
    q.ws = p.ws; q.ws_slice = p.ws_slice; q.nslices = slices_per_out;
    q.C = (float*)p.C; q.c_b0 = a->c_b0; q.c_rs = a->c_rs;
    for (int g = 0; g < 4; ++g) q.Cg[g] = p.slices_per_batch ? (float*)p.Cg[g] : nullptr;
    q.M = a->M; q.N = a->N; q.accumulate = a->accumulate;
    launch_splitk_reduce(q, n_out, s);
    CALM_LAUNCH_CHECK();
    return 0;
}

// query != nullptr: plan only and report the workspace size of the launch (calm_gemm_workspace_bytes);
// plan != nullptr: plan only and describe the launch (calm_gemm_describe)
static int gemm_run(const calm_gemm_args* a, void* stream, int64_t* query, calm_gemm_plan* plan = nullptr) {
    if (!a || !a->A || !a->B || !a->C) return CALM_E_INVAL;
    if (a->M <= 0 || a->N <= 0 || a->K <= 0 || a->batch0 <= 0 || a->batch1 <= 0) return CALM_E_INVAL;
    if (a->dtype != CALM_F32 && a->dtype != CALM_BF16 && a->dtype != CALM_BF16X3) return CALM_E_UNSUPP;
    for (int t : {a->c_type, a->aux_type, a->r_type})
        if (t != CALM_ST_F32 && t != CALM_ST_BF16) return CALM_E_INVAL;
    for (int t : {a->a_type, a->b_type})
        if (t < CALM_ST_F32 || t > CALM_ST_FP8_E5M2) return CALM_E_INVAL;
    const bool fp8 = a->a_type >= CALM_ST_FP8_E4M3 || a->b_type >= CALM_ST_FP8_E4M3;
    const bool any_bf16_tensor = a->a_type || a->b_type || a->c_type || a->aux_type || a->r_type;      // (or fp8)
    if (any_bf16_tensor && a->dtype != CALM_BF16) return CALM_E_UNSUPP;       // bf16 tensors: bf16 matrix pipe only
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
    p.A = a->A; p.B = a->B; p.C = a->C;
    p.a_type = a->a_type; p.b_type = a->b_type; p.c_type = a->c_type; p.aux_type = a->aux_type; p.r_type = a->r_type;
    p.dq_a = fp8 ? a->a_dq : nullptr; p.dq_b = fp8 ? a->b_dq : nullptr;
    p.M = a->M; p.N = a->N; p.K = a->K;
    p.batch1 = a->batch1;
    p.a_rs = a->a_rs; p.a_cs = a->a_cs; p.a_b0 = a->a_b0; p.a_b1 = a->a_b1;
    p.b_rs = a->b_rs; p.b_cs = a->b_cs; p.b_b0 = a->b_b0; p.b_b1 = a->b_b1;
    p.c_rs = a->c_rs; p.c_b0 = a->c_b0; p.c_b1 = a->c_b1;
    p.alpha = a->alpha; p.inv_scale = a->inv_scale; p.bias = a->bias; p.col_scale = a->col_scale;
    p.residual = a->residual; p.r_rs = a->r_rs; p.r_b0 = a->r_b0; p.r_b1 = a->r_b1;
    p.C_pre = a->C_pre; p.aux = a->aux;
    p.act = a->act; p.accumulate = a->accumulate;
    p.n_group = a->n_group;
    p.reduce_group = a->n_group && a->reduce_batch;
    for (int g = 0; g < 4; ++g) {
        const bool on = g < a->n_group;
        p.Ag[g] = on ? a->A_group[g] : nullptr;
        p.Bg[g] = on ? a->B_group[g] : nullptr;
        p.Cg[g] = on ? a->C_group[g] : nullptr;
        p.Sg[g] = on ? a->inv_scale_group[g] : nullptr;
    }
    {   // vector epilogue: every tensor the epilogue touches addressable in aligned groups of 4 columns
        auto m4 = [](int64_t x) { return (x & 3) == 0; };
        bool ev = m4(a->N) && m4(a->c_rs) && m4(a->c_b0) && m4(a->c_b1) && aligned16(a->C) &&
                  (!a->C_pre || aligned16(a->C_pre)) && (!a->aux || aligned16(a->aux)) &&
                  (!a->bias || aligned16(a->bias)) && (!a->col_scale || aligned16(a->col_scale)) &&
                  (!a->residual || (aligned16(a->residual) && m4(a->r_rs) && m4(a->r_b0) && m4(a->r_b1)));
        for (int g = 0; g < a->n_group; ++g) ev = ev && (!a->C_group[g] || aligned16(a->C_group[g]));
        p.epi_vec = ev && CALM_GEMM_VEC_EPILOGUE;
    }
    const int batch = a->batch0 * a->batch1;
    const bool akc = a->a_cs == 1;
    const bool bkc = a->b_cs == 1;
    if (fp8) {
        // fp8 family: both operands fp8 (B e4m3; A e4m3 or e5m2), k-contiguous, K and row strides multiples of 16 bytes,
        // one launch per call (no k-split, no groups), dequantisation factors on the device
        if (a->b_type != CALM_ST_FP8_E4M3 || a->a_type < CALM_ST_FP8_E4M3 || !a->a_dq || !a->b_dq) return CALM_E_INVAL;
        if (a->dtype != CALM_BF16 || !akc || !bkc || a->n_group || a->reduce_batch || a->split_k > 1) return CALM_E_UNSUPP;
        if ((a->K & 15) || (a->a_rs & 15) || (a->b_rs & 15) || (a->a_b0 & 15) || (a->a_b1 & 15) || (a->b_b0 & 15) ||
            (a->b_b1 & 15) || !aligned16(a->A) || !aligned16(a->B))
            return CALM_E_LAYOUT;
        if (batch > 65535) return CALM_E_UNSUPP;
        p.tiles_m = (a->M + WBM - 1) / WBM;
        p.tiles_n = (a->N + WBN - 1) / WBN;
        p.kpb = 0; p.kb_total = 0; p.kb_per_z = 0; p.atomic = 0; p.slices_per_batch = 0; p.reduce_group = 0;
        p.ws = nullptr; p.ws_slice = 0;
        if (query) {
            *query = 0;
            return 0;
        }
        if (plan) {
            *plan = calm_gemm_plan{5, WBM, WBN, 64, p.tiles_m, p.tiles_n, 1, p.tiles_m * p.tiles_n * batch,
                                   p.tiles_m * p.tiles_n * batch, 0, 0, WTHREADS};
            return 0;
        }
        return launch_fp8(p, dim3(p.tiles_m * p.tiles_n, batch), s);
    }
    // 16-byte staging vectors hold 4 fp32 or 8 bf16 elements: sizes / strides of an operand must be multiples of that
    const int64_t ea = a->a_type == CALM_ST_BF16 ? 7 : 3, eb = a->b_type == CALM_ST_BF16 ? 7 : 3;
    auto mult = [](int64_t x, int64_t mask) { return (x & mask) == 0; };
    bool vec = aligned16(a->A) && aligned16(a->B) && mult(a->a_b0, ea) && mult(a->a_b1, ea) && mult(a->b_b0, eb) &&
               mult(a->b_b1, eb);
    for (int g = 0; g < a->n_group; ++g) vec = vec && aligned16(a->A_group[g]) && aligned16(a->B_group[g]);
    vec = vec && (akc ? (mult(a->K, ea) && mult(a->a_rs, ea)) : (mult(a->M, ea) && mult(a->a_cs, ea)));
    vec = vec && (bkc ? (mult(a->K, eb) && mult(a->b_rs, eb)) : (mult(a->N, eb) && mult(a->b_cs, eb)));
    // bf16-operand kernels need the 16-byte staging path; anything else runs on the exact fp32 kernels — which
    // only read fp32 tensors: a bf16 tensor in a launch that cannot be vectorised is the caller's layout error
    if (!vec && any_bf16_tensor) return CALM_E_LAYOUT;
    const int family = vec ? a->dtype : CALM_F32;
    if (family == CALM_BF16 && a->a_type == CALM_ST_BF16 && a->b_type == CALM_ST_BF16 && pipe_enabled()) {
        const int rc = pipe_run(a, p, akc, bkc, false, s, query, plan);
        if (rc != PIPE_DECLINED) return rc;
    }
    if (vec && a->dtype == CALM_F32 && !any_bf16_tensor && (pipe32_mode() == 1 || (pipe32_mode() == 2 && akc && bkc))) {
        const int rc = pipe_run(a, p, akc, bkc, true, s, query, plan);
        if (rc != PIPE_DECLINED) return rc;
    }
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
        // every slice adds its whole tile onto the SAME output with atomics: beyond about one workgroup per CU the
        // contention costs more than the shorter slices save (scripts/ab_reduce_split.py, 256 images: 80x176x528 51.6 us
        // at 512 slices, 33.4 at 128; 224x176x528 69.7 -> 56.5; 128x80x240 24.4 -> 19.7)
        // (measured on the bf16-operand kernels; the fp32 kernels, 16x slower per k-block, keep their full round of slices)
        if (a->split_k <= 1 && family == CALM_BF16) {
            const int cap = tiles == 1 ? 128 : 256 / tiles;
            if (nsplit > cap) nsplit = cap;
        }
        if (nsplit < 1) nsplit = 1;
    } else if (k_split) {
        if (batch != 1) return CALM_E_UNSUPP;
        nsplit = a->split_k > 1 ? a->split_k : split_slots / tiles;
        const int max_split = (p.kpb + 15) / 16;
        if (nsplit > max_split) nsplit = max_split;
        // small outputs (the 128-row kernels' share of the weight gradients): every slice adds its tiles onto the same
        // output — about one workgroup per CU is the optimum (264 x 240 x 45056: 41.9 us at 88 slices, 29.7 at 32)
        // (bf16-operand kernels only: as above)
        if (a->split_k <= 1 && family == CALM_BF16 && !wide && nsplit > 256 / tiles) nsplit = 256 / tiles;
        if (nsplit < 1) nsplit = 1;
        p.kb_total = p.kpb;
        p.atomic = nsplit > 1;
    } else {
        p.kb_total = batch * p.kpb;
    }
    if (p.atomic && a->c_type != CALM_ST_F32) return CALM_E_UNSUPP;            // k-slices combine in fp32
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
    if (p.atomic && (deterministic() || (slices_per_out >= CALM_GEMM_WS_MIN_SLICES && p.ws_slice >= 100000)))     // tiny outputs: the second launch costs more than their atomics
        ws_need = (int64_t)sizeof(float) * gy * p.ws_slice;
    if (query) {
        *query = ws_need;
        return 0;
    }
    const bool use_ws = ws_need > 0 && a->workspace && a->workspace_bytes >= ws_need && aligned16(a->workspace);
    if (plan) {
        const int fam = wide ? 2 : family == CALM_F32 ? 0 : 1;
        *plan = calm_gemm_plan{fam, wide ? WBM : BM, bn, family == CALM_F32 ? BK : CK, p.tiles_m, p.tiles_n,
                               p.atomic ? slices_per_out : 1, tiles * gy, tiles * gy, 0, use_ws ? 1 : 0,
                               wide ? WTHREADS : NTHREADS};
        return 0;
    }
    if (use_ws) {
        p.ws = (float*)a->workspace;
    } else if (p.atomic && !a->accumulate) {
        for (int g = 0; g < n_out; ++g) {
            float* out = (float*)(p.slices_per_batch ? p.Cg[g] : p.C);
            hipError_t e;
            if (a->c_rs == a->N) e = hipMemsetAsync(out, 0, sizeof(float) * (size_t)a->M * a->N, s);
            else e = hipMemset2DAsync(out, sizeof(float) * a->c_rs, 0, sizeof(float) * a->N, a->M, s);
            if (e != hipSuccess) return (int)e;
        }
    }

    auto launch_main = [&]() -> int {
        if (wide) return launch_bf16_wide(p, grid, akc, bkc, s);
        if (family == CALM_BF16) return launch_bf16(p, grid, bn, akc, bkc, 1, s);
        if (family == CALM_BF16X3) return launch_bf16(p, grid, bn, akc, bkc, 3, s);
        return launch_f32(p, grid, bn, akc, bkc, vec, s);
    };
    const int rc = launch_main();
    if (rc || !use_ws) return rc;

    ReduceP q;
    q.ws = p.ws; q.ws_slice = p.ws_slice; q.nslices = slices_per_out;
    q.C = (float*)p.C; q.c_b0 = a->c_b0; q.c_rs = a->c_rs;
    for (int g = 0; g < 4; ++g) q.Cg[g] = p.slices_per_batch ? (float*)p.Cg[g] : nullptr;
    q.M = a->M; q.N = a->N; q.accumulate = a->accumulate;
    launch_splitk_reduce(q, n_out, s);
    CALM_LAUNCH_CHECK();
    return 0;
}

extern "C" int calm_gemm(const calm_gemm_args* a, void* stream) { return gemm_run(a, stream, nullptr); }

extern "C" int calm_gemm_set_option(int32_t option, int32_t value) {
    if (option != CALM_GEMM_OPT_PIPE && option != CALM_GEMM_OPT_PIPE32 && option != CALM_GEMM_OPT_DETERMINISTIC)
        return CALM_E_INVAL;
    if (value < 0 || value > (option == CALM_GEMM_OPT_PIPE32 ? 2 : 1)) return CALM_E_INVAL;
    return pipe_option(option).exchange(value);
}

extern "C" int calm_gemm_describe(const calm_gemm_args* a, calm_gemm_plan* plan) {
    if (!plan) return CALM_E_INVAL;
    return gemm_run(a, nullptr, nullptr, plan);
}

extern "C" int64_t calm_gemm_workspace_bytes(const calm_gemm_args* a) {
    int64_t bytes = 0;
    return gemm_run(a, nullptr, &bytes) == 0 ? bytes : 0;
}
