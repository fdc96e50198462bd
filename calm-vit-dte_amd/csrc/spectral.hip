// Spectral norm for every wrapped Linear/Conv of the model in three launches (the reference runs
// ~3 ATen calls per layer per forward: 882 mv + 883 div).
//
// Plan blob (built on the host by calm_sn_plan, copied to device memory by the caller):
//   [SnLayerDev x n_layers][SnWork x n_work][SnWorkA x n_work_a]
// Work item of phases B/C = (layer, row chunk); of phase A = (layer, 64-column group).  Phases (training):
//   A: t[c]   = sum_r W[r,c] u[r]                            (W^T u; a block owns its columns over ALL rows and adds
//                                                             its four row-interleaved partial sums in a fixed order:
//                                                             no atomics, so u, v, sigma are bit-reproducible and
//                                                             identical on every data-parallel rank)
//   B: v      = t / max(|t|, eps);  s[r] = W[r,:] . v       (every block renormalises t itself)
//   C: u      = s / max(|s|, eps);  sigma = u . s           (one block per layer)
// eval: phase B uses the stored v, phase C the stored u (sigma = u . W v, no update).
#include "common.h"

namespace {

constexpr int NT = 256;
constexpr int ROWS_PER_WORK = 16;     // rows per work item (A/B: 64 -> 16, four times the blocks in flight over the 160 MB of weights)

struct SnLayerDev {
    const float* w; float* u; float* v; float* sigma;
    int rows, cols;
    long t_off, s_off;     // offsets into scratch (floats)
};
struct SnWork { int layer, row0, nrows, first; };
struct SnWorkA { int layer, col0; };
constexpr int COLS_PER_WORK_A = 64;

__global__ __launch_bounds__(NT) void sn_phase_a(const SnLayerDev* __restrict__ layers, const SnWorkA* __restrict__ work,
                                                 float* __restrict__ scratch) {
    __shared__ float part[NT / 64][COLS_PER_WORK_A];
    const SnWorkA wk = work[blockIdx.x];
    const SnLayerDev L = layers[wk.layer];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;        // wave g walks rows g, g+4, ... (a 256-byte row segment per load)
    const int c = min(wk.col0 + lane, L.cols - 1);                  // columns past the end are clamped, not stored
    const float* wc = L.w + c;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int r = g;
    for (; r + 12 < L.rows; r += 16) {                              // four independent loads in flight
        s0 = fmaf(wc[(long)r * L.cols], L.u[r], s0);
        s1 = fmaf(wc[(long)(r + 4) * L.cols], L.u[r + 4], s1);
        s2 = fmaf(wc[(long)(r + 8) * L.cols], L.u[r + 8], s2);
        s3 = fmaf(wc[(long)(r + 12) * L.cols], L.u[r + 12], s3);
    }
    for (; r < L.rows; r += 4) s0 = fmaf(wc[(long)r * L.cols], L.u[r], s0);
    part[g][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g == 0 && wk.col0 + lane < L.cols)
        scratch[L.t_off + wk.col0 + lane] = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}

__global__ __launch_bounds__(NT) void sn_phase_b(const SnLayerDev* __restrict__ layers, const SnWork* __restrict__ work,
                                                 float* __restrict__ scratch, int training, float eps) {
    __shared__ float red[4];
    const SnWork wk = work[blockIdx.x];
    const SnLayerDev L = layers[wk.layer];
    const float* vin = training ? scratch + L.t_off : L.v;
    float inv = 1.0f;
    if (training) {
        float q = 0.f;
        for (int c = threadIdx.x; c < L.cols; c += NT) q += vin[c] * vin[c];
        q = block_sum_256(q, red);
        inv = 1.0f / fmaxf(sqrtf(q), eps);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* s = scratch + L.s_off;
    for (int r = wk.row0 + wave; r < wk.row0 + wk.nrows; r += NT / 64) {
        const float* wr = L.w + (long)r * L.cols;
        float acc = 0.f;
        for (int c = lane; c < L.cols; c += 64) acc += wr[c] * (vin[c] * inv);
        acc = wave_sum(acc);
        if (lane == 0) s[r] = acc;
    }
    if (training && wk.first) {
        // safe: no block reads L.v in this launch when training (they all read t)
        for (int c = threadIdx.x; c < L.cols; c += NT) L.v[c] = vin[c] * inv;
    }
}

__global__ __launch_bounds__(NT) void sn_phase_c(const SnLayerDev* __restrict__ layers, const float* __restrict__ scratch,
                                                 int training, float eps) {
    __shared__ float red[4];
    const SnLayerDev L = layers[blockIdx.x];
    const float* s = scratch + L.s_off;
    if (training) {
        float q = 0.f;
        for (int r = threadIdx.x; r < L.rows; r += NT) q += s[r] * s[r];
        q = block_sum_256(q, red);
        const float inv = 1.0f / fmaxf(sqrtf(q), eps);
        float d = 0.f;
        for (int r = threadIdx.x; r < L.rows; r += NT) {
            const float un = s[r] * inv;
            L.u[r] = un;
            d += un * s[r];
        }
        d = block_sum_256(d, red);
        if (threadIdx.x == 0) L.sigma[0] = d;
    } else {
        float d = 0.f;
        for (int r = threadIdx.x; r < L.rows; r += NT) d += L.u[r] * s[r];
        d = block_sum_256(d, red);
        if (threadIdx.x == 0) L.sigma[0] = d;
    }
}

// ---- weight gradient through W_orig / sigma ------------------------------------------------
// pass 1: rowdot[r] = sum_k G[r,k] W[r,k] / sigma   (one wave per row)
__global__ __launch_bounds__(NT) void sn_bwd_rowdot(const float* __restrict__ G, const float* __restrict__ w,
                                                    const float* __restrict__ sigma, float* __restrict__ rowdot,
                                                    int rows, int cols) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * (NT / 64);
    const float inv = 1.0f / sigma[0];
    for (int r = wave; r < rows; r += nwaves) {
        float acc = 0.f;
        for (int c = lane; c < cols; c += 64) acc += G[(long)r * cols + c] * w[(long)r * cols + c];
        acc = wave_sum(acc);
        if (lane == 0) rowdot[r] = acc * inv;
    }
}

// pass 2: dW[r,k] = (ls[r] G[r,k] - dot * u[r] v[k]) / sigma,  dot = sum_r ls[r] rowdot[r]
__global__ __launch_bounds__(NT) void sn_bwd_apply(const float* __restrict__ G, const float* __restrict__ u,
                                                   const float* __restrict__ v, const float* __restrict__ sigma,
                                                   const float* __restrict__ ls, const float* __restrict__ rowdot,
                                                   float* __restrict__ dw, float* __restrict__ d_ls, int rows, int cols) {
    __shared__ float red[4];
    float d = 0.f;
    for (int r = threadIdx.x; r < rows; r += NT) d += (ls ? ls[r] : 1.0f) * rowdot[r];
    const float dot = block_sum_256(d, red);
    const float inv = 1.0f / sigma[0];
    const long total = (long)rows * cols;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int r = (int)(i / cols);
        const int c = (int)(i - (long)r * cols);
        dw[i] = ((ls ? ls[r] : 1.0f) * G[i] - dot * u[r] * v[c]) * inv;
    }
    if (d_ls && blockIdx.x == 0)
        for (int r = threadIdx.x; r < rows; r += NT) d_ls[r] = rowdot[r];
}

// ---- per-step bf16 copies of all weights (bf16 pipeline) ------------------------------------------------------------
constexpr int CAST_CHUNK = 16384;            // elements per work item (64 per thread)
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(NT) void cast_bf16_kernel(const calm_cast_entry* __restrict__ E, const int* __restrict__ chunk_entry) {
    const calm_cast_entry e = E[chunk_entry[blockIdx.x]];
    const long i0 = (long)(blockIdx.x - e.chunk0) * CAST_CHUNK;
    const long i1 = min(i0 + (long)CAST_CHUNK, (long)e.numel);
    __bf16* dst = reinterpret_cast<__bf16*>(e.dst);
    if (((reinterpret_cast<uintptr_t>(e.src) & 15) | (reinterpret_cast<uintptr_t>(e.dst) & 7)) == 0) {
        const long v1 = i0 + ((i1 - i0) & ~3L);
        for (long i = i0 + 4 * threadIdx.x; i < v1; i += 4 * NT) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(e.src + i);
            *reinterpret_cast<bf16x4*>(dst + i) = (bf16x4){(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
        }
        for (long i = v1 + threadIdx.x; i < i1; i += NT) dst[i] = (__bf16)e.src[i];
    } else {
        for (long i = i0 + threadIdx.x; i < i1; i += NT) dst[i] = (__bf16)e.src[i];
    }
}

typedef __bf16 bf16x8c __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(NT) void cast_one_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, long n8, long n) {
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n8; i += (long)gridDim.x * NT) {
        const f32x4 a = reinterpret_cast<const f32x4*>(src)[2 * i], b = reinterpret_cast<const f32x4*>(src)[2 * i + 1];
        reinterpret_cast<bf16x8c*>(dst)[i] = (bf16x8c){(__bf16)a[0], (__bf16)a[1], (__bf16)a[2], (__bf16)a[3],
                                                       (__bf16)b[0], (__bf16)b[1], (__bf16)b[2], (__bf16)b[3]};
    }
    for (long i = 8 * n8 + (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) dst[i] = (__bf16)src[i];
}

__global__ __launch_bounds__(NT) void cast_f32_kernel(const __bf16* __restrict__ src, float* __restrict__ dst, long n) {
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) dst[i] = (float)src[i];
}

}  // namespace

extern "C" {

int32_t calm_cast_chunk_elems(void) { return CAST_CHUNK; }

int calm_cast_f32_one(const void* src, float* dst, int64_t n, void* stream) {
    if (!src || !dst || n <= 0) return CALM_E_INVAL;
    long g = (n + NT - 1) / NT;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(cast_f32_kernel, dim3((int)g), dim3(NT), 0, as_stream(stream), reinterpret_cast<const __bf16*>(src), dst,
                       (long)n);
    CALM_LAUNCH_CHECK();
    return 0;
}

int calm_cast_bf16_one(const float* src, void* dst, int64_t n, void* stream) {
    if (!src || !dst || n <= 0) return CALM_E_INVAL;
    const bool vec = aligned16(src) && aligned16(dst);
    const long n8 = vec ? n / 8 : 0;
    long g = (n / 8 + NT - 1) / NT;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(cast_one_kernel, dim3((int)g), dim3(NT), 0, as_stream(stream), src, reinterpret_cast<__bf16*>(dst), n8,
                       (long)n);
    CALM_LAUNCH_CHECK();
    return 0;
}

int calm_cast_bf16(const calm_cast_entry* entries_dev, const int32_t* chunk_entry_dev, int32_t n_chunks, void* stream) {
    if (!entries_dev || !chunk_entry_dev || n_chunks <= 0) return CALM_E_INVAL;
    hipLaunchKernelGGL(cast_bf16_kernel, dim3(n_chunks), dim3(NT), 0, as_stream(stream), entries_dev, chunk_entry_dev);
    CALM_LAUNCH_CHECK();
    return 0;
}

int calm_sn_plan(const calm_sn_layer* layers, int32_t n, void* blob_host, calm_sn_plan_info* info) {
    if (!layers || n <= 0 || !info) return CALM_E_INVAL;
    int n_work = 0, n_work_a = 0;
    long scratch = 0;
    for (int i = 0; i < n; ++i) {
        if (!layers[i].w || !layers[i].u || !layers[i].v || !layers[i].sigma || layers[i].rows <= 0 ||
            layers[i].cols <= 0)
            return CALM_E_INVAL;
        n_work += (layers[i].rows + ROWS_PER_WORK - 1) / ROWS_PER_WORK;
        n_work_a += (layers[i].cols + COLS_PER_WORK_A - 1) / COLS_PER_WORK_A;
        scratch += layers[i].rows + layers[i].cols;
    }
    info->n_layers = n;
    info->n_work = n_work;
    info->n_work_a = n_work_a;
    info->reserved = 0;
    info->scratch_floats = scratch;
    info->blob_bytes = (int64_t)sizeof(SnLayerDev) * n + (int64_t)sizeof(SnWork) * n_work +
                       (int64_t)sizeof(SnWorkA) * n_work_a;
    if (!blob_host) return 0;
    SnLayerDev* L = reinterpret_cast<SnLayerDev*>(blob_host);
    SnWork* W = reinterpret_cast<SnWork*>(L + n);
    SnWorkA* WA = reinterpret_cast<SnWorkA*>(W + n_work);
    long off = 0;
    int wi = 0, wa = 0;
    for (int i = 0; i < n; ++i) {
        L[i].w = layers[i].w; L[i].u = layers[i].u; L[i].v = layers[i].v; L[i].sigma = layers[i].sigma;
        L[i].rows = layers[i].rows; L[i].cols = layers[i].cols;
        L[i].t_off = off; off += layers[i].cols;
        L[i].s_off = off; off += layers[i].rows;
        for (int r0 = 0; r0 < layers[i].rows; r0 += ROWS_PER_WORK) {
            W[wi].layer = i; W[wi].row0 = r0;
            W[wi].nrows = layers[i].rows - r0 < ROWS_PER_WORK ? layers[i].rows - r0 : ROWS_PER_WORK;
            W[wi].first = r0 == 0;
            ++wi;
        }
        for (int c0 = 0; c0 < layers[i].cols; c0 += COLS_PER_WORK_A) {
            WA[wa].layer = i; WA[wa].col0 = c0;
            ++wa;
        }
    }
    return 0;
}

int calm_sn_power_iter(const void* plan_dev, const calm_sn_plan_info* info, int32_t training, float eps,
                       float* scratch, void* stream) {
    if (!plan_dev || !info || !scratch || info->n_layers <= 0 || info->n_work <= 0 || info->n_work_a <= 0)
        return CALM_E_INVAL;
    hipStream_t s = as_stream(stream);
    const SnLayerDev* L = reinterpret_cast<const SnLayerDev*>(plan_dev);
    const SnWork* W = reinterpret_cast<const SnWork*>(L + info->n_layers);
    if (training) {
        const SnWorkA* WA = reinterpret_cast<const SnWorkA*>(W + info->n_work);
        hipLaunchKernelGGL(sn_phase_a, dim3(info->n_work_a), dim3(NT), 0, s, L, WA, scratch);
        CALM_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(sn_phase_b, dim3(info->n_work), dim3(NT), 0, s, L, W, scratch, training, eps);
    CALM_LAUNCH_CHECK();
    hipLaunchKernelGGL(sn_phase_c, dim3(info->n_layers), dim3(NT), 0, s, L, (const float*)scratch, training, eps);
    CALM_LAUNCH_CHECK();
    return 0;
}

int calm_sn_weight_bwd(const float* G, const float* w_orig, const float* u, const float* v, const float* sigma,
                       const float* ls, float* d_w_orig, float* d_ls, int32_t rows, int32_t cols, float* scratch,
                       void* stream) {
    if (!G || !w_orig || !u || !v || !sigma || !d_w_orig || !scratch || rows <= 0 || cols <= 0) return CALM_E_INVAL;
    hipStream_t s = as_stream(stream);
    int g1 = (rows + 3) / 4;
    if (g1 > 1024) g1 = 1024;
    hipLaunchKernelGGL(sn_bwd_rowdot, dim3(g1), dim3(NT), 0, s, G, w_orig, sigma, scratch, rows, cols);
    CALM_LAUNCH_CHECK();
    long total = (long)rows * cols;
    int g2 = (int)((total + NT * 4 - 1) / (NT * 4));
    if (g2 > 1024) g2 = 1024;
    if (g2 < 1) g2 = 1;
    hipLaunchKernelGGL(sn_bwd_apply, dim3(g2), dim3(NT), 0, s, G, u, v, sigma, ls, (const float*)scratch, d_w_orig,
                       d_ls, rows, cols);
    CALM_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
