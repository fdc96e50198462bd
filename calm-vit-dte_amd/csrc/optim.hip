// Optimizer-side step of the training iteration in three launches over ALL parameters
// (distributed_trainer_cls.py:88-96,158: GradScaler.unscale_ + inf check, clip_grad_norm_(1.0), AdamW(lr, betas, wd),
// zero_grad) — the reference issues one foreach/fused ATen call per stage over 521 tensors; here also the
// spectral-norm weight-gradient correction of every deferred layer is folded in (otherwise 2 launches per layer in
// backward, ~600 per step):
//     dW_orig = (G - <G, W_orig/sigma> u v^T) / sigma          G = gradient w.r.t. the normalised weight
//   pass 1  optim_stats     per chunk: sum g^2, non-finite flag; deferred layers also <G, W_orig> and u^T G v — stored
//                           as per-chunk partials (no atomics: every sum below has a fixed order, so the clip
//                           coefficient and the correction coefficients are bit-identical on all data-parallel ranks)
//   pass 2  optim_finalize  per tensor: its chunks' partials summed in chunk order; c = <G,W>/sigma,
//                           ||dW||^2 = (|G|^2 - 2c u^T G v + c^2 |u|^2 |v|^2)/sigma^2;
//                           total norm, clip coefficient = min(1, max_norm / (norm + 1e-6)), found_inf; the device
//                           step counter advances unless found_inf (torch's fused AdamW: steps -= found_inf)
//   pass 3  optim_update    g' = ((G - c u_i v_j)/sigma) * clip / grad_scale; AdamW exactly as torch.optim.AdamW:
//                           p *= 1 - lr*wd; m = b1 m + (1-b1) g'; v = b2 v + (1-b2) g'^2;
//                           p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps);   skipped entirely if found_inf
// Work item = chunk of CHUNK consecutive elements of one tensor (table built by the host).
#include "common.h"

namespace {

constexpr int NT = 256;
constexpr int CHUNK = 16384;                 // elements per work item (64 per thread)
constexpr int ST = 6;                        // per-tensor scratch: sum g^2, <G,W>, u^T G v, |u|^2, |v|^2, c
constexpr int CP = 3;                        // per-chunk partials: sum g^2, <G,W>, u^T G v

struct Globals { float total_sq, norm, clip, found_inf; };
// scratch layout: [ST x n_tensors][Globals][CP x n_chunks]

__device__ __forceinline__ bool finite_f(float x) { return fabsf(x) <= 3.402823466e38f; }   // false for inf and NaN

__global__ __launch_bounds__(NT) void optim_stats(const calm_optim_tensor* __restrict__ T, const int* __restrict__ chunk_tensor,
                                                  float* __restrict__ stats, Globals* __restrict__ G,
                                                  float* __restrict__ chunk_part) {
    __shared__ float red[4];
    const int t = chunk_tensor[blockIdx.x];
    const calm_optim_tensor e = T[t];
    const long i0 = (long)(blockIdx.x - e.chunk0) * CHUNK;
    const long i1 = min(i0 + (long)CHUNK, (long)e.numel);
    float s2 = 0.f, gw = 0.f, guv = 0.f;
    bool bad = false;
    if (e.sn_sigma) {
        for (long i = i0 + threadIdx.x; i < i1; i += NT) {
            const float g = e.grad[i];
            // rows * cols < 2^31 for a spectral-norm layer (checked by the host): 32-bit division, not the 64-bit sequence
            const unsigned r = (unsigned)i / (unsigned)e.cols, c = (unsigned)i - r * (unsigned)e.cols;
            s2 += g * g; gw += g * e.param[i]; guv += g * e.sn_u[r] * e.sn_v[c];
            bad |= !finite_f(g);
        }
    } else {
        for (long i = i0 + threadIdx.x; i < i1; i += NT) {
            const float g = e.grad[i];
            s2 += g * g;
            bad |= !finite_f(g);
        }
    }
    s2 = block_sum_256(s2, red);
    if (threadIdx.x == 0) chunk_part[CP * blockIdx.x] = s2;
    if (e.sn_sigma) {
        gw = block_sum_256(gw, red);
        guv = block_sum_256(guv, red);
        if (threadIdx.x == 0) { chunk_part[CP * blockIdx.x + 1] = gw; chunk_part[CP * blockIdx.x + 2] = guv; }
        if (i0 == 0) {                         // the tensor's first chunk also measures u and v
            float uu = 0.f, vv = 0.f;
            for (int r = threadIdx.x; r < e.rows; r += NT) uu += e.sn_u[r] * e.sn_u[r];
            for (int c = threadIdx.x; c < e.cols; c += NT) vv += e.sn_v[c] * e.sn_v[c];
            uu = block_sum_256(uu, red);
            vv = block_sum_256(vv, red);
            if (threadIdx.x == 0) { stats[ST * t + 3] = uu; stats[ST * t + 4] = vv; }
        }
    }
    if (bad) G->found_inf = 1.f;              // benign race: every writer stores the same value
}

__global__ __launch_bounds__(NT) void optim_finalize(const calm_optim_tensor* __restrict__ T, int n, float* __restrict__ stats,
                                                     Globals* __restrict__ G, float max_norm, const float* __restrict__ grad_scale,
                                                     float* __restrict__ out, const float* __restrict__ chunk_part,
                                                     int* __restrict__ step_dev) {
    __shared__ float red[4];
    float acc = 0.f;
    for (int t = threadIdx.x; t < n; t += NT) {
        const calm_optim_tensor e = T[t];
        const int nch = (int)((e.numel + CHUNK - 1) / CHUNK);
        float n2 = 0.f, gw = 0.f, guv = 0.f;
        for (int k = 0; k < nch; ++k) {                             // fixed order: chunk 0, 1, 2, ...
            const float* cp = chunk_part + (long)CP * (e.chunk0 + k);
            n2 += cp[0];
            if (e.sn_sigma) { gw += cp[1]; guv += cp[2]; }
        }
        if (e.sn_sigma) {
            const float uu = stats[ST * t + 3], vv = stats[ST * t + 4];
            const float sg = e.sn_sigma[0];
            const float c = gw / sg;                                // <G, W_orig / sigma>
            stats[ST * t + 5] = c;
            n2 = (n2 - 2.f * c * guv + c * c * uu * vv) / (sg * sg);
        }
        acc += fmaxf(n2, 0.f);
    }
    acc = block_sum_256(acc, red);
    if (threadIdx.x == 0) {
        const float inv_scale = grad_scale ? 1.0f / grad_scale[0] : 1.0f;
        const float norm = sqrtf(acc) * inv_scale;
        const bool bad = G->found_inf != 0.f || !finite_f(norm);
        float clip = max_norm > 0.f ? fminf(1.0f, max_norm / (norm + 1e-6f)) : 1.0f;
        G->total_sq = acc;
        G->norm = norm;
        G->clip = clip * inv_scale;
        G->found_inf = bad ? 1.f : 0.f;
        out[0] = norm;
        out[1] = bad ? 1.f : 0.f;
        if (step_dev && !bad) step_dev[0] += 1;      // a skipped step does not advance the bias-correction count
    }
}

__global__ __launch_bounds__(NT) void optim_update(const calm_optim_tensor* __restrict__ T, const int* __restrict__ chunk_tensor,
                                                   const float* __restrict__ stats, const Globals* __restrict__ G,
                                                   calm_optim_hparams hp, const int* __restrict__ step_dev,
                                                   const float* __restrict__ lr_dev) {
    if (G->found_inf != 0.f) return;          // the reference's scaler.step() skips optimizer.step() on inf/NaN
    if (step_dev) hp.step = step_dev[0];      // already advanced by optim_finalize for this (un-skipped) step
    if (lr_dev) hp.lr = lr_dev[0];            // a captured step follows the LR schedule through this device scalar
    const int t = chunk_tensor[blockIdx.x];
    const calm_optim_tensor e = T[t];
    const long i0 = (long)(blockIdx.x - e.chunk0) * CHUNK;
    const long i1 = min(i0 + (long)CHUNK, (long)e.numel);
    const float mul = G->clip;
    const float decay = 1.0f - hp.lr * hp.weight_decay;
    const float bc1 = 1.0f - powf(hp.beta1, (float)hp.step), bc2 = 1.0f - powf(hp.beta2, (float)hp.step);
    const float step_size = hp.lr / bc1, inv_sqrt_bc2 = 1.0f / sqrtf(bc2);
    const bool sn = e.sn_sigma != nullptr;
    const float c = sn ? stats[ST * t + 5] : 0.f, inv_sg = sn ? 1.0f / e.sn_sigma[0] : 1.f;
    for (long i = i0 + threadIdx.x; i < i1; i += NT) {
        float g = e.grad[i];
        if (sn) {
            const unsigned r = (unsigned)i / (unsigned)e.cols, cc = (unsigned)i - r * (unsigned)e.cols;
            g = (g - c * e.sn_u[r] * e.sn_v[cc]) * inv_sg;
        }
        g *= mul;
        float p = e.param[i] * decay;
        const float m = e.exp_avg[i] + (g - e.exp_avg[i]) * (1.0f - hp.beta1);          // lerp_
        const float v = e.exp_avg_sq[i] * hp.beta2 + g * g * (1.0f - hp.beta2);
        const float denom = sqrtf(v) * inv_sqrt_bc2 + hp.eps;
        p -= step_size * (m / denom);
        e.param[i] = p; e.exp_avg[i] = m; e.exp_avg_sq[i] = v;
    }
}

}  // namespace

extern "C" {

int32_t calm_optim_chunk_elems(void) { return CHUNK; }

int calm_optim_step(const calm_optim_tensor* tensors_dev, int32_t n_tensors, const int32_t* chunk_tensor_dev,
                    int32_t n_chunks, float* scratch, const calm_optim_hparams* hp, const float* grad_scale,
                    float* stats_out, int32_t* step_dev, const float* lr_dev, void* stream) {
    if (!tensors_dev || !chunk_tensor_dev || !scratch || !hp || !stats_out || n_tensors <= 0 || n_chunks <= 0)
        return CALM_E_INVAL;
    if ((!step_dev && hp->step < 1) || hp->lr < 0.f || hp->beta1 < 0.f || hp->beta1 >= 1.f || hp->beta2 < 0.f || hp->beta2 >= 1.f)
        return CALM_E_INVAL;
    hipStream_t s = as_stream(stream);
    const size_t scratch_bytes = sizeof(float) * ST * (size_t)n_tensors + sizeof(Globals);
    hipError_t e = hipMemsetAsync(scratch, 0, scratch_bytes, s);
    if (e != hipSuccess) return (int)e;
    Globals* G = reinterpret_cast<Globals*>(scratch + ST * (size_t)n_tensors);
    float* chunk_part = scratch + ST * (size_t)n_tensors + sizeof(Globals) / sizeof(float);     // every entry is written
    hipLaunchKernelGGL(optim_stats, dim3(n_chunks), dim3(NT), 0, s, tensors_dev, chunk_tensor_dev, scratch, G, chunk_part);
    CALM_LAUNCH_CHECK();
    hipLaunchKernelGGL(optim_finalize, dim3(1), dim3(NT), 0, s, tensors_dev, n_tensors, scratch, G, hp->max_norm,
                       grad_scale, stats_out, (const float*)chunk_part, step_dev);
    CALM_LAUNCH_CHECK();
    hipLaunchKernelGGL(optim_update, dim3(n_chunks), dim3(NT), 0, s, tensors_dev, chunk_tensor_dev,
                       (const float*)scratch, (const Globals*)G, *hp, (const int*)step_dev, lr_dev);
    CALM_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
