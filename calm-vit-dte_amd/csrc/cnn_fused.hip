// Fused CNN residual of Block.proj / ViT.proj (Vi_Tools_CNN_less_V2.py:378-385,400-403; CALM_ViT_V2.py:60-67,80-83)
//   out = x + conv1x1_{32->3}( gelu( dwconv3x3( gelu( conv1x1_{3->32}(x) ) ) ) )
// on the token grid viewed as a channels-last image [B,S,S,3].
//
// HBM traffic is the 3-channel input and output only (24 B/pixel forward, 36 B/pixel backward): the
// 32-channel hidden maps live in LDS for one 16x16 pixel tile (+halo) and are RECOMPUTED in backward
// instead of being stored (the unfused form wrote/read four [B,S,S,32] maps: ~1 KB/pixel).  The work is
// instruction-issue bound (measured: a cheaper erf changed nothing), so a thread owns a channel PAIR:
// 8-byte LDS accesses and packed fp32 math halve the instruction count.  The channel reductions of the two
// 1x1 convs go through an LDS image re-read with the thread = pixel mapping.
#include "common.h"

namespace {

constexpr int CH = 32;          // hidden channels of proj
constexpr int T = 16;           // tile edge (pixels)
constexpr int PS = CH + 2;      // LDS floats per pixel (even: 8-byte aligned channel pairs; 34 spreads adjacent pixels over banks)

typedef float f32x2 __attribute__((ext_vector_type(2)));

struct CnnW {
    const float* w0; const float* s0; const float* b0;   // [32,3]  (/sigma0), [32]
    const float* w2; const float* s2; const float* b2;   // [32,9]  (/sigma2), [32]
    const float* w4; const float* s4; const float* b4;   // [3,32]  (/sigma4), [3]
};

__device__ __forceinline__ f32x2 gelu2(f32x2 z) { return (f32x2){gelu_erf_f(z[0]), gelu_erf_f(z[1])}; }
__device__ __forceinline__ f32x2 gelu2_grad(f32x2 z) { return (f32x2){gelu_erf_grad_f(z[0]), gelu_erf_grad_f(z[1])}; }
__device__ __forceinline__ void gelu2_both(f32x2 z, f32x2& val, f32x2& grad) {      // one exp / rcp per element for both
    float c0, g0, c1, g1;
    gelu_parts_f(z[0], c0, g0);
    gelu_parts_f(z[1], c1, g1);
    val = (f32x2){z[0] * c0, z[1] * c1};
    grad = (f32x2){fmaf(z[0] * 0.39894228040143267794f, g0, c0), fmaf(z[1] * 0.39894228040143267794f, g1, c1)};
}
__device__ __forceinline__ f32x2 ld2(const float* p) { return *reinterpret_cast<const f32x2*>(p); }
__device__ __forceinline__ void st2(float* p, f32x2 v) { *reinterpret_cast<f32x2*>(p) = v; }

// Thread = (channel pair c2 = tid&15 -> channels 2*c2, 2*c2+1 ; pixel group g = tid>>4): every LDS access of the
// hidden maps is one 8-byte ds_read/ds_write per lane and the per-channel math is packed (v_pk_fma_f32).

// ------------------------------------------------------------------------------------------ forward
constexpr int FW_NT = 256;
constexpr int FG = FW_NT / 16;  // 16 pixel groups
constexpr int FH = T + 2;       // h1 region edge (halo 1)

// res: 1.f = out = x + proj(x) (the Block / ViT forward), 0.f = the bare proj(x)
__global__ __launch_bounds__(FW_NT) void cnn_fwd_kernel(const float* __restrict__ x, CnnW W, float* __restrict__ out,
                                                        int B, int S, int tiles_per_side, long n_tiles, float res) {
    __shared__ float xs[FH * FH * 3];
    __shared__ __attribute__((aligned(16))) float hs[FH * FH * PS];   // h1 on 18x18; later h2 on the 16x16 tile
    __shared__ float w4s[3 * CH + 3];

    const int tid = threadIdx.x, c2 = tid & 15, g = tid >> 4;
    const int c = 2 * c2;
    const float i0 = 1.0f / W.s0[0], i2 = 1.0f / W.s2[0], i4 = 1.0f / W.s4[0];
    f32x2 w0r[3], w2r[9];
#pragma unroll
    for (int i = 0; i < 3; ++i) w0r[i] = (f32x2){W.w0[c * 3 + i], W.w0[(c + 1) * 3 + i]} * i0;
#pragma unroll
    for (int k = 0; k < 9; ++k) w2r[k] = (f32x2){W.w2[c * 9 + k], W.w2[(c + 1) * 9 + k]} * i2;
    const f32x2 b0r = {W.b0[c], W.b0[c + 1]}, b2r = {W.b2[c], W.b2[c + 1]};
    if (tid < 3 * CH) w4s[tid] = W.w4[tid] * i4;
    if (tid < 3) w4s[3 * CH + tid] = W.b4[tid];

    // the input region of the NEXT tile is requested right after this tile's has been handed to LDS, so its global
    // latency runs under the GELU work instead of in front of it (stall counters: 45 % of the wave cycles waited)
    constexpr int FPRE = (FH * FH * 3 + FW_NT - 1) / FW_NT;
    float pre[FPRE];
    auto request = [&](long tile) {
        const int b = (int)(tile / (tiles_per_side * tiles_per_side));
        const int tt = (int)(tile - (long)b * tiles_per_side * tiles_per_side);
        const int y0 = (tt / tiles_per_side) * T, x0 = (tt % tiles_per_side) * T;
        const float* xb = x + (long)b * S * S * 3;
#pragma unroll
        for (int k = 0; k < FPRE; ++k) {
            const int e = tid + k * FW_NT;
            const int p = e / 3, ch = e - 3 * p;
            const int yy = y0 - 1 + p / FH, xx = x0 - 1 + p % FH;
            pre[k] = (e < FH * FH * 3 && yy >= 0 && yy < S && xx >= 0 && xx < S) ? xb[((long)yy * S + xx) * 3 + ch] : 0.f;
        }
    };
    if (blockIdx.x < n_tiles) request(blockIdx.x);
    for (long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int b = (int)(tile / (tiles_per_side * tiles_per_side));
        const int tt = (int)(tile - (long)b * tiles_per_side * tiles_per_side);
        const int y0 = (tt / tiles_per_side) * T, x0 = (tt % tiles_per_side) * T;
        __syncthreads();                                   // previous tile's LDS fully consumed
#pragma unroll
        for (int k = 0; k < FPRE; ++k)
            if (tid + k * FW_NT < FH * FH * 3) xs[tid + k * FW_NT] = pre[k];
        __syncthreads();
        if (tile + gridDim.x < n_tiles) request(tile + gridDim.x);
        // (out-of-image pixels: computed unconditionally on the zero-padded x and multiplied by 0 — a select makes the
        // compiler wrap the GELU in an exec-mask branch, which keeps two unrolled iterations from interleaving)
#pragma unroll 2
        for (int p = g; p < FH * FH; p += FG) {
            const int yy = y0 - 1 + p / FH, xx = x0 - 1 + p % FH;
            const float in = (yy >= 0 && yy < S && xx >= 0 && xx < S) ? 1.f : 0.f;
            const f32x2 z = w0r[0] * xs[3 * p] + w0r[1] * xs[3 * p + 1] + w0r[2] * xs[3 * p + 2] + b0r;
            st2(&hs[p * PS + c], gelu2(z) * in);                       // zero padding applies to the dwconv INPUT
        }
        __syncthreads();
        f32x2 h2[T * T / FG];
#pragma unroll
        for (int k = 0; k < T * T / FG; ++k) {
            const int q = g + FG * k;
            const int qy = q / T, qx = q % T;
            f32x2 acc = b2r;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) acc += w2r[ky * 3 + kx] * ld2(&hs[((qy + ky) * FH + qx + kx) * PS + c]);
            h2[k] = gelu2(acc);
        }
        __syncthreads();                                   // everyone done reading h1
#pragma unroll
        for (int k = 0; k < T * T / FG; ++k) st2(&hs[(g + FG * k) * PS + c], h2[k]);
        __syncthreads();
        {
            const int q = tid, qy = q / T, qx = q % T;
            const int yy = y0 + qy, xx = x0 + qx;
            if (yy < S && xx < S) {
                float o0 = w4s[3 * CH], o1 = w4s[3 * CH + 1], o2 = w4s[3 * CH + 2];
#pragma unroll
                for (int cc = 0; cc < CH; cc += 2) {
                    const f32x2 h = ld2(&hs[q * PS + cc]);
                    o0 += w4s[cc] * h[0] + w4s[cc + 1] * h[1];
                    o1 += w4s[CH + cc] * h[0] + w4s[CH + cc + 1] * h[1];
                    o2 += w4s[2 * CH + cc] * h[0] + w4s[2 * CH + cc + 1] * h[1];
                }
                const int pc = ((qy + 1) * FH + qx + 1) * 3;
                float* ob = out + ((long)b * S * S + (long)yy * S + xx) * 3;
                ob[0] = fmaf(res, xs[pc], o0); ob[1] = fmaf(res, xs[pc + 1], o1); ob[2] = fmaf(res, xs[pc + 2], o2);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ backward
constexpr int BW_NT = 512;
constexpr int BG = BW_NT / 16;   // 32 pixel groups
constexpr int H2 = T + 4;        // h1 region edge (halo 2) = 20
constexpr int H1 = T + 2;        // dh2p region edge (halo 1) = 18

constexpr int CNN_PART_N = 17 * CH + 3, CNN_PART_STRIDE = 560;      // one row of weight-gradient partials per workgroup
static_assert(CNN_PART_N <= CNN_PART_STRIDE, "partial row");
__global__ __launch_bounds__(BW_NT) void cnn_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, CnnW W,
                                                        float* __restrict__ dx, float* __restrict__ part, int B, int S,
                                                        int tiles_per_side, long n_tiles, float res) {
    __shared__ float xs[H2 * H2 * 3];
    __shared__ float dys[H1 * H1 * 3];
    __shared__ __attribute__((aligned(16))) float h1s[H2 * H2 * PS];   // h1 on 20x20; later dh1p on the tile
    __shared__ __attribute__((aligned(16))) float d2s[H1 * H1 * PS];   // dL/d(h2 pre-activation) on 18x18
    __shared__ float wts[3 * CH];                                       // w0 (by [c][i]) / sigma
    __shared__ float red[BG * CH];
    __shared__ __attribute__((aligned(16))) float g1s[T * T * PS];     // gelu'(h1 pre-activation) on the tile's own pixels

    const int tid = threadIdx.x, c2 = tid & 15, g = tid >> 4;
    const int c = 2 * c2;
    const float i0 = 1.0f / W.s0[0], i2 = 1.0f / W.s2[0], i4 = 1.0f / W.s4[0];
    f32x2 w0r[3], w2r[9], w4r[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        w0r[i] = (f32x2){W.w0[c * 3 + i], W.w0[(c + 1) * 3 + i]} * i0;
        w4r[i] = (f32x2){W.w4[i * CH + c], W.w4[i * CH + c + 1]} * i4;
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) w2r[k] = (f32x2){W.w2[c * 9 + k], W.w2[(c + 1) * 9 + k]} * i2;
    const f32x2 b0r = {W.b0[c], W.b0[c + 1]}, b2r = {W.b2[c], W.b2[c + 1]};
    if (tid < 3 * CH) wts[tid] = W.w0[tid] * i0;

    const f32x2 zero2 = {0.f, 0.f};
    f32x2 a_g4[3] = {zero2, zero2, zero2}, a_g2[9], a_gb2 = zero2, a_g0[3] = {zero2, zero2, zero2}, a_gb0 = zero2;
    float a_gb4[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 9; ++k) a_g2[k] = zero2;

    // next tile's x / dy regions requested right after this tile's have been handed to LDS (see the forward kernel)
    constexpr int XPRE = (H2 * H2 * 3 + BW_NT - 1) / BW_NT, DPRE = (H1 * H1 * 3 + BW_NT - 1) / BW_NT;
    float prex[XPRE], pred[DPRE];
    auto request = [&](long tile) {
        const int b = (int)(tile / (tiles_per_side * tiles_per_side));
        const int tt = (int)(tile - (long)b * tiles_per_side * tiles_per_side);
        const int y0 = (tt / tiles_per_side) * T, x0 = (tt % tiles_per_side) * T;
        const float* xb = x + (long)b * S * S * 3;
        const float* gb = dy + (long)b * S * S * 3;
#pragma unroll
        for (int k = 0; k < XPRE; ++k) {
            const int e = tid + k * BW_NT;
            const int p = e / 3, ch = e - 3 * p;
            const int yy = y0 - 2 + p / H2, xx = x0 - 2 + p % H2;
            prex[k] = (e < H2 * H2 * 3 && yy >= 0 && yy < S && xx >= 0 && xx < S) ? xb[((long)yy * S + xx) * 3 + ch] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < DPRE; ++k) {
            const int e = tid + k * BW_NT;
            const int p = e / 3, ch = e - 3 * p;
            const int yy = y0 - 1 + p / H1, xx = x0 - 1 + p % H1;
            pred[k] = (e < H1 * H1 * 3 && yy >= 0 && yy < S && xx >= 0 && xx < S) ? gb[((long)yy * S + xx) * 3 + ch] : 0.f;
        }
    };
    if (blockIdx.x < n_tiles) request(blockIdx.x);
    for (long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int b = (int)(tile / (tiles_per_side * tiles_per_side));
        const int tt = (int)(tile - (long)b * tiles_per_side * tiles_per_side);
        const int y0 = (tt / tiles_per_side) * T, x0 = (tt % tiles_per_side) * T;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < XPRE; ++k)
            if (tid + k * BW_NT < H2 * H2 * 3) xs[tid + k * BW_NT] = prex[k];
#pragma unroll
        for (int k = 0; k < DPRE; ++k)
            if (tid + k * BW_NT < H1 * H1 * 3) dys[tid + k * BW_NT] = pred[k];
        __syncthreads();
        if (tile + gridDim.x < n_tiles) request(tile + gridDim.x);
        // h1 on the 20x20 region; on the tile's own 16x16 pixels the derivative gelu'(z) as well (value and derivative
        // share their exponential): the conv0 backward below needs it, and recomputing z and a second GELU-class
        // evaluation there cost more than keeping 256 x 32 floats in LDS
#pragma unroll 2
        for (int p = g; p < H2 * H2; p += BG) {
            const int py = p / H2, px = p % H2;
            const int yy = y0 - 2 + py, xx = x0 - 2 + px;
            const float in = (yy >= 0 && yy < S && xx >= 0 && xx < S) ? 1.f : 0.f;
            const f32x2 z = w0r[0] * xs[3 * p] + w0r[1] * xs[3 * p + 1] + w0r[2] * xs[3 * p + 2] + b0r;
            f32x2 val, grad;
            gelu2_both(z, val, grad);
            st2(&h1s[p * PS + c], val * in);
            if (py >= 2 && py < T + 2 && px >= 2 && px < T + 2) st2(&g1s[((py - 2) * T + px - 2) * PS + c], grad);
        }
        __syncthreads();
        // dL/dh2p on the 18x18 region (+ weight grads of conv4 / dwconv on the tile's own pixels)
#pragma unroll 2
        for (int r = g; r < H1 * H1; r += BG) {
            const int ry = r / H1, rx = r % H1;
            const int yy = y0 - 1 + ry, xx = x0 - 1 + rx;
            const bool in = yy >= 0 && yy < S && xx >= 0 && xx < S;
            f32x2 h1v[9];
            f32x2 z = b2r;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    h1v[ky * 3 + kx] = ld2(&h1s[((ry + ky) * H2 + rx + kx) * PS + c]);
                    z += w2r[ky * 3 + kx] * h1v[ky * 3 + kx];
                }
            const float d0 = dys[3 * r], d1 = dys[3 * r + 1], d2 = dys[3 * r + 2];
            const f32x2 dh2 = w4r[0] * d0 + w4r[1] * d1 + w4r[2] * d2;
            f32x2 h2, dgelu;
            gelu2_both(z, h2, dgelu);
            const f32x2 dz = dh2 * dgelu;                  // dy is zero outside the image, hence so is dh2
            st2(&d2s[r * PS + c], dz);
            // weight gradients only from the tile's own pixels: a 0/1 factor instead of a branch
            const float own = (in && ry >= 1 && ry <= T && rx >= 1 && rx <= T) ? 1.f : 0.f;
            const f32x2 h2o = h2 * own, dzo = dz * own;
            a_g4[0] += h2o * d0; a_g4[1] += h2o * d1; a_g4[2] += h2o * d2;
            a_gb2 += dzo;
#pragma unroll
            for (int k = 0; k < 9; ++k) a_g2[k] += dzo * h1v[k];
        }
        __syncthreads();                                   // d2s complete, h1s no longer read
        // dL/dh1p on the tile; staged over the h1 image for the channel reduction of conv0^T
#pragma unroll 2
        for (int q = g; q < T * T; q += BG) {
            const int qy = q / T, qx = q % T;
            const int yy = y0 + qy, xx = x0 + qx;
            const float in = (yy < S && xx < S) ? 1.f : 0.f;
            f32x2 dh1 = zero2;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
                    dh1 += w2r[ky * 3 + kx] * ld2(&d2s[((qy + 2 - ky) * H1 + qx + 2 - kx) * PS + c]);
            const int px = ((qy + 2) * H2 + qx + 2) * 3;
            const float x0v = xs[px], x1v = xs[px + 1], x2v = xs[px + 2];
            const f32x2 dz = dh1 * ld2(&g1s[q * PS + c]) * in;
            st2(&h1s[q * PS + c], dz);
            a_gb0 += dz;
            a_g0[0] += dz * x0v; a_g0[1] += dz * x1v; a_g0[2] += dz * x2v;
        }
        __syncthreads();
        if (tid < T * T) {
            const int q = tid, qy = q / T, qx = q % T;
            const int yy = y0 + qy, xx = x0 + qx;
            if (yy < S && xx < S) {
                const int pd = ((qy + 1) * H1 + qx + 1) * 3;
                float o0 = dys[pd], o1 = dys[pd + 1], o2 = dys[pd + 2];
                a_gb4[0] += o0; a_gb4[1] += o1; a_gb4[2] += o2;
                o0 *= res; o1 *= res; o2 *= res;                       // the skip connection's share of dx
#pragma unroll
                for (int cc = 0; cc < CH; cc += 2) {
                    const f32x2 d = ld2(&h1s[q * PS + cc]);
                    o0 += wts[cc * 3] * d[0] + wts[cc * 3 + 3] * d[1];
                    o1 += wts[cc * 3 + 1] * d[0] + wts[cc * 3 + 4] * d[1];
                    o2 += wts[cc * 3 + 2] * d[0] + wts[cc * 3 + 5] * d[1];
                }
                float* ob = dx + ((long)b * S * S + (long)yy * S + xx) * 3;
                ob[0] = o0; ob[1] = o1; ob[2] = o2;
            }
        }
    }
    // ---- block reduction of the weight-gradient accumulators (fixed order) into the block's row of partials:
    //      [g0 3CH | gb0 CH | g2 9CH | gb2 CH | g4 3CH | gb4 3], combined over the blocks by calm_reduce_partials ----
    float* const prow = part + (long)blockIdx.x * CNN_PART_STRIDE;
    float* const g0 = prow, * const gb0 = prow + 3 * CH, * const g2 = prow + 4 * CH, * const gb2 = prow + 13 * CH,
               * const g4 = prow + 14 * CH, * const gb4 = prow + 17 * CH;
    auto reduce_c = [&](f32x2 v, float* dst0, int stride) {   // sum over the 32 pixel groups for channels c, c+1
        __syncthreads();
        red[g * CH + c] = v[0];
        red[g * CH + c + 1] = v[1];
        __syncthreads();
        if (tid < CH) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < BG; ++k) s += red[k * CH + tid];
            dst0[tid * stride] = s;
        }
    };
#pragma unroll
    for (int o = 0; o < 3; ++o) reduce_c(a_g4[o], g4 + o * CH, 1);
#pragma unroll
    for (int k = 0; k < 9; ++k) reduce_c(a_g2[k], g2 + k, 9);
    reduce_c(a_gb2, gb2, 1);
#pragma unroll
    for (int i = 0; i < 3; ++i) reduce_c(a_g0[i], g0 + i, 3);
    reduce_c(a_gb0, gb0, 1);
#pragma unroll
    for (int o = 0; o < 3; ++o) {
        __syncthreads();
        float v = wave_sum(a_gb4[o]);
        if ((tid & 63) == 0) red[tid >> 6] = v;
        __syncthreads();
        if (tid == 0) {
            float s = 0.f;
            for (int k = 0; k < BW_NT / 64; ++k) s += red[k];
            gb4[o] = s;
        }
    }
}

}  // namespace

extern "C" {

int calm_cnn_residual_fwd(const float* x, const float* w0, const float* s0, const float* b0, const float* w2,
                          const float* s2, const float* b2, const float* w4, const float* s4, const float* b4,
                          float* out, int32_t B, int32_t S, int32_t hidden, int32_t residual, void* stream) {
    if (!x || !w0 || !s0 || !b0 || !w2 || !s2 || !b2 || !w4 || !s4 || !b4 || !out || B <= 0 || S <= 0)
        return CALM_E_INVAL;
    if (hidden != CH) return CALM_E_UNSUPP;
    const int tps = (S + T - 1) / T;
    const long n_tiles = (long)B * tps * tps;
    const int grid = (int)(n_tiles < 256 * 6 ? n_tiles : 256 * 6);
    CnnW W{w0, s0, b0, w2, s2, b2, w4, s4, b4};
    hipLaunchKernelGGL(cnn_fwd_kernel, dim3(grid), dim3(FW_NT), 0, as_stream(stream), x, W, out, B, S, tps, n_tiles,
                       residual ? 1.f : 0.f);
    CALM_LAUNCH_CHECK();
    return 0;
}

int calm_cnn_residual_bwd(const float* dy, const float* x, const float* w0, const float* s0, const float* b0,
                          const float* w2, const float* s2, const float* b2, const float* w4, const float* s4,
                          const float* b4, float* dx, float* g0, float* gb0, float* g2, float* gb2, float* g4,
                          float* gb4, int32_t B, int32_t S, int32_t hidden, int32_t residual, float* partials,
                          void* stream) {
    if (!dy || !x || !w0 || !s0 || !b0 || !w2 || !s2 || !b2 || !w4 || !s4 || !b4 || !dx || !g0 || !gb0 || !g2 ||
        !gb2 || !g4 || !gb4 || !partials || B <= 0 || S <= 0)
        return CALM_E_INVAL;
    if (hidden != CH) return CALM_E_UNSUPP;
    const int tps = (S + T - 1) / T;
    const long n_tiles = (long)B * tps * tps;
    const int grid = (int)(n_tiles < 256 ? n_tiles : 256);
    CnnW W{w0, s0, b0, w2, s2, b2, w4, s4, b4};
    hipLaunchKernelGGL(cnn_bwd_kernel, dim3(grid), dim3(BW_NT), 0, as_stream(stream), dy, x, W, dx, partials, B, S, tps,
                       n_tiles, residual ? 1.f : 0.f);
    CALM_LAUNCH_CHECK();
    CalmReduceDst d{};
    d.out[0] = g0; d.out[1] = gb0; d.out[2] = g2; d.out[3] = gb2; d.out[4] = g4; d.out[5] = gb4;
    d.begin[0] = 0; d.begin[1] = 3 * CH; d.begin[2] = 4 * CH; d.begin[3] = 13 * CH; d.begin[4] = 14 * CH;
    d.begin[5] = 17 * CH; d.begin[6] = CNN_PART_N; d.nseg = 6;
    hipLaunchKernelGGL(calm_reduce_partials_kernel, dim3((CNN_PART_N + 63) / 64), dim3(CALM_RED_THREADS), 0, as_stream(stream),
                       partials, grid, CNN_PART_N, CNN_PART_STRIDE, d);
    CALM_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
