// Fused CNN residual of Block.proj / ViT.proj (Vi_Tools_CNN_less_V2.py:378-385,400-403; CALM_ViT_V2.py:60-67,80-83)
//   out = x + conv1x1_{32->3}( gelu( dwconv3x3( gelu( conv1x1_{3->32}(x) ) ) ) )
// on the token grid viewed as a channels-last image [B,S,S,3].
//
// HBM traffic is the 3-channel input and output only (24 B/pixel forward, 36 B/pixel backward): the
// 32-channel hidden maps live in LDS for one 16x16 pixel tile (+halo) and are RECOMPUTED in backward
// instead of being stored (the unfused form wrote/read four [B,S,S,32] maps: ~1 KB/pixel).  The work is
// VALU (erf-GELU) bound; thread = (channel c = tid&31, pixel group g = tid>>5), so LDS accesses of a
// 32-lane half are 32 consecutive floats (conflict-free), and the channel reductions of the two 1x1 convs
// go through an LDS image re-read with the thread = pixel mapping.
#include "common.h"

namespace {

constexpr int CH = 32;          // hidden channels of proj
constexpr int T = 16;           // tile edge (pixels)

struct CnnW {
    const float* w0; const float* s0; const float* b0;   // [32,3]  (/sigma0), [32]
    const float* w2; const float* s2; const float* b2;   // [32,9]  (/sigma2), [32]
    const float* w4; const float* s4; const float* b4;   // [3,32]  (/sigma4), [3]
};

// ------------------------------------------------------------------------------------------ forward
constexpr int FW_NT = 256;
constexpr int FH = T + 2;       // h1 region edge (halo 1)

__global__ __launch_bounds__(FW_NT) void cnn_fwd_kernel(const float* __restrict__ x, CnnW W, float* __restrict__ out,
                                                        int B, int S, int tiles_per_side, long n_tiles) {
    __shared__ float xs[FH * FH * 3];
    __shared__ float hs[FH * FH * CH];        // h1 on the 18x18 region; later h2 on the 16x16 tile, stride 33
    __shared__ float w4s[3 * CH + 3];

    const int tid = threadIdx.x, c = tid & 31, g = tid >> 5;
    const float i0 = 1.0f / W.s0[0], i2 = 1.0f / W.s2[0], i4 = 1.0f / W.s4[0];
    float w0r[3], w2r[9];
#pragma unroll
    for (int i = 0; i < 3; ++i) w0r[i] = W.w0[c * 3 + i] * i0;
#pragma unroll
    for (int k = 0; k < 9; ++k) w2r[k] = W.w2[c * 9 + k] * i2;
    const float b0r = W.b0[c], b2r = W.b2[c];
    if (tid < 3 * CH) w4s[tid] = W.w4[tid] * i4;
    if (tid < 3) w4s[3 * CH + tid] = W.b4[tid];

    for (long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int b = (int)(tile / (tiles_per_side * tiles_per_side));
        const int tt = (int)(tile - (long)b * tiles_per_side * tiles_per_side);
        const int y0 = (tt / tiles_per_side) * T, x0 = (tt % tiles_per_side) * T;
        const float* xb = x + (long)b * S * S * 3;
        __syncthreads();                                   // previous tile's LDS fully consumed
        for (int e = tid; e < FH * FH * 3; e += FW_NT) {
            const int p = e / 3, ch = e - 3 * p;
            const int yy = y0 - 1 + p / FH, xx = x0 - 1 + p % FH;
            xs[e] = (yy >= 0 && yy < S && xx >= 0 && xx < S) ? xb[((long)yy * S + xx) * 3 + ch] : 0.f;
        }
        __syncthreads();
        for (int p = g; p < FH * FH; p += FW_NT / 32) {
            const int yy = y0 - 1 + p / FH, xx = x0 - 1 + p % FH;
            const bool in = yy >= 0 && yy < S && xx >= 0 && xx < S;
            const float z = w0r[0] * xs[3 * p] + w0r[1] * xs[3 * p + 1] + w0r[2] * xs[3 * p + 2] + b0r;
            hs[p * CH + c] = in ? gelu_erf_f(z) : 0.f;   // zero padding applies to the dwconv INPUT
        }
        __syncthreads();
        float h2[T * T / (FW_NT / 32)];
#pragma unroll
        for (int k = 0; k < T * T / (FW_NT / 32); ++k) {
            const int q = g + (FW_NT / 32) * k;
            const int qy = q / T, qx = q % T;
            float acc = b2r;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) acc += w2r[ky * 3 + kx] * hs[((qy + ky) * FH + qx + kx) * CH + c];
            h2[k] = gelu_erf_f(acc);
        }
        __syncthreads();                                   // everyone done reading h1
#pragma unroll
        for (int k = 0; k < T * T / (FW_NT / 32); ++k) hs[(g + (FW_NT / 32) * k) * (CH + 1) + c] = h2[k];
        __syncthreads();
        {
            const int q = tid, qy = q / T, qx = q % T;
            const int yy = y0 + qy, xx = x0 + qx;
            if (yy < S && xx < S) {
                float o0 = w4s[3 * CH], o1 = w4s[3 * CH + 1], o2 = w4s[3 * CH + 2];
#pragma unroll
                for (int cc = 0; cc < CH; ++cc) {
                    const float h = hs[q * (CH + 1) + cc];
                    o0 += w4s[cc] * h; o1 += w4s[CH + cc] * h; o2 += w4s[2 * CH + cc] * h;
                }
                const int pc = ((qy + 1) * FH + qx + 1) * 3;
                float* ob = out + ((long)b * S * S + (long)yy * S + xx) * 3;
                ob[0] = o0 + xs[pc]; ob[1] = o1 + xs[pc + 1]; ob[2] = o2 + xs[pc + 2];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ backward
constexpr int BW_NT = 512;
constexpr int BG = BW_NT / 32;   // 16 pixel groups
constexpr int H2 = T + 4;        // h1 region edge (halo 2) = 20
constexpr int H1 = T + 2;        // dh2p region edge (halo 1) = 18

__global__ __launch_bounds__(BW_NT) void cnn_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, CnnW W,
                                                        float* __restrict__ dx, float* __restrict__ g0,
                                                        float* __restrict__ gb0, float* __restrict__ g2,
                                                        float* __restrict__ gb2, float* __restrict__ g4,
                                                        float* __restrict__ gb4, int B, int S, int tiles_per_side,
                                                        long n_tiles) {
    __shared__ float xs[H2 * H2 * 3];
    __shared__ float dys[H1 * H1 * 3];
    __shared__ float h1s[H2 * H2 * CH];       // h1 on 20x20; later dh1p on the tile with stride 33
    __shared__ float d2s[H1 * H1 * CH];       // dL/d(h2 pre-activation) on 18x18
    __shared__ float wts[3 * CH * 2];         // w4 (by [o][c]) and w0 (by [c][i]) / sigma
    __shared__ float red[BG * CH];

    const int tid = threadIdx.x, c = tid & 31, g = tid >> 5;
    const float i0 = 1.0f / W.s0[0], i2 = 1.0f / W.s2[0], i4 = 1.0f / W.s4[0];
    float w0r[3], w2r[9], w4r[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { w0r[i] = W.w0[c * 3 + i] * i0; w4r[i] = W.w4[i * CH + c] * i4; }
#pragma unroll
    for (int k = 0; k < 9; ++k) w2r[k] = W.w2[c * 9 + k] * i2;
    const float b0r = W.b0[c], b2r = W.b2[c];
    if (tid < 3 * CH) { wts[tid] = W.w4[tid] * i4; wts[3 * CH + tid] = W.w0[tid] * i0; }

    float a_g4[3] = {0.f, 0.f, 0.f}, a_g2[9], a_gb2 = 0.f, a_g0[3] = {0.f, 0.f, 0.f}, a_gb0 = 0.f;
    float a_gb4[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 9; ++k) a_g2[k] = 0.f;

    for (long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int b = (int)(tile / (tiles_per_side * tiles_per_side));
        const int tt = (int)(tile - (long)b * tiles_per_side * tiles_per_side);
        const int y0 = (tt / tiles_per_side) * T, x0 = (tt % tiles_per_side) * T;
        const float* xb = x + (long)b * S * S * 3;
        const float* gb = dy + (long)b * S * S * 3;
        __syncthreads();
        for (int e = tid; e < H2 * H2 * 3; e += BW_NT) {
            const int p = e / 3, ch = e - 3 * p;
            const int yy = y0 - 2 + p / H2, xx = x0 - 2 + p % H2;
            xs[e] = (yy >= 0 && yy < S && xx >= 0 && xx < S) ? xb[((long)yy * S + xx) * 3 + ch] : 0.f;
        }
        for (int e = tid; e < H1 * H1 * 3; e += BW_NT) {
            const int p = e / 3, ch = e - 3 * p;
            const int yy = y0 - 1 + p / H1, xx = x0 - 1 + p % H1;
            dys[e] = (yy >= 0 && yy < S && xx >= 0 && xx < S) ? gb[((long)yy * S + xx) * 3 + ch] : 0.f;
        }
        __syncthreads();
        // h1 on the 20x20 region
        for (int p = g; p < H2 * H2; p += BG) {
            const int yy = y0 - 2 + p / H2, xx = x0 - 2 + p % H2;
            const bool in = yy >= 0 && yy < S && xx >= 0 && xx < S;
            const float z = w0r[0] * xs[3 * p] + w0r[1] * xs[3 * p + 1] + w0r[2] * xs[3 * p + 2] + b0r;
            h1s[p * CH + c] = in ? gelu_erf_f(z) : 0.f;
        }
        __syncthreads();
        // dL/dh2p on the 18x18 region (+ weight grads of conv4 / dwconv on the tile's own pixels)
        for (int r = g; r < H1 * H1; r += BG) {
            const int ry = r / H1, rx = r % H1;
            const int yy = y0 - 1 + ry, xx = x0 - 1 + rx;
            const bool in = yy >= 0 && yy < S && xx >= 0 && xx < S;
            float h1v[9];
            float z = b2r;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    h1v[ky * 3 + kx] = h1s[((ry + ky) * H2 + rx + kx) * CH + c];
                    z += w2r[ky * 3 + kx] * h1v[ky * 3 + kx];
                }
            const float d0 = dys[3 * r], d1 = dys[3 * r + 1], d2 = dys[3 * r + 2];
            const float dh2 = w4r[0] * d0 + w4r[1] * d1 + w4r[2] * d2;
            const float dz = in ? dh2 * gelu_erf_grad_f(z) : 0.f;
            d2s[r * CH + c] = dz;
            const bool own = in && ry >= 1 && ry <= T && rx >= 1 && rx <= T;
            if (own) {
                const float h2 = gelu_erf_f(z);
                a_g4[0] += d0 * h2; a_g4[1] += d1 * h2; a_g4[2] += d2 * h2;
                a_gb2 += dz;
#pragma unroll
                for (int k = 0; k < 9; ++k) a_g2[k] += dz * h1v[k];
            }
        }
        __syncthreads();                                   // d2s complete, h1s no longer read
        // dL/dh1p on the tile; staged (stride 33) over the h1 image for the channel reduction of conv0^T
        for (int q = g; q < T * T; q += BG) {
            const int qy = q / T, qx = q % T;
            const int yy = y0 + qy, xx = x0 + qx;
            const bool in = yy < S && xx < S;
            float dh1 = 0.f;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
                    dh1 += w2r[ky * 3 + kx] * d2s[((qy + 2 - ky) * H1 + qx + 2 - kx) * CH + c];
            const int px = ((qy + 2) * H2 + qx + 2) * 3;
            const float x0v = xs[px], x1v = xs[px + 1], x2v = xs[px + 2];
            const float z = w0r[0] * x0v + w0r[1] * x1v + w0r[2] * x2v + b0r;
            const float dz = in ? dh1 * gelu_erf_grad_f(z) : 0.f;
            h1s[q * (CH + 1) + c] = dz;
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
#pragma unroll
                for (int cc = 0; cc < CH; ++cc) {
                    const float d = h1s[q * (CH + 1) + cc];
                    o0 += wts[3 * CH + cc * 3] * d; o1 += wts[3 * CH + cc * 3 + 1] * d; o2 += wts[3 * CH + cc * 3 + 2] * d;
                }
                float* ob = dx + ((long)b * S * S + (long)yy * S + xx) * 3;
                ob[0] = o0; ob[1] = o1; ob[2] = o2;
            }
        }
    }
    // ---- block reduction of the weight-gradient accumulators, one atomic per element per block ----
    auto reduce_c = [&](float v, float* dst) {             // sum over the 16 pixel groups for channel c
        __syncthreads();
        red[g * CH + c] = v;
        __syncthreads();
        if (g == 0) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < BG; ++k) s += red[k * CH + c];
            atomicAdd(dst, s);
        }
    };
#pragma unroll
    for (int o = 0; o < 3; ++o) reduce_c(a_g4[o], g4 + o * CH + c);
#pragma unroll
    for (int k = 0; k < 9; ++k) reduce_c(a_g2[k], g2 + c * 9 + k);
    reduce_c(a_gb2, gb2 + c);
#pragma unroll
    for (int i = 0; i < 3; ++i) reduce_c(a_g0[i], g0 + c * 3 + i);
    reduce_c(a_gb0, gb0 + c);
#pragma unroll
    for (int o = 0; o < 3; ++o) {
        __syncthreads();
        float v = wave_sum(a_gb4[o]);
        if ((tid & 63) == 0) red[tid >> 6] = v;
        __syncthreads();
        if (tid == 0) {
            float s = 0.f;
            for (int k = 0; k < BW_NT / 64; ++k) s += red[k];
            atomicAdd(gb4 + o, s);
        }
    }
}

}  // namespace

extern "C" {

int calm_cnn_residual_fwd(const float* x, const float* w0, const float* s0, const float* b0, const float* w2,
                          const float* s2, const float* b2, const float* w4, const float* s4, const float* b4,
                          float* out, int32_t B, int32_t S, int32_t hidden, void* stream) {
    if (!x || !w0 || !s0 || !b0 || !w2 || !s2 || !b2 || !w4 || !s4 || !b4 || !out || B <= 0 || S <= 0)
        return CALM_E_INVAL;
    if (hidden != CH) return CALM_E_UNSUPP;
    const int tps = (S + T - 1) / T;
    const long n_tiles = (long)B * tps * tps;
    const int grid = (int)(n_tiles < 256 * 6 ? n_tiles : 256 * 6);
    CnnW W{w0, s0, b0, w2, s2, b2, w4, s4, b4};
    hipLaunchKernelGGL(cnn_fwd_kernel, dim3(grid), dim3(FW_NT), 0, as_stream(stream), x, W, out, B, S, tps, n_tiles);
    CALM_LAUNCH_CHECK();
    return 0;
}

int calm_cnn_residual_bwd(const float* dy, const float* x, const float* w0, const float* s0, const float* b0,
                          const float* w2, const float* s2, const float* b2, const float* w4, const float* s4,
                          const float* b4, float* dx, float* g0, float* gb0, float* g2, float* gb2, float* g4,
                          float* gb4, int32_t B, int32_t S, int32_t hidden, void* stream) {
    if (!dy || !x || !w0 || !s0 || !b0 || !w2 || !s2 || !b2 || !w4 || !s4 || !b4 || !dx || !g0 || !gb0 || !g2 ||
        !gb2 || !g4 || !gb4 || B <= 0 || S <= 0)
        return CALM_E_INVAL;
    if (hidden != CH) return CALM_E_UNSUPP;
    const int tps = (S + T - 1) / T;
    const long n_tiles = (long)B * tps * tps;
    const int grid = (int)(n_tiles < 256 ? n_tiles : 256);
    CnnW W{w0, s0, b0, w2, s2, b2, w4, s4, b4};
    hipLaunchKernelGGL(cnn_bwd_kernel, dim3(grid), dim3(BW_NT), 0, as_stream(stream), dy, x, W, dx, g0, gb0, g2, gb2,
                       g4, gb4, B, S, tps, n_tiles);
    CALM_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
