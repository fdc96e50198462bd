// Tokenisation (bit-exact index permutations) and the depthwise 3x3 stencil of Block.proj,
// all on channels-last token grids: a token tensor [B,S,3S] IS a [B,S,S,3] image.
#include "common.h"

namespace {

constexpr int NT = 256;
constexpr int MAX_BLOCKS = 2048;

inline int grid_for(int64_t work_items, int per_block) {
    int64_t g = (work_items + per_block - 1) / per_block;
    if (g > MAX_BLOCKS) g = MAX_BLOCKS;
    if (g < 1) g = 1;
    return (int)g;
}

// rows[b,i,3j+c] = img[b,c,i,j]
__global__ __launch_bounds__(NT) void image_to_rows_kernel(const float* __restrict__ img, float* __restrict__ rows,
                                                           int B, int S) {
    const long total = (long)B * S * S * 3;
    const long plane = (long)S * S;
    for (long o = (long)blockIdx.x * NT + threadIdx.x; o < total; o += (long)gridDim.x * NT) {
        const long pix = o / 3;
        const int c = (int)(o - pix * 3);
        const long b = pix / plane;
        const long ij = pix - b * plane;
        rows[o] = img[(b * 3 + c) * plane + ij];
    }
}

__global__ __launch_bounds__(NT) void rows_to_image_kernel(const float* __restrict__ rows, float* __restrict__ img,
                                                           int B, int S) {
    const long total = (long)B * S * S * 3;
    const long plane = (long)S * S;
    for (long o = (long)blockIdx.x * NT + threadIdx.x; o < total; o += (long)gridDim.x * NT) {
        const long bc = o / plane;
        const long ij = o - bc * plane;
        const long b = bc / 3;
        const int c = (int)(bc - b * 3);
        img[o] = rows[(b * plane + ij) * 3 + c];
    }
}

// out[b,j,i,:] = in[b,i,j,:]  (3 floats per pixel), 32x32-pixel LDS tiles, both sides coalesced
__global__ __launch_bounds__(NT) void grid_transpose_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                            int S) {
    __shared__ float tile[32][97];
    const int b = blockIdx.z;
    const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
    const float* src = in + (long)b * S * S * 3;
    float* dst = out + (long)b * S * S * 3;
    for (int e = threadIdx.x; e < 32 * 96; e += NT) {
        const int ii = e / 96, f = e - ii * 96;          // f = 3*jj + c
        const int i = i0 + ii, j = j0 + f / 3;
        if (i < S && j < S) tile[ii][f] = src[((long)i * S + j0) * 3 + f];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 32 * 96; e += NT) {
        const int jj = e / 96, f = e - jj * 96;          // f = 3*ii + c
        const int ii = f / 3, c = f - 3 * ii;
        const int j = j0 + jj, i = i0 + ii;
        if (i < S && j < S) dst[((long)j * S + i0) * 3 + f] = tile[ii][3 * jj + c];
    }
}

// The same with 16-byte global accesses (S % 4 == 0: every row segment of a tile starts 16-byte aligned and its valid part
// is a whole number of float4).  The scalar form above moves 3.3-3.4 TB/s (PMC, round 4) where the LayerNorm / RoPE / cast
// kernels move 5-5.9: it issues one 4-byte access per element and a workgroup has 24 KB in flight.
__global__ __launch_bounds__(NT) void grid_transpose_vec_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                                int S) {
    __shared__ float tile[32][97];
    const int b = blockIdx.z;
    const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
    const float* src = in + (long)b * S * S * 3;
    float* dst = out + (long)b * S * S * 3;
    const int row_floats = 3 * S;
    constexpr int NLD = (32 * 24) / NT;
    f32x4 v[NLD];
#pragma unroll
    for (int k = 0; k < NLD; ++k) {                        // the three requests of a thread first, then the LDS writes
        const int e = threadIdx.x + k * NT;
        const int ii = e / 24, f = 4 * (e - ii * 24);
        const bool in = i0 + ii < S && 3 * j0 + f < row_floats;
        const int ic = in ? i0 + ii : i0, fc = in ? f : 0;                  // (clamped: loaded, not stored)
        v[k] = *reinterpret_cast<const f32x4*>(src + ((long)ic * S + j0) * 3 + fc);
    }
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        const int e = threadIdx.x + k * NT;
        const int ii = e / 24, f = 4 * (e - ii * 24);
        if (i0 + ii < S && 3 * j0 + f < row_floats) {
            tile[ii][f] = v[k][0]; tile[ii][f + 1] = v[k][1]; tile[ii][f + 2] = v[k][2]; tile[ii][f + 3] = v[k][3];
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < (32 * 24) / NT; ++k) {
        const int e = threadIdx.x + k * NT;
        const int jj = e / 24, f = 4 * (e - jj * 24);
        const int j = j0 + jj;
        if (j < S && 3 * i0 + f < row_floats) {
            f32x4 o;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int ii = (f + q) / 3, c = (f + q) - 3 * ii;
                o[q] = tile[ii][3 * jj + c];
            }
            *reinterpret_cast<f32x4*>(dst + ((long)j * S + i0) * 3 + f) = o;
        }
    }
}

// depthwise 3x3, channels-last [B,S,S,C]
__global__ __launch_bounds__(NT) void dwconv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ inv_scale,
                                                        const float* __restrict__ bias, float* __restrict__ y,
                                                        float* __restrict__ y_pre, int act, int B, int S, int C) {
    const long total = (long)B * S * S * C;
    const float sc = inv_scale ? 1.0f / inv_scale[0] : 1.0f;
    for (long o = (long)blockIdx.x * NT + threadIdx.x; o < total; o += (long)gridDim.x * NT) {
        const int c = (int)(o % C);
        const long pix = o / C;
        const int xx = (int)(pix % S);
        const long t = pix / S;
        const int yy = (int)(t % S);
        const long b = t / S;
        float acc = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int sy = yy + ky - 1;
            if (sy < 0 || sy >= S) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int sx = xx + kx - 1;
                if (sx < 0 || sx >= S) continue;
                acc += w[c * 9 + ky * 3 + kx] * x[((b * S + sy) * S + sx) * C + c];
            }
        }
        float v = acc * sc + (bias ? bias[c] : 0.f);
        if (y_pre) y_pre[o] = v;
        if (act == CALM_ACT_GELU) v = gelu_erf_f(v);
        y[o] = v;
    }
}

constexpr int DW_MAXC = 64;

__global__ __launch_bounds__(NT) void dwconv_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ x,
                                                        const float* __restrict__ w,
                                                        const float* __restrict__ inv_scale, float* __restrict__ dx,
                                                        float* __restrict__ dw, float* __restrict__ db, int B, int S,
                                                        int C) {
    __shared__ float acc_s[DW_MAXC * 10];
    for (int i = threadIdx.x; i < C * 10; i += NT) acc_s[i] = 0.f;
    __syncthreads();
    const long total = (long)B * S * S * C;
    const float sc = inv_scale ? 1.0f / inv_scale[0] : 1.0f;
    // stride is a multiple of C (host guarantees NT*gridDim % C == 0), so a thread keeps its channel
    const long stride = (long)gridDim.x * NT;
    const long first = (long)blockIdx.x * NT + threadIdx.x;
    const int c = (int)(first % C);
    float wreg[9], gw[9], gb = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) { wreg[k] = w[c * 9 + k] * sc; gw[k] = 0.f; }
    for (long o = first; o < total; o += stride) {
        const long pix = o / C;
        const int xx = (int)(pix % S);
        const long t = pix / S;
        const int yy = (int)(t % S);
        const long b = t / S;
        const float g = dz[o];
        gb += g;
        float acc = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                // weight gradient: dz[y,x] * x[y+ky-1, x+kx-1]
                const int sy = yy + ky - 1, sx = xx + kx - 1;
                if (sy >= 0 && sy < S && sx >= 0 && sx < S)
                    gw[ky * 3 + kx] += g * x[((b * S + sy) * S + sx) * C + c];
                // input gradient: dx[y,x] = sum w[ky,kx] * dz[y-ky+1, x-kx+1]
                const int ty = yy - ky + 1, tx = xx - kx + 1;
                if (ty >= 0 && ty < S && tx >= 0 && tx < S)
                    acc += wreg[ky * 3 + kx] * dz[((b * S + ty) * S + tx) * C + c];
            }
        }
        dx[o] = acc;
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) atomicAdd(&acc_s[c * 10 + k], gw[k]);
    atomicAdd(&acc_s[c * 10 + 9], gb);
    __syncthreads();
    for (int i = threadIdx.x; i < C * 10; i += NT) {
        const int cc = i / 10, k = i - cc * 10;
        if (k < 9) atomicAdd(dw + cc * 9 + k, acc_s[i]);
        else atomicAdd(db + cc, acc_s[i]);
    }
}

// Device-side batch collate: ToDtype(float32, scale=True) + Normalize(mean, std) + per-sample horizontal
// flip + the batch-level CutMix / MixUp of distributed_trainer_cls.py:58-61,128-139 in one pass over uint8 images.
//   out[b,c,y,x] = w * n(img[b]) + (1 - w) * n(img[(b - 1) mod B])       n(v) = (v/255 - mean[c]) / std[c]
//   MixUp : w = lam everywhere;  CutMix: w = 0 inside the box [y1,y2) x [x1,x2), 1 outside (partner = batch rolled by 1)
// TOKENS: the output is the row-token tensor [B, H, 3W] the first Block consumes (rows[b,i,3j+c] = img[b,c,i,j],
// Vi_Tools_CNN_less_V2.py:389-391) instead of the [B,3,H,W] image; crop: per-sample top-left corner of the H x W window
// inside the Hs x Ws source (RandomCrop, distributed_trainer_cls.py:130), NULL = (0, 0).
template <bool TOKENS>
__global__ __launch_bounds__(NT) void collate_mix_kernel(const unsigned char* __restrict__ img, int Hs, int Ws,
                                                         const int* __restrict__ crop,
                                                         const unsigned char* __restrict__ flip, float* __restrict__ out,
                                                         int B, int H, int W, int mode, float lam, int y1, int y2,
                                                         int x1, int x2, float m0, float m1, float m2, float i0, float i1,
                                                         float i2) {
    const long total = (long)B * 3 * H * W;
    const long splane = (long)Hs * Ws;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        int x, y, c, b;
        if (TOKENS) {                       // i = (b, y, 3 x + c)
            c = (int)(i % 3);
            x = (int)((i / 3) % W);
            y = (int)((i / (3L * W)) % H);
            b = (int)(i / (3L * W * H));
        } else {                            // i = (b, c, y, x)
            x = (int)(i % W);
            y = (int)((i / W) % H);
            c = (int)((i / ((long)W * H)) % 3);
            b = (int)(i / ((long)W * H * 3));
        }
        const int pb = b == 0 ? B - 1 : b - 1;
        const float mean = c == 0 ? m0 : c == 1 ? m1 : m2, inv = c == 0 ? i0 : c == 1 ? i1 : i2;
        const int xs = flip && flip[b] ? W - 1 - x : x;
        const int xp = flip && flip[pb] ? W - 1 - x : x;
        const int oy = crop ? crop[2 * b] : 0, ox = crop ? crop[2 * b + 1] : 0;
        const float own = ((float)img[((long)b * 3 + c) * splane + (long)(oy + y) * Ws + ox + xs] * (1.0f / 255.0f) - mean) * inv;
        float v = own;
        if (mode != 0) {
            const int py = crop ? crop[2 * pb] : 0, px = crop ? crop[2 * pb + 1] : 0;
            const float oth =
                ((float)img[((long)pb * 3 + c) * splane + (long)(py + y) * Ws + px + xp] * (1.0f / 255.0f) - mean) * inv;
            if (mode == 1) v = own * lam + oth * (1.0f - lam);                 // MixUp: x.roll(1,0)*(1-lam) + x*lam
            else if (y >= y1 && y < y2 && x >= x1 && x < x2) v = oth;          // CutMix box
        }
        out[i] = v;
    }
}

}  // namespace

extern "C" {

int calm_collate_crop_mix(const uint8_t* img_u8, int32_t Hs, int32_t Ws, const int32_t* crop_yx, const uint8_t* flip,
                          float* out, int32_t B, int32_t H, int32_t W, int32_t out_tokens, int32_t mode, float lam,
                          const int32_t* box, const float* mean, const float* std, void* stream) {
    if (!img_u8 || !out || !mean || !std || B <= 0 || H <= 0 || W <= 0 || Hs < H || Ws < W || mode < 0 || mode > 2) return CALM_E_INVAL;
    if (mode == 2 && !box) return CALM_E_INVAL;
    if (!crop_yx && (Hs != H || Ws != W)) return CALM_E_INVAL;
    int g = grid_for((int64_t)B * 3 * H * W, NT * 4);
    const int y1 = box ? box[0] : 0, y2 = box ? box[1] : 0, x1 = box ? box[2] : 0, x2 = box ? box[3] : 0;
    if (out_tokens)
        hipLaunchKernelGGL(collate_mix_kernel<true>, dim3(g), dim3(NT), 0, as_stream(stream), img_u8, Hs, Ws, crop_yx, flip, out, B, H,
                           W, mode, lam, y1, y2, x1, x2, mean[0], mean[1], mean[2], 1.0f / std[0], 1.0f / std[1], 1.0f / std[2]);
    else
        hipLaunchKernelGGL(collate_mix_kernel<false>, dim3(g), dim3(NT), 0, as_stream(stream), img_u8, Hs, Ws, crop_yx, flip, out, B, H,
                           W, mode, lam, y1, y2, x1, x2, mean[0], mean[1], mean[2], 1.0f / std[0], 1.0f / std[1], 1.0f / std[2]);
    CALM_LAUNCH_CHECK();
    return 0;
}

int calm_collate_mix(const uint8_t* img_u8, const uint8_t* flip, float* out, int32_t B, int32_t H, int32_t W,
                     int32_t mode, float lam, const int32_t* box, const float* mean, const float* std, void* stream) {
    return calm_collate_crop_mix(img_u8, H, W, nullptr, flip, out, B, H, W, 0, mode, lam, box, mean, std, stream);
}

int calm_image_to_rows(const float* img, float* rows, int32_t B, int32_t S, void* stream) {
    if (!img || !rows || B <= 0 || S <= 0) return CALM_E_INVAL;
    hipLaunchKernelGGL(image_to_rows_kernel, dim3(grid_for((int64_t)B * S * S * 3, NT * 4)), dim3(NT), 0,
                       as_stream(stream), img, rows, B, S);
    CALM_LAUNCH_CHECK();
    return 0;
}

int calm_rows_to_image(const float* rows, float* img, int32_t B, int32_t S, void* stream) {
    if (!img || !rows || B <= 0 || S <= 0) return CALM_E_INVAL;
    hipLaunchKernelGGL(rows_to_image_kernel, dim3(grid_for((int64_t)B * S * S * 3, NT * 4)), dim3(NT), 0,
                       as_stream(stream), rows, img, B, S);
    CALM_LAUNCH_CHECK();
    return 0;
}

int calm_grid_transpose(const float* in, float* out, int32_t B, int32_t S, void* stream) {
    if (!in || !out || B <= 0 || S <= 0 || in == out) return CALM_E_INVAL;
    if (B > 65535) return CALM_E_UNSUPP;
    const int t = (S + 31) / 32;
    static_assert((32 * 24) % NT == 0, "vector tile map");
    if ((S & 3) == 0 && aligned16(in) && aligned16(out))
        hipLaunchKernelGGL(grid_transpose_vec_kernel, dim3(t, t, B), dim3(NT), 0, as_stream(stream), in, out, S);
    else
        hipLaunchKernelGGL(grid_transpose_kernel, dim3(t, t, B), dim3(NT), 0, as_stream(stream), in, out, S);
    CALM_LAUNCH_CHECK();
    return 0;
}

int calm_dwconv3x3_fwd(const float* x, const float* w, const float* inv_scale, const float* bias, float* y,
                       float* y_pre, int32_t act, int32_t B, int32_t S, int32_t C, void* stream) {
    if (!x || !w || !y || B <= 0 || S <= 0 || C <= 0) return CALM_E_INVAL;
    if (act != CALM_ACT_NONE && act != CALM_ACT_GELU) return CALM_E_UNSUPP;
    hipLaunchKernelGGL(dwconv_fwd_kernel, dim3(grid_for((int64_t)B * S * S * C, NT * 2)), dim3(NT), 0,
                       as_stream(stream), x, w, inv_scale, bias, y, y_pre, act, B, S, C);
    CALM_LAUNCH_CHECK();
    return 0;
}

int calm_dwconv3x3_bwd(const float* dz, const float* x, const float* w, const float* inv_scale, float* dx,
                       float* dw, float* db, int32_t B, int32_t S, int32_t C, void* stream) {
    if (!dz || !x || !w || !dx || !dw || !db || B <= 0 || S <= 0 || C <= 0) return CALM_E_INVAL;
    if (C > DW_MAXC || NT % C != 0) return CALM_E_UNSUPP;
    int g = grid_for((int64_t)B * S * S * C, NT * 8);
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL(dwconv_bwd_kernel, dim3(g), dim3(NT), 0, as_stream(stream), dz, x, w, inv_scale, dx, dw, db, B,
                       S, C);
    CALM_LAUNCH_CHECK();
    return 0;
}

int calm_abi_version(void) { return CALM_ABI_VERSION; }
const char* calm_build_info(void) { return "libcalmvit_hip gfx950 fp32-mfma " __DATE__; }

}  // extern "C"
