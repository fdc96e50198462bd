// HBM-bound streaming kernels of the CALM-ViT path: LayerNorm, learned RoPE + head assembly,
// row softmax, latent sampling/KL, and small helpers.  All are one-wave-per-row or grid-stride
// kernels with coalesced (16-byte where alignment allows) accesses and wave shuffles for the
// row reductions; cross-row reductions (dw, d_inv_freq, kl, bias grads) are summed per block in registers / LDS in a
// fixed order, leave the block as one row of caller-provided scratch and are combined by calm_reduce_partials
// (common.h) in workgroup order: no atomics, every result repeats bit for bit (ABI v7).
#include "common.h"

namespace {

constexpr int NT = 256;
constexpr int MAX_BLOCKS = 2048;   // 256 CUs x 8 blocks

inline int grid_for(int64_t work_items, int per_block) {
    int64_t g = (work_items + per_block - 1) / per_block;
    if (g > MAX_BLOCKS) g = MAX_BLOCKS;
    if (g < 1) g = 1;
    return (int)g;
}

// ------------------------------------------------------------------ LayerNorm
// one wave per row, rows grid-strided over waves
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
// Y16 / G16: the normalised output / the incoming gradient is a bf16 tensor (bf16 pipeline: the LayerNorm output only
// feeds GEMMs, which round their operands to bf16 anyway — it is rounded once, here, when stored)
template <bool Y16>
__global__ __launch_bounds__(NT) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                    void* __restrict__ y_, float* __restrict__ mean,
                                                    float* __restrict__ rstd, long rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * (NT / 64);
    for (long row = wave; row < rows; row += nwaves) {
        const float* xr = x + row * D;
        float s = 0.f;
        for (int c = lane; c < D; c += 64) s += xr[c];
        const float mu = wave_sum(s) / D;
        float q = 0.f;
        for (int c = lane; c < D; c += 64) { const float d = xr[c] - mu; q += d * d; }
        const float rs = rsqrtf(wave_sum(q) / D + eps);
        for (int c = lane; c < D; c += 64) {
            const float o = (xr[c] - mu) * rs * w[c];
            if constexpr (Y16) reinterpret_cast<__bf16*>(y_)[row * D + c] = (__bf16)o;
            else reinterpret_cast<float*>(y_)[row * D + c] = o;
        }
        if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
    }
}

constexpr int LN_MAXC = 32;   // columns per lane kept in registers for dw (D <= 2048)

template <bool G16>
__global__ __launch_bounds__(NT) void ln_bwd_kernel(const void* __restrict__ dy_, const float* __restrict__ x,
                                                    const float* __restrict__ w, const float* __restrict__ mean,
                                                    const float* __restrict__ rstd, float* __restrict__ dx,
                                                    float* __restrict__ dw_part, const float* __restrict__ dx_add,
                                                    long rows, int D) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * (NT / 64);
    float dwacc[LN_MAXC];
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) dwacc[i] = 0.f;
    for (long row = wave; row < rows; row += nwaves) {
        const float* xr = x + row * D;
        auto gr = [&](int c) -> float {
            if constexpr (G16) return (float)reinterpret_cast<const __bf16*>(dy_)[row * D + c];
            else return reinterpret_cast<const float*>(dy_)[row * D + c];
        };
        const float mu = mean[row], rs = rstd[row];
        float c1 = 0.f, c2 = 0.f;
        for (int c = lane; c < D; c += 64) {
            const float g = gr(c) * w[c];
            const float xh = (xr[c] - mu) * rs;
            c1 += g; c2 += g * xh;
        }
        c1 = wave_sum(c1) / D; c2 = wave_sum(c2) / D;
        float* dr = dx + row * D;
#pragma unroll
        for (int i = 0; i < LN_MAXC; ++i) {
            const int c = lane + 64 * i;
            if (c < D) {
                const float xh = (xr[c] - mu) * rs;
                const float gy = gr(c);
                dr[c] = rs * (gy * w[c] - c1 - xh * c2) + (dx_add ? dx_add[row * D + c] : 0.f);
                dwacc[i] += gy * xh;
            }
        }
    }
    __shared__ float red[(NT / 64) * 64 * LN_MAXC];
    const int wv_id = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) red[wv_id * (64 * LN_MAXC) + lane + 64 * i] = dwacc[i];
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += NT) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NT / 64; ++k) s += red[k * (64 * LN_MAXC) + c];
        dw_part[(long)blockIdx.x * D + c] = s;
    }
}

// Vectorised variants: the whole row lives in registers as NV float4 per lane (D <= 256*NV, D % 4 == 0),
// one HBM pass per tensor, 16-byte coalesced accesses.
template <int NV, bool Y16>
__global__ __launch_bounds__(NT) void ln_fwd_vec_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        void* __restrict__ y_, float* __restrict__ mean,
                                                        float* __restrict__ rstd, long rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * (NT / 64);
    const int nv4 = D >> 2;
    f32x4 wv[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c4 = lane + 64 * i;
        wv[i] = c4 < nv4 ? reinterpret_cast<const f32x4*>(w)[c4] : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    for (long row = wave; row < rows; row += nwaves) {
        const f32x4* xr = reinterpret_cast<const f32x4*>(x + row * D);
        f32x4 xv[NV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c4 = lane + 64 * i;
            xv[i] = c4 < nv4 ? xr[c4] : (f32x4){0.f, 0.f, 0.f, 0.f};
            s += xv[i][0] + xv[i][1] + xv[i][2] + xv[i][3];
        }
        const float mu = wave_sum(s) / D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (lane + 64 * i < nv4) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = xv[i][e] - mu; q += d * d; }
            }
        }
        const float rs = rsqrtf(wave_sum(q) / D + eps);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c4 = lane + 64 * i;
            if (c4 < nv4) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (xv[i][e] - mu) * rs * wv[i][e];
                if constexpr (Y16) {
                    bf16x4 ob = {(__bf16)o[0], (__bf16)o[1], (__bf16)o[2], (__bf16)o[3]};
                    reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(y_) + row * D)[c4] = ob;
                } else {
                    reinterpret_cast<f32x4*>(reinterpret_cast<float*>(y_) + row * D)[c4] = o;
                }
            }
        }
        if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
    }
}

template <int NV, bool G16>
__global__ __launch_bounds__(NT) void ln_bwd_vec_kernel(const void* __restrict__ dy_, const float* __restrict__ x,
                                                        const float* __restrict__ w, const float* __restrict__ mean,
                                                        const float* __restrict__ rstd, float* __restrict__ dx,
                                                        float* __restrict__ dw_part, const float* __restrict__ dx_add,
                                                        long rows, int D) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * (NT / 64);
    const int nv4 = D >> 2;
    f32x4 wv[NV], dwacc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c4 = lane + 64 * i;
        wv[i] = c4 < nv4 ? reinterpret_cast<const f32x4*>(w)[c4] : (f32x4){0.f, 0.f, 0.f, 0.f};
        dwacc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    for (long row = wave; row < rows; row += nwaves) {
        const f32x4* xr = reinterpret_cast<const f32x4*>(x + row * D);
        const float mu = mean[row], rs = rstd[row];
        f32x4 xh[NV], gv[NV], av[NV];
        float c1 = 0.f, c2 = 0.f;
        // Every load of the row is issued before anything is converted or masked: chunk indices past the row are CLAMPED
        // (loaded, then zeroed by a select) instead of branched around, and bf16 gradients are converted after the last
        // request — with the conversion inside a masked branch the compiler put `s_waitcnt vmcnt(0)` behind each of the NV
        // gradient loads (ISA, round 4): three exposed round trips per row on top of the row's own, which is why the bf16
        // variant moved 4.9 TB/s where the fp32 one moves 5.7.  The skip-connection gradient is requested together with x
        // and dy: one memory latency per row, not a second one behind the two wave reductions.
        const f32x4* ar = dx_add ? reinterpret_cast<const f32x4*>(dx_add + row * D) : nullptr;
        f32x4 xv[NV];
        bf16x4 graw[G16 ? NV : 1];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c4 = lane + 64 * i, cc = c4 < nv4 ? c4 : nv4 - 1;
            xv[i] = xr[cc];
            if constexpr (G16) graw[i] = reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(dy_) + row * D)[cc];
            else gv[i] = reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(dy_) + row * D)[cc];
            av[i] = ar ? ar[cc] : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c4 = lane + 64 * i;
            const bool in = c4 < nv4;
            if constexpr (G16) gv[i] = (f32x4){(float)graw[i][0], (float)graw[i][1], (float)graw[i][2], (float)graw[i][3]};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                gv[i][e] = in ? gv[i][e] : 0.f;
                xh[i][e] = in ? (xv[i][e] - mu) * rs : 0.f;
                const float g = gv[i][e] * wv[i][e];
                c1 += g; c2 += g * xh[i][e];
            }
        }
        c1 = wave_sum(c1) / D; c2 = wave_sum(c2) / D;
        f32x4* dr = reinterpret_cast<f32x4*>(dx + row * D);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c4 = lane + 64 * i;
            if (c4 < nv4) {
                f32x4 o = av[i];                                           // gradient of the skip connection around LN
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] += rs * (gv[i][e] * wv[i][e] - c1 - xh[i][e] * c2);
                    dwacc[i][e] += gv[i][e] * xh[i][e];
                }
                dr[c4] = o;
            }
        }
    }
    // block-level reduction of dw over the 4 waves (fixed order), then the block's row of partials
    __shared__ float red[(NT / 64) * 256 * NV * 4];
    const int wv_id = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c4 = lane + 64 * i;
#pragma unroll
        for (int e = 0; e < 4; ++e) red[wv_id * (256 * NV) + 4 * c4 + e] = dwacc[i][e];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += NT) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NT / 64; ++k) s += red[k * (256 * NV) + c];
        dw_part[(long)blockIdx.x * D + c] = s;
    }
}

// ------------------------------------------------------------------ RoPE
__global__ void rope_table_kernel(const float* __restrict__ inv_freq, float* __restrict__ table, int S, int half) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S * half) return;
    const int s = i / half, j = i - s * half;
    const float ang = (float)s * inv_freq[j];
    table[i] = cosf(ang);
    table[S * half + i] = sinf(ang);
}

// element access by storage type (CALM_ST_*): the bf16 pipeline keeps projection outputs / attention operands as bf16
__device__ __forceinline__ float ldt(const void* p, long i, int type) {
    return type == CALM_ST_BF16 ? (float)reinterpret_cast<const __bf16*>(p)[i] : reinterpret_cast<const float*>(p)[i];
}
__device__ __forceinline__ void stt(void* p, long i, float v, int type) {
    if (type == CALM_ST_BF16) reinterpret_cast<__bf16*>(p)[i] = (__bf16)v;
    else reinterpret_cast<float*>(p)[i] = v;
}

__global__ __launch_bounds__(NT) void rope_fwd_kernel(const void* __restrict__ content, const void* __restrict__ xr,
                                                      const float* __restrict__ table, void* __restrict__ out,
                                                      long nrows, int S, int H, int dc, int dr, int content_type,
                                                      int xr_type, int out_type) {
    const int half = dr >> 1;
    const int wd = dc + half;                 // work items per row: dc copies + half rotations
    const long total = nrows * wd;
    const float* cosT = table;
    const float* sinT = table + (long)S * half;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const long row = i / wd;
        const int j = (int)(i - row * wd);
        const long o = row * (dc + dr);
        if (j < dc) {
            stt(out, o + j, ldt(content, row * dc + j, content_type), out_type);
        } else {
            const int jj = j - dc;
            const int s = (int)((row / H) % S);
            const float c = cosT[s * half + jj], sn = sinT[s * half + jj];
            const float x1 = ldt(xr, row * dr + jj, xr_type), x2 = ldt(xr, row * dr + jj + half, xr_type);
            stt(out, o + dc + jj, x1 * c - x2 * sn, out_type);
            stt(out, o + dc + jj + half, x2 * c + x1 * sn, out_type);
        }
    }
}

constexpr int ROPE_MAX_HALF = 256;
#ifndef CALM_ROPE_VEC
#define CALM_ROPE_VEC 1        // 0: always the one-element kernels (A/B runs)
#endif
#ifndef CALM_ROPE_VW4
#define CALM_ROPE_VW4 1
#endif
#ifndef CALM_ROPE_BWD_GRID
#define CALM_ROPE_BWD_GRID 1024
#endif

// ---- row-walking form ----
// The one-element-per-thread kernels above cost a 64-bit division per element, an LDS atomic per rotated pair and, on
// bf16 tensors, 2-byte accesses: near the HBM roofline on fp32 tensors at round 1's sizes, 2-3x off it on bf16 ones.
// Here a thread owns a fixed rotation pair (VW = 2 adjacent pairs when dc and dr/2 are even: 4-byte bf16 / 8-byte fp32
// accesses) — so the angle gradient accumulates in registers — plus the content columns congruent to it, and walks
// rows: one 32-bit division per row for the position, no divergence between the copy and the rotation, whole rows
// contiguous per workgroup pass.
template <int VW> struct RopeVec { typedef float type __attribute__((ext_vector_type(VW))); };
template <> struct RopeVec<1> { typedef float type; };
template <int VW>
__device__ __forceinline__ typename RopeVec<VW>::type ldtv(const void* p, unsigned i, int type) {
    if constexpr (VW == 1) {
        return ldt(p, i, type);
    } else {
        typedef typename RopeVec<VW>::type fvec;
        if (type == CALM_ST_BF16) {
            typedef __bf16 bvec __attribute__((ext_vector_type(VW)));
            const bvec b = *reinterpret_cast<const bvec*>(reinterpret_cast<const __bf16*>(p) + i);
            fvec r;
#pragma unroll
            for (int e = 0; e < VW; ++e) r[e] = (float)b[e];
            return r;
        }
        return *reinterpret_cast<const fvec*>(reinterpret_cast<const float*>(p) + i);
    }
}
template <int VW>
__device__ __forceinline__ void sttv(void* p, unsigned i, typename RopeVec<VW>::type v, int type) {
    if constexpr (VW == 1) {
        stt(p, i, v, type);
    } else {
        typedef typename RopeVec<VW>::type fvec;
        if (type == CALM_ST_BF16) {
            typedef __bf16 bvec __attribute__((ext_vector_type(VW)));
            bvec b;
#pragma unroll
            for (int e = 0; e < VW; ++e) b[e] = (__bf16)v[e];
            *reinterpret_cast<bvec*>(reinterpret_cast<__bf16*>(p) + i) = b;
        } else {
            *reinterpret_cast<fvec*>(reinterpret_cast<float*>(p) + i) = v;
        }
    }
}
template <int VW>
__device__ __forceinline__ typename RopeVec<VW>::type ldf(const float* p) {
    return *reinterpret_cast<const typename RopeVec<VW>::type*>(p);
}

// (h, s) of row = (b S + s) H + h under row += stride, without divisions in the loop
struct RopePos {
    int h, s, dh, ds, H, S;
    __device__ __forceinline__ RopePos(int row0, int stride, int H_, int S_) : H(H_), S(S_) {
        const int bs = row0 / H_;
        h = row0 - bs * H_;
        s = bs % S_;
        const int dbs = stride / H_;
        dh = stride - dbs * H_;
        ds = dbs % S_;
    }
    __device__ __forceinline__ void advance() {
        h += dh;
        const int carry = h >= H ? 1 : 0;
        h -= carry ? H : 0;
        s += ds + carry;                      // < 2 S
        s -= s >= S ? S : 0;
    }
};

// TT: storage type of ALL tensors of the call as a compile-time constant (CALM_ST_F32 / CALM_ST_BF16), or -1 = per-tensor
// run-time types.  With run-time types every load and store of the row loop sat behind its own scalar branch (ISA, round 4:
// 20 branches per row, each access in its own basic block with its own wait): the accesses of a row could not overlap,
// which is what held the bf16 calls (2-8 bytes per access) at 1.5-3.5 TB/s.  The model only ever mixes nothing.
template <int TT> __device__ __forceinline__ int rope_type(int runtime_type) { return TT < 0 ? runtime_type : TT; }

template <int VW, int TT>
__global__ __launch_bounds__(NT) void rope_fwd_vec_kernel(const void* __restrict__ content, const void* __restrict__ xr,
                                                          const float* __restrict__ table, void* __restrict__ out,
                                                          int nrows, int S, int H, int dc, int dr, int content_type_,
                                                          int xr_type_, int out_type_) {
    typedef typename RopeVec<VW>::type vec;
    const int content_type = rope_type<TT>(content_type_), xr_type = rope_type<TT>(xr_type_), out_type = rope_type<TT>(out_type_);
    const int half = dr >> 1, ir = half / VW, ic = dc / VW;             // rotation / content items per row
    const int rpb = NT / ir;                                            // rows per workgroup pass
    const int r_in = threadIdx.x / ir, j = threadIdx.x - r_in * ir;
    if (r_in >= rpb) return;
    const int jj = VW * j;
    const float* cosT = table + jj;
    const float* sinT = table + (long)S * half + jj;
    // position s = (row / H) % S kept incrementally (the row advances by a launch constant): three divisions per
    // thread instead of two per row — the kernel is bound by instruction issue, not by bytes
    RopePos pos(blockIdx.x * rpb + r_in, gridDim.x * rpb, H, S);
    // element offsets as 32-bit unsigned values from the (uniform) tensor bases: the loads and stores then take the
    // scalar-base + 32-bit-offset form instead of a 64-bit multiply-add per access (host: every tensor < 2^31 elements)
    for (int row = blockIdx.x * rpb + r_in; row < nrows; row += gridDim.x * rpb, pos.advance()) {
        const unsigned o = (unsigned)row * (unsigned)(dc + dr), xo = (unsigned)row * (unsigned)dr + jj;
        const int s = pos.s;
        const vec c = ldf<VW>(cosT + s * half), sn = ldf<VW>(sinT + s * half);
        const vec x1 = ldtv<VW>(xr, xo, xr_type), x2 = ldtv<VW>(xr, xo + half, xr_type);
        for (int k = j; k < ic; k += ir)
            sttv<VW>(out, o + VW * k, ldtv<VW>(content, (unsigned)row * (unsigned)dc + VW * k, content_type), out_type);
        sttv<VW>(out, o + dc + jj, x1 * c - x2 * sn, out_type);
        sttv<VW>(out, o + dc + jj + half, x2 * c + x1 * sn, out_type);
    }
}

template <int VW, int TT>
__global__ __launch_bounds__(NT) void rope_bwd_vec_kernel(const void* __restrict__ d_out, const void* __restrict__ xr,
                                                          const float* __restrict__ table, void* __restrict__ d_content,
                                                          void* __restrict__ d_xr, float* __restrict__ dif_part,
                                                          int nrows, int S, int H, int dc, int dr, int dout_type_,
                                                          int xr_type_, int dcontent_type_, int dxr_type_) {
    typedef typename RopeVec<VW>::type vec;
    const int dout_type = rope_type<TT>(dout_type_), xr_type = rope_type<TT>(xr_type_),
              dcontent_type = rope_type<TT>(dcontent_type_), dxr_type = rope_type<TT>(dxr_type_);
    __shared__ float facc[NT * VW];                    // [row lane][rotation pair]: rpb * half <= NT * VW floats
    const int half = dr >> 1, ir = half / VW, ic = dc / VW;
    const int rpb = NT / ir;
    const int r_in = threadIdx.x / ir, j = threadIdx.x - r_in * ir;
    const int jj = VW * j;
    const float* cosT = table + jj;
    const float* sinT = table + (long)S * half + jj;
    vec acc = vec(0.f);
    if (r_in < rpb) {
        RopePos pos(blockIdx.x * rpb + r_in, gridDim.x * rpb, H, S);
        for (int row = blockIdx.x * rpb + r_in; row < nrows; row += gridDim.x * rpb, pos.advance()) {
            const unsigned go = (unsigned)row * (unsigned)(dc + dr), xo = (unsigned)row * (unsigned)dr + jj;
            const int s = pos.s;
            const vec c = ldf<VW>(cosT + s * half), sn = ldf<VW>(sinT + s * half);
            const vec g1 = ldtv<VW>(d_out, go + dc + jj, dout_type), g2 = ldtv<VW>(d_out, go + dc + jj + half, dout_type);
            const vec x1 = ldtv<VW>(xr, xo, xr_type), x2 = ldtv<VW>(xr, xo + half, xr_type);
            for (int k = j; k < ic; k += ir)
                sttv<VW>(d_content, (unsigned)row * (unsigned)dc + VW * k, ldtv<VW>(d_out, go + VW * k, dout_type), dcontent_type);
            sttv<VW>(d_xr, xo, g1 * c + g2 * sn, dxr_type);
            sttv<VW>(d_xr, xo + half, g2 * c - g1 * sn, dxr_type);
            // d/d(angle): y1 = x1 c - x2 s, y2 = x2 c + x1 s ; angle = s * inv_freq[jj]
            const vec dang = g1 * (-x1 * sn - x2 * c) + g2 * (-x2 * sn + x1 * c);
            acc += dang * (float)s;
        }
        if constexpr (VW == 1) {
            facc[r_in * half + jj] = acc;
        } else {
#pragma unroll
            for (int e = 0; e < VW; ++e) facc[r_in * half + jj + e] = acc[e];
        }
    }
    __syncthreads();
    // the row lanes of the block in lane order, then the block's row of partials (calm_reduce_partials adds the blocks)
    for (int k = threadIdx.x; k < half; k += NT) {
        float s = 0.f;
        for (int r = 0; r < rpb; ++r) s += facc[r * half + k];
        dif_part[(long)blockIdx.x * half + k] = s;
    }
}

// ------------------------------------------------------------------ softmax (one wave per row)
constexpr int SM_MAXC = 16;   // cols <= 1024

__global__ __launch_bounds__(NT) void softmax_fwd_kernel(float* __restrict__ x, long rows, int cols) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * (NT / 64);
    for (long row = wave; row < rows; row += nwaves) {
        float* xr = x + row * cols;
        float v[SM_MAXC];
        float m = -INFINITY;
#pragma unroll
        for (int i = 0; i < SM_MAXC; ++i) {
            const int c = lane + 64 * i;
            v[i] = c < cols ? xr[c] : -INFINITY;
            m = fmaxf(m, v[i]);
        }
        m = wave_max(m);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < SM_MAXC; ++i) {
            const int c = lane + 64 * i;
            v[i] = c < cols ? expf(v[i] - m) : 0.f;
            s += v[i];
        }
        const float inv = 1.0f / wave_sum(s);
#pragma unroll
        for (int i = 0; i < SM_MAXC; ++i) {
            const int c = lane + 64 * i;
            if (c < cols) xr[c] = v[i] * inv;
        }
    }
}

__global__ __launch_bounds__(NT) void softmax_bwd_kernel(const float* __restrict__ p, float* __restrict__ dp,
                                                         long rows, int cols) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * (NT / 64);
    for (long row = wave; row < rows; row += nwaves) {
        const float* pr = p + row * cols;
        float* gr = dp + row * cols;
        float pv[SM_MAXC], gv[SM_MAXC];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < SM_MAXC; ++i) {
            const int c = lane + 64 * i;
            pv[i] = c < cols ? pr[c] : 0.f;
            gv[i] = c < cols ? gr[c] : 0.f;
            s += pv[i] * gv[i];
        }
        s = wave_sum(s);
#pragma unroll
        for (int i = 0; i < SM_MAXC; ++i) {
            const int c = lane + 64 * i;
            if (c < cols) gr[c] = pv[i] * (gv[i] - s);
        }
    }
}

// softmax backward of all H heads of one (image, query) row by one wave, with the head-sum of the results — the
// gradient of the head-broadcast mask (Vi_Tools:291) — kept in registers: saves sum_heads' re-read of dL.
__global__ __launch_bounds__(NT) void softmax_bwd_heads_kernel(const float* __restrict__ p, float* __restrict__ dp,
                                                               float* __restrict__ dm, long n_bq, int H, int Sq, int cols) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * (NT / 64);
    for (long bq = wave; bq < n_bq; bq += nwaves) {
        const long b = bq / Sq, i = bq - b * Sq;
        float msum[SM_MAXC];
#pragma unroll
        for (int u = 0; u < SM_MAXC; ++u) msum[u] = 0.f;
        for (int hh = 0; hh < H; ++hh) {
            const long row = (b * H + hh) * Sq + i;
            const float* pr = p + row * cols;
            float* gr = dp + row * cols;
            float pv[SM_MAXC], gv[SM_MAXC];
            float s = 0.f;
#pragma unroll
            for (int u = 0; u < SM_MAXC; ++u) {
                const int c = lane + 64 * u;
                pv[u] = c < cols ? pr[c] : 0.f;
                gv[u] = c < cols ? gr[c] : 0.f;
                s += pv[u] * gv[u];
            }
            s = wave_sum(s);
#pragma unroll
            for (int u = 0; u < SM_MAXC; ++u) {
                const int c = lane + 64 * u;
                const float d = pv[u] * (gv[u] - s);
                if (c < cols) gr[c] = d;
                msum[u] += d;
            }
        }
        float* mr = dm + bq * cols;
#pragma unroll
        for (int u = 0; u < SM_MAXC; ++u) {
            const int c = lane + 64 * u;
            if (c < cols) mr[c] = msum[u];
        }
    }
}

__global__ __launch_bounds__(NT) void sum_heads_kernel(const float* __restrict__ dl, float* __restrict__ dm,
                                                       int B, int H, long per_head) {
    const long total = (long)B * per_head;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const long b = i / per_head, r = i - b * per_head;
        const float* src = dl + b * H * per_head + r;
        float s = 0.f;
        for (int h = 0; h < H; ++h) s += src[h * per_head];
        dm[i] = s;
    }
}

// ------------------------------------------------------------------ latent sampling + KL
__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }

__global__ __launch_bounds__(NT) void latent_fwd_kernel(const float* __restrict__ mv, const float* __restrict__ noise,
                                                        float* __restrict__ z, float* __restrict__ std_out,
                                                        float* __restrict__ kl_part, long rows, int mvh) {
    __shared__ float red[4];
    const long total = rows * mvh;
    float acc = 0.f;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const long row = i / mvh;
        const int c = (int)(i - row * mvh);
        const float mean = mv[row * 2 * mvh + c];
        const float raw = mv[row * 2 * mvh + mvh + c];
        const float sd = softplus_f(raw) + 1e-6f;
        z[i] = noise ? mean + noise[i] * sd : mean;
        std_out[i] = sd;
        acc += 1.0f + 2.0f * logf(sd) - mean * mean - sd * sd;
    }
    acc = block_sum_256(acc, red);
    if (threadIdx.x == 0) kl_part[blockIdx.x] = acc;
}

// Four elements per thread, 16-byte accesses, 32-bit index arithmetic and hardware exp / log (mvh % 4 == 0, aligned
// tensors, < 2^31 elements).  The one-element kernel above ran at 1.9 TB/s (PMC, round 4: 54 us per call at Base-224's
// [20480, 240]): a 64-bit division per element and three libm calls.  softplus(x) = max(x, 0) + log1p(e^{-|x|}) with
// log1p(t) = t - t^2/2 + t^3/3 below 1e-3 (error < 3e-10 relative) and log(1 + t) above (error <= 6e-5 relative at
// t = 1e-3, shrinking with t): the standard deviation keeps >= 4 digits over the whole range, its logarithm 1e-4 absolute.
__device__ __forceinline__ float softplus_fast(float x) {
    if (x > 20.f) return x;
    const float t = __expf(-fabsf(x));
    const float l = t < 1e-3f ? t * fmaf(t, fmaf(t, 0.33333334f, -0.5f), 1.0f) : __logf(1.0f + t);
    return fmaxf(x, 0.f) + l;
}
__global__ __launch_bounds__(NT) void latent_fwd_vec_kernel(const float* __restrict__ mv, const float* __restrict__ noise,
                                                            float* __restrict__ z, float* __restrict__ std_out,
                                                            float* __restrict__ kl_part, unsigned rows, unsigned mvh) {
    __shared__ float red[4];
    const unsigned q = mvh >> 2, total4 = rows * q;
    float acc = 0.f;
    for (unsigned i = blockIdx.x * NT + threadIdx.x; i < total4; i += gridDim.x * NT) {
        const unsigned row = i / q, c = 4u * (i - row * q);
        const f32x4 mean = *reinterpret_cast<const f32x4*>(mv + (size_t)row * 2 * mvh + c);
        const f32x4 raw = *reinterpret_cast<const f32x4*>(mv + (size_t)row * 2 * mvh + mvh + c);
        f32x4 nz = {0.f, 0.f, 0.f, 0.f};
        if (noise) nz = *reinterpret_cast<const f32x4*>(noise + (size_t)4 * i);
        f32x4 sd, zz;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            sd[e] = softplus_fast(raw[e]) + 1e-6f;
            zz[e] = fmaf(nz[e], sd[e], mean[e]);
            acc += 1.0f + 2.0f * __logf(sd[e]) - mean[e] * mean[e] - sd[e] * sd[e];
        }
        *reinterpret_cast<f32x4*>(z + (size_t)4 * i) = zz;
        *reinterpret_cast<f32x4*>(std_out + (size_t)4 * i) = sd;
    }
    acc = block_sum_256(acc, red);
    if (threadIdx.x == 0) kl_part[blockIdx.x] = acc;
}

__global__ __launch_bounds__(NT) void latent_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ d_kl_sum,
                                                        const float* __restrict__ mv, const float* __restrict__ noise,
                                                        const float* __restrict__ std_in, float* __restrict__ dmv,
                                                        long rows, int mvh) {
    const long total = rows * mvh;
    const float dk = d_kl_sum ? d_kl_sum[0] : 0.f;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const long row = i / mvh;
        const int c = (int)(i - row * mvh);
        const float mean = mv[row * 2 * mvh + c];
        const float raw = mv[row * 2 * mvh + mvh + c];
        const float sd = std_in[i];
        const float g = dz ? dz[i] : 0.f;
        const float dmean = g - 2.0f * dk * mean;
        const float dstd = (noise ? g * noise[i] : 0.f) + dk * (2.0f / sd - 2.0f * sd);
        const float sig = raw > 20.f ? 1.0f : 1.0f / (1.0f + expf(-raw));
        dmv[row * 2 * mvh + c] = dmean;
        dmv[row * 2 * mvh + mvh + c] = dstd * sig;
    }
}

// ------------------------------------------------------------------ helpers
__global__ __launch_bounds__(NT) void add_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                 float* __restrict__ out, long n4, long n) {
    const long stride = (long)gridDim.x * NT;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n4; i += stride) {
        const f32x4 x = reinterpret_cast<const f32x4*>(a)[i];
        const f32x4 y = reinterpret_cast<const f32x4*>(b)[i];
        reinterpret_cast<f32x4*>(out)[i] = x + y;
    }
    for (long i = 4 * n4 + (long)blockIdx.x * NT + threadIdx.x; i < n; i += stride) out[i] = a[i] + b[i];
}

__global__ __launch_bounds__(NT) void gelu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n) {
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) y[i] = gelu_erf_f(x[i]);
}

__global__ __launch_bounds__(NT) void gelu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ z,
                                                      float* __restrict__ dz, long n) {
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT)
        dz[i] = dy[i] * gelu_erf_grad_f(z[i]);
}

constexpr int COLSUM_MAXC = 4096;
#ifndef COLSUM_GRID
#define COLSUM_GRID 128      // A/B on [57344,448]: 66 us (scalar form) -> 28.6 us; 256 blocks: 39.6 (atomics), 128x256 threads: 34
#endif
__global__ __launch_bounds__(NT) void colsum_kernel(const void* __restrict__ x, float* __restrict__ part, long rows,
                                                    int cols, int x_type) {
    // block owns the rows r = blockIdx.x (mod gridDim.x); partial row of the block -> part[blockIdx.x * cols ..]
    if (cols >= NT) {
        // thread owns columns tid, tid+256, ...
        for (int c = threadIdx.x; c < cols; c += NT) {
            float s = 0.f;
            for (long r = blockIdx.x; r < rows; r += gridDim.x) s += ldt(x, r * cols + c, x_type);
            part[(long)blockIdx.x * cols + c] = s;
        }
    } else {
        // NT / cols row lanes, each thread one column of its rows; the lanes meet in LDS in lane order
        __shared__ float acc[NT];
        const int rl_n = NT / cols, rl = threadIdx.x / cols, c = threadIdx.x - rl * cols;
        float s = 0.f;
        if (rl < rl_n)
            for (long r = (long)blockIdx.x * rl_n + rl; r < rows; r += (long)gridDim.x * rl_n) s += ldt(x, r * cols + c, x_type);
        acc[threadIdx.x] = s;
        __syncthreads();
        if (threadIdx.x < cols) {
            float t = 0.f;
            for (int k = 0; k < rl_n; ++k) t += acc[k * cols + threadIdx.x];
            part[(long)blockIdx.x * cols + threadIdx.x] = t;
        }
    }
}

// 16-byte form: TX lanes across the row (one float4 each, several passes if the row is longer), 1024/TX row lanes,
// four rows in flight per thread; partial sums meet in LDS in lane order, one row of partials per block.
constexpr int CS_NT = 1024;          // 16 waves per block: few blocks (few atomics per column), many rows in flight
// X16: x is a bf16 tensor (8-byte loads of 4 elements; the sums stay fp32)
template <int TX, bool X16>
__global__ __launch_bounds__(CS_NT) void colsum_vec_kernel(const void* __restrict__ x_, float* __restrict__ prow, long rows,
                                                        int cols) {
    auto ld = [&](long r, int c4) -> f32x4 {
        if constexpr (X16) {
            const bf16x4 v = reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(x_) + r * cols)[c4];
            return (f32x4){(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
        } else {
            return reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(x_) + r * cols)[c4];
        }
    };
    constexpr int TY = CS_NT / TX;
    __shared__ f32x4 part[CS_NT];
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int c4n = cols >> 2;
    const long rstep = (long)gridDim.x * TY;
    for (int c4 = tx; c4 < ((c4n + TX - 1) / TX) * TX; c4 += TX) {
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
        if (c4 < c4n) {
            long r = (long)blockIdx.x * TY + ty;
            for (; r + 3 * rstep < rows; r += 4 * rstep) {
                a0 += ld(r, c4);
                a1 += ld(r + rstep, c4);
                a2 += ld(r + 2 * rstep, c4);
                a3 += ld(r + 3 * rstep, c4);
            }
            for (; r < rows; r += rstep) a0 += ld(r, c4);
        }
        part[threadIdx.x] = (a0 + a1) + (a2 + a3);
        __syncthreads();
        if (ty == 0 && c4 < c4n) {
            f32x4 s = part[tx];
#pragma unroll
            for (int k = 1; k < TY; ++k) s += part[k * TX + tx];
            *reinterpret_cast<f32x4*>(prow + (long)blockIdx.x * cols + 4 * c4) = s;
        }
        __syncthreads();
    }
}

template <bool O16>
__global__ __launch_bounds__(NT) void row_scale_kernel(const float* __restrict__ x, const float* __restrict__ s,
                                                       void* __restrict__ out_, int rows, int cols) {
    const long total = (long)rows * cols;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const float v = x[i] * s[i / cols];
        if constexpr (O16) reinterpret_cast<__bf16*>(out_)[i] = (__bf16)v;
        else reinterpret_cast<float*>(out_)[i] = v;
    }
}

__global__ __launch_bounds__(NT) void mean_seq_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int B,
                                                          int S, int D) {
    const long total = (long)B * D;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const long b = i / D;
        const int d = (int)(i - b * D);
        const float* src = x + b * S * D + d;
        float s = 0.f;
        for (int t = 0; t < S; ++t) s += src[(long)t * D];
        y[i] = s / S;
    }
}

__global__ __launch_bounds__(NT) void mean_seq_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int B,
                                                          int S, int D) {
    const long total = (long)B * S * D;
    const float inv = 1.0f / S;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const long b = i / ((long)S * D);
        const int d = (int)(i % D);
        dx[i] = dy[b * D + d] * inv;
    }
}

}  // namespace

extern "C" {

int calm_layernorm_fwd(const float* x, const float* w, void* y, float* mean, float* rstd, int64_t rows, int32_t D,
                       float eps, int32_t y_type, void* stream) {
    if (!x || !w || !y || !mean || !rstd || rows <= 0 || D <= 0) return CALM_E_INVAL;
    if (y_type != CALM_ST_F32 && y_type != CALM_ST_BF16) return CALM_E_INVAL;
    const bool y16 = y_type == CALM_ST_BF16;
    hipStream_t s = as_stream(stream);
    const dim3 g(grid_for(rows, NT / 64)), b(NT);
    if ((D & 3) == 0 && D <= 256 * 5 && aligned16(x) && aligned16(y) && aligned16(w)) {
        const int nv = (D / 4 + 63) / 64;
#define LN_FWD(NVV)                                                                                                  \
    do {                                                                                                             \
        if (y16) hipLaunchKernelGGL((ln_fwd_vec_kernel<NVV, true>), g, b, 0, s, x, w, y, mean, rstd, (long)rows, D, eps); \
        else hipLaunchKernelGGL((ln_fwd_vec_kernel<NVV, false>), g, b, 0, s, x, w, y, mean, rstd, (long)rows, D, eps);    \
    } while (0)
        switch (nv) {
            case 1: LN_FWD(1); break;
            case 2: LN_FWD(2); break;
            case 3: LN_FWD(3); break;
            case 4: LN_FWD(4); break;
            default: LN_FWD(5); break;
        }
#undef LN_FWD
        CALM_LAUNCH_CHECK();
        return 0;
    }
    if (y16) hipLaunchKernelGGL(ln_fwd_kernel<true>, g, b, 0, s, x, w, y, mean, rstd, (long)rows, D, eps);
    else hipLaunchKernelGGL(ln_fwd_kernel<false>, g, b, 0, s, x, w, y, mean, rstd, (long)rows, D, eps);
    CALM_LAUNCH_CHECK();
    return 0;
}

static int ln_bwd_grid(int64_t rows) {
    const int g = grid_for(rows, NT / 64);
    return g > 512 ? 512 : g;
}

int calm_layernorm_bwd(const void* dy, const float* x, const float* w, const float* mean, const float* rstd,
                       float* dx, float* dw, const float* dx_add, int64_t rows, int32_t D, int32_t dy_type,
                       float* partials, void* stream) {
    if (!dy || !x || !w || !mean || !rstd || !dx || !dw || !partials || rows <= 0 || D <= 0) return CALM_E_INVAL;
    if (dy_type != CALM_ST_F32 && dy_type != CALM_ST_BF16) return CALM_E_INVAL;
    if (D > 64 * LN_MAXC) return CALM_E_UNSUPP;
    const bool g16 = dy_type == CALM_ST_BF16;
    const int g = ln_bwd_grid(rows);
    hipStream_t s = as_stream(stream);
    const dim3 gd(g), b(NT);
    if ((D & 3) == 0 && D <= 256 * 5 && aligned16(x) && aligned16(dy) && aligned16(dx) && aligned16(w) &&
        aligned16(dx_add)) {
        const int nv = (D / 4 + 63) / 64;
#define LN_BWD(NVV)                                                                                                        \
    do {                                                                                                                   \
        if (g16) hipLaunchKernelGGL((ln_bwd_vec_kernel<NVV, true>), gd, b, 0, s, dy, x, w, mean, rstd, dx, partials, dx_add, (long)rows, D); \
        else hipLaunchKernelGGL((ln_bwd_vec_kernel<NVV, false>), gd, b, 0, s, dy, x, w, mean, rstd, dx, partials, dx_add, (long)rows, D);    \
    } while (0)
        switch (nv) {
            case 1: LN_BWD(1); break;
            case 2: LN_BWD(2); break;
            case 3: LN_BWD(3); break;
            case 4: LN_BWD(4); break;
            default: LN_BWD(5); break;
        }
#undef LN_BWD
        CALM_LAUNCH_CHECK();
        calm_reduce_partials(partials, g, D, dw, s);
        CALM_LAUNCH_CHECK();
        return 0;
    }
    if (g16) hipLaunchKernelGGL(ln_bwd_kernel<true>, gd, b, 0, s, dy, x, w, mean, rstd, dx, partials, dx_add, (long)rows, D);
    else hipLaunchKernelGGL(ln_bwd_kernel<false>, gd, b, 0, s, dy, x, w, mean, rstd, dx, partials, dx_add, (long)rows, D);
    CALM_LAUNCH_CHECK();
    calm_reduce_partials(partials, g, D, dw, s);
    CALM_LAUNCH_CHECK();
    return 0;
}

static bool st_ok(int t) { return t == CALM_ST_F32 || t == CALM_ST_BF16; }
// row-walking RoPE kernels: 32-bit row index; VW = 2 when column pairs do not straddle the content / half boundaries
static bool rope_vec_ok(long nrows, int dc, int dr) {
    return CALM_ROPE_VEC && nrows * (dc + dr) < (1L << 31) && dr / 2 <= NT && dr / 2 <= ROPE_MAX_HALF;      // 32-bit element offsets
}
static int rope_vw(int dc, int dr) {     // widest column group that does not straddle the content / half boundaries
    const int half = dr / 2;
    return ((dc & 3) == 0 && (half & 3) == 0 && CALM_ROPE_VW4) ? 4 : ((dc & 1) == 0 && (half & 1) == 0) ? 2 : 1;
}
static int rope_vec_grid(long nrows, int dc, int dr) {
    const int rpb = NT / (dr / 2 / rope_vw(dc, dr));
    const long blocks = (nrows + rpb - 1) / rpb;
    return (int)(blocks < 256 * 16 ? blocks : 256 * 16);
}

int calm_rope_fwd(const void* content, const void* xr, const float* inv_freq, float* table, void* out, int32_t B,
                  int32_t S, int32_t H, int32_t dc, int32_t dr, int32_t content_type, int32_t xr_type, int32_t out_type,
                  void* stream) {
    if (!xr || !inv_freq || !table || !out || B <= 0 || S <= 0 || H <= 0 || dc < 0 || dr <= 0 || (dr & 1))
        return CALM_E_INVAL;
    if (dc > 0 && !content) return CALM_E_INVAL;
    if (!st_ok(content_type) || !st_ok(xr_type) || !st_ok(out_type)) return CALM_E_INVAL;
    const int half = dr / 2;
    hipLaunchKernelGGL(rope_table_kernel, dim3((S * half + 255) / 256), dim3(256), 0, as_stream(stream), inv_freq,
                       table, S, half);
    CALM_LAUNCH_CHECK();
    const long nrows = (long)B * S * H;
    if (rope_vec_ok(nrows, dc, dr)) {
        const dim3 gv(rope_vec_grid(nrows, dc, dr));
        const bool same = xr_type == out_type && (dc == 0 || content_type == out_type);
        const int tt = !same ? -1 : out_type;
#define ROPE_FWD(VWV, TTV)                                                                                              \
    hipLaunchKernelGGL((rope_fwd_vec_kernel<VWV, TTV>), gv, dim3(NT), 0, as_stream(stream), content, xr, table, out,   \
                       (int)nrows, S, H, dc, dr, content_type, xr_type, out_type)
#define ROPE_FWD_T(VWV)                                                                                                 \
    do {                                                                                                                \
        if (tt == CALM_ST_BF16) ROPE_FWD(VWV, CALM_ST_BF16);                                                            \
        else if (tt == CALM_ST_F32) ROPE_FWD(VWV, CALM_ST_F32);                                                         \
        else ROPE_FWD(VWV, -1);                                                                                         \
    } while (0)
        if (rope_vw(dc, dr) == 4) ROPE_FWD_T(4);
        else if (rope_vw(dc, dr) == 2) ROPE_FWD_T(2);
        else ROPE_FWD_T(1);
#undef ROPE_FWD_T
#undef ROPE_FWD
        CALM_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(rope_fwd_kernel, dim3(grid_for(nrows * (dc + half), NT)), dim3(NT), 0, as_stream(stream),
                       content, xr, table, out, nrows, S, H, dc, dr, content_type, xr_type, out_type);
    CALM_LAUNCH_CHECK();
    return 0;
}

static int rope_bwd_grid(long nrows, int dc, int dr) {
    int gv = rope_vec_grid(nrows, dc, dr);
    // every workgroup ends with a row of dr/2 partials: few workgroups for small launches (A/B at S=80 with the
    // atomics of rounds 1-3: 1024 -> 25 us, 2048 -> 36, 4096 -> 60), ~16 row passes per workgroup for large ones
    const int want = gv / 16;
    return want < CALM_ROPE_BWD_GRID ? (gv < CALM_ROPE_BWD_GRID ? gv : CALM_ROPE_BWD_GRID) : (want < 2048 ? want : 2048);
}

int calm_rope_bwd(const void* d_out, const void* xr, const float* table, void* d_content, void* d_xr,
                  float* d_inv_freq, int32_t B, int32_t S, int32_t H, int32_t dc, int32_t dr, int32_t dout_type,
                  int32_t xr_type, int32_t dcontent_type, int32_t dxr_type, float* partials, void* stream) {
    if (!d_out || !xr || !table || !d_xr || !d_inv_freq || !partials || B <= 0 || S <= 0 || H <= 0 || dc < 0 ||
        dr <= 0 || (dr & 1))
        return CALM_E_INVAL;
    if (dc > 0 && !d_content) return CALM_E_INVAL;
    if (!st_ok(dout_type) || !st_ok(xr_type) || !st_ok(dcontent_type) || !st_ok(dxr_type)) return CALM_E_INVAL;
    const long nrows = (long)B * S * H;
    // (the one-element-per-thread backward of round 1 combined its angle gradients with LDS atomics; the row-walking
    // kernel serves every shape of the path: dr/2 <= 256 rotation pairs, tensors below 2^31 elements)
    if (dr / 2 > ROPE_MAX_HALF || dr / 2 > NT || nrows * (dc + dr) >= (1L << 31)) return CALM_E_UNSUPP;
    const int gv = rope_bwd_grid(nrows, dc, dr);
    const bool same = xr_type == dout_type && dxr_type == dout_type && (dc == 0 || dcontent_type == dout_type);
    const int tt = !same ? -1 : dout_type;
#define ROPE_BWD(VWV, TTV)                                                                                              \
    hipLaunchKernelGGL((rope_bwd_vec_kernel<VWV, TTV>), dim3(gv), dim3(NT), 0, as_stream(stream), d_out, xr, table,    \
                       d_content, d_xr, partials, (int)nrows, S, H, dc, dr, dout_type, xr_type, dcontent_type, dxr_type)
#define ROPE_BWD_T(VWV)                                                                                                 \
    do {                                                                                                                \
        if (tt == CALM_ST_BF16) ROPE_BWD(VWV, CALM_ST_BF16);                                                            \
        else if (tt == CALM_ST_F32) ROPE_BWD(VWV, CALM_ST_F32);                                                         \
        else ROPE_BWD(VWV, -1);                                                                                         \
    } while (0)
    if (rope_vw(dc, dr) == 4) ROPE_BWD_T(4);
    else if (rope_vw(dc, dr) == 2) ROPE_BWD_T(2);
    else ROPE_BWD_T(1);
#undef ROPE_BWD_T
#undef ROPE_BWD
    CALM_LAUNCH_CHECK();
    calm_reduce_partials(partials, gv, dr / 2, d_inv_freq, as_stream(stream));
    CALM_LAUNCH_CHECK();
    return 0;
}

int calm_softmax_fwd(float* x, int64_t rows, int32_t cols, void* stream) {
    if (!x || rows <= 0 || cols <= 0) return CALM_E_INVAL;
    if (cols > 64 * SM_MAXC) return CALM_E_UNSUPP;
    hipLaunchKernelGGL(softmax_fwd_kernel, dim3(grid_for(rows, NT / 64)), dim3(NT), 0, as_stream(stream), x,
                       (long)rows, cols);
    CALM_LAUNCH_CHECK();
    return 0;
}

int calm_softmax_bwd(const float* p, float* dp, int64_t rows, int32_t cols, void* stream) {
    if (!p || !dp || rows <= 0 || cols <= 0) return CALM_E_INVAL;
    if (cols > 64 * SM_MAXC) return CALM_E_UNSUPP;
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3(grid_for(rows, NT / 64)), dim3(NT), 0, as_stream(stream), p, dp,
                       (long)rows, cols);
    CALM_LAUNCH_CHECK();
    return 0;
}

int calm_softmax_bwd_heads(const float* p, float* dp, float* dm, int32_t B, int32_t H, int32_t Sq, int32_t cols,
                           void* stream) {
    if (!p || !dp || !dm || B <= 0 || H <= 0 || Sq <= 0 || cols <= 0) return CALM_E_INVAL;
    if (cols > 64 * SM_MAXC) return CALM_E_UNSUPP;
    const long n_bq = (long)B * Sq;
    hipLaunchKernelGGL(softmax_bwd_heads_kernel, dim3(grid_for(n_bq, NT / 64)), dim3(NT), 0, as_stream(stream), p, dp, dm,
                       n_bq, H, Sq, cols);
    CALM_LAUNCH_CHECK();
    return 0;
}

int calm_sum_heads(const float* dl, float* dm, int32_t B, int32_t H, int64_t per_head, void* stream) {
    if (!dl || !dm || B <= 0 || H <= 0 || per_head <= 0) return CALM_E_INVAL;
    hipLaunchKernelGGL(sum_heads_kernel, dim3(grid_for((int64_t)B * per_head, NT)), dim3(NT), 0, as_stream(stream),
                       dl, dm, B, H, (long)per_head);
    CALM_LAUNCH_CHECK();
    return 0;
}

static int latent_grid(int64_t rows, int mvh) {
    const int g = grid_for(rows * mvh, NT * 4);        // (four elements per thread in the vector kernel)
    return g > 2048 ? 2048 : g;
}

int calm_latent_fwd(const float* mv, const float* noise, float* z, float* std_out, float* kl_sum, int64_t rows,
                    int32_t mvh, float* partials, void* stream) {
    if (!mv || !z || !std_out || !kl_sum || !partials || rows <= 0 || mvh <= 0) return CALM_E_INVAL;
    const int g = latent_grid(rows, mvh);
    if ((mvh & 3) == 0 && rows * mvh < (1ll << 31) && aligned16(mv) && aligned16(z) && aligned16(std_out) &&
        aligned16(noise))
        hipLaunchKernelGGL(latent_fwd_vec_kernel, dim3(g), dim3(NT), 0, as_stream(stream), mv, noise, z, std_out, partials,
                           (unsigned)rows, (unsigned)mvh);
    else
        hipLaunchKernelGGL(latent_fwd_kernel, dim3(g), dim3(NT), 0, as_stream(stream), mv, noise, z, std_out, partials,
                           (long)rows, mvh);
    CALM_LAUNCH_CHECK();
    calm_reduce_partials(partials, g, 1, kl_sum, as_stream(stream));
    CALM_LAUNCH_CHECK();
    return 0;
}

int calm_latent_bwd(const float* dz, const float* d_kl_sum, const float* mv, const float* noise, const float* std_in,
                    float* dmv, int64_t rows, int32_t mvh, void* stream) {
    if (!mv || !std_in || !dmv || rows <= 0 || mvh <= 0) return CALM_E_INVAL;
    hipLaunchKernelGGL(latent_bwd_kernel, dim3(grid_for(rows * mvh, NT)), dim3(NT), 0, as_stream(stream), dz,
                       d_kl_sum, mv, noise, std_in, dmv, (long)rows, mvh);
    CALM_LAUNCH_CHECK();
    return 0;
}

int calm_add(const float* a, const float* b, float* out, int64_t n, void* stream) {
    if (!a || !b || !out || n <= 0) return CALM_E_INVAL;
    const bool vec = aligned16(a) && aligned16(b) && aligned16(out);
    const long n4 = vec ? n / 4 : 0;
    hipLaunchKernelGGL(add_kernel, dim3(grid_for(vec ? n4 + 1 : n, NT)), dim3(NT), 0, as_stream(stream), a, b, out,
                       n4, (long)n);
    CALM_LAUNCH_CHECK();
    return 0;
}

int calm_gelu_fwd(const float* x, float* y, int64_t n, void* stream) {
    if (!x || !y || n <= 0) return CALM_E_INVAL;
    hipLaunchKernelGGL(gelu_fwd_kernel, dim3(grid_for(n, NT)), dim3(NT), 0, as_stream(stream), x, y, (long)n);
    CALM_LAUNCH_CHECK();
    return 0;
}

int calm_gelu_bwd(const float* dy, const float* z, float* dz, int64_t n, void* stream) {
    if (!dy || !z || !dz || n <= 0) return CALM_E_INVAL;
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3(grid_for(n, NT)), dim3(NT), 0, as_stream(stream), dy, z, dz, (long)n);
    CALM_LAUNCH_CHECK();
    return 0;
}

// (vector form?, workgroups) of a column-sum launch
static int colsum_grid(int64_t rows, int cols, bool vec, int* tx_out) {
    if (vec) {
        const int c4n = cols >> 2;
        const int tx = c4n <= 32 ? 32 : c4n <= 64 ? 64 : c4n <= 128 ? 128 : 256;
        const int ty = CS_NT / tx;
        const long gl = (rows + (long)ty * 8 - 1) / ((long)ty * 8);          // >= 8 rows per row lane
        if (tx_out) *tx_out = tx;
        return (int)(gl < 1 ? 1 : gl > COLSUM_GRID ? COLSUM_GRID : gl);      // few blocks: every block ends in a row of partials
    }
    const long per = cols >= NT ? 1 : NT / cols;                              // rows per block pass
    long g = (rows + per * 8 - 1) / (per * 8);
    return (int)(g < 1 ? 1 : g > 512 ? 512 : g);
}

int calm_colsum(const void* x, float* out, int64_t rows, int32_t cols, int32_t x_type, float* partials, void* stream) {
    if (!x || !out || !partials || rows <= 0 || cols <= 0) return CALM_E_INVAL;
    if (!st_ok(x_type)) return CALM_E_INVAL;
    if (cols > COLSUM_MAXC) return CALM_E_UNSUPP;
    const bool x16 = x_type == CALM_ST_BF16;
    hipStream_t s = as_stream(stream);
    if ((cols & 3) == 0 && aligned16(x) && aligned16(partials) && rows >= 64) {
        int tx = 0;
        const int gv = colsum_grid(rows, cols, true, &tx);
#define CS_LAUNCH(TXV)                                                                                              \
    do {                                                                                                            \
        if (x16) hipLaunchKernelGGL((colsum_vec_kernel<TXV, true>), dim3(gv), dim3(CS_NT), 0, s, x, partials, (long)rows, cols); \
        else hipLaunchKernelGGL((colsum_vec_kernel<TXV, false>), dim3(gv), dim3(CS_NT), 0, s, x, partials, (long)rows, cols);    \
    } while (0)
        switch (tx) {
            case 32: CS_LAUNCH(32); break;
            case 64: CS_LAUNCH(64); break;
            case 128: CS_LAUNCH(128); break;
            default: CS_LAUNCH(256); break;
        }
#undef CS_LAUNCH
        CALM_LAUNCH_CHECK();
        calm_reduce_partials(partials, gv, cols, out, s);
        CALM_LAUNCH_CHECK();
        return 0;
    }
    const int g = colsum_grid(rows, cols, false, nullptr);
    hipLaunchKernelGGL(colsum_kernel, dim3(g), dim3(NT), 0, s, x, partials, (long)rows, cols, x_type);
    CALM_LAUNCH_CHECK();
    calm_reduce_partials(partials, g, cols, out, s);
    CALM_LAUNCH_CHECK();
    return 0;
}

/* floats of `partials` scratch an entry point with a cross-workgroup reduction needs (upper bound over its kernel
 * variants); op = CALM_RED_*; (rows, cols) as documented at the enum */
int64_t calm_reduce_scratch_floats(int32_t op, int64_t rows, int32_t cols) {
    if (rows <= 0 || cols <= 0) return 0;
    switch (op) {
        case CALM_RED_LAYERNORM_BWD: return (int64_t)ln_bwd_grid(rows) * cols;
        case CALM_RED_ROPE_BWD: return (int64_t)4096 * (cols / 2);
        case CALM_RED_LATENT_FWD: return latent_grid(rows, cols);
        case CALM_RED_COLSUM: {
            const int64_t a = (int64_t)colsum_grid(rows, cols, false, nullptr) * cols;
            const int64_t b = (cols & 3) == 0 && rows >= 64 ? (int64_t)colsum_grid(rows, cols, true, nullptr) * cols : 0;
            return a > b ? a : b;
        }
        case CALM_RED_CNN_BWD: return (int64_t)256 * 560;       // cnn_fused.hip: grid <= 256 rows of CNN_PART_STRIDE
        default: return 0;
    }
}

int calm_row_scale(const float* x, const float* s, void* out, int32_t rows, int32_t cols, int32_t out_type,
                   void* stream) {
    if (!x || !s || !out || rows <= 0 || cols <= 0) return CALM_E_INVAL;
    if (out_type != CALM_ST_F32 && out_type != CALM_ST_BF16) return CALM_E_INVAL;
    const dim3 g(grid_for((int64_t)rows * cols, NT)), b(NT);
    if (out_type == CALM_ST_BF16) hipLaunchKernelGGL(row_scale_kernel<true>, g, b, 0, as_stream(stream), x, s, out, rows, cols);
    else hipLaunchKernelGGL(row_scale_kernel<false>, g, b, 0, as_stream(stream), x, s, out, rows, cols);
    CALM_LAUNCH_CHECK();
    return 0;
}

int calm_mean_seq_fwd(const float* x, float* y, int32_t B, int32_t S, int32_t D, void* stream) {
    if (!x || !y || B <= 0 || S <= 0 || D <= 0) return CALM_E_INVAL;
    hipLaunchKernelGGL(mean_seq_fwd_kernel, dim3(grid_for((int64_t)B * D, NT)), dim3(NT), 0, as_stream(stream), x, y,
                       B, S, D);
    CALM_LAUNCH_CHECK();
    return 0;
}

int calm_mean_seq_bwd(const float* dy, float* dx, int32_t B, int32_t S, int32_t D, void* stream) {
    if (!dy || !dx || B <= 0 || S <= 0 || D <= 0) return CALM_E_INVAL;
    hipLaunchKernelGGL(mean_seq_bwd_kernel, dim3(grid_for((int64_t)B * S * D, NT)), dim3(NT), 0, as_stream(stream),
                       dy, dx, B, S, D);
    CALM_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
