// fp8 kernel family of calm_gemm (BASELINE.json configs[4]: "bf16 + fp8 MFMA GEMMs") and the per-tensor quantisers.
//
// Forward and input-gradient products of the Linear layers with BOTH operands in OCP fp8 in HBM (gfx950: e4m3fn /
// e5m2 — not MI300's fnuz encodings), one scale per tensor, fp32 accumulation on v_mfma_f32_32x32x16_fp8_fp8 /
// _bf8_fp8:
//     C = (A_q B_q^T) * dq_a * dq_b * alpha / sigma ...        A_q = fp8(A * FP8_MAX / amax(A)),  dq_a = amax(A) / FP8_MAX
// Both operands are k-contiguous (the input gradient multiplies by a transposed fp8 copy of the weight), so every
// MFMA fragment is one 8-byte LDS read.  256x128x64 tile per 512-thread workgroup (8 waves as 4x2, each 64x64), images
// [row][64 bytes] with a 72-byte row stride (32 lanes x 8 bytes land on 64 distinct banks), register-prefetch double
// buffer as in the bf16 family; the epilogue is the shared one (bias, GELU / GELU', LayerScale, residual, bf16 or fp32
// outputs).  Non-scaled fp8 MFMAs run at the bf16 rate: what fp8 buys here is half the operand bytes.
// Weight gradients stay on the bf16 kernels.
#include "gemm_common.h"

namespace calm_gemm_detail {

constexpr int FK = 64;             // k-tile in elements (= bytes)
constexpr int F_LD = 72;           // bytes per image row

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// ROWS-row operand tile of a k-contiguous fp8 matrix: NV 16-byte vectors per thread and k-tile
template <int ROWS>
struct FCursor {
    static constexpr int NV = ROWS * FK / (16 * WTHREADS);      // 2 (256 rows) or 1 (128 rows)
    const unsigned char* base;
    unsigned off[NV];
    __device__ __forceinline__ void init(const unsigned char* origin, long rs, int row0, int nrows_all, int k0) {
        const int tid = threadIdx.x;
        const int last = min(nrows_all - row0, ROWS) - 1;
        base = origin + (long)row0 * rs + k0;
#pragma unroll
        for (int i = 0; i < NV; ++i) off[i] = (unsigned)(min((tid >> 2) + (WTHREADS / 4) * i, last) * rs + 16 * (tid & 3));
    }
    __device__ __forceinline__ void load(int k_left, u32x4 (&reg)[NV]) {
        const int tid = threadIdx.x;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (16 * (tid & 3) < k_left) v = *reinterpret_cast<const u32x4*>(base + off[i]);      // K % 16 == 0
            reg[i] = v;
        }
        base += FK;
    }
    __device__ __forceinline__ void store(unsigned char* __restrict__ img, const u32x4 (&reg)[NV]) const {
        const int tid = threadIdx.x;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            unsigned char* d = img + ((tid >> 2) + (WTHREADS / 4) * i) * F_LD + 16 * (tid & 3);
            *reinterpret_cast<u32x2*>(d) = (u32x2){reg[i][0], reg[i][1]};
            *reinterpret_cast<u32x2*>(d + 8) = (u32x2){reg[i][2], reg[i][3]};
        }
    }
};

template <bool A_BF8>
__global__ __launch_bounds__(WTHREADS, 4) void gemm_fp8w_kernel(const GemmP p) {
    constexpr int MT = 2, NT = 2;
    __shared__ __attribute__((aligned(16))) unsigned char lds_a[2][WBM * F_LD];
    __shared__ __attribute__((aligned(16))) unsigned char lds_b[2][WBN * F_LD];

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
    const int z = blockIdx.y;                                   // batch entry
    const int b0 = z / p.batch1, b1 = z - b0 * p.batch1;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    FCursor<WBM> ca;
    FCursor<WBN> cb;
    u32x4 ra[FCursor<WBM>::NV], rb[FCursor<WBN>::NV];
    ca.init(reinterpret_cast<const unsigned char*>(p.A) + b0 * p.a_b0 + b1 * p.a_b1, p.a_rs, m0, p.M, 0);
    cb.init(reinterpret_cast<const unsigned char*>(p.B) + b0 * p.b_b0 + b1 * p.b_b1, p.b_rs, n0, p.N, 0);
    const int nkb = (p.K + FK - 1) / FK;
    ca.load(p.K, ra);
    cb.load(p.K, rb);
    ca.store(lds_a[0], ra);
    cb.store(lds_b[0], rb);
    __syncthreads();
    int buf = 0;
    for (int kb = 0; kb < nkb; ++kb) {
        const bool more = kb + 1 < nkb;
        if (more) {
            ca.load(p.K - FK * (kb + 1), ra);
            cb.load(p.K - FK * (kb + 1), rb);
        }
#pragma unroll
        for (int s = 0; s < FK / 16; ++s) {
            long af[MT], bf[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i)
                af[i] = *reinterpret_cast<const long*>(lds_a[buf] + (wm * 64 + 32 * i + r) * F_LD + 16 * s + 8 * h);
#pragma unroll
            for (int j = 0; j < NT; ++j)
                bf[j] = *reinterpret_cast<const long*>(lds_b[buf] + (wn * 64 + 32 * j + r) * F_LD + 16 * s + 8 * h);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    if constexpr (A_BF8) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf8_fp8(af[i], bf[j], acc[i][j], 0, 0, 0);
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(af[i], bf[j], acc[i][j], 0, 0, 0);
                }
        }
        if (more) {
            ca.store(lds_a[buf ^ 1], ra);
            cb.store(lds_b[buf ^ 1], rb);
        }
        __syncthreads();
        buf ^= 1;
    }
    static_assert(sizeof(lds_a) >= 4096 * (WTHREADS / 64), "the epilogue's per-wave scratch lives in the A stages");
    gemm_epilogue<MT, NT, true>(p, acc, m0, n0, wm, wn, r, h, z, 0, (lds_float*)(&lds_a[0][0]) + 1024 * wave);
}

int launch_fp8(const GemmP& p, dim3 grid, hipStream_t s) {
    if (p.a_type == CALM_ST_FP8_E5M2) hipLaunchKernelGGL(gemm_fp8w_kernel<true>, grid, dim3(WTHREADS), 0, s, p);
    else hipLaunchKernelGGL(gemm_fp8w_kernel<false>, grid, dim3(WTHREADS), 0, s, p);
    CALM_LAUNCH_CHECK();
    return 0;
}

// ---- per-tensor quantisers ---------------------------------------------------------------------------------------
namespace {
constexpr int QNT = 256;
__device__ __forceinline__ float ldq(const void* p, long i, int type) {
    return type == CALM_ST_BF16 ? (float)reinterpret_cast<const __bf16*>(p)[i] : reinterpret_cast<const float*>(p)[i];
}

// amax over the tensor: non-negative floats order like their bit patterns -> one atomicMax per block
__global__ __launch_bounds__(QNT) void amax_kernel(const void* __restrict__ x, int x_type, long n, unsigned* __restrict__ amax_bits) {
    __shared__ float red[4];
    float m = 0.f;
    bool bad = false;                                        // fmaxf drops NaN: a non-finite input must stay visible
    for (long i = (long)blockIdx.x * QNT + threadIdx.x; i < n; i += (long)gridDim.x * QNT) {
        const float a = fabsf(ldq(x, i, x_type));
        bad = bad || !(a <= 3.402823466e38f);                // NaN or infinity
        m = fmaxf(m, a);
    }
    m = block_max_256(m, red);
    if (threadIdx.x == 0) atomicMax(amax_bits, __float_as_uint(m));
    // (as unsigned bits a quiet NaN compares above every finite value and above +inf: it survives the atomic max)
    if (__syncthreads_or(bad) && threadIdx.x == 0) atomicMax(amax_bits, 0x7FC00000u);
}

// q = fp8(x * FP8_MAX / amax); state[1] = amax / FP8_MAX (the factor the GEMM epilogue multiplies back in).
// 4 elements per thread and trip: two v_cvt_pk_{fp8,bf8}_f32 fill one 32-bit word.
template <bool BF8>
__global__ __launch_bounds__(QNT) void quant_kernel(const void* __restrict__ x, int x_type, long n, float* __restrict__ state,
                                                    unsigned* __restrict__ q) {
    constexpr float FMAX = BF8 ? 57344.0f : 448.0f;
    const float amax = state[0];
    const float sc = amax > 0.f ? FMAX / amax : 1.0f;
    // a NaN / infinity among the inputs (amax is then NaN): the dequantisation factor becomes NaN, so every output of the
    // product that consumes this tensor is NaN — the clamp below would otherwise turn the offending element into a
    // finite -FMAX and hide the divergence from the loss and from GradScaler's inf check
    if (blockIdx.x == 0 && threadIdx.x == 0) state[1] = !(amax <= 3.402823466e38f) ? __uint_as_float(0x7FC00000u) : amax > 0.f ? amax / FMAX : 1.0f;
    const long n4 = n >> 2;                                  // n % 4 == 0 (checked by the host)
    for (long i = (long)blockIdx.x * QNT + threadIdx.x; i < n4; i += (long)gridDim.x * QNT) {
        // clamped: amax * (FMAX / amax) may round a hair above FMAX, which the conversion would turn into NaN
        auto sc1 = [&](long k) { return fminf(fmaxf(ldq(x, k, x_type) * sc, -FMAX), FMAX); };
        const float a = sc1(4 * i), b = sc1(4 * i + 1), c = sc1(4 * i + 2), d = sc1(4 * i + 3);
        int w = 0;
        if constexpr (BF8) {
            w = __builtin_amdgcn_cvt_pk_bf8_f32(a, b, w, false);
            w = __builtin_amdgcn_cvt_pk_bf8_f32(c, d, w, true);
        } else {
            w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
            w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
        }
        q[i] = (unsigned)w;
    }
}

// byte transpose [rows][cols] -> [cols][rows] (the fp8 weight copy the input-gradient product reads k-contiguously)
__global__ __launch_bounds__(256) void transpose_u8_kernel(const unsigned char* __restrict__ in, unsigned char* __restrict__ out,
                                                           int rows, int cols) {
    __shared__ unsigned char tile[64][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;              // 64 x 4 threads, 64 x 64 tile
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int rr = r0 + ty + 4 * k, cc = c0 + tx;
        if (rr < rows && cc < cols) tile[ty + 4 * k][tx] = in[(long)rr * cols + cc];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int rr = c0 + ty + 4 * k, cc = r0 + tx;                      // out row = in column
        if (rr < cols && cc < rows) out[(long)rr * rows + cc] = tile[tx][ty + 4 * k];
    }
}
}  // namespace

}  // namespace calm_gemm_detail

using namespace calm_gemm_detail;

extern "C" {

int calm_quantize_fp8(const void* x, int32_t x_type, int64_t n, void* q, int32_t q_type, float* state, void* stream) {
    if (!x || !q || !state || n <= 0 || (n & 3)) return CALM_E_INVAL;
    if (x_type != CALM_ST_F32 && x_type != CALM_ST_BF16) return CALM_E_INVAL;
    if (q_type != CALM_ST_FP8_E4M3 && q_type != CALM_ST_FP8_E5M2) return CALM_E_INVAL;
    hipStream_t s = as_stream(stream);
    hipError_t e = hipMemsetAsync(state, 0, 2 * sizeof(float), s);
    if (e != hipSuccess) return (int)e;
    long g = (n / 8 + QNT - 1) / QNT;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(amax_kernel, dim3((int)g), dim3(QNT), 0, s, x, x_type, (long)n, reinterpret_cast<unsigned*>(state));
    CALM_LAUNCH_CHECK();
    if (q_type == CALM_ST_FP8_E5M2)
        hipLaunchKernelGGL(quant_kernel<true>, dim3((int)g), dim3(QNT), 0, s, x, x_type, (long)n, state, reinterpret_cast<unsigned*>(q));
    else
        hipLaunchKernelGGL(quant_kernel<false>, dim3((int)g), dim3(QNT), 0, s, x, x_type, (long)n, state, reinterpret_cast<unsigned*>(q));
    CALM_LAUNCH_CHECK();
    return 0;
}

int calm_transpose_u8(const void* in, void* out, int32_t rows, int32_t cols, void* stream) {
    if (!in || !out || rows <= 0 || cols <= 0) return CALM_E_INVAL;
    hipLaunchKernelGGL(transpose_u8_kernel, dim3((cols + 63) / 64, (rows + 63) / 64), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const unsigned char*>(in), reinterpret_cast<unsigned char*>(out), rows, cols);
    CALM_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
