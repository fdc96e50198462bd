// Store-throughput microbenchmark: one 512-thread workgroup per CU writes `iters` tiles of 256 x 224 bf16 (112 KiB, rows of a
// [rows][672] bf16 matrix) in the access patterns an MFMA epilogue can produce.  Prints GB/s per CU and chip-wide.
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/store_rate.hip -o scripts/micro/store_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// PAT 0: per instruction 16 rows x 32 B, 8 B per lane (the swapped-operand accumulator layout, one 16x16 tile)
// PAT 1: 16 B per lane, 2 rows x 448 B per instruction (56 lanes active)
// PAT 2: 16 B per lane, 16 rows x 64 B per instruction (two adjacent 16x16 tiles per lane pair)
// PAT 3: 16 B per lane, fully contiguous 1 KiB per instruction (ceiling)
// PAT 4: 8 B per lane, 4 rows x 128 B per instruction
// PAT 5: PAT 1 with non-temporal stores;  PAT 6: PAT 1 through a buffer descriptor with sc1 (write-through)
template <int PAT>
__global__ __launch_bounds__(512, 2) void k(char* out, int iters, long tile_stride_rows) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const long RS = 1344;    // bytes per matrix row
    for (int it = 0; it < iters; ++it) {
        // tile origin: 256 rows x 448 B; workgroups walk disjoint row ranges
        char* t = out + ((long)(it * gridDim.x + blockIdx.x) * 256) * RS + (blockIdx.x % 3) * 448;
        char* w = t + (long)(wm * 64) * RS + wn * 224;     // wave strip: 64 rows x 224 B
        if (PAT == 0) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 7; ++nt) {
                    u32x2 v = {(unsigned)it, (unsigned)lane};
                    *(u32x2*)(w + (long)(16 * mt + (lane & 15)) * RS + 32 * nt + 8 * (lane >> 4)) = v;
                }
        } else if (PAT == 1) {
            // wave strip 64 rows x 224 B = 14 x 16 B per row; 4 rows per instruction (56 lanes)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                u32x4 v = {(unsigned)it, (unsigned)lane, 1u, 2u};
                const int r = 4 * i + lane / 14, c = lane % 14;
                if (lane < 56) *(u32x4*)(w + (long)r * RS + 16 * c) = v;
            }
        } else if (PAT == 2) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 7; nt += 2) {
                    u32x4 v = {(unsigned)it, (unsigned)lane, 1u, 2u};
                    if (nt + 1 < 7 || (lane >> 4) < 2)
                        *(u32x4*)(w + (long)(16 * mt + (lane & 15)) * RS + 32 * nt + 16 * (lane >> 4)) = v;
                }
        } else if (PAT == 3) {
            char* c = out + ((long)(it * gridDim.x + blockIdx.x) * 256) * RS;
#pragma unroll
            for (int i = 0; i < 14; ++i) {
                u32x4 v = {(unsigned)it, (unsigned)lane, 1u, 2u};
                *(u32x4*)(c + (long)(wave * 14 + i) * 1024 + lane * 16) = v;
            }
        } else if (PAT == 5) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                u32x4 v = {(unsigned)it, (unsigned)lane, 1u, 2u};
                const int r = 4 * i + lane / 14, c = lane % 14;
                if (lane < 56) __builtin_nontemporal_store(v, (u32x4*)(w + (long)r * RS + 16 * c));
            }
        } else if (PAT == 6) {
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, 0x7FFFFFF0, 0x00020000);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                u32x4 v = {(unsigned)it, (unsigned)lane, 1u, 2u};
                const int r = 4 * i + lane / 14, c = lane % 14;
                const unsigned off = lane < 56 ? (unsigned)((w - out) + (long)r * RS + 16 * c) : 0xFFFFFFFFu;
                __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 16);
            }
        } else if (PAT == 4) {
#pragma unroll
            for (int i = 0; i < 28; ++i) {
                u32x2 v = {(unsigned)it, (unsigned)lane};
                const int r = 4 * (i % 16) + (lane >> 4), c = (i / 16) * 128 + 8 * (lane & 15);
                if (c < 224) *(u32x2*)(w + (long)r * RS + c) = v;
            }
        }
    }
}

int main(int argc, char** argv) {
    const int iters = 30, grid = argc > 1 ? atoi(argv[1]) : 256;
    const size_t bytes = (size_t)iters * grid * 256 * 1344 + (1 << 20);
    char* d; CK(hipMalloc(&d, bytes));
    CK(hipMemset(d, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int pat = 0; pat < 7; ++pat) {
        std::vector<float> ts;
        for (int rep = 0; rep < 7; ++rep) {
            CK(hipEventRecord(e0));
            switch (pat) {
            case 0: hipLaunchKernelGGL(k<0>, dim3(grid), dim3(512), 0, 0, d, iters, 0); break;
            case 1: hipLaunchKernelGGL(k<1>, dim3(grid), dim3(512), 0, 0, d, iters, 0); break;
            case 2: hipLaunchKernelGGL(k<2>, dim3(grid), dim3(512), 0, 0, d, iters, 0); break;
            case 3: hipLaunchKernelGGL(k<3>, dim3(grid), dim3(512), 0, 0, d, iters, 0); break;
            case 4: hipLaunchKernelGGL(k<4>, dim3(grid), dim3(512), 0, 0, d, iters, 0); break;
            case 5: hipLaunchKernelGGL(k<5>, dim3(grid), dim3(512), 0, 0, d, iters, 0); break;
            case 6: hipLaunchKernelGGL(k<6>, dim3(grid), dim3(512), 0, 0, d, iters, 0); break;
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms);
        }
        std::sort(ts.begin(), ts.end());
        const double t = ts[ts.size() / 2] * 1e-3, b = (double)iters * grid * 256 * 448;
        printf("pattern %d: %.1f us per tile per CU, %.1f GB/s per CU, %.2f TB/s chip\n", pat, t / iters * 1e6, b / t / grid / 1e9, b / t / 1e12);
    }
    return 0;
}
