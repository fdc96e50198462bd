// Standalone check + timing of calm_gemm on bf16 tensors (no torch): links libcalmvit_hip.so through the C-ABI.
//   build: hipcc -O2 --offload-arch=gfx950 scripts/micro/pipe_gemm_check.cpp -Iinclude -Lcalm-vit-dte_amd -lcalmvit_hip \
//          -Wl,-rpath,'$ORIGIN/../../calm-vit-dte_amd' -o scripts/micro/pipe_gemm_check
//   run:   scripts/micro/pipe_gemm_check [M N K akc bkc [batch]]      (no arguments: a built-in list of shapes)
// Each shape: one launch checked against a CPU double-precision reference on sampled outputs, then timed (median of 20).
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "calm_vit.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

static uint16_t f2bf(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    u += 0x7FFF + ((u >> 16) & 1);
    return (uint16_t)(u >> 16);
}
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

struct Rng { uint64_t s; float next() { s = s * 6364136223846793005ull + 1442695040888963407ull; return ((s >> 40) / (float)(1 << 24)) * 2.f - 1.f; } };

static int run(int M, int N, int K, int akc, int bkc, int batch, int c_bf16) {
    const size_t na = (size_t)M * K * batch, nb = (size_t)N * K * batch, nc = (size_t)M * N * batch;
    std::vector<uint16_t> A(na), B(nb);
    Rng r{12345u + (uint64_t)M * 31 + N * 7 + K};
    for (auto& x : A) x = f2bf(r.next());
    for (auto& x : B) x = f2bf(r.next());
    void *dA, *dB, *dC, *dW = nullptr;
    CK(hipMalloc(&dA, na * 2)); CK(hipMalloc(&dB, nb * 2)); CK(hipMalloc(&dC, nc * 4));
    CK(hipMemcpy(dA, A.data(), na * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), nb * 2, hipMemcpyHostToDevice));
    CK(hipMemset(dC, 0x7f, nc * 4));
    calm_gemm_args g; memset(&g, 0, sizeof g);
    g.A = dA; g.B = dB; g.C = dC; g.M = M; g.N = N; g.K = K; g.batch0 = batch; g.batch1 = 1;
    if (akc) { g.a_rs = K; g.a_cs = 1; } else { g.a_rs = 1; g.a_cs = M; }
    if (bkc) { g.b_rs = K; g.b_cs = 1; } else { g.b_rs = 1; g.b_cs = N; }
    g.a_b0 = (int64_t)M * K; g.b_b0 = (int64_t)N * K; g.c_rs = N; g.c_b0 = (int64_t)M * N;
    g.alpha = 1.f; g.dtype = CALM_BF16; g.a_type = CALM_ST_BF16; g.b_type = CALM_ST_BF16; g.c_type = c_bf16 ? CALM_ST_BF16 : CALM_ST_F32;
    g.split_k = (M >= 4096) ? 1 : 0;
    int64_t ws = calm_gemm_workspace_bytes(&g);
    const bool stamp = getenv("STAMP") && ws == 0;
    if (stamp) ws = 1 << 20;
    if (ws > 0) { CK(hipMalloc(&dW, ws)); CK(hipMemset(dW, 0, ws)); g.workspace = dW; g.workspace_bytes = ws; }
    int rc = calm_gemm(&g, nullptr);
    if (rc) { printf("calm_gemm rc=%d\n", rc); return 1; }
    CK(hipDeviceSynchronize());
    std::vector<uint8_t> C(nc * 4);
    CK(hipMemcpy(C.data(), dC, nc * (c_bf16 ? 2 : 4), hipMemcpyDeviceToHost));
    // sampled check (all rows/cols near the edges + a random sample)
    double worst = 0, scale = 0;
    Rng q{99};
    const int nsamp = 20000;
    for (int s = 0; s < nsamp; ++s) {
        int b = (int)((q.next() * 0.5f + 0.5f) * batch) % batch;
        int m = s < 2000 ? (M - 1 - (s % std::min(M, 40))) : (int)((q.next() * 0.5f + 0.5f) * M) % M;
        int n = (s % 3 == 0) ? (N - 1 - (s % std::min(N, 24))) : (int)((q.next() * 0.5f + 0.5f) * N) % N;
        double acc = 0;
        for (int k = 0; k < K; ++k) {
            const float a = bf2f(A[(size_t)b * M * K + (akc ? (size_t)m * K + k : (size_t)k * M + m)]);
            const float w = bf2f(B[(size_t)b * N * K + (bkc ? (size_t)n * K + k : (size_t)k * N + n)]);
            acc += (double)a * w;
        }
        const size_t ci = (size_t)b * M * N + (size_t)m * N + n;
        const float got = c_bf16 ? bf2f(((uint16_t*)C.data())[ci]) : ((float*)C.data())[ci];
        worst = std::max(worst, std::fabs((double)got - acc));
        scale = std::max(scale, std::fabs(acc));
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> t;
    for (int i = 0; i < 23; ++i) {
        CK(hipEventRecord(e0, nullptr));
        calm_gemm(&g, nullptr);
        CK(hipEventRecord(e1, nullptr));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (i >= 3) t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    const double med = t[t.size() / 2], fl = 2.0 * M * N * K * batch;
    const double tol = c_bf16 ? 8e-3 : 3e-5 * std::sqrt((double)K);
    printf("M=%6d N=%5d K=%6d b=%3d %s/%s c=%s: err %.2e (scale %.1f) %s | %8.1f us %7.1f TF\n", M, N, K, batch,
           akc ? "k" : "r", bkc ? "k" : "r", c_bf16 ? "bf16" : "f32", worst / (scale + 1e-30), scale,
           worst <= tol * scale ? "ok" : "MISMATCH", med * 1e3, fl / med / 1e9);
    fflush(stdout);
    if (stamp) {   // diagnostic library (-DCALM_PIPE_STAMP): per workgroup, wave 0 / 7, item: {k-loop cycles, of which waiting, epilogue cycles, start}
        std::vector<unsigned long long> st((1 << 20) / 8);
        CK(hipMemcpy(st.data(), dW, 1 << 20, hipMemcpyDeviceToHost));
        unsigned long long tmin = ~0ull;
        for (int b = 0; b < 256; ++b) if (st[(b * 2 * 8) * 4 + 3]) tmin = std::min(tmin, st[(b * 2 * 8) * 4 + 3]);
        for (int w = 0; w < 2; ++w)
            for (int it = 0; it < 4; ++it) {
                double s0 = 0, s1 = 0, s2 = 0, s3 = 0; int n = 0;
                for (int b = 0; b < 256; ++b) {
                    const unsigned long long* d = &st[((b * 2 + w) * 8 + it) * 4];
                    if (!d[3]) continue;
                    s0 += d[0]; s1 += d[1]; s2 += d[2]; s3 += (double)(d[3] - tmin); ++n;
                }
                if (n) printf("   wave %d item %d (%3d wgs): k-loop %7.0f cyc (waiting %7.0f)  epilogue %7.0f  start +%7.0f\n", w ? 7 : 0, it, n, s0 / n, s1 / n, s2 / n, s3 / n);
            }
    }
    hipFree(dA); hipFree(dB); hipFree(dC); if (dW) hipFree(dW);
    return worst <= tol * scale ? 0 : 1;
}

int main(int argc, char** argv) {
    if (argc >= 6) return run(atoi(argv[1]), atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), argc > 6 ? atoi(argv[6]) : 1, argc > 7 ? atoi(argv[7]) : 1);
    int bad = 0;
    const int small[][6] = {{256, 256, 64, 1, 1, 1}, {256, 256, 128, 1, 1, 1}, {1024, 672, 672, 1, 1, 1}, {1000, 136, 72, 1, 1, 2},
                            {1024, 672, 1344, 1, 0, 1}, {672, 672, 8192, 0, 0, 1}, {528, 1056, 4104, 0, 0, 1}};
    for (auto& s : small) bad += run(s[0], s[1], s[2], s[3], s[4], s[5], 0);
    const int big[][3] = {{57344, 672, 672}, {57344, 1344, 672}, {57344, 672, 1344}, {45056, 528, 528}, {45056, 1056, 528},
                          {45056, 528, 1056}, {32768, 384, 384}, {32768, 768, 384}, {20480, 240, 240}, {20480, 480, 240}};
    for (auto& s : big) {
        bad += run(s[0], s[1], s[2], 1, 1, 1, 1);          // forward
        bad += run(s[0], s[2], s[1], 1, 0, 1, 1);          // data gradient: M x K' = (M x N') (N' x K')
        bad += run(s[1], s[2], s[0], 0, 0, 1, 0);          // weight gradient
    }
    printf(bad ? "FAILED: %d shapes\n" : "all ok\n", bad);
    return bad ? 1 : 0;
}
