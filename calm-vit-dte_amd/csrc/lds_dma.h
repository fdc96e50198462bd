// LDS-DMA helpers shared by the pipelined GEMM family (gemm_bf16p.h) and the pipelined attention forward
// (attention_bf16_fwd2.h): global -> LDS without VGPRs (global_load_lds_dwordx4: lane l's 16 bytes land at LDS byte
// M0 + 16 l).  Inline asm on purpose — issued through the builtin, hipcc (ROCm 7.2) drains the loads with
// s_waitcnt vmcnt(0) before the first ds_read that follows; the callers place counted waits themselves.
#pragma once
#include "common.h"

namespace calm_lds_dma {

// 16 zero bytes in device memory (one copy per translation unit: no relocatable device code): source of the LDS-DMA
// lanes that fall past K in the last k-tile
static __device__ __attribute__((aligned(16))) unsigned calm_zero_block[4];

// one LDS-DMA instruction: lane l's 16 bytes at `base + voff` -> LDS byte lds_dst + 16 l (lds_dst wave-uniform)
__device__ __forceinline__ void glds16(const void* base, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    // `base` is wave-uniform by construction; say so (an "s" operand the compiler holds in VGPRs does not assemble)
    const unsigned long long b64 = reinterpret_cast<unsigned long long>(base);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b64);           // (the builtin returns int:
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b64 >> 32));   //  widen as unsigned)
    const unsigned long long bu = ((unsigned long long)hi << 32) | lo;
    lds_dst = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_dst);
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(bu), "s"(lds_dst) : "memory");
}
// wave-uniform 64-bit value as an SGPR pair (an "s" asm operand the compiler holds in VGPRs does not assemble)
__device__ __forceinline__ unsigned long long pipe_uniform64(const void* p) {
    const unsigned long long b64 = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b64);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b64 >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
// lean form for the k-loop: `base` already uniform (pipe_uniform64, once per k-tile), M0 declared clobbered instead
// of saved and restored — three instructions per piece
__device__ __forceinline__ void glds16_u(unsigned long long base, unsigned voff, unsigned lds_dst) {
    lds_dst = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_dst);
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :: "v"(voff), "s"(base), "s"(lds_dst) : "memory", "m0");
}
// the same with a full per-lane address (last k-tile of a reduction whose length is not a multiple of 64)
__device__ __forceinline__ void glds16_addr(const void* addr, unsigned lds_dst) {
    unsigned keep;
    lds_dst = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_dst);
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(addr), "s"(lds_dst) : "memory");
}

// s_waitcnt vmcnt(k) with the largest implemented k <= n (n wave-uniform, computed at run time from a ledger of the
// wave's vector-memory instructions: waiting for more than necessary is always safe)
#define CALM_VMW(k) asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory")
__device__ __forceinline__ void vm_wait_le(int n) {
    if (n >= 16) {
        if (n >= 56) CALM_VMW(56);
        else if (n >= 48) CALM_VMW(48);
        else if (n >= 40) CALM_VMW(40);
        else if (n >= 32) CALM_VMW(32);
        else if (n >= 28) CALM_VMW(28);
        else if (n >= 24) CALM_VMW(24);
        else if (n >= 20) CALM_VMW(20);
        else CALM_VMW(16);
    } else if (n >= 8) {
        if (n >= 14) CALM_VMW(14);
        else if (n >= 12) CALM_VMW(12);
        else if (n >= 10) CALM_VMW(10);
        else CALM_VMW(8);
    } else {
        if (n >= 7) CALM_VMW(7);
        else if (n >= 6) CALM_VMW(6);
        else if (n >= 5) CALM_VMW(5);
        else if (n >= 4) CALM_VMW(4);
        else if (n >= 3) CALM_VMW(3);
        else if (n >= 2) CALM_VMW(2);
        else if (n >= 1) CALM_VMW(1);
        else CALM_VMW(0);
    }
}

// buffer descriptor over `bytes` bytes at `base` (wave-uniform): stores at offset 0xFFFFFFFF are dropped by the range
// check, so a masked lane still ISSUES its store and the wave's store count stays a constant of the code
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, long bytes) {
    const unsigned long long b = reinterpret_cast<unsigned long long>(base);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
    const int n = __builtin_amdgcn_readfirstlane((int)(bytes > 0xFFFFFFF0l ? 0xFFFFFFF0l : bytes));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, n, 0x00020000);
}

}  // namespace calm_lds_dma
