// Pipelined persistent bf16-tensor GEMM (gemm_bf16p.h): operand layouts A k-contiguous, B k-contiguous.
#include "gemm_bf16p.h"

namespace calm_gemm_detail {
int launch_pipe_kk(const GemmP& p, int mt, int nt, int grid, hipStream_t s) {
    return launch_pipe_layout<false, true, true>(p, mt, nt, grid, s);
}
}  // namespace calm_gemm_detail
