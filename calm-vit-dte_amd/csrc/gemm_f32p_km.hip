// Pipelined persistent fp32-tensor GEMM (gemm_bf16p.h, element type float): operand layouts A k-contiguous, B row-contiguous.
#include "gemm_bf16p.h"

namespace calm_gemm_detail {
int launch_pipe32_km(const GemmP& p, int mt, int nt, int grid, hipStream_t s) {
    return launch_pipe_layout<true, true, false>(p, mt, nt, grid, s);
}
}  // namespace calm_gemm_detail
