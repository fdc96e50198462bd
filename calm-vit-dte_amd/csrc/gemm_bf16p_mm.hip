// Pipelined persistent bf16-tensor GEMM (gemm_bf16p.h): operand layouts A row-contiguous, B row-contiguous.
#include "gemm_bf16p.h"

namespace calm_gemm_detail {
int launch_pipe_mm(const GemmP& p, int mt, int nt, int grid, hipStream_t s) {
    return launch_pipe_layout<false, false, false>(p, mt, nt, grid, s);
}
}  // namespace calm_gemm_detail
