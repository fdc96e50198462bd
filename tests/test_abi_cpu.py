"""The drop-in boundary without a GPU: libcalmvit_hip.so builds / loads on a CPU-only host, exports every entry point
include/calm_vit.h declares, the ctypes binding (calm-vit-dte_amd/_lib.py) names exactly those entry points, and the ABI
version of the library is the header's.  No compute entry point is called here."""
import ctypes
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "calm_vit.h")
LIB = os.path.join(ROOT, "calm-vit-dte_amd", "libcalmvit_hip.so")


def _declared():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"^\s*(?:int|int32_t|int64_t|const char\*)\s+(calm_\w+)\s*\(", text, flags=re.M)))


def _library():
    if not os.path.exists(LIB):
        sys.path.insert(0, ROOT)
        import __graft_entry__
        __graft_entry__.build()
    return ctypes.CDLL(LIB)


def test_header_declares_the_path_entry_points():
    names = _declared()
    for must in ("calm_gemm", "calm_attention_fwd", "calm_attention_bwd", "calm_layernorm_fwd", "calm_layernorm_bwd",
                 "calm_rope_fwd", "calm_sn_power_iter", "calm_cnn_residual_fwd", "calm_cnn_residual_bwd",
                 "calm_optim_step", "calm_collate_mix", "calm_abi_version"):
        assert must in names, must
    assert len(names) >= 35


def test_library_exports_every_declared_symbol_and_binding_matches():
    lib = _library()
    names = _declared()
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    sys.path.insert(0, ROOT)
    from importlib import import_module
    binding = import_module("calm_vit_dte_amd._lib")
    assert sorted(binding.SIGNATURES) == names          # the Python side binds exactly the declared C-ABI
    version = int(re.search(r"#define\s+CALM_ABI_VERSION\s+(\d+)", open(HEADER).read()).group(1))
    lib.calm_abi_version.restype = ctypes.c_int
    assert lib.calm_abi_version() == version
    lib.calm_build_info.restype = ctypes.c_char_p
    assert b"gfx950" in lib.calm_build_info()


def test_struct_layouts_match_the_header():
    """sizeof of the by-pointer structs as the C compiler lays them out vs the ctypes mirrors."""
    import subprocess
    import tempfile
    sys.path.insert(0, ROOT)
    from importlib import import_module
    binding = import_module("calm_vit_dte_amd._lib")
    src = '#include <stdio.h>\n#include "calm_vit.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(calm_gemm_args), ' \
          'sizeof(calm_sn_layer), sizeof(calm_sn_plan_info), sizeof(calm_optim_tensor), sizeof(calm_optim_hparams), ' \
          'sizeof(calm_gemm_plan));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        c, exe = os.path.join(d, "s.c"), os.path.join(d, "s")
        open(c, "w").write(src)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe], check=True)
        sizes = [int(x) for x in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split()]
    mirrors = [binding.GemmArgs, binding.SnLayer, binding.SnPlanInfo, binding.OptimTensor, binding.OptimHparams, binding.GemmPlan]
    assert sizes == [ctypes.sizeof(m) for m in mirrors]


def test_host_side_planning_entry_points_answer_without_a_gpu():
    """The planning half of the C-ABI is host code (no launch, no device query): the scratch a reduction entry point
    needs (calm_reduce_scratch_floats, ABI v7) and what a calm_gemm launch decomposes into (calm_gemm_describe).  The
    pointers handed to describe are never dereferenced."""
    sys.path.insert(0, ROOT)
    from importlib import import_module
    binding = import_module("calm_vit_dte_amd._lib")
    lib = binding.load()
    for op in (binding.RED_LAYERNORM_BWD, binding.RED_ROPE_BWD, binding.RED_LATENT_FWD, binding.RED_COLSUM, binding.RED_CNN_BWD):
        for rows, cols in ((1, 4), (57344, 672), (20480, 240), (3, 1344)):
            need = int(lib.calm_reduce_scratch_floats(op, rows, cols))
            assert 0 < need <= (1 << 22), (op, rows, cols, need)          # a few MB at most: the backend keeps 4 MB per stream
    assert int(lib.calm_reduce_scratch_floats(99, 10, 10)) == 0
    g = binding.GemmArgs()
    M, N, K = 57344, 1344, 672                                  # the bench's MLP forward on bf16 tensors
    g.A, g.B, g.C = 0x10000, 0x20000, 0x30000
    g.M, g.N, g.K = M, N, K
    g.a_rs, g.a_cs, g.b_rs, g.b_cs, g.c_rs = K, 1, K, 1, N
    g.batch0 = g.batch1 = 1
    g.alpha = 1.0
    g.dtype = 1
    g.a_type = g.b_type = g.c_type = binding.ST_BF16
    g.split_k = 1
    plan = binding.GemmPlan()
    assert lib.calm_gemm_describe(ctypes.byref(g), ctypes.byref(plan)) == 0
    assert plan.family == 3 and plan.tile_k == 64 and plan.grid == 256 and plan.threads == 512
    assert plan.tiles_m * plan.tile_m >= M and plan.tiles_n * plan.tile_n >= N and plan.items == plan.tiles_m * plan.tiles_n
    g.dtype, g.a_type, g.b_type, g.c_type = 0, 0, 0, 0          # the same product on fp32 tensors: 128-row tiles
    assert lib.calm_gemm_describe(ctypes.byref(g), ctypes.byref(plan)) == 0
    assert plan.family == 0 and plan.tile_m == 128 and plan.tile_n in (96, 128)
