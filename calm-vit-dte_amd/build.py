"""Compile the HIP sources in csrc/ into libcalmvit_hip.so (in-tree, gfx950 only).

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so is
git-ignored but travels to the GPU box with the repo snapshot.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcalmvit_hip.so")
SOURCES = ["gemm.hip", "gemm_f32.hip", "gemm_bf16.hip", "gemm_bf16p_kk.hip", "gemm_bf16p_km.hip", "gemm_bf16p_mm.hip", "gemm_f32p_kk.hip", "gemm_f32p_km.hip", "gemm_f32p_mm.hip", "gemm_fp8.hip", "norm_act.hip", "spectral.hip", "tokens_conv.hip", "cnn_fused.hip",
           "attention_fused.hip", "attention_bf16.hip", "optim.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the CALM-ViT HIP library cannot be built")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False):
    """Build (if stale) and return the path of libcalmvit_hip.so."""
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")]
    headers.append(os.path.join(HERE, "..", "include", "calm_vit.h"))
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + headers):
            cmd = [hipcc] + FLAGS + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            jobs.append((src, subprocess.Popen(cmd)))          # one compiler process per stale translation unit
        objs.append(o)
    failed = [src for src, pr in jobs if pr.wait() != 0]
    if failed:
        raise RuntimeError(f"hipcc failed for {failed}")
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build_library(verbose=True))
