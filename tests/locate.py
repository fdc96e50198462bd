"""Self-locating GEMM mismatch reports (VERDICT r3 #1 / ADVICE r3 high).

Round 3 saw `test_bf16_pipeline_linears_at_bench_size` fail twice with the same figure on ONE box of the pool — and the
failing output was not kept, so nothing could tell "one bad CU / LDS bank" from "one item position" from "first k-tile
after an epilogue".  `check_gemm()` therefore turns any bound violation of a calm_gemm output into a report that names
WHERE the wrong elements are in terms of the launch's own decomposition (calm_gemm_describe, ABI v7):

  * the count of offending elements and their (row, col) extent,
  * per offending element: tile, work item, persistent workgroup slot (item % 256), XCD (slot % 8), the round of that
    workgroup (item // 256), wave (4 x 2 layout of the pipelined family), 16-row strip mt and 16-column strip nt,
  * histograms over workgroup slot / XCD / wave / round — "all bad elements in one slot" is a CU, "spread over slots but
    one round" is an item position, "every tile's first rows" is an epilogue / first-k-tile problem,
  * the box identity (device name, CU count, ROCm / driver versions, whether the .so was built on this box),
  * optionally the bitwise diff against the same launch on the 256x128 family (CALM_GEMM_OPT_PIPE = 0).

The report is printed, attached to the AssertionError and written to gpurun_out/mismatch_<name>.json so that one
occurrence anywhere (the builder's gpurun calls or the driver's round-end run) localises the defect.
"""
import json
import os
import platform
import time
from collections import Counter

import torch


def box_identity():
    ident = {"host": platform.node(), "time": time.strftime("%Y-%m-%dT%H:%M:%S")}
    try:
        p = torch.cuda.get_device_properties(0)
        ident.update(device=p.name, cus=p.multi_processor_count, gcn_arch=getattr(p, "gcnArchName", "?"),
                     total_mem_gib=round(p.total_memory / 2 ** 30, 1), hip=torch.version.hip, torch=torch.__version__)
    except Exception as e:                                   # noqa: BLE001
        ident["device_error"] = repr(e)
    for path, key in (("/sys/module/amdgpu/version", "amdgpu_driver"), ("/opt/rocm/.info/version", "rocm")):
        try:
            ident[key] = open(path).read().strip()
        except OSError:
            pass
    try:
        import calm_vit_dte_amd as calm
        lib = calm._lib.LIB_PATH
        st = os.stat(lib)
        ident["lib"] = {"path": lib, "bytes": st.st_size, "mtime": time.strftime("%Y-%m-%dT%H:%M:%S", time.localtime(st.st_mtime)),
                        "build_info": calm._lib.load().calm_build_info().decode()}
    except Exception as e:                                   # noqa: BLE001
        ident["lib_error"] = repr(e)
    return ident


def item_of_linear(lin, n_items):
    """Inverse of gemm_bf16p.h::decode's XCD-aware order: the item index a persistent workgroup sees for linear tile
    index `lin` (items with equal index mod 8 are consecutive tiles)."""
    if n_items < 8:
        return lin
    q, rem = n_items >> 3, n_items & 7
    big = rem * (q + 1)
    if lin < big:
        x, idx = divmod(lin, q + 1)
    else:
        x, idx = divmod(lin - big, q)
        x += rem
    return idx * 8 + x


def locate(rows, cols, plan, batch_index=0, n_cols=None):
    """Map output elements (rows[i], cols[i]) of batch entry / k-slice `batch_index` to the launch decomposition."""
    tm, tn = plan["tile_m"], plan["tile_n"]
    tiles = plan["tiles_m"] * plan["tiles_n"]
    out = []
    for r, c in zip(rows, cols):
        t = (r // tm) * plan["tiles_n"] + c // tn
        lin = batch_index * tiles + t
        rec = {"row": int(r), "col": int(c), "tile": int(t)}
        if plan["family"] in (3, 4):                         # persistent pipelined families
            item = item_of_linear(lin, plan["items"])
            slot = item % plan["grid"]
            mt_w, nt_w = tm // 4, tn // 2                    # wave tile: 4 x 2 waves
            wave = ((r % tm) // mt_w) * 2 + (c % tn) // nt_w
            rec.update(item=int(item), slot=int(slot), xcd=int(slot % 8), round=int(item // plan["grid"]), wave=int(wave),
                       strip_mt=int((r % mt_w) // 16), strip_nt=int((c % nt_w) // 16), row_in_strip=int(r % 16))
        else:
            rec.update(item=int(lin), slot=int(lin % 256), xcd=int(lin % 8), round=int(lin // 256))
        out.append(rec)
    return out


def report(name, got, ref, bound, plan, extra=None, max_list=40):
    """Build (and persist) the report for a [M, N] output `got` against `ref` under the normalised inf-norm `bound`."""
    got32, ref32 = got.detach().float(), ref.detach().float()
    scale = float(ref32.abs().max())
    err = (got32 - ref32).abs()
    bad = err > bound * scale
    nbad = int(bad.sum())
    rep = {"name": name, "bound": bound, "scale": scale, "max_err_rel": float(err.max()) / max(scale, 1e-30),
           "n_bad": nbad, "shape": list(got.shape), "plan": plan, "box": box_identity()}
    if nbad:
        idx = bad.nonzero()
        rows, cols = idx[:, 0].tolist(), idx[:, 1].tolist()
        rep["row_extent"] = [min(rows), max(rows)]
        rep["col_extent"] = [min(cols), max(cols)]
        order = err[bad].argsort(descending=True)[:max_list].tolist()
        loc = locate([rows[i] for i in order], [cols[i] for i in order], plan)
        for rec, i in zip(loc, order):
            rec["err_rel"] = float(err[rows[i], cols[i]]) / max(scale, 1e-30)
            rec["got"], rec["ref"] = float(got32[rows[i], cols[i]]), float(ref32[rows[i], cols[i]])
        rep["worst"] = loc
        allloc = locate(rows[:20000], cols[:20000], plan)
        for key in ("slot", "xcd", "round", "wave", "strip_mt", "strip_nt", "row_in_strip", "tile"):
            if key in allloc[0]:
                rep["hist_" + key] = dict(Counter(r[key] for r in allloc).most_common(12))
    if extra:
        rep.update(extra)
    try:
        out_dir = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "gpurun_out")
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, f"mismatch_{name}.json"), "w") as f:
            json.dump(rep, f, indent=1)
    except OSError:
        pass
    return rep


def summary(rep):
    keep = {k: rep[k] for k in rep if k.startswith("hist_") or k in ("name", "n_bad", "max_err_rel", "bound", "row_extent",
                                                                       "col_extent", "plan", "family_diff")}
    keep["box"] = {k: rep["box"].get(k) for k in ("host", "device", "cus", "rocm", "amdgpu_driver")}
    keep["worst3"] = rep.get("worst", [])[:3]
    return json.dumps(keep)


def check_gemm(name, be, got, ref, bound, gemm_args, gemm_kwargs, cross_family=True):
    """Assert max|got - ref| / max|ref| < bound for the calm_gemm launch described by (gemm_args, gemm_kwargs) — the
    arguments `be.gemm` was called with, `got` being its C.  On a violation: self-locating report (see module doc)."""
    got32, ref32 = got.detach().float(), ref.detach().float()
    e = float((got32 - ref32).abs().max()) / max(float(ref32.abs().max()), 1e-30)
    if e < bound:
        return e
    plan = be.gemm_describe(*gemm_args, **gemm_kwargs)
    extra = {}
    if cross_family and plan["family"] == 3:
        # the same launch on the 256x128 / 128-row kernels (families agree bit for bit on every healthy box seen so far)
        alt = torch.empty_like(got)
        args = list(gemm_args)
        args[2] = alt
        prev = be.gemm_set_option(be.GEMM_OPT_PIPE, 0)
        try:
            be.gemm(*args, **gemm_kwargs)
            torch.cuda.synchronize()
        finally:
            be.gemm_set_option(be.GEMM_OPT_PIPE, prev)
        diff = (alt.view(torch.int16) != got.view(torch.int16)) if got.dtype == torch.bfloat16 else (alt != got)
        nd = int(diff.sum())
        extra["family_diff"] = {"n_diff_bits": nd, "alt_max_err_rel": float((alt.float() - ref32).abs().max()) / max(float(ref32.abs().max()), 1e-30)}
        if nd:
            idx = diff.nonzero()
            rows, cols = idx[:, 0].tolist()[:20000], idx[:, 1].tolist()[:20000]
            loc = locate(rows, cols, plan)
            for key in ("slot", "xcd", "round", "wave", "strip_mt", "strip_nt", "tile"):
                extra["family_diff"]["hist_" + key] = dict(Counter(r[key] for r in loc).most_common(12))
    rep = report(name, got, ref, bound, plan, extra)
    msg = summary(rep)
    print("GEMM MISMATCH REPORT " + msg)
    raise AssertionError(f"{name}: rel err {e:.3e} >= {bound:.3e}; {msg}")
