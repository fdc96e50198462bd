"""Spectral-normed weights with the reference's state-dict surface.

The reference wraps every Linear/Conv in hook-based `torch.nn.utils.spectral_norm`
(Vi_Tools_CNN_less_V2.py:137-205,380-384; CALM_ViT_V2.py:50-52,62-66), which yields the keys
`<name>.weight_orig` (Parameter), `<name>.weight_u`, `<name>.weight_v` (buffers) [+ `<name>.bias`]
and runs one power iteration per training forward.  Here the iteration of ALL layers under the
outermost running module is done by three kernel launches (calm_sn_power_iter) when that module's
forward starts; each layer then consumes its device-resident sigma, which the GEMM epilogues divide
by (W_orig / sigma is never materialised).
"""
import math

import torch

from .backend import bf16_pipeline, get_backend

W16_GEN_ATTR = "_calm_w16_gen"
W16_ATTR = "_calm_w16"      # on a weight_orig Parameter: its bf16 copy for this step (bf16 pipeline), read by ops._wop

_scope_depth = 0


class SpectralWeight(torch.nn.Module):
    """Holds weight_orig / weight_u / weight_v (/ bias) for one spectral-normed layer."""

    def __init__(self, weight_shape, bias=False):
        super().__init__()
        weight_shape = tuple(weight_shape)
        rows = weight_shape[0]
        cols = 1
        for d in weight_shape[1:]:
            cols *= d
        bound = 1.0 / math.sqrt(cols)            # nn.Linear / nn.Conv2d default init
        self.weight_orig = torch.nn.Parameter(torch.empty(weight_shape).uniform_(-bound, bound))
        if bias:
            self.bias = torch.nn.Parameter(torch.empty(rows).uniform_(-bound, bound))
        else:
            self.register_parameter("bias", None)
        self.register_buffer("weight_u", torch.nn.functional.normalize(torch.randn(rows), dim=0, eps=1e-12))
        self.register_buffer("weight_v", torch.nn.functional.normalize(torch.randn(cols), dim=0, eps=1e-12))
        self.register_buffer("_sigma", torch.ones(1), persistent=False)
        self.rows, self.cols = rows, cols
        self._fresh = False
        self._own_plan = None

    def matrix(self):
        return self.weight_orig.view(self.rows, self.cols)

    def sn_tensors(self):
        return (self.weight_orig.detach().view(self.rows, self.cols), self.weight_u, self.weight_v, self._sigma)

    def sigma(self):
        """sigma for this forward: taken from the enclosing batched update, or (layer used on its
        own) computed now with a one-layer plan."""
        if self._fresh:
            self.__dict__["_fresh"] = False
            return self._sigma
        be = get_backend()
        t = self.sn_tensors()
        key = tuple(x.data_ptr() for x in t)
        if self._own_plan is None or self._own_plan.key != key:
            self._own_plan = be.sn_plan([t])
        be.sn_power_iter(self._own_plan, self.training)
        _refresh_bf16_weights(self, [self])
        return self._sigma


class SNLinear(SpectralWeight):
    """sn(nn.Linear(in, out, bias)).  forward(x) = x W_orig^T / sigma + bias."""

    def __init__(self, in_features, out_features, bias=False):
        super().__init__((out_features, in_features), bias=bias)
        self.in_features, self.out_features = in_features, out_features

    def forward(self, x, act=0, ls=None, residual=None, out16=False):
        from .ops import SNLinearFn
        return SNLinearFn.apply(x, self.weight_orig, self.bias, ls, residual, self.weight_u, self.weight_v,
                                self.sigma(), act, out16)

    def extra_repr(self):
        return f"in_features={self.in_features}, out_features={self.out_features}, bias={self.bias is not None}"


class SNConv2d(SpectralWeight):
    """sn(nn.Conv2d(...)) parameter holder for Block.proj / ViT.proj; the convolutions themselves run
    fused in ops.CnnResidualFn on the channels-last token grid."""

    def __init__(self, in_channels, out_channels, kernel_size, groups=1, padding=0, bias=True):
        super().__init__((out_channels, in_channels // groups, kernel_size, kernel_size), bias=bias)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.groups, self.padding = kernel_size, groups, padding

    def forward(self, x):
        raise NotImplementedError("SNConv2d layers are executed through the fused CNN residual "
                                  "(CnnResidual module); call the parent Block / ViT instead")

    def extra_repr(self):
        return (f"{self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}, "
                f"groups={self.groups}, padding={self.padding}")


class sn_scope:
    """Context entered by every module forward of the path.  The OUTERMOST one runs the batched
    power iteration for all SpectralWeight layers below its module."""

    def __init__(self, module):
        self.module = module

    def __enter__(self):
        global _scope_depth
        _scope_depth += 1
        if _scope_depth == 1:
            _batched_update(self.module)
        return self

    def __exit__(self, *exc):
        global _scope_depth
        _scope_depth -= 1
        if _scope_depth == 0:
            for m in self.module.__dict__.get("_sn_layers") or ():
                m.__dict__["_fresh"] = False      # a sigma is valid for the forward it was computed in only
                                                  # (through __dict__: nn.Module.__setattr__ costs 2 us x 296 layers x 2 per step)
        return False


def _batched_update(root):
    layers = root.__dict__.get("_sn_layers")
    if layers is None:
        layers = [m for m in root.modules() if isinstance(m, SpectralWeight)]
        root.__dict__["_sn_layers"] = layers
    if not layers:
        return
    be = get_backend()
    tensors = [m.sn_tensors() for m in layers]
    key = tuple(x.data_ptr() for t in tensors for x in t)
    plan = root.__dict__.get("_sn_plan")
    if plan is None or plan.key != key:
        plan = be.sn_plan(tensors)
        root.__dict__["_sn_plan"] = plan
    be.sn_power_iter(plan, root.training)
    _refresh_bf16_weights(root, layers)
    for m in layers:
        m.__dict__["_fresh"] = True


def _refresh_bf16_weights(root, layers):
    """bf16 pipeline: one launch rewrites the bf16 copy of every weight below `root` (W_orig rounded to nearest even —
    what autocast's cast of a Linear weight does; sigma stays a fp32 epilogue factor).  The copies live in one flat
    buffer per root; each weight_orig Parameter carries its view (W16_ATTR) for ops to use as the GEMM operand."""
    if not bf16_pipeline():
        return
    be = get_backend()
    ws = [m.weight_orig for m in layers]
    key = tuple(w.data_ptr() for w in ws)
    st = root.__dict__.get("_w16_state")
    if st is None or st["key"] != key:
        offs, n = [], 0
        for w in ws:
            offs.append(n)
            n += (w.numel() + 7) & ~7                          # 16-byte aligned slices
        flat = torch.empty(n, dtype=torch.bfloat16, device=ws[0].device)
        views = [flat[o:o + w.numel()].view(w.shape) for o, w in zip(offs, ws)]
        plan = be.cast_plan([(w.detach().reshape(-1), v.view(-1)) for w, v in zip(ws, views)])
        st = {"key": key, "flat": flat, "views": views, "plan": plan, "gen": 0}
        root.__dict__["_w16_state"] = st
    be.cast_run(st["plan"])
    # the copies are rewritten IN PLACE by every forward and reach backward outside save_for_backward (no autograd
    # version counter): a generation number lets a backward detect that a later forward has replaced the weights its
    # forward multiplied with (forward, optimizer step, forward, then the first forward's backward)
    st["gen"] += 1
    for w, v in zip(ws, st["views"]):
        setattr(w, W16_ATTR, v)
        setattr(w, W16_GEN_ATTR, st["gen"])
