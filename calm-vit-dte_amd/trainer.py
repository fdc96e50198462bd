"""Per-rank training step and single-node data parallelism for the CALM-ViT path.

Replaces the Spark TorchDistributor launcher + DDP wrapper of
/root/reference/CALM-ViT/distributed_trainer_cls.py (train(): 25-114, __main__: 116-175) with
torch.distributed over RCCL (one process per GPU, env:// rendezvous from RANK/LOCAL_RANK/WORLD_SIZE
as set by torchrun):

  * `init_distributed()`               <- dist.init_process_group(...)                 (cls:46-51)
  * `sync_module_states(model)`        <- DDP ctor broadcast of params+buffers         (cls:55)
  * `BucketedGradReducer`              <- DDP's bucketed gradient all-reduce           (cls:55,87):
        gradients are packed into a few large flat buckets (sized for xGMI: point-to-point links,
        per-link-bound rings -> fewer, larger collectives) as soon as backward has produced them, and
        all-reduced (mean) on a SIDE HIP stream while the rest of backward keeps running.
  * `TrainStep`                        <- one iteration of the step loop               (cls:79-96):
        forward, CE with soft targets, backward, unscale / clip_grad_norm_(1.0), AdamW, zero_grad.

The model also works under stock `torch.nn.parallel.DistributedDataParallel` (that is what the
reference's train() does with it); the reducer here is the MI355X-tuned equivalent.
"""
import os
import weakref

import torch
import torch.distributed as dist


def init_distributed(use_gpu=True):
    """Returns (rank, local_rank, world_size).  No-op (0,0,1) when not launched by torchrun."""
    if int(os.environ.get("WORLD_SIZE", "1")) <= 1:
        return 0, 0, 1
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local_rank = int(os.environ.get("LOCAL_RANK", rank))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not dist.is_initialized():
        backend = os.environ.get("CALM_DIST_BACKEND", "nccl" if use_gpu else "gloo")
        if use_gpu and backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        elif use_gpu:
            # rehearsal mode: several ranks may share one GPU (CALM_LOCAL_DEVICE) and exchange through gloo
            torch.cuda.set_device(int(os.environ.get("CALM_LOCAL_DEVICE", local_rank)))
            dist.init_process_group(backend=backend)
        else:
            dist.init_process_group(backend="gloo")
    return rank, local_rank, world


def _flat_groups(tensors, max_bytes):
    groups, cur, size = [], [], 0
    for t in tensors:
        nb = t.numel() * t.element_size()
        if cur and size + nb > max_bytes:
            groups.append(cur)
            cur, size = [], 0
        cur.append(t)
        size += nb
    if cur:
        groups.append(cur)
    return groups


@torch.no_grad()
def sync_module_states(model, src=0, bucket_bytes=256 << 20):
    """Broadcast parameters and buffers from rank `src` (what DDP's constructor does)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    tensors = [p.data for p in model.parameters()] + [b.data for b in model.buffers()]
    by_type = {}
    for t in tensors:
        by_type.setdefault((t.dtype, t.device), []).append(t)
    for group in by_type.values():
        for chunk in _flat_groups(group, bucket_bytes):
            flat = torch.cat([t.reshape(-1) for t in chunk])
            dist.broadcast(flat, src)
            off = 0
            for t in chunk:
                t.copy_(flat[off:off + t.numel()].view_as(t))
                off += t.numel()


def _bucket_groups(params, bucket_bytes, tail_bytes):
    """Parameters in gradient-arrival order -> buckets of ~bucket_bytes; the LAST bucket (the gradients that arrive
    when backward is about to end, whose all-reduce nothing is left to overlap with) holds at most ~tail_bytes."""
    sizes = [p.numel() * p.element_size() for p in params]
    tail, acc = len(params), 0
    while tail > 1 and acc + sizes[tail - 1] <= tail_bytes:
        tail -= 1
        acc += sizes[tail]
    if tail == len(params):                     # the last parameter alone exceeds tail_bytes
        tail = len(params) - 1
    groups = _flat_groups(params[:tail], bucket_bytes) if tail > 0 else []
    groups.append(params[tail:])
    return groups


class BucketedGradReducer:
    """Mean all-reduce of all gradients in large flat buckets, overlapped with backward.

    Parameters are bucketed in reverse registration order (~ the order backward produces their gradients): 64 MiB
    buckets — xGMI rings are per-link bound, so few large collectives — and a small last one, because the tail of
    backward has nothing left to hide a collective behind.  When the last gradient of a bucket has been accumulated,
    the bucket is packed and its all-reduce is launched on a side stream (RCCL over xGMI on the GPU: ReduceOp.AVG, the
    mean inside the collective; gloo on CPU in tests: SUM then a scale).  `finish()` (call after backward) makes the
    compute stream wait for the collectives and re-points every `p.grad` at its slice of the reduced bucket — no copy
    back: the optimizer-side step reads the buckets in place, and since those addresses never change its gradient
    pointer table is uploaded once.  (The gradients are PACKED into the bucket by one `_foreach_copy_` per bucket: the
    optimizer-side step releases `.grad`, so autograd's AccumulateGrad adopts the freshly produced gradient buffers —
    keeping `.grad` pointed at the bucket views across steps would instead make autograd add into them, one extra
    read-modify-write launch per parameter.)"""

    def __init__(self, model, bucket_mb=64, process_group=None, tail_mb=8, force=False):
        """force: run the buckets and collectives in a world of one as well (a single-GPU RCCL rehearsal of exactly the
        code path the 8-GPU job takes; needs an initialised process group)."""
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.enabled = self.world > 1 or (force and dist.is_initialized())
        self.buckets = []
        self._works = []
        if not self.enabled:
            return
        dev = self.params[0].device
        self.on_gpu = dev.type == "cuda"
        self.avg_in_collective = dist.get_backend(process_group) == "nccl"
        self.side = torch.cuda.Stream(device=dev) if self.on_gpu else None
        for group in _bucket_groups(list(reversed(self.params)), bucket_mb << 20, tail_mb << 20):
            n = sum(p.numel() for p in group)
            flat = torch.zeros(n, dtype=group[0].dtype, device=dev)
            views, off = [], 0
            for p in group:
                views.append(flat[off:off + p.numel()].view_as(p))
                off += p.numel()
            self.buckets.append({"params": group, "flat": flat, "views": views, "pending": len(group)})
        self._where = {}
        for bi, b in enumerate(self.buckets):
            for p in b["params"]:
                self._where[p] = bi
                p.register_post_accumulate_grad_hook(self._hook)

    def _hook(self, p):
        b = self.buckets[self._where[p]]
        b["pending"] -= 1
        if b["pending"] == 0:
            self._launch(b)

    def _launch(self, b):
        grads = [p.grad for p in b["params"]]
        op = dist.ReduceOp.AVG if self.avg_in_collective else dist.ReduceOp.SUM
        if self.on_gpu:
            # the side stream starts after everything enqueued so far: the gradients exist, and the previous step's
            # optimizer-side kernels (which read this bucket in place) have been issued before them.  The collective is
            # issued SYNCHRONOUSLY (async_op=False) on the side stream: for ProcessGroupNCCL that blocks nobody — it
            # makes RCCL's stream wait for the side stream, enqueues the all-reduce there and makes the side stream wait
            # for its end event — and it is the form that a stream capture supports (fork from the capturing stream by
            # wait_stream, collectives on the fork with async_op=False, join by wait_stream: no Work object outlives
            # the call, nothing is waited on from a non-captured stream; round 3 used async_op=True + work.wait() in
            # finish() and ended in a segfault inside capture_end, gpurun_out/rccl.log)
            self.side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.side):
                torch._foreach_copy_(b["views"], grads)
                dist.all_reduce(b["flat"], op=op, group=self.pg, async_op=False)
                if not self.avg_in_collective:
                    b["flat"].mul_(1.0 / self.world)
            work = None
        else:
            torch._foreach_copy_(b["views"], grads)
            work = dist.all_reduce(b["flat"], op=op, group=self.pg, async_op=True)
        self._works.append((work, b))

    @torch.no_grad()
    def finish(self):
        if not self.enabled:
            return
        for b in self.buckets:                      # bucket not launched by the hooks: some of its
            if b["pending"] > 0:                    # parameters got no gradient this step
                for p in b["params"]:
                    if p.grad is None:
                        p.grad = torch.zeros_like(p)
                self._launch(b)
        inv = 1.0 / self.world
        for work, b in self._works:
            if work is not None:                    # CPU (gloo) path: asynchronous collectives, waited for here
                work.wait()
                if not self.avg_in_collective:
                    b["flat"].mul_(inv)
            for p, v in zip(b["params"], b["views"]):
                p.grad = v                          # the reduced gradient, in place in its bucket
        if self.on_gpu:
            torch.cuda.current_stream().wait_stream(self.side)
        self._works.clear()
        for b in self.buckets:
            b["pending"] = len(b["params"])


def soft_target_cross_entropy(logits, soft_targets):
    """torch.nn.CrossEntropyLoss() with class-probability targets (cls:63,86; CutMix/MixUp labels)."""
    return torch.nn.functional.cross_entropy(logits, soft_targets)


class TrainStep:
    """One iteration of distributed_trainer_cls.py:79-96 (per rank)."""

    def __init__(self, model, optimizer, reducer=None, max_norm=1.0, scaler=None, autocast_dtype=None):
        self.model, self.opt, self.reducer = model, optimizer, reducer
        self.max_norm = max_norm
        self.scaler = scaler
        self.autocast_dtype = autocast_dtype          # torch.bfloat16: the reference's `with autocast(...)` (cls:84)
        self.params = [p for p in model.parameters() if p.requires_grad]

    def __call__(self, x, y_soft):
        with torch.autocast(device_type="cuda", dtype=self.autocast_dtype or torch.bfloat16,
                            enabled=self.autocast_dtype is not None and x.is_cuda):        # cls:84
            y_hat, _ = self.model(x)                                       # cls:85
            loss = soft_target_cross_entropy(y_hat.squeeze(), y_soft)      # cls:86
        return self._finish(loss, y_hat)

    def _finish(self, loss, y_hat):
        """backward, gradient exchange, unscale / clip / optimizer step, zero_grad (cls:87-96)."""
        if self.scaler is not None:
            self.scaler.scale(loss).backward()                             # cls:87
        else:
            loss.backward()
        if self.reducer is not None:
            self.reducer.finish()
        if isinstance(self.opt, FusedClipAdamW):                           # cls:88-96 in three launches
            self.opt.max_norm = self.max_norm
            if self.scaler is not None and self.scaler.is_enabled():
                # GradScaler semantics on device, no host sync: unscale + inf check + skip-on-inf happen inside
                # calm_optim_step; the scale then backs off (x backoff) on inf/NaN or grows (x growth) after
                # growth_interval clean steps, exactly as scaler.step() / scaler.update() would do it
                sc = self.scaler
                if not hasattr(self, "_clean_steps"):
                    self._clean_steps = torch.zeros((), dtype=torch.int32, device=loss.device)
                    self._one = torch.ones((), dtype=torch.float32, device=loss.device)
                scale = sc.scale(self._one)                      # the current scale as a device tensor, no host sync
                stats = self.opt.step(grad_scale=scale)
                bad = stats[1] > 0
                # the counter is updated IN PLACE: a captured step (GraphedTrainStep) replays these kernels on the
                # addresses seen at capture, so a rebound tensor would leave every replay reading the pre-capture value
                # (the scale then never grows — or doubles on every replay from recycled memory)
                self._clean_steps.copy_(torch.where(bad, torch.zeros_like(self._clean_steps), self._clean_steps + 1))
                grow = self._clean_steps >= sc.get_growth_interval()
                new_scale = torch.where(bad, scale * sc.get_backoff_factor(),
                                        torch.where(grow, scale * sc.get_growth_factor(), scale))
                self._clean_steps.copy_(torch.where(grow, torch.zeros_like(self._clean_steps), self._clean_steps))
                sc.update(new_scale.detach())          # GradScaler.update(tensor) copies into its scale tensor in place
            else:
                self.opt.step()
            return loss.detach(), y_hat.detach()
        if self.scaler is not None:
            self.scaler.unscale_(self.opt)                                 # cls:88
        torch.nn.utils.clip_grad_norm_(self.params, max_norm=self.max_norm, error_if_nonfinite=False)   # cls:92
        if self.scaler is not None:
            self.scaler.step(self.opt)                                     # cls:93-94
            self.scaler.update()
        else:
            self.opt.step()
        self.opt.zero_grad()                                               # cls:96
        return loss.detach(), y_hat.detach()


class RegTrainStep(TrainStep):
    """One iteration of the generative trainer (distributed_trainer_reg.py:71-95): the `generate=True` model returns
    tokens [B,S,3S]; they are viewed as the image [B,3,S,S] (:78-79), loss = HuberLoss(img, x) + 0.1 * kl_loss
    (:81,87), then the same scale / clip(1.0) / optimizer step as the classification trainer."""

    def __init__(self, model, optimizer, reducer=None, max_norm=1.0, scaler=None, autocast_dtype=None, kl_weight=0.1):
        super().__init__(model, optimizer, reducer, max_norm=max_norm, scaler=scaler, autocast_dtype=autocast_dtype)
        self.kl_weight = kl_weight

    def __call__(self, x, _y=None):
        S = x.shape[-1]
        with torch.autocast(device_type="cuda", dtype=self.autocast_dtype or torch.bfloat16,
                            enabled=self.autocast_dtype is not None and x.is_cuda):        # reg:76
            y_hat, kl = self.model(x)                                      # reg:77
            img = y_hat.reshape(-1, S, S, 3).permute(0, 3, 1, 2)           # reg:78-79
            loss = torch.nn.functional.huber_loss(img, x) + kl * self.kl_weight   # reg:81,87
        return self._finish(loss, img)


class DeviceCollate:
    """The batch-level part of the reference's input pipeline on the device:
    ToDtype(scale=True) + Normalize + RandomHorizontalFlip + RandomChoice([CutMix(alpha=1.0), MixUp(alpha=0.8)])
    with 1000-way soft labels (distributed_trainer_cls.py:58-61,128-139, torchvision.transforms.v2 semantics), one kernel
    pass over the uint8 batch (calm_collate_mix).  The decode / resize / colour augmentations stay with the loader.
    torchvision is not installed in the build image, so the semantics are restated from its documentation."""

    MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)

    def __init__(self, num_classes=1000, cutmix_alpha=1.0, mixup_alpha=0.8, flip_p=0.5, seed=None):
        import numpy as np
        self.num_classes, self.cutmix_alpha, self.mixup_alpha, self.flip_p = num_classes, cutmix_alpha, mixup_alpha, flip_p
        self.rng = np.random.default_rng(seed)

    def draw(self, B, H, W):
        """(mode, lam, box, flips): the random decisions of one batch (host side, tiny)."""
        rng = self.rng
        mode = 2 if rng.random() < 0.5 else 1                               # RandomChoice([cut_mix, mix_up])
        alpha = self.cutmix_alpha if mode == 2 else self.mixup_alpha
        lam = float(rng.beta(alpha, alpha))
        box = None
        if mode == 2:
            box, lam = SoftMixCollate.cutmix_box(lam, int(rng.integers(0, W)), int(rng.integers(0, H)), H, W)
        flips = torch.from_numpy((rng.random(B) < self.flip_p).astype("uint8"))
        return mode, lam, box, flips

    def __call__(self, img_u8, labels, decisions=None, crop=None, tokens=False):
        """crop=(H, W): RandomCrop of every sample to H x W inside the (resized) source, corners drawn uniformly as
        torchvision's RandomCrop.get_params does (cls:130); tokens=True: the batch comes out as the row tokens
        [B, H, 3W] of the first Block (Vi_Tools_CNN_less_V2.py:389-391) — feed it to `model.autoencoder` / a ViT whose
        first Block skips the tokenisation — instead of the image [B,3,H,W]."""
        from .backend import get_backend
        B, _, Hs, Ws = img_u8.shape
        H, W = crop if crop is not None else (Hs, Ws)
        mode, lam, box, flips = decisions if decisions is not None else self.draw(B, H, W)
        corners = None
        if crop is not None:
            import numpy as np
            corners = torch.from_numpy(np.stack([self.rng.integers(0, Hs - H + 1, B), self.rng.integers(0, Ws - W + 1, B)],
                                                axis=1).astype("int32")).to(img_u8.device)
        out = torch.empty((B, H, 3 * W) if tokens else (B, 3, H, W), dtype=torch.float32, device=img_u8.device)
        if crop is None and not tokens:
            get_backend().collate_mix(img_u8, flips.to(img_u8.device), out, mode, lam, box, self.MEAN, self.STD)
        else:
            get_backend().collate_crop_mix(img_u8, corners, flips.to(img_u8.device), out, mode, lam, box, self.MEAN, self.STD,
                                           tokens=tokens)
        onehot = torch.nn.functional.one_hot(labels, self.num_classes).to(torch.float32)
        y = onehot * lam + onehot.roll(1, 0) * (1.0 - lam)
        self.last_corners = corners
        return out, y


def evaluate(model, batches):
    """Top-1 accuracy over (x, labels) batches in eval mode (CALM_ViT_V2.py:228-239)."""
    was_training = model.training
    model.eval()
    correct = total = 0
    with torch.no_grad():
        for x, labels in batches:
            y_hat, _ = model(x)
            correct += int((y_hat.reshape(x.shape[0], -1).argmax(dim=1) == labels).sum())
            total += int(labels.numel())
    model.train(was_training)
    return correct / max(total, 1)


def save_samples(imgs, out_dir, prefix="sample_"):
    """CALM_ViT_V2.py:113-118: sigmoid of the generated images [B,3,H,W], one 8-bit RGB PNG per image
    (`<out_dir>/sample_<i>.png`).  Returns the paths.  The PNG is written with zlib only (no plotting dependency)."""
    import struct
    import zlib
    os.makedirs(out_dir, exist_ok=True)
    x = torch.sigmoid(imgs.detach().float()).permute(0, 2, 3, 1)            # HWC, as the reference hands to imsave
    u8 = (x * 255.0).round().clamp_(0, 255).to(torch.uint8).cpu().numpy()

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    paths = []
    for i, img in enumerate(u8):
        h, w, _ = img.shape
        raw = b"".join(b"\x00" + img[r].tobytes() for r in range(h))         # filter type 0 per scanline
        png = (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0))
               + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))
        path = os.path.join(out_dir, f"{prefix}{i}.png")
        with open(path, "wb") as f:
            f.write(png)
        paths.append(path)
    return paths


def make_optimizer(model, lr=3.1e-3, weight_decay=0.02, betas=(0.9, 0.98), capturable=False):
    """optim.AdamW(model.parameters(), lr=3.1e-3, weight_decay=0.02, betas=(0.9, 0.98)) (cls:146,158)."""
    params = [p for p in model.parameters() if p.requires_grad]
    fused = params[0].is_cuda
    return torch.optim.AdamW(params, lr=lr, weight_decay=weight_decay, betas=betas, fused=fused,
                             capturable=capturable and fused)


def _clear_deferred(refs, attr, token):
    """Remove the deferral mark from the parameters that still carry THIS optimizer's token: a mark re-set by a newer
    optimizer on the same model is not the dropped one's to clear (its finalizer may run long after the new one exists)."""
    for r in refs:
        p = r()
        if p is not None and getattr(p, attr, None) == token:
            delattr(p, attr)


class FusedClipAdamW(torch.optim.Optimizer):
    """unscale + inf/NaN check + clip_grad_norm_(max_norm) + AdamW + zero_grad of distributed_trainer_cls.py:88-96 as
    THREE kernel launches over all parameters (calm_optim_step), with the spectral-norm weight-gradient correction
    folded in: while an instance is live, the backward of every spectral-normed layer that is not combined with a
    LayerScale leaves the gradient w.r.t. the normalised weight in `weight_orig.grad` (≈600 tiny launches per step
    less) and this step applies dW_orig = (G - <G, W/sigma> u v^T)/sigma on the fly — so between backward and step()
    those `.grad`s are NOT the reference's gradients; mean all-reduce commutes with the correction (it is linear and
    u, v, sigma, W are replicated), so the BucketedGradReducer / DDP run unchanged in between.  close() restores the
    in-backward correction.  Same update rule, hyper-parameters and state names as torch.optim.AdamW.

    It IS a torch.optim.Optimizer (one param group): `CosineAnnealingLR(optimizer, T_max=epochs, eta_min=1e-6)` of the
    reference's train() (cls:52,108-109) wraps it, and the learning rate a step uses is `param_groups[0]["lr"]`."""

    def __init__(self, model, lr=3.1e-3, weight_decay=0.02, betas=(0.9, 0.98), eps=1e-8, max_norm=1.0, defer_sn=True):
        from . import ops
        from .backend import get_backend
        from .spectral_norm import SpectralWeight
        self.be = get_backend()
        self.max_norm = max_norm
        params = [p for p in model.parameters() if p.requires_grad]
        super().__init__(params, dict(lr=lr, weight_decay=weight_decay, betas=tuple(betas), eps=eps))
        self.params = params
        self.exp_avg = [torch.zeros_like(p) for p in self.params]
        self.exp_avg_sq = [torch.zeros_like(p) for p in self.params]
        sn = {}
        if defer_sn:
            for m in model.modules():
                if isinstance(m, SpectralWeight) and not getattr(m, "layer_scaled", False):
                    sn[id(m.weight_orig)] = m
        records, self._deferred, self._sn_tensors = [], [], []
        for p, ea, eas in zip(self.params, self.exp_avg, self.exp_avg_sq):
            m = sn.get(id(p))
            info = None
            if m is not None:
                info = (m.weight_u, m.weight_v, m._sigma, m.rows, m.cols)
                self._deferred.append(p)
                self._sn_tensors.append((m, m.weight_u.data_ptr(), m.weight_v.data_ptr(), m._sigma.data_ptr()))
            records.append({"param": p.data, "exp_avg": ea, "exp_avg_sq": eas, "sn": info})
        self._plan = self.be.optim_plan(records)
        # the deferral is a mark on the PARAMETER OBJECT (ops reads it in forward), not a set of raw addresses: it
        # survives a re-materialised buffer and cannot be inherited by an unrelated tensor at a recycled address
        # The mark carries its owner: a second optimizer built over the same model takes the marks over, and the first
        # one's finalizer (or close()) then leaves them alone — clearing them would make backward correct the gradient
        # AND this step correct it again.
        self._token = id(self)
        self._defer_attr = ops.DEFER_ATTR
        for p in self._deferred:
            setattr(p, ops.DEFER_ATTR, self._token)
        # an optimizer that is dropped without close() must not leave its layers deferred (their backward would hand
        # out un-corrected gradients with nobody left to correct them)
        self._finalizer = weakref.finalize(self, _clear_deferred, [weakref.ref(p) for p in self._deferred], ops.DEFER_ATTR,
                                           self._token)
        self.stats = torch.zeros(2, dtype=torch.float32, device=self.params[0].device)   # [grad norm, found_inf]
        # learning rate as a device scalar for captured steps (GraphedTrainStep sets lr_on_device and refreshes it from
        # param_groups[0]["lr"] before a replay whenever a scheduler has changed it); eager steps pass the host value
        self.lr_on_device = False
        self._lr_dev = torch.full((1,), float(lr), dtype=torch.float32, device=self.params[0].device)
        self._lr_dev_value = float(lr)

    def sync_lr_to_device(self):
        """Write param_groups[0]["lr"] to the device scalar a captured step reads (no-op when unchanged)."""
        lr = float(self.lr)
        if lr != self._lr_dev_value:
            self._lr_dev.fill_(lr)
            self._lr_dev_value = lr

    @property
    def step_count(self):
        """Completed (un-skipped) optimizer steps: the counter lives on the device and does not advance when an
        inf/NaN gradient skips the update — torch.optim.AdamW under a GradScaler.  Reading it synchronises."""
        return int(self._plan.step_dev.item())

    # the hyper-parameters live in the (single) param group, where LR schedulers read and write them
    lr = property(lambda self: self.param_groups[0]["lr"], lambda self, v: self.param_groups[0].__setitem__("lr", v))
    weight_decay = property(lambda self: self.param_groups[0]["weight_decay"])
    betas = property(lambda self: self.param_groups[0]["betas"])
    eps = property(lambda self: self.param_groups[0]["eps"])

    def close(self):
        self._finalizer()
        self._deferred = []

    def _check_plan(self):
        """The plan caches raw addresses: refuse to update orphaned storage after the model was moved / re-materialised
        (model.to(...), .float(), a replaced buffer) once the optimizer exists."""
        import numpy as np
        now = np.fromiter((p.data_ptr() for p in self.params), dtype=np.uint64, count=len(self.params))
        ok = np.array_equal(now, self._plan.param_ptrs)
        for m, pu, pv, ps in self._sn_tensors:
            ok = ok and m.weight_u.data_ptr() == pu and m.weight_v.data_ptr() == pv and m._sigma.data_ptr() == ps
        if not ok:
            raise RuntimeError("FusedClipAdamW: parameters or spectral-norm buffers were moved or replaced after the "
                               "optimizer was built (model.to / .float / load with assign=True); build the optimizer "
                               "after the model is on its final device")

    @torch.no_grad()
    def step(self, closure=None, grad_scale=None):
        """One optimizer-side step on the current `.grad`s; grads are released (set to None) afterwards.
        grad_scale: device scalar the loss was multiplied by (GradScaler), or None.  Returns the stats tensor."""
        if closure is not None:
            raise NotImplementedError("FusedClipAdamW.step takes no closure")
        self._check_plan()
        for p in self._deferred:
            if getattr(p, self._defer_attr, None) != self._token:
                raise RuntimeError("FusedClipAdamW: another optimizer took over (or cleared) the deferred spectral-norm "
                                   "correction of this model; only the newest optimizer of a model may step")
        grads = []
        for p in self.params:
            g = p.grad
            if g is None:
                g = p.grad = torch.zeros_like(p)
            elif not g.is_contiguous():
                g = p.grad = g.contiguous()
            grads.append(g)
        hp = (float(self.lr), self.betas[0], self.betas[1], self.eps, self.weight_decay, self.max_norm or 0.0, 0)
        if self.lr_on_device and not torch.cuda.is_current_stream_capturing():
            self.sync_lr_to_device()
        self.be.optim_step(self._plan, grads, hp, grad_scale, self.stats, lr_dev=self._lr_dev if self.lr_on_device else None)
        for p in self.params:
            p.grad = None
        return self.stats

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            p.grad = None

    def state_dict(self):
        return {"step": self.step_count, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq,
                "hparams": dict(lr=self.lr, weight_decay=self.weight_decay, betas=self.betas, eps=self.eps,
                                max_norm=self.max_norm)}

    def load_state_dict(self, sd):
        if "hparams" in sd:
            self.lr = sd["hparams"]["lr"]
        self._plan.step_dev.fill_(int(sd["step"]))
        for dst, src in zip(self.exp_avg, sd["exp_avg"]):
            dst.copy_(src)
        for dst, src in zip(self.exp_avg_sq, sd["exp_avg_sq"]):
            dst.copy_(src)


class GraphedTrainStep:
    """The same step captured once into a hipGraph and replayed: every kernel of the library only enqueues on the
    current stream (no allocation, no host sync), so forward + loss + backward + gradient exchange + unscale / clip /
    AdamW is one graph launch — with the library's fused optimizer-side step (FusedClipAdamW: its step counter, plan
    and learning rate live on the device) or a capturable torch optimizer, with or without the reference trainer's
    autocast(bfloat16) + GradScaler (the scale update is device arithmetic, see TrainStep._finish).  Removes the ~3000
    host-side launches per step: the host's share of a step drops from tens of milliseconds to one graph launch, which
    is what keeps 8 ranks on one host from becoming host-bound.  N > 1: pass the BucketedGradReducer — its bucket
    copies and RCCL all-reduces are captured with the step (round 4; the collectives are issued in the
    capture-compatible form, see BucketedGradReducer._launch); without one the constructor refuses a world of more than
    one rank.  Inputs are copied into static buffers.  A learning-rate scheduler acts on replays: FusedClipAdamW reads
    the rate from a device scalar (calm_optim_step's lr_dev, ABI v7) that __call__ refreshes from
    param_groups[0]["lr"] before the replay; capturable torch optimizers keep theirs on the device already.

    The warm-up steps in front of the capture (allocator, lazy plans) are real training steps on the example batch.
    restore_after_warmup=True (FusedClipAdamW only) undoes them: parameters, buffers (spectral-norm u / v), optimizer
    moments, the device step counter, the GradScaler state and the CUDA RNG state are put back IN PLACE after the
    capture, so that the first replay is step 1 of the run — the replayed trajectory then equals the eager one."""

    def __init__(self, model, optimizer, example_x, example_y, max_norm=1.0, warmup=3, scaler=None, autocast_dtype=None,
                 reducer=None, restore_after_warmup=False):
        """reducer: a BucketedGradReducer whose bucket copies and RCCL all-reduces are captured with the step (round 4:
        the collectives are issued in the capture-compatible form, see BucketedGradReducer._launch; exercised in a world
        of one in a child process by scripts/rccl_capture_check.py — DESIGN.md section 6 records the outcome)."""
        if reducer is None and dist.is_initialized() and dist.get_world_size() > 1:
            # the captured step would contain no gradient exchange: replicas would silently train apart
            raise RuntimeError("GraphedTrainStep captures a single-GPU step unless it is given the BucketedGradReducer "
                               "to capture with it; with world_size > 1 pass reducer=... or use TrainStep")
        self.inner = TrainStep(model, optimizer, reducer, max_norm=max_norm, scaler=scaler, autocast_dtype=autocast_dtype)
        self.opt = optimizer
        if isinstance(optimizer, FusedClipAdamW):
            optimizer.lr_on_device = True          # replays read the learning rate from a device scalar
        self.x = example_x.clone()
        self.y = example_y.clone()
        snap = None
        if restore_after_warmup:
            if not isinstance(optimizer, FusedClipAdamW):
                raise ValueError("restore_after_warmup needs FusedClipAdamW (its whole state is a known set of tensors)")
            live = list(model.parameters()) + list(model.buffers()) + optimizer.exp_avg + optimizer.exp_avg_sq + \
                [optimizer._plan.step_dev]
            snap = ([(t, t.detach().clone()) for t in live], torch.cuda.get_rng_state(example_x.device),
                    None if scaler is None or not scaler.is_enabled() else scaler.get_scale())
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                     # warm-up on a side stream (allocator + lazy plans)
            for _ in range(max(warmup, 1 if scaler is not None else 0)):
                self.inner(self.x, self.y)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss, self.y_hat = self.inner(self.x, self.y)
        if snap is not None:                               # in place: the graph replays on these addresses
            with torch.no_grad():
                for t, c in snap[0]:
                    t.copy_(c)
            torch.cuda.set_rng_state(snap[1], example_x.device)
            if snap[2] is not None:
                scaler.update(new_scale=float(snap[2]))                # fill_ on the scale tensor the graph reads
                if hasattr(self.inner, "_clean_steps"):
                    self.inner._clean_steps.zero_()                    # (TrainStep._finish keeps the growth count itself)
            torch.cuda.synchronize()

    def __call__(self, x, y_soft):
        self.x.copy_(x, non_blocking=True)
        self.y.copy_(y_soft, non_blocking=True)
        if isinstance(self.opt, FusedClipAdamW):
            self.opt.sync_lr_to_device()           # a scheduler's new rate reaches the replay without re-capture
        self.graph.replay()
        return self.loss, self.y_hat


class SoftMixCollate:
    """collate_fn of the reference's DataLoader (distributed_trainer_cls.py:58-62): default_collate, then
    RandomChoice([CutMix(num_classes, alpha=1.0), MixUp(num_classes, alpha=0.8)]) on the batch — torchvision.transforms.v2
    semantics restated (torchvision is not installed here; tests/golden/mix_vectors.json pins them): the partner of
    sample i is sample i-1 (`roll(1, 0)`), lam ~ Beta(alpha, alpha);
      MixUp : x = lam x + (1 - lam) x_rolled,                      y = lam onehot + (1 - lam) onehot_rolled
      CutMix: a box of area ratio (1 - lam) centred at a uniform point, clipped to the image, is pasted from the
              partner; lam is then corrected to 1 - box_area / image_area.
    Host side, float images [B,3,H,W]; `DeviceCollate` is the uint8-on-device form of the same decisions."""

    def __init__(self, num_classes=1000, cutmix_alpha=1.0, mixup_alpha=0.8, seed=None):
        self.num_classes, self.cutmix_alpha, self.mixup_alpha = num_classes, cutmix_alpha, mixup_alpha
        self.seed = seed
        self._rng = None
        self._rng_worker = None

    @property
    def rng(self):
        """The generator of THIS process: DataLoader workers each receive a pickled copy of the collate object, so a
        generator built in __init__ would hand every worker the same (mode, lam, box) sequence.  It is
        created on first use from (seed, worker id) — torchvision's transforms draw from the per-worker-seeded torch RNG
        in the reference (num_workers=5, cls:62)."""
        import numpy as np
        info = torch.utils.data.get_worker_info()
        wid = info.id if info is not None else -1
        if self._rng is None or self._rng_worker != wid:
            self._rng = np.random.default_rng(None if self.seed is None else [int(self.seed), wid + 1])
            self._rng_worker = wid
        return self._rng

    def __getstate__(self):
        st = dict(self.__dict__)
        st["_rng"], st["_rng_worker"] = None, None        # never ship generator state to a worker
        return st

    @staticmethod
    def cutmix_box(lam, cx, cy, H, W):
        """torchvision v2 CutMix._get_params: r = 0.5 sqrt(1 - lam); box = centre -+ (r W, r H) truncated to ints and
        clipped; returns (y1, y2, x1, x2) and the corrected lam."""
        r = 0.5 * (1.0 - lam) ** 0.5
        rw, rh = int(r * W), int(r * H)
        x1, y1, x2, y2 = max(cx - rw, 0), max(cy - rh, 0), min(cx + rw, W), min(cy + rh, H)
        return (y1, y2, x1, x2), 1.0 - (x2 - x1) * (y2 - y1) / float(W * H)

    def mix(self, x, labels, mode, lam, box=None):
        onehot = torch.nn.functional.one_hot(labels, self.num_classes).to(torch.float32)
        xr = x.roll(1, 0)
        if mode == 1:
            out = x * lam + xr * (1.0 - lam)
        else:
            y1, y2, x1, x2 = box
            out = x.clone()
            out[..., y1:y2, x1:x2] = xr[..., y1:y2, x1:x2]
        return out, onehot * lam + onehot.roll(1, 0) * (1.0 - lam)

    def __call__(self, batch):
        x = torch.stack([b[0] for b in batch])
        labels = torch.as_tensor([int(b[1]) for b in batch])
        H, W = x.shape[-2:]
        mode = 2 if self.rng.random() < 0.5 else 1
        alpha = self.cutmix_alpha if mode == 2 else self.mixup_alpha
        lam = float(self.rng.beta(alpha, alpha))
        box = None
        if mode == 2:
            box, lam = self.cutmix_box(lam, int(self.rng.integers(0, W)), int(self.rng.integers(0, H)), H, W)
        return self.mix(x, labels, mode, lam, box)


def train(initializer, optimizer, scheduler=None, use_gpu=True, dataset=None, epochs=15, batch_size=128,
          checkpoint_path=None, num_classes=1000, num_workers=0, collate_fn="mix", log_every=100, max_steps=None,
          destroy_process_group=True, device_collate=False, crop=None, graph=False, selfcheck="raise"):
    """Per-rank training job: the reference's `train(initializer, optimizer, scheduler, use_gpu, dataset, epochs,
    batch_size)` (distributed_trainer_cls.py:25-114) on torch.distributed + RCCL instead of Spark's TorchDistributor —
    start one process per GPU with `python -m torch.distributed.run --nproc-per-node N ...` (RANK / LOCAL_RANK /
    WORLD_SIZE from the environment, cls:48-50).

      * init_process_group (nccl on GPUs, gloo on CPU)                                   cls:46
      * CosineAnnealingLR(optimizer, T_max=epochs, eta_min=1e-6), stepped once per epoch  cls:52,108-109
        (scheduler=None / True: built here as the reference does — it builds this schedule whatever it is handed;
        a scheduler OBJECT passed in is kept and stepped instead; scheduler=False: constant rate)
      * model.to(device); parameters + buffers broadcast from rank 0; gradients mean-all-reduced in buckets on a side
        stream (sync_module_states + BucketedGradReducer = what DDP(model) does)          cls:54-55
      * DistributedSampler(dataset, shuffle=True, seed=2006), set_epoch(epoch)           cls:56-57,73
      * DataLoader with the CutMix / MixUp soft-label collate                            cls:58-62
      * per step: autocast(bf16) forward, CE, GradScaler, clip_grad_norm_(1.0), optimizer step, zero_grad   cls:79-96
      * rank 0: loss / dominant-class accuracy print every 100 steps, state_dict checkpoint per epoch      cls:97-107
      * returns the model on the CPU after destroy_process_group()                       cls:112-114

    optimizer: a torch optimizer over `initializer.parameters()` (the reference hands in AdamW(lr=3.1e-3,
    weight_decay=0.02, betas=(0.9, 0.98))), or the string "fused" for FusedClipAdamW with those hyper-parameters (built
    here, after the model is on its device).  checkpoint_path: the reference writes /config/Codebase/models/model_cls.pth.

    device_collate=True: the dataset yields uint8 images [3,Hs,Ws] and integer labels; the DataLoader only
    stacks them (default collate), the uint8 batch goes host-to-device as it is (a quarter of the fp32 bytes) and
    `DeviceCollate` does ToDtype + Normalize + RandomCrop(crop) + flip + CutMix / MixUp in one kernel pass, writing the
    first Block's row tokens [B,S,3S] directly — the model's first Block takes them without the image_to_rows pass
    (cls:58-62,128-139; Vi_Tools:389-391).

    selfcheck ("raise" | "fallback" | None; GPU only, once per process): before the first step the box is asked whether
    the two bf16 GEMM families agree on it (HipBackend.selfcheck_bf16_gemm — round 3 saw one box of the pool on which the
    default pipelined family returned a deterministic wrong gradient); "raise" stops the job with the pattern of the
    differing elements, "fallback" warns and trains on the 256x128 family.

    graph=True (GPU, optimizer "fused" / FusedClipAdamW): the step — forward, loss, backward, the bucketed RCCL
    all-reduces of a world > 1, unscale / clip / AdamW — is captured into a hipGraph on the first batch
    (GraphedTrainStep with restore_after_warmup: the capture's warm-up steps are undone) and replayed per batch: one
    graph launch of host work per step instead of ~3000 kernel launches.  The DataLoader then drops the last partial
    batch of an epoch (a graph has one batch shape); a batch of any other shape would run the eager step."""
    from torch.utils.data import DataLoader, DistributedSampler
    if device_collate and not use_gpu:                     # (argument errors before any process group exists)
        raise ValueError("device_collate=True needs use_gpu=True (the collate is a HIP kernel)")
    if graph and not (use_gpu and (optimizer == "fused" or isinstance(optimizer, FusedClipAdamW))):
        raise ValueError('graph=True needs use_gpu=True and optimizer="fused" (FusedClipAdamW)')
    rank, local_rank, world = init_distributed(use_gpu)
    device = torch.device(f"cuda:{local_rank}" if use_gpu else "cpu")
    if use_gpu:
        torch.cuda.set_device(device)
    if use_gpu and selfcheck:
        from .backend import get_backend
        be = get_backend()
        if not getattr(be, "_selfcheck_done", False):
            be.selfcheck_bf16_gemm(on_mismatch=selfcheck)
            be._selfcheck_done = True
            torch.cuda.empty_cache()
    model = initializer.to(device)
    if optimizer == "fused":
        optimizer = FusedClipAdamW(model)
    if scheduler is None or scheduler is True:
        scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max=epochs, eta_min=1e-6)      # cls:52
    elif scheduler is False:
        scheduler = None
    sync_module_states(model)
    reducer = BucketedGradReducer(model) if world > 1 else None
    if world > 1:
        sampler = DistributedSampler(dataset, num_replicas=world, rank=rank, shuffle=True, seed=2006)       # cls:56
    else:
        sampler = DistributedSampler(dataset, num_replicas=1, rank=0, shuffle=True, seed=2006)
    sampler.set_epoch(0)
    dcoll = None
    if device_collate:
        if not use_gpu:
            raise ValueError("device_collate=True needs use_gpu=True (the collate is a HIP kernel)")
        dcoll = DeviceCollate(num_classes=num_classes, seed=2006 + rank)
        collate_fn = None                                   # default_collate: stack uint8 images and labels
    elif collate_fn == "mix":
        collate_fn = SoftMixCollate(num_classes=num_classes, seed=2006 + rank)
    loader = DataLoader(dataset, batch_size=batch_size, sampler=sampler, collate_fn=collate_fn, num_workers=num_workers,
                        pin_memory=use_gpu, persistent_workers=num_workers > 0, drop_last=bool(graph))
    scaler = torch.amp.GradScaler("cuda", enabled=use_gpu)                                                 # cls:64
    step = TrainStep(model, optimizer, reducer, max_norm=1.0, scaler=scaler if use_gpu else None,
                     autocast_dtype=torch.bfloat16 if use_gpu else None)
    model.train()
    n_steps = 0
    gstep = None
    try:
        for epoch in range(epochs):
            sampler.set_epoch(epoch)
            model.train()
            epoch_loss = 0.0
            for i, (x, y) in enumerate(loader):
                x, y = x.to(device, non_blocking=True), y.to(device, non_blocking=True)
                if dcoll is not None:
                    x, y = dcoll(x, y.long(), crop=crop, tokens=True)   # uint8 batch -> row tokens + soft labels
                if graph and gstep is None:
                    gstep = GraphedTrainStep(model, optimizer, x, y, max_norm=1.0, scaler=scaler,
                                             autocast_dtype=torch.bfloat16, reducer=reducer, restore_after_warmup=True)
                if gstep is not None and x.shape == gstep.x.shape and y.shape == gstep.y.shape:
                    loss, y_hat = gstep(x, y)
                else:
                    loss, y_hat = step(x, y)
                epoch_loss += loss.item()                                                                  # cls:97
                if rank == 0 and local_rank == 0 and i % log_every == 0:
                    correct = (y_hat.reshape(y.shape[0], -1).argmax(1) == y.argmax(1)).sum().item()
                    print(f"Epoch: {epoch + 1}, Batch: {i + 1}, Device: [{rank}, {local_rank}], Loss: {float(loss)}, "
                          f"Accuracy: {100.0 * correct / y.size(0):.4f}%")
                n_steps += 1
                if max_steps is not None and n_steps >= max_steps:
                    break
            if rank == 0 and local_rank == 0 and checkpoint_path:
                os.makedirs(os.path.dirname(os.path.abspath(checkpoint_path)), exist_ok=True)
                torch.save(model.state_dict(), checkpoint_path)                                            # cls:105-107
            if scheduler is not None:
                scheduler.step()                                                                           # cls:108-109
            if max_steps is not None and n_steps >= max_steps:
                break
    finally:
        if gstep is not None:              # the captured graph holds RCCL kernels: released before the process group
            gstep = None
            import gc
            gc.collect()
            torch.cuda.synchronize()
        if isinstance(optimizer, FusedClipAdamW):
            optimizer.close()
    model = model.to("cpu")                                                                                # cls:112
    if destroy_process_group and dist.is_initialized():
        dist.destroy_process_group()
    return model
