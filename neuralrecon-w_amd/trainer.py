"""The optimiser side of the train step, laid out for one GPU per process.

The reference's recipe (train.py:21-25,61; utils/__init__.py:23-31; neuconw_system.py:178-184) is
`Adam(lr, eps=1e-7)` over embedding + NeuconW + NeRF, a global grad-norm clip of 0.99 and, under DDP, a
gradient all-reduce.  With ~90 parameter tensors that is three passes of many small launches per step.
Here every trainable parameter is RE-SEATED into one flat fp32 buffer and every `.grad` into one flat
gradient buffer (the renderer writes its weight-norm backward straight into it, renderer.py `_grad_views`),
so that per step there is ONE zero-fill, ONE all-reduce, ONE norm and ONE Adam update, with the same
arithmetic per element as the stock optimiser.  `state_dict()` keys and shapes of the modules are unchanged
(checkpoint format of utils/__init__.py:64-98); parameters stay ordinary `nn.Parameter`s.
"""
import torch
import torch.distributed as dist


class FlatParams:
    """Flat parameter / gradient storage for `modules` (an iterable of nn.Module).

    Order: the renderer's packed-network parameters first (their gradient slice is adopted by the renderer
    as its persistent flat gradient buffer), then every remaining parameter in module order.  Parameters
    that never receive a gradient (the reference's dead layers: NeuconW.xyz_encoding_final,
    NeRF.views_linears) keep a zero gradient, for which Adam's update is exactly zero -- the same end state
    as the stock optimiser skipping them."""

    def __init__(self, modules, renderer=None):
        modules = list(modules)
        seen, params = set(), []
        first = list(renderer._params()) if renderer is not None else []
        for p in first + [p for m in modules for p in m.parameters()]:
            if id(p) not in seen and p.requires_grad:
                seen.add(id(p))
                params.append(p)
        if not params:
            raise ValueError("no trainable parameters")
        dev = params[0].device
        for p in params:
            if p.device != dev or p.dtype != torch.float32:
                raise ValueError("FlatParams needs fp32 parameters on one device")
        n = sum(p.numel() for p in params)
        self.params = params
        self.flat = torch.nn.Parameter(torch.empty(n, device=dev, dtype=torch.float32))
        self.flat_grad = torch.zeros(n, device=dev, dtype=torch.float32)
        self.flat.grad = self.flat_grad
        self.slices = {}
        off = 0
        with torch.no_grad():
            for p in params:
                k = p.numel()
                view = self.flat.data[off:off + k].view(p.shape)
                view.copy_(p.data)
                p.data = view
                p.grad = self.flat_grad[off:off + k].view(p.shape)
                self.slices[id(p)] = (off, k)
                off += k
        # the packed-weight caches key on parameter version counters; updates now arrive through `flat`
        for m in modules:
            for sub in m.modules():
                sub.__dict__["_ncw_version_srcs"] = sub.__dict__.get("_ncw_version_srcs", ()) + (self,)
        self._epoch = 0
        if renderer is not None and first:
            nr = sum(p.numel() for p in first)
            renderer.adopt_grad_buffer(first, self.flat_grad[:nr])
        self.renderer = renderer

    @property
    def _version(self):
        return (self.flat._version, self._epoch)

    def mark_updated(self):
        """Call after an optimiser step on `flat`: fused optimiser kernels do not bump tensor version
        counters, and the packed-weight caches must see that the parameters changed."""
        self._epoch += 1

    def zero_grad(self):
        """One fill; `.grad` tensors stay the same views (autograd and the renderer accumulate in place)."""
        self.flat_grad.zero_()
        for p in self.params:  # a foreign optimizer.zero_grad(set_to_none=True) would have dropped the views
            if p.grad is None:
                off, k = self.slices[id(p)]
                p.grad = self.flat_grad[off:off + k].view(p.shape)

    def allreduce(self, world_size=None, group=None, async_op=False, force=False):
        """Mean of the flat gradient over ranks: ONE all-reduce (RCCL over xGMI), in place.  async_op=True returns a
        handle whose wait() completes the mean; RCCL averages in the collective (ReduceOp.AVG), gloo sums and the division
        is one more launch.  TrainStep waits for it right away (the step is SYNCHRONOUS in the exchange: every gradient
        becomes final in one weight-gradient launch at the very end of the backward).  `force`: issue the
        collective in a one-rank group too (tests push the RCCL path through a single GPU)."""
        if not dist.is_available() or not dist.is_initialized():
            return None
        world_size = dist.get_world_size(group) if world_size is None else world_size
        if world_size == 1 and not force:
            return None
        avg = dist.get_backend(group) == "nccl"
        work = dist.all_reduce(self.flat_grad, op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM, group=group,
                               async_op=True)
        h = _ReduceHandle(work, None if avg else (self.flat_grad, world_size))
        if async_op:
            return h
        h.wait()
        return None

    def broadcast(self, src=0, group=None):
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.broadcast(self.flat.data, src=src, group=group)


class _ReduceHandle:
    def __init__(self, work, div):
        self.work, self.div = work, div

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
            if self.div is not None:
                self.div[0].div_(self.div[1])


class FlatAdam:
    """Adam over trainer.FlatParams with the global-norm clip folded in: `ncw_adam_step_dev` (a one-thread prologue +
    ONE elementwise launch, after one norm reduction) instead of torch's multi-tensor norm / clamp / mul / fused-Adam
    sequence.  Same arithmetic as torch.optim.Adam(lr, betas, eps, weight_decay=0, amsgrad=False) after
    torch.nn.utils.clip_grad_norm_(params, clip) (tests/test_gpu_trainer.py).

    Everything step-dependent lives on the DEVICE (`state`: NcwAdamState): the applied-step counter (a step skipped for a
    non-finite gradient norm does not advance Adam's bias correction), the bias corrections (computed in double like
    torch's non-capturable path), the clip coefficient -- so the update is HIP-graph capturable -- and, in the fp16 mode,
    the loss scale (`loss_scale`: the renderer's device {scale, 1 / scale}): halved after a non-finite step, doubled after
    `growth_interval` consecutive clean steps (torch.cuda.amp.GradScaler's policy and defaults)."""

    def __init__(self, flat_params, lr, betas=(0.9, 0.999), eps=1e-7, clip=None, loss_scale=None, growth_interval=2000,
                 scale_min=1.0, scale_max=65536.0):
        self.fp = flat_params
        # the learning rate is DEVICE data (ncw_adam_step_dev reads `lr_dev` at run time), so a schedule keeps working
        # when the step is replayed from a HIP graph: assigning `opt.lr` refills the device scalar
        self.lr_dev = torch.empty(1, device=flat_params.flat_grad.device, dtype=torch.float32)
        self.lr = float(lr)
        self.betas, self.eps, self.clip = (float(betas[0]), float(betas[1])), float(eps), clip
        self.exp_avg = torch.zeros_like(flat_params.flat_grad)
        self.exp_avg_sq = torch.zeros_like(flat_params.flat_grad)
        self.state = torch.zeros(8, device=flat_params.flat_grad.device, dtype=torch.int32)  # NcwAdamState
        self.loss_scale, self.growth_interval = loss_scale, int(growth_interval)
        self.scale_min, self.scale_max = float(scale_min), float(scale_max)

    @property
    def lr(self):
        return self._lr

    @lr.setter
    def lr(self, v):
        self._lr = float(v)
        self.lr_dev.fill_(self._lr)  # an ordinary stream-ordered fill: also valid between graph replays

    # applied (non-skipped) steps: Adam's t.  Lives on the device; reading it synchronises (checkpoints, tests)
    @property
    def step_count(self):
        return int(self.state[0])

    @step_count.setter
    def step_count(self, v):
        self.state[0] = int(v)

    @property
    def skipped_steps(self):
        return int(self.state[2])

    def step(self):
        from . import lib as L

        fp = self.fp
        if not fp.flat_grad.is_cuda:
            raise L.NeuconwHipError("FlatAdam: parameters are not on a GPU; there is no CPU fallback")
        b1, b2 = self.betas
        need_norm = self.clip is not None or self.loss_scale is not None
        norm = None
        if need_norm:  # ||flat_grad||_2 as one fixed-order launch (ncw_grad_norm) instead of ATen's reduction kernels
            lib = L.get_lib()
            if self.__dict__.get("_norm_scratch") is None or self._norm_scratch.device != fp.flat_grad.device:
                self._norm_scratch = torch.zeros(int(lib.ncw_grad_norm_scratch_floats()), device=fp.flat_grad.device, dtype=torch.float32)
                self._norm = torch.empty(1, device=fp.flat_grad.device, dtype=torch.float32)
            norm = self._norm
            L.check(lib.ncw_grad_norm(L.ptr(fp.flat_grad), fp.flat_grad.numel(), L.ptr(self._norm_scratch), L.ptr(norm),
                                      L.stream_ptr(fp.flat_grad.device)), "ncw_grad_norm")
            norm = norm[0]
        L.check(L.get_lib().ncw_adam_step_dev(
            L.ptr(fp.flat.data), L.ptr(fp.flat_grad), L.ptr(self.exp_avg), L.ptr(self.exp_avg_sq), fp.flat_grad.numel(),
            L.ptr(self.state), L.ptr(norm), self.lr, L.ptr(self.lr_dev), b1, b2, self.eps,
            float(self.clip) if self.clip is not None else 0.0, L.ptr(self.loss_scale), self.growth_interval,
            self.scale_min, self.scale_max, L.stream_ptr(fp.flat_grad.device)), "ncw_adam_step_dev")
        fp.mark_updated()
        # a fresh 4-byte tensor (device copy, no sync): `_norm` is overwritten by the next step, a caller keeping norms across
        # steps must not alias it
        return norm.clone() if norm is not None else None

    def state_dict(self):
        return {"step": self.step_count, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "lr": self.lr,
                "betas": self.betas, "eps": self.eps}

    def torch_state_dict(self, param_order):
        """The state in torch.optim.Adam's own `state_dict()` layout for the parameters in `param_order` (the order
        the reference's `get_optimizer` hands them to Adam: utils/__init__.py:10-31 over [embedding_a, {neuconw, nerf}],
        neuconw_system.py:70-136) -- what a PyTorch-Lightning checkpoint stores under `optimizer_states[0]`, so the
        moments load into a stock torch.optim.Adam built the reference's way.  Parameters that are not in the flat storage
        are skipped."""
        state, ids = {}, []
        for i, p in enumerate(param_order):
            ids.append(i)
            sl = self.fp.slices.get(id(p))
            if sl is None:
                continue
            off, k = sl
            state[i] = {"step": torch.tensor(float(self.step_count)),
                        "exp_avg": self.exp_avg[off:off + k].view(p.shape).detach().cpu().clone(),
                        "exp_avg_sq": self.exp_avg_sq[off:off + k].view(p.shape).detach().cpu().clone()}
        group = {"lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": 0, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "params": ids}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd, param_order=None):
        """Accepts this class' flat state or a torch.optim.Adam state_dict (with `param_order` as above)."""
        if "param_groups" in sd:
            if param_order is None:
                raise ValueError("a torch.optim.Adam state_dict needs param_order")
            step = 0
            for i, p in enumerate(param_order):
                st, sl = sd["state"].get(i), self.fp.slices.get(id(p))
                if st is None or sl is None:
                    continue
                off, k = sl
                self.exp_avg[off:off + k].copy_(st["exp_avg"].reshape(-1))
                self.exp_avg_sq[off:off + k].copy_(st["exp_avg_sq"].reshape(-1))
                step = max(step, int(float(st["step"])))
            self.step_count = step
            return
        self.step_count = int(sd["step"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])


class TrainStep:
    """render -> loss -> backward -> all-reduce -> clip -> Adam, the timed region of SURVEY 8(d).

    loss_fn(outputs, targets) -> scalar is the user's (NeuconWLoss, losses.py:21-43).

    capture=True records the step ONCE into HIP graphs (torch.cuda.CUDAGraph) after `capture_warmup` eager
    steps and replays it afterwards: the launches of a step (C-ABI kernels, the loss, autograd glue, norm, clip + Adam)
    are submitted as one graph launch.  Everything that varies between steps is device data: the batch is copied
    into static buffers, `cos_anneal_ratio` is a device scalar read by the compositor kernels, Adam's step
    counter, bias corrections and the fp16 loss scale live in device state (FlatAdam / ncw_adam_step_dev), the
    sampler's jitter comes from torch's graph-safe Philox state.  Requirements: fixed batch shape, a loss_fn without host syncs, renderer.sync_free (set here).
    With world_size > 1 the gradient all-reduce stays an eager RCCL call between two graphs."""

    def __init__(self, renderer, modules, loss_fn, lr, eps=1e-7, betas=(0.9, 0.999), clip=0.99, world_size=1,
                 group=None, capture=False, capture_warmup=3, native_optimizer=None):
        self.renderer, self.loss_fn, self.clip = renderer, loss_fn, clip
        self.world_size, self.group = world_size, group
        self.capture, self.capture_warmup = bool(capture), int(capture_warmup)
        self._n_eager, self._graphs, self._static = 0, None, None
        self.fp = FlatParams(modules, renderer)
        if world_size > 1:  # DDP's constructor-time broadcast; a world_size-1 TrainStep inside a larger job (one rank's own
            self.fp.broadcast(group=group)  # diagnostics, e.g. bench.py's parity legs) must not enter a collective
        kw = dict(lr=lr, eps=eps, betas=betas)
        # clip + Adam as ONE C-ABI call (FlatAdam: device-resident step state, so it is graph-capturable too);
        # native_optimizer=False keeps stock torch.optim.Adam over the flat buffer (eager only)
        self.native = True if native_optimizer is None else bool(native_optimizer)
        if self.capture:
            renderer.sync_free = True
            if not self.native:
                raise ValueError("TrainStep(capture=True) needs the native optimiser (torch's path checks the gradient "
                                 "norm on the host)")
        if self.native:
            from . import lib as L

            ls = renderer.loss_scale.tensor(self.fp.flat_grad.device) if getattr(renderer, "prec", None) == L.PREC_F16 else None
            self.opt = FlatAdam(self.fp, clip=clip, loss_scale=ls, **kw)
            return
        try:
            self.opt = torch.optim.Adam([self.fp.flat], fused=True, **kw)
        except (RuntimeError, TypeError, ValueError):
            self.opt = torch.optim.Adam([self.fp.flat], **kw)

    # ---- the two halves of a step (the all-reduce sits between them) ---------------------------------
    def _fwd_bwd(self, rays, ts, label, targets, background_rgb, cos_anneal_ratio, render_kw):
        self.fp.zero_grad()
        out = self.renderer.render(rays, ts, label, background_rgb=background_rgb,
                                   cos_anneal_ratio=cos_anneal_ratio, **render_kw)
        loss = self.loss_fn(out, targets)
        loss.backward()
        return loss, out

    def _update(self):
        if self.native:
            # clip + Adam + mark_updated.  The gradient norm stays on the device (no sync); a NON-FINITE norm makes
            # ncw_adam_step_dev skip the update and halve the fp16 loss scale -- callers that log can look at
            # last_grad_norm / opt.skipped_steps now and then
            self.last_grad_norm = self.opt.step()
            return
        # stock torch optimiser (eager only): the same guard on the host -- one overflowed fp16 step must not turn every
        # parameter and both Adam moments into NaN through the clip coefficient
        norm = torch.nn.utils.clip_grad_norm_([self.fp.flat], self.clip if self.clip is not None else float("inf"))  # train.py:61
        self.last_grad_norm = norm
        if not bool(torch.isfinite(norm)):
            self.skipped_steps = getattr(self, "skipped_steps", 0) + 1
            ls = getattr(self.renderer, "loss_scale", None)
            if ls is not None and ls.buf is not None:
                ls.set(max(ls.value() * 0.5, 1.0))
            return
        self.opt.step()
        self.fp.mark_updated()

    def eager_step(self, rays, ts, label, targets, background_rgb=None, cos_anneal_ratio=0.0, **render_kw):
        loss, out = self._fwd_bwd(rays, ts, label, targets, background_rgb, cos_anneal_ratio, render_kw)
        # ONE all-reduce of the flat gradient (train.py:53-55 accelerator='ddp'), issued asynchronously: the collective
        # runs on RCCL's stream while the host queues the norm / optimiser launches, which wait for it on the device
        h = self.fp.allreduce(self.world_size, self.group, async_op=True)
        if h is not None:
            h.wait()
        self._update()
        return loss, out

    # ---- graph capture ---------------------------------------------------------------------------
    def _capture(self, rays, ts, label, targets, background_rgb, render_kw):
        dev = rays.device
        st = dict(rays=rays.clone(), ts=ts.clone(), label=label.clone(), targets=targets.clone(),
                  bg=None if background_rgb is None else background_rgb.clone(),
                  cos=torch.zeros(1, device=dev, dtype=torch.float32))
        self._render_kw = dict(render_kw)
        torch.cuda.synchronize(dev)
        g1 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1):
            loss, out = self._fwd_bwd(st["rays"], st["ts"], st["label"], st["targets"], st["bg"], st["cos"],
                                      self._render_kw)
            if self.world_size == 1:
                self._update()
        g2 = None
        if self.world_size > 1:
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2, pool=g1.pool()):
                self._update()
        st["loss"], st["out"] = loss.detach(), {k: (v.detach() if torch.is_tensor(v) else v) for k, v in out.items()}
        self._graphs, self._static = (g1, g2), st

    def __call__(self, rays, ts, label, targets, background_rgb=None, cos_anneal_ratio=0.0, **render_kw):
        if not self.capture:
            return self.eager_step(rays, ts, label, targets, background_rgb, cos_anneal_ratio, **render_kw)
        if self._graphs is None:
            if self._n_eager < self.capture_warmup:  # arenas, descriptor tables and Adam state must exist first
                self._n_eager += 1
                loss, out = self.eager_step(rays, ts, label, targets, background_rgb, cos_anneal_ratio, **render_kw)
                # detached, like the replays: an autograd graph of an eager step that is still alive at capture
                # time keeps its AccumulateGrad nodes (bound to the eager stream) in use, and the engine then
                # synchronises the capturing stream with that stream -- hipStreamEndCapture crashes on it
                return loss.detach(), {k: (v.detach() if torch.is_tensor(v) else v) for k, v in out.items()}
            import gc
            gc.collect()
            self._capture(rays, ts, label, targets, background_rgb, render_kw)
            # the capture pass itself does not execute anything: fall through and replay this step
        st = self._static
        if dict(render_kw) != self._render_kw or rays.shape != st["rays"].shape:
            raise ValueError("captured TrainStep: batch shape and render keywords are fixed at capture time")
        st["rays"].copy_(rays, non_blocking=True)
        st["ts"].copy_(ts, non_blocking=True)
        st["label"].copy_(label, non_blocking=True)
        st["targets"].copy_(targets, non_blocking=True)
        if st["bg"] is not None:
            st["bg"].copy_(background_rgb, non_blocking=True)
        st["cos"].fill_(float(cos_anneal_ratio))
        g1, g2 = self._graphs
        g1.replay()
        if g2 is not None:
            self.fp.allreduce(self.world_size, self.group)
            g2.replay()
        self.fp._epoch += 1
        return st["loss"], st["out"]


# ---------------------------------------------------------------------------------------------------
# Checkpoint I/O in the reference's format.  The reference saves PyTorch-Lightning checkpoints
# whose `state_dict` holds `embedding_a.*`, `neuconw.*`, `nerf.*` (train.py:32-38, neuconw_system.py:376-400)
# and reads them back by prefix (utils/__init__.py:64-98 `extract_model_state_dict` / `load_ckpt`, used by
# tools/extract_mesh.py:130-134).  These two functions write / read exactly that layout; the optimiser state is
# written in torch.optim.Adam's own state_dict layout in the reference's parameter order, so weights (through the
# reference's load_ckpt) and Adam moments (through torch.optim.Adam.load_state_dict) move in both directions
# (tests/test_checkpoint_io.py).
# ---------------------------------------------------------------------------------------------------
def reference_param_order(embedding_a, neuconw, nerf):
    """`get_parameters([embedding_a, {"neuconw": neuconw, "nerf": nerf}])` of the reference (utils/__init__.py:10-21)."""
    return list(embedding_a.parameters()) + list(neuconw.parameters()) + list(nerf.parameters())


def resume_state(optimizer, renderer=None, epoch=0, step_in_epoch=0, generator_state=None):
    """What a bit-for-bit continuation needs beyond weights and Adam moments (the checkpoint's `ncw_resume` entry): the epoch
    and the batches of it already consumed, the batch generator's state at the START of that epoch (the epoch's permutation
    is drawn from it), and the fp16 dynamic loss scale with its clean / skipped step counters (NcwAdamState)."""
    st = {"epoch": int(epoch), "step_in_epoch": int(step_in_epoch),
          "generator_state": None if generator_state is None else generator_state.detach().cpu().clone(),
          # the sampler's jitter (renderer.py:477-480,536-541 `torch.rand`) draws from the device's global generator
          "cuda_rng_state": torch.cuda.get_rng_state(optimizer.fp.flat_grad.device).clone()
          if hasattr(optimizer, "fp") and optimizer.fp.flat_grad.is_cuda else None,
          "cpu_rng_state": torch.get_rng_state().clone()}
    if torch.is_tensor(getattr(optimizer, "state", None)):  # FlatAdam: device-resident NcwAdamState {step, good, skipped, ...}
        s_ = optimizer.state.detach().cpu()   # (torch.optim.Adam also has `.state`: a dict -- nothing to record for it)
        st.update(adam_good=int(s_[1]), adam_skipped=int(s_[2]))
    ls = getattr(renderer, "loss_scale", None)
    if ls is not None:
        st["loss_scale"] = float(ls.value())
    return st


def apply_resume_state(st, optimizer, renderer=None):
    """Restores the optimiser-side entries of `resume_state`; returns (epoch, step_in_epoch, generator_state)."""
    if torch.is_tensor(getattr(optimizer, "state", None)) and "adam_good" in st:
        optimizer.state[1] = int(st["adam_good"])
        optimizer.state[2] = int(st["adam_skipped"])
    ls = getattr(renderer, "loss_scale", None)
    if ls is not None and st.get("loss_scale"):
        ls.set(float(st["loss_scale"]))
    if st.get("cuda_rng_state") is not None and hasattr(optimizer, "fp") and optimizer.fp.flat_grad.is_cuda:
        torch.cuda.set_rng_state(st["cuda_rng_state"], optimizer.fp.flat_grad.device)
    if st.get("cpu_rng_state") is not None:
        torch.set_rng_state(st["cpu_rng_state"])
    return int(st.get("epoch", 0)), int(st.get("step_in_epoch", 0)), st.get("generator_state")


def save_checkpoint(path, embedding_a, neuconw, nerf, optimizer=None, global_step=0, extra=None, epoch=0, lr_scheduler=None):
    """Writes {'state_dict': {prefix.key: tensor}, 'global_step', 'epoch', ['optimizer_states': [...]]} plus the keys
    PyTorch-Lightning 1.4.8's `resume_from_checkpoint` looks up ('lr_schedulers', 'callbacks',
    'pytorch-lightning_version').  TESTED round trips: the reference's `load_ckpt` / `extract_model_state_dict` (weights)
    and a stock `torch.optim.Adam` (moments), tests/test_checkpoint_io.py; PL's own restore path is not exercised here
    (PL is not installable in this image)."""
    sd = {}
    for prefix, mod in (("embedding_a", embedding_a), ("neuconw", neuconw), ("nerf", nerf)):
        for k, v in mod.state_dict().items():
            sd[prefix + "." + k] = v.detach().cpu().clone()   # clone: parameters may be views of the flat buffer
    # lr_scheduler: the per-epoch scheduler's state in torch's `_LRScheduler.state_dict()` layout (what PL stores under
    # 'lr_schedulers'; utils/__init__.py:45-61): {"last_epoch", "_step_count", "_last_lr", ...} -- None for 'none'
    ckpt = {"state_dict": sd, "global_step": int(global_step), "epoch": int(epoch),
            "lr_schedulers": [] if lr_scheduler is None else [dict(lr_scheduler)], "callbacks": {},
            "pytorch-lightning_version": "1.4.8"}
    if optimizer is not None:
        if hasattr(optimizer, "torch_state_dict"):  # FlatAdam -> torch.optim.Adam's layout in the reference's parameter
            # order (utils/__init__.py:10-31 over [embedding_a, {neuconw, nerf}])
            ckpt["optimizer_states"] = [optimizer.torch_state_dict(reference_param_order(embedding_a, neuconw, nerf))]
        else:
            ckpt["optimizer_states"] = [optimizer.state_dict()]
    if extra:
        ckpt.update(extra)
    torch.save(ckpt, path)
    return ckpt


def load_checkpoint(path, embedding_a=None, neuconw=None, nerf=None, strict=True, flat_params=None):
    """The reference's load_ckpt (utils/__init__.py:79-98) for the three modules at once.  With `flat_params`
    (trainer.FlatParams) the packed-weight caches are told that the parameters changed."""
    ckpt = torch.load(path, map_location="cpu")
    sd = ckpt["state_dict"] if "state_dict" in ckpt else ckpt  # a PL checkpoint or a bare state_dict
    for prefix, mod in (("embedding_a", embedding_a), ("neuconw", neuconw), ("nerf", nerf)):
        if mod is None:
            continue
        sub = {k[len(prefix) + 1:]: v for k, v in sd.items() if k.startswith(prefix + ".")}
        with torch.no_grad():
            mod.load_state_dict(sub, strict=strict)  # copies INTO the existing storage (flat views stay views)
    if flat_params is not None:
        flat_params.mark_updated()
    return ckpt
