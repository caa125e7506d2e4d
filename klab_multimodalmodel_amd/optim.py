"""Fused Adam for the native MyModel.

Drop-in for the reference's `torch.optim.Adam(model.module.transformer.parameters(), lr=args.lr)` (ref/train.py:28): same
constructor arguments, same update rule (no amsgrad, L2 weight decay), `param_groups` / LR schedulers / `zero_grad` /
`state_dict` as usual.  When every parameter belongs to one klab MyModel whose gradients live in the engine's flat buffer
(the direct-gradient mode that `MyModel._direct_grads` / klab DDP use), `step()` is ONE kernel over all tensors that also
rewrites the bf16 copies the next forward's GEMMs read -- the engine's separate fp32->bf16 cast pass is skipped.  In every
other situation (foreign parameters, several param groups with different hyper-parameters, amsgrad, gradients that are not
views of the flat buffer) it silently delegates to `torch.optim.Adam`, so it is always safe to use.

`step_in_backward=True` (opt-in; ONLY for loops in which every `backward()` is followed by `step()`, i.e. the reference's loop
with its default `--accumulation_steps 1`, ref/train.py:61-67, ref/modules/config.py:16): the update of backward segment 0
(LM head / shared embedding / decoder, ~70 % of the trainable parameters) is enqueued on a side stream as soon as that
segment's gradients are final (and, under klab DDP, averaged), so that the HBM-bound Adam kernel runs underneath the
latency-bound encoder backward; `step()` then updates segment 1 and joins.  Same arithmetic, same result; the first step and
every step after a change of circumstances (`_fast_ok`) run the ordinary way.  Measured on configs[1] (one MI355X): 6.67 ms/step
against 6.63 without -- the co-running streaming kernel slows the chain by as much as it hides -- so nothing turns it on by default.
"""
import weakref

import torch
from torch.optim import Adam as _TorchAdam
from torch.optim import Optimizer


class FusedAdam(Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False, *, maximize=False,
                 step_in_backward=False):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("invalid Adam hyper-parameter")
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad, maximize=maximize)
        super().__init__(params, defaults)
        self._steps = 0
        self._m = self._v = None
        self._owner = None
        self._fallback = None
        self._fb_reason = None
        self.step_in_backward = bool(step_in_backward)
        self._opt_stream = None
        self._bw_token = None  # forward token of the backward whose segment 0 was already updated
        self.in_backward_updates = 0  # how many segment-0 updates ran underneath a backward (tests, bench)

    def _hyper(self, steps):
        g = self.param_groups[0]
        b1, b2 = g["betas"]
        return (float(g["lr"]), float(b1), float(b2), float(g["eps"]), float(g["weight_decay"]), 1.0 - b1 ** steps, 1.0 - b2 ** steps)

    @torch.no_grad()
    def _segment_ready(self, model, seg):
        """called by the model's backward when segment `seg`'s gradients are final on the current stream (and their all-reduce,
        if any, is enqueued): update segment 0 underneath the rest of the backward"""
        if seg != 0 or self._fallback is not None or self._m is None or len(self.param_groups) != 1:
            return
        if self._bw_token is not None:
            # a second backward() before step(): gradient accumulation (ref/train.py --accumulation_steps > 1).  The first
            # micro-batch's gradients were already applied to segment 0 underneath its backward -- the update is not what Adam
            # would have done with the accumulated sum.  The contract is one backward per step: say so instead of training wrong.
            raise RuntimeError("FusedAdam(step_in_backward=True) supports exactly one backward() per step(); with gradient "
                               "accumulation construct it with step_in_backward=False")
        g = self.param_groups[0]
        if g["amsgrad"] or g["maximize"] or self._owner is None or self._owner() is not model:
            return
        flat = model._flat.get("main")
        if flat is None or self._m.shape != flat.shape or self._m.device != flat.device:
            return
        cur = torch.cuda.current_stream(flat.device)
        if self._opt_stream is None:
            self._opt_stream = torch.cuda.Stream(device=flat.device)
        self._opt_stream.wait_stream(cur)
        with torch.cuda.stream(self._opt_stream):
            red = getattr(model, "_reducer", None)
            if red is not None:
                red.finish_segment(0, flat.device)  # the current stream is the optimizer stream here
            model._engine.adam_step(self._m, self._v, *self._hyper(self._steps + 1), segment=0)
        self._bw_token = model._fwd_token
        self.in_backward_updates += 1
        model._pending_opt_stream = self._opt_stream

    # ---- which model owns these parameters -------------------------------------------------------------------------
    def _find_owner(self):
        owner = None
        for g in self.param_groups:
            for p in g["params"]:
                ref = getattr(p, "_klab_owner", None)
                m = ref() if ref is not None else None
                if m is None or (owner is not None and m is not owner):
                    return None
                owner = m
        return owner

    def _fast_ok(self):
        """(model, None) when the one-kernel path applies, else (None, reason)."""
        if len(self.param_groups) != 1:
            return None, "several param groups"
        g = self.param_groups[0]
        if g["amsgrad"] or g["maximize"]:
            return None, "amsgrad / maximize"
        model = self._find_owner()
        if model is None:
            return None, "parameters not owned by one klab MyModel"
        if model._flat.get("main") is None or model._engine.shape is None:
            return None, "engine not bound yet"
        views = {id(p): v for p, v, mn in (model._views or []) if mn == "main"}
        mine = {id(p) for p in g["params"]}
        if mine != {id(p) for p in model.transformer.ordered() if p.requires_grad} or set(views) != mine:
            return None, "not exactly the trainable T5 parameters"
        for p in g["params"]:
            v = views[id(p)]
            if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                return None, "gradients are not views of the flat buffer"
        return model, None

    # ---- torch.optim API --------------------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        model, why = (None, "fallback already active") if self._fallback is not None else self._fast_ok()
        if model is None:
            if self._bw_token is not None:
                raise RuntimeError(f"FusedAdam(step_in_backward=True): segment 0 was updated during backward() but step() cannot "
                                   f"take the fused path any more ({why})")
            for grp in self.param_groups:  # gradients may still be in flight (klab DDP overlap_optimizer): join before torch reads them
                for p in grp["params"]:
                    ref = getattr(p, "_klab_owner", None)
                    owner = ref() if ref is not None else None
                    red = getattr(owner, "_pending_reduce", None) if owner is not None else None
                    if red is not None:
                        owner._pending_reduce = None
                        red.finish()
            self._fb_reason = why
            self._step_fallback()
            return loss
        flat = model._flat["main"]
        if self._m is None or self._m.shape != flat.shape or self._m.device != flat.device:
            if self._bw_token is not None:
                raise RuntimeError("FusedAdam(step_in_backward=True): the optimizer state changed between backward() and step()")
            self._m = torch.zeros_like(flat)
            self._v = torch.zeros_like(flat)
        self._steps += 1
        hyper = self._hyper(self._steps)
        seg0_done = self._bw_token is not None  # (reset by every step, set by every backward: one backward per step is the contract)
        self._bw_token = None
        if seg0_done:  # segment 0 was updated during the backward: the compute stream continues behind it
            torch.cuda.current_stream(flat.device).wait_stream(self._opt_stream)
            model._pending_opt_stream = None

        def launch():
            red = getattr(model, "_pending_reduce", None)
            if red is not None:  # klab DDP(overlap_optimizer=True): segment 0 is updated while segment 1 is still being reduced
                model._pending_reduce = None
                for seg in (0, 1):
                    if seg == 0 and seg0_done:
                        continue
                    red.finish_segment(seg)
                    model._engine.adam_step(self._m, self._v, *hyper, segment=seg)
                red.finish()  # any further segment (Swin) and the bookkeeping
            elif seg0_done:
                model._engine.adam_step(self._m, self._v, *hyper, segment=1)
            else:
                model._engine.adam_step(self._m, self._v, *hyper)

        launch()
        model._note_optimizer_step()
        self._owner = weakref.ref(model)
        model._optimizer_in_backward = weakref.ref(self) if self.step_in_backward else None
        return loss

    def _step_fallback(self):
        if self._fallback is None:
            self._fallback = _TorchAdam(self.param_groups, fused=all(p.is_cuda for g in self.param_groups for p in g["params"]) or None)
            self._fallback.param_groups = self.param_groups  # share the group dicts (LR schedulers act on ours)
            if self._m is not None:  # carry the fused state over (one-way)
                model = self._owner() if self._owner is not None else None
                if model is not None:
                    for p, v, mn in model._views:
                        if mn != "main" or not p.requires_grad:
                            continue
                        off = v.storage_offset()
                        self._fallback.state[p] = {
                            "step": torch.tensor(float(self._steps), device=p.device if self._fallback.defaults.get("fused") else "cpu"),
                            "exp_avg": self._m[off:off + p.numel()].view(p.shape).clone(),
                            "exp_avg_sq": self._v[off:off + p.numel()].view(p.shape).clone()}
        self._fallback.step()

    def state_dict(self):
        """torch.optim.Adam-compatible: per-parameter `step`, `exp_avg`, `exp_avg_sq` (copies)."""
        if self._fallback is not None:
            return self._fallback.state_dict()
        model = self._owner() if self._owner is not None else None
        if model is not None and self._m is not None:
            for p, v, mn in model._views:
                if mn == "main" and p.requires_grad:
                    off = v.storage_offset()
                    self.state[p] = {"step": torch.tensor(float(self._steps)),
                                     "exp_avg": self._m[off:off + p.numel()].view(p.shape).clone(),
                                     "exp_avg_sq": self._v[off:off + p.numel()].view(p.shape).clone()}
        return super().state_dict()

    def load_state_dict(self, state_dict):
        """torch.optim.Adam-compatible.  The model must have been bound to a batch shape (one forward) so that the flat state
        layout is known; otherwise the loaded state is kept per parameter and the torch fallback continues from it."""
        super().load_state_dict(state_dict)
        model = self._find_owner()
        steps = 0
        if model is not None and model._flat.get("main") is not None:
            model._grad_targets()  # builds the parameter -> flat-buffer views if no backward has run yet
        if model is not None and model._views:
            flat = model._flat["main"]
            self._m, self._v = torch.zeros_like(flat), torch.zeros_like(flat)
            for p, v, mn in model._views:
                st = self.state.get(p)
                if mn != "main" or not st:
                    continue
                off = v.storage_offset()
                self._m[off:off + p.numel()].view(p.shape).copy_(st["exp_avg"])
                self._v[off:off + p.numel()].view(p.shape).copy_(st["exp_avg_sq"])
                steps = max(steps, int(float(st["step"])))
            self._owner = weakref.ref(model)
            self._steps = steps
            self._fallback = None
        else:  # no flat layout yet: continue with torch.optim.Adam on the loaded per-parameter state
            self._fallback = _TorchAdam(self.param_groups)
            self._fallback.param_groups = self.param_groups
            self._fallback.state = self.state
