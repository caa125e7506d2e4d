"""Drop-in mirror of the reference's `models/model.py::MyModel` (ref/models/model.py:8-42).

Same constructor (`MyModel(args)`), same `forward(images, source_encoding, target_encoding=None,
return_loss=True)`, same `save` / `load`, same attributes (`.transformer`, `.image_model`,
`.language_model`) and the same state-dict key schema as the HuggingFace modules the reference
wraps -- but forward + backward run in the native engine (libklab_mm.so), not in `transformers`.
`train.py` (ref/train.py) works unchanged: `.to(device)`, `DDP(model)`, `Adam(model.module.
transformer.parameters())`, `model.module.transformer.train()/eval()`, `loss.item()`,
`loss.backward()`, `model.module.save(...)`.
"""
import os
from typing import Dict, List

import torch
from torch import nn

from .. import hf_io
from ..engine import Engine, SwinConfig, T5Config

TIED_T5 = ("encoder.embed_tokens.weight", "decoder.embed_tokens.weight", "lm_head.weight")  # HF/t5:902-906


class _Node(nn.Module):
    """anonymous container so that dotted HuggingFace names become real module paths."""


class HFTree(nn.Module):
    """Parameters registered under HuggingFace state-dict names."""

    def __init__(self, specs, tied: Dict[str, str] = None):
        super().__init__()
        self._order: List[str] = []
        for spec in specs:
            self._register(spec.name, nn.Parameter(torch.empty(spec.shape, dtype=torch.float32)))
            self._order.append(spec.name)
        for alias, target in (tied or {}).items():
            self._register(alias, self.get_parameter(target))

    def _register(self, dotted, param):
        parts = dotted.split(".")
        mod = self
        for p in parts[:-1]:
            if not hasattr(mod, p):
                mod.add_module(p, _Node())
            mod = getattr(mod, p)
        mod.register_parameter(parts[-1], param)

    def ordered(self):
        return [self.get_parameter(n) for n in self._order]


def _init_t5_(tree: HFTree, cfg: T5Config, gen):
    """statistics of HF `_init_weights` (HF/t5:562-616), initializer_factor 1."""
    d, dk, h, ff = cfg.d_model, cfg.d_kv, cfg.num_heads, cfg.d_ff
    with torch.no_grad():
        for n, p in tree.named_parameters():
            if n in TIED_T5:
                continue
            if n.endswith("layer_norm.weight"):
                p.fill_(1.0)
                continue
            std = 1.0
            if n.endswith(".q.weight"):
                std = (d * dk) ** -0.5
            elif n.endswith(".k.weight") or n.endswith(".v.weight"):
                std = d ** -0.5
            elif n.endswith(".o.weight"):
                std = (h * dk) ** -0.5
            elif n.endswith("wi.weight"):
                std = d ** -0.5
            elif n.endswith("wo.weight"):
                std = ff ** -0.5
            elif n.endswith("relative_attention_bias.weight"):
                std = d ** -0.5
            p.copy_(torch.randn(p.shape, generator=gen) * std)


def _init_swin_(tree: HFTree, cfg: SwinConfig, gen):
    """statistics of HF `_init_weights` (HF/swinv2:873-886): N(0, 0.02) weights, zero biases, unit LayerNorm, log(10) scale."""
    with torch.no_grad():
        for n, p in tree.named_parameters():
            if n.endswith("logit_scale"):
                p.fill_(float(torch.log(torch.tensor(10.0))))
            elif "norm" in n.split(".")[-2] or n.startswith("layernorm"):
                p.fill_(1.0 if n.endswith("weight") else 0.0)
            elif n.endswith("bias"):
                p.zero_()
            else:
                p.copy_(torch.randn(p.shape, generator=gen) * 0.02)


class _LossFn(torch.autograd.Function):
    """autograd boundary: one node for the whole forward; backward runs the engine's segments."""

    @staticmethod
    def forward(ctx, model, want_grad, pixels, src, tgt, *params):
        eng = model._engine_for(pixels, src, tgt)
        # a fresh tensor per step, written by the cross-entropy kernel itself (no copy launch behind it)
        out = torch.empty(1, device=pixels.device, dtype=torch.float32)
        direct_loss = eng.set_loss_out(out)
        # the engine keeps a device-side counter RNG: the base only (re)seeds it, every forward advances it
        try:
            eng.forward(pixels, src, tgt, training=int(model.transformer.training) | (2 if model._frozen_unchanged() else 0) | (4 if model._trainable_current() else 0),
                        seed=model._seed_base, want_grad=want_grad)
        except Exception:
            eng.set_loss_out(None)  # a forward that failed before its cross-entropy must not leave the one-shot pointer armed
            raise
        ctx.model = model
        ctx.eng = eng
        ctx.nparams = len(params)
        model._fwd_token += 1
        ctx.token = model._fwd_token
        return out[0] if direct_loss else eng.loss_view[0].clone()

    @staticmethod
    def backward(ctx, gout):
        model, eng = ctx.model, ctx.eng
        if ctx.token != model._fwd_token or eng is not model._engine:
            raise RuntimeError("klab MyModel: backward() must follow its own forward() (activations live in one workspace)")
        g = gout.detach().to(torch.float32).contiguous().view(1)
        direct = model._direct_grads
        targets = model._grad_targets()
        nseg = 3 if model.args.image_model_train else 2
        if not direct:
            for seg in range(nseg):
                eng.backward(seg, g)
                if model._segment_hook is not None:
                    model._segment_hook(seg)
        else:
            # .grad tensors alias the flat buffers the engine is about to overwrite.  Per flat buffer: parameters whose
            # .grad is None get fresh gradients; parameters that still hold a gradient (accumulation steps, or the Swin
            # gradients the reference never zeroes) keep the running sum.
            state, held = {}, {}
            for mname in ("main", "swin"):
                ps = [p for p, _v, mn in model._views if mn == mname and p.requires_grad]
                if model._flat.get(mname) is None or not ps:
                    continue
                n_set = sum(p.grad is not None for p in ps)
                state[mname] = "fresh" if n_set == 0 else ("all" if n_set == len(ps) else "mixed")
                if n_set:
                    held[mname] = model._flat[mname].clone()
            seg_model = ("main", "main", "swin")

            def settle(mname):
                st = state.get(mname)
                if st == "all":
                    model._flat[mname].add_(held[mname])  # every .grad aliases the flat buffer: one fused add
                elif st is not None:
                    for p, view, mn in model._views:
                        if mn != mname or not p.requires_grad:
                            continue
                        if p.grad is None:
                            p.grad = view
                        elif st == "mixed":
                            off = view.storage_offset()
                            view.add_(held[mname][off:off + view.numel()].view(view.shape))

            for seg in range(nseg):
                eng.backward(seg, g)
                last_of_model = seg == nseg - 1 or seg_model[seg + 1] != seg_model[seg]
                if state.get(seg_model[seg]) == "fresh":
                    if model._segment_hook is not None:
                        model._segment_hook(seg)
                    opt = model._optimizer_in_backward() if model._optimizer_in_backward is not None else None
                    if opt is not None and seg_model[seg] == "main":
                        opt._segment_ready(model, seg)  # optim.FusedAdam(step_in_backward=True)
                    if last_of_model:
                        settle(seg_model[seg])
                elif last_of_model:
                    settle(seg_model[seg])
                    if model._segment_hook is not None:
                        for s2 in range(nseg):
                            if seg_model[s2] == seg_model[seg]:
                                # the running sums were added on the current stream AFTER the engine's per-layer events:
                                # the reducer must order itself behind the stream, not behind those events
                                model._segment_hook(s2, False)
            return (None,) * (5 + ctx.nparams)
        grads = []
        views = {id(p): v for p, v in targets}
        for p in model._trainable():
            grads.append(views[id(p)])
        return (None, None, None, None, None) + tuple(grads)


class MyModel(nn.Module):
    def __init__(self, args, _configs=None, _state_dicts=None, _seed=0, dtype=None):
        super().__init__()
        self.args = args
        self.result_dir = args.result_dir
        if _configs is None:
            lang_cfg, lang_sd = hf_io.resolve(args.language_model_name, "t5")          # ref/models/model.py:14
            swin_cfg, swin_sd = hf_io.resolve(args.image_model_name, "swin")           # :15
            main_cfg, main_sd = hf_io.resolve(args.transformer_model_name, "t5")       # :17
        else:
            swin_cfg, lang_cfg, main_cfg = _configs
            swin_sd, lang_sd, main_sd = _state_dicts or (None, None, None)
        self.swin_cfg, self.lang_cfg, self.main_cfg = swin_cfg, lang_cfg, main_cfg
        dt = dtype or os.environ.get("KLAB_DTYPE", "bf16")
        # "fp8" (BASELINE configs[4]): bf16 storage and backward, forward Linear GEMMs on fp8 MFMA with per-row scales
        self.compute_dtype = {"bf16": torch.bfloat16, "fp32": torch.float32, "fp8": "fp8", torch.bfloat16: torch.bfloat16,
                              torch.float32: torch.float32}[dt]
        self._engine = Engine(swin_cfg, lang_cfg, main_cfg, self.compute_dtype, bool(args.image_model_train))
        tied = {a: "shared.weight" for a in TIED_T5}
        self.language_model = HFTree(self._engine.params["lang"], {"encoder.embed_tokens.weight": "shared.weight"})
        self.image_model = HFTree(self._engine.params["swin"])
        self.transformer = HFTree(self._engine.params["main"], tied)
        gen = torch.Generator().manual_seed(_seed)
        for tree, cfg, sd, init in ((self.image_model, swin_cfg, swin_sd, _init_swin_), (self.language_model, lang_cfg, lang_sd, _init_t5_),
                                    (self.transformer, main_cfg, main_sd, _init_t5_)):
            if sd is None:
                init(tree, cfg, gen)
            else:
                self._load_tree(tree, sd)
        self.language_model.requires_grad_(False)                    # ref/models/model.py:14
        self.image_model.requires_grad_(bool(args.image_model_train))  # :15
        # from_pretrained returns eval-mode modules and the reference only ever toggles .transformer
        self.language_model.eval()
        self.image_model.eval()
        self.transformer.eval()
        self._bound_key = None
        self._flat = {}
        self._views = None
        self._direct_grads = False
        self._segment_hook = None
        self._optimizer_in_backward = None  # weakref to an optim.FusedAdam(step_in_backward=True) after its first fused step
        self._pending_opt_stream = None  # stream of an in-backward optimizer update that step() has not joined yet
        self._reducer = None  # klab DDP's SegmentReducer
        self._pending_reduce = None  # klab DDP(overlap_optimizer=True): reducer whose last all-reduces are not joined yet
        self.use_graph = os.environ.get("KLAB_GRAPH", "0") == "1"  # hipGraph replay of the engine's launch sequences
        self._seed_base = torch.initial_seed() & 0xFFFFFFFF
        self._pending_rng = None  # (base, counter) restored by load_checkpoint, applied at the next bind
        self._frozen_fp = None
        self._train_fp = None  # version fingerprint of the trainable T5 right after a klab FusedAdam step (bf16 copies current)
        self._fwd_token = 0
        import weakref
        for p in self.transformer.parameters():  # lets klab_multimodalmodel_amd.optim.FusedAdam find its engine
            p._klab_owner = weakref.ref(self)

    # ---- weights -----------------------------------------------------------------------------
    @staticmethod
    def _load_tree(tree, sd):
        own = tree.state_dict()
        missing = [k for k in own if k not in sd and k not in TIED_T5 and k != "encoder.embed_tokens.weight"]
        if "shared.weight" not in sd and "encoder.embed_tokens.weight" in sd:
            sd = dict(sd)
            sd["shared.weight"] = sd["encoder.embed_tokens.weight"]
            missing = [k for k in missing if k != "shared.weight"]
        if missing:
            raise RuntimeError(f"Missing key(s) in state_dict: {missing[:5]}{'...' if len(missing) > 5 else ''}")
        with torch.no_grad():
            for k, p in tree.named_parameters():
                if k in sd:
                    if tuple(sd[k].shape) != tuple(p.shape):
                        raise RuntimeError(f"size mismatch for {k}: checkpoint {tuple(sd[k].shape)} vs model {tuple(p.shape)}")
                    p.copy_(sd[k].to(torch.float32))

    def _apply(self, fn, *a, **k):  # .to()/.cuda() replace parameter storage => rebind lazily
        self._bound_key = None
        return super()._apply(fn, *a, **k)

    def train(self, mode: bool = True):
        # nn.Module.train() would flip the frozen towers too; DDP(model) / model.train() callers expect the
        # reference's effective behaviour only if they call it -- train.py never does (ref/train.py:52 toggles
        # .transformer only).  Keep torch semantics: propagate, but Swin / language encoder have no train-mode ops here.
        return super().train(mode)

    # ---- engine plumbing ---------------------------------------------------------------------
    def _trainable(self):
        ps = list(self.transformer.ordered())
        if self.args.image_model_train:
            ps += list(self.image_model.ordered())
        return [p for p in ps if p.requires_grad]

    def _engine_for(self, pixels, src, tgt):
        B, Ls = src.shape
        Lt = tgt.shape[1]
        dev = pixels.device
        if dev.type != "cuda":
            raise RuntimeError("klab MyModel runs on an MI355X (cuda/HIP device) only; there is no CPU fallback")
        key = (B, Ls, Lt, dev)
        if self._bound_key != key:
            eng = self._engine
            tensors = {"swin": [p.data for p in self.image_model.ordered()], "lang": [p.data for p in self.language_model.ordered()],
                       "main": [p.data for p in self.transformer.ordered()]}
            for ts in tensors.values():
                for t in ts:
                    if t.device != dev:
                        raise RuntimeError(f"Expected all tensors to be on the same device, but found {t.device} and {dev}")
            if "main" not in self._flat or self._flat["main"].device != dev:
                self._flat["main"] = torch.zeros(eng.grad_elems["main"], device=dev)
                self._flat["swin"] = torch.zeros(max(eng.grad_elems["swin"], 8), device=dev) if self.args.image_model_train else None
                self._views = None
            eng.bind(B, Ls, Lt, tensors, self._flat["main"], self._flat["swin"], dev)
            eng.set_graph(self.use_graph)
            if self._pending_rng is not None:  # load_checkpoint before the first forward: continue the saved dropout stream
                eng.set_rng(*self._pending_rng)
                self._pending_rng = None
            self._bound_key = key
            self._frozen_fp = None
            self._train_fp = None
        return self._engine

    def _note_optimizer_step(self):
        """called by optim.FusedAdam: the step kernel rewrote the masters AND their compute-dtype copies."""
        self._train_fp = sum(p._version for p in self.transformer.parameters())

    def _trainable_current(self):
        """True when nothing wrote the trainable T5 since the last fused optimizer step (tensor version counters) and the
        engine was not re-bound: the forward may then skip its fp32 -> bf16 weight cast."""
        if self._train_fp is None:
            return False
        if sum(p._version for p in self.transformer.parameters()) != self._train_fp:
            self._train_fp = None
            return False
        return True

    def _frozen_unchanged(self):
        """True when no frozen-tower parameter was written since the previous forward (tensor version counters):
        the engine may then keep its bf16 copies and the Swin position-bias tables."""
        ps = list(self.language_model.parameters())
        if not self.args.image_model_train:
            ps += list(self.image_model.parameters())
        fp = sum(p._version for p in ps)
        same = fp == self._frozen_fp
        self._frozen_fp = fp
        return same

    def _grad_targets(self):
        if self._views is None:
            out = []
            for mname, tree in (("main", self.transformer), ("swin", self.image_model)):
                flat = self._flat.get(mname)
                if flat is None:
                    continue
                for spec, p in zip(self._engine.params[mname], tree.ordered()):
                    if spec.grad_off >= 0:
                        out.append((p, flat[spec.grad_off:spec.grad_off + p.numel()].view(p.shape), mname))
            self._views = out
        return [(p, v) for p, v, _m in self._views if p.requires_grad]

    def flat_grads(self, model="main"):
        return self._flat.get(model)

    # ---- the reference surface -----------------------------------------------------------------
    def forward(self, images, source_encoding, target_encoding=None, return_loss=True):
        if self._pending_opt_stream is not None:  # an in-backward optimizer update nobody joined through step()
            torch.cuda.current_stream().wait_stream(self._pending_opt_stream)
            self._pending_opt_stream = None
        if self._pending_reduce is not None:  # nobody consumed the last backward's gradients through FusedAdam: join now
            red, self._pending_reduce = self._pending_reduce, None
            red.finish()
        pixels = images["pixel_values"]
        src = source_encoding["input_ids"]
        if pixels.dim() != 4 or pixels.shape[1] != self.swin_cfg.num_channels:
            raise ValueError("Make sure that the channel dimension of the pixel values match with the one set in the configuration.")
        if pixels.shape[2] != self.swin_cfg.image_size or pixels.shape[3] != self.swin_cfg.image_size:
            raise NotImplementedError(f"images must be {self.swin_cfg.image_size}x{self.swin_cfg.image_size} (the configured size)")
        pixels = pixels.to(torch.float32).contiguous()
        src = src.to(torch.int64).contiguous()
        if return_loss:
            tgt = target_encoding["input_ids"].to(torch.int64).contiguous()
            params = self._trainable()
            # (grad mode is off inside Function.forward, so decide here whether d logits must be produced)
            want_grad = torch.is_grad_enabled() and len(params) > 0
            return _LossFn.apply(self, want_grad, pixels, src, tgt, *params)
        return self.generate(pixels, src)

    @torch.no_grad()
    def generate(self, pixels, src, max_length=20, kv_cache=True):
        """greedy decoding with HF's default generation settings (ref/models/model.py:28: max_length 20, no sampling).
        Prefill = one evaluation-mode forward (Swin, both encoders, the cross K/V of all layers, decoder position 0); every
        further token runs the decoder over ONE new position against the per-layer K/V cache (`klab_engine_decode_step`,
        SURVEY §8 row f-3; HF/t5:308-332).  kv_cache=False keeps the round-1 form -- decoder + LM head over the whole prefix per
        token -- as the cross-check of the cache."""
        B = src.shape[0]
        cfg = self.main_cfg
        steps = max_length - 1
        tgt = torch.full((B, steps), cfg.pad_token_id, dtype=torch.int64, device=src.device)
        done = torch.zeros(B, dtype=torch.bool, device=src.device)
        was_training = self.transformer.training
        self.transformer.eval()
        try:
            nxt = None
            for t in range(steps):
                eng = self._engine_for(pixels, src, tgt)
                if t == 0 or not kv_cache:
                    eng.forward(pixels, src, tgt, training=(8 if t > 0 else 0) | (2 if t > 0 else 0), seed=self._seed_base, want_grad=False)
                    logits = eng.buffer("logits").view(B, steps, -1)[:, t].float()
                else:
                    eng.decode_step(t, nxt)
                    logits = eng.buffer("logits_step").float()
                nxt = logits.argmax(-1)
                nxt = torch.where(done, torch.full_like(nxt, cfg.pad_token_id), nxt).contiguous()
                tgt[:, t] = nxt
                done |= nxt == cfg.eos_token_id
                if bool(done.all()):
                    tgt = tgt[:, :t + 1]
                    break
        finally:
            self.transformer.train(was_training)
        start = torch.full((B, 1), cfg.decoder_start_token_id, dtype=torch.int64, device=src.device)
        return torch.cat([start, tgt], dim=1)

    def _join_pending_update(self):
        """an optimizer update still running on its own stream (optim.FusedAdam(step_in_backward=True)) writes the weights:
        anything that reads them on the current stream waits for it first"""
        if self._pending_opt_stream is not None:
            torch.cuda.current_stream().wait_stream(self._pending_opt_stream)
            self._pending_opt_stream = None

    def state_dict(self, *a, **k):
        self._join_pending_update()
        return super().state_dict(*a, **k)

    def save(self, result_name="best.pth"):
        self._join_pending_update()
        result_path = os.path.join(self.args.result_dir, result_name)
        checkpoints = {'transformer': self.transformer.state_dict()}
        if self.args.image_model_train:
            checkpoints['image_model'] = self.image_model.state_dict()
        torch.save(checkpoints, result_path)

    def load(self, result_name="best.pth"):
        result_path = os.path.join(self.args.result_dir, result_name)
        checkpoints = torch.load(result_path)
        self.transformer.load_state_dict(checkpoints['transformer'])
        if self.args.image_model_train:
            self.image_model.load_state_dict(checkpoints['image_model'])
