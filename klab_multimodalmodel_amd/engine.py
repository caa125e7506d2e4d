"""Host-side handle on the native engine (klab_engine_* of include/klab_mm.h).

Builds the C config structs, mirrors the engine's parameter table (HuggingFace state-dict names),
computes the input-independent integer/float tables with the reference's own arithmetic
(T5 relative-position buckets HF/t5:216-262; Swin-V2 CPB coords/index HF/swinv2:457-492) and binds
caller-owned torch storage (parameters, flat gradient buffers, one workspace) to the plan.
"""
import ctypes as C
import math
from dataclasses import dataclass
from typing import List, Optional, Sequence

import torch

from . import _lib as L


# ---- config mirrors (field names follow HF/swinv2cfg:56-73 and HF/t5cfg:44-62) -----------------
@dataclass
class SwinConfig:
    image_size: int = 224
    patch_size: int = 4
    num_channels: int = 3
    embed_dim: int = 96
    depths: Sequence[int] = (2, 2, 6, 2)
    num_heads: Sequence[int] = (3, 6, 12, 24)
    window_size: int = 7
    pretrained_window_sizes: Sequence[int] = (0, 0, 0, 0)
    mlp_ratio: float = 4.0
    qkv_bias: bool = True
    layer_norm_eps: float = 1e-5

    @classmethod
    def from_dict(cls, d):
        keys = cls.__dataclass_fields__.keys()
        c = cls(**{k: d[k] for k in keys if k in d and d[k] is not None})
        if isinstance(c.image_size, (list, tuple)):
            if c.image_size[0] != c.image_size[1]:
                raise NotImplementedError("non-square images are outside the scoped configs")
            c.image_size = c.image_size[0]
        if isinstance(c.patch_size, (list, tuple)):
            c.patch_size = c.patch_size[0]
        return c

    @property
    def hidden_size(self):
        return int(self.embed_dim * 2 ** (len(self.depths) - 1))


@dataclass
class T5Config:
    vocab_size: int = 32128
    d_model: int = 512
    d_kv: int = 64
    d_ff: int = 2048
    num_layers: int = 6
    num_decoder_layers: Optional[int] = None
    num_heads: int = 8
    relative_attention_num_buckets: int = 32
    relative_attention_max_distance: int = 128
    dropout_rate: float = 0.1
    layer_norm_epsilon: float = 1e-6
    feed_forward_proj: str = "relu"
    decoder_start_token_id: Optional[int] = 0
    pad_token_id: Optional[int] = 0
    eos_token_id: int = 1
    scale_decoder_outputs: bool = True

    def __post_init__(self):
        if self.num_decoder_layers is None:
            self.num_decoder_layers = self.num_layers

    @classmethod
    def from_dict(cls, d):
        keys = cls.__dataclass_fields__.keys()
        c = cls(**{k: d[k] for k in keys if k in d and d[k] is not None})
        if "decoder_start_token_id" in d and d["decoder_start_token_id"] is None:
            c.decoder_start_token_id = None
        if d.get("tie_word_embeddings", None) is False:  # HF/t5cfg:82-83
            c.scale_decoder_outputs = False
        return c


# ---- C structs -------------------------------------------------------------------------------
class CSwinCfg(C.Structure):
    _fields_ = [("image_size", C.c_int), ("patch", C.c_int), ("in_ch", C.c_int), ("embed_dim", C.c_int), ("n_stages", C.c_int),
                ("depths", C.c_int * 8), ("heads", C.c_int * 8), ("window", C.c_int), ("pretrained_window", C.c_int * 8),
                ("mlp_ratio", C.c_int), ("qkv_bias", C.c_int), ("ln_eps", C.c_float)]


class CT5Cfg(C.Structure):
    _fields_ = [("vocab", C.c_int), ("d_model", C.c_int), ("d_kv", C.c_int), ("n_heads", C.c_int), ("d_ff", C.c_int),
                ("n_layers", C.c_int), ("n_dec_layers", C.c_int), ("rel_buckets", C.c_int), ("rel_max_dist", C.c_int),
                ("dropout", C.c_float), ("ln_eps", C.c_float), ("start_id", C.c_int), ("pad_id", C.c_int),
                ("scale_decoder_outputs", C.c_int)]


class CModelCfg(C.Structure):
    _fields_ = [("swin", CSwinCfg), ("lang", CT5Cfg), ("main", CT5Cfg), ("dtype", C.c_int), ("train_swin", C.c_int)]


ENGINE_SIGS = {
    "klab_engine_create": ([C.POINTER(CModelCfg)], C.c_void_p),
    "klab_engine_destroy": ([C.c_void_p], None),
    "klab_engine_num_params": ([C.c_void_p, C.c_int], C.c_int),
    "klab_engine_param_info": ([C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_long), C.POINTER(C.c_int),
                                C.POINTER(C.c_long)], C.c_int),
    "klab_engine_grad_elems": ([C.c_void_p, C.c_int], C.c_long),
    "klab_engine_segment": ([C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_long), C.POINTER(C.c_long)], C.c_int),
    "klab_engine_num_buckets": ([C.c_void_p, C.c_int], C.c_int),
    "klab_engine_bucket": ([C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_long), C.POINTER(C.c_long)], C.c_int),
    "klab_engine_bucket_wait": ([C.c_void_p, C.c_int, C.c_int, C.c_void_p], C.c_int),
    "klab_engine_set_bucket_events": ([C.c_void_p, C.c_int], C.c_int),
    "klab_engine_workspace_bytes": ([C.c_void_p, C.c_int, C.c_int, C.c_int], C.c_size_t),
    "klab_engine_bind": ([C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                          C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p),
                          C.POINTER(C.c_void_p), C.c_void_p], C.c_int),
    "klab_engine_forward": ([C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_int, C.c_void_p], C.c_int),
    "klab_engine_decode_step": ([C.c_void_p, C.c_int, C.c_void_p, C.c_void_p], C.c_int),
    "klab_engine_backward": ([C.c_void_p, C.c_int, C.c_void_p, C.c_void_p], C.c_int),
    "klab_engine_set_graph": ([C.c_void_p, C.c_int], C.c_int),
    "klab_engine_get_rng": ([C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_void_p], C.c_int),
    "klab_engine_set_rng": ([C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p], C.c_int),
    "klab_engine_adam_step": ([C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                               C.c_float, C.c_void_p], C.c_int),
    "klab_engine_adam_step_segment": ([C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                       C.c_float, C.c_float, C.c_void_p], C.c_int),
    "klab_engine_probe_enable": ([C.c_void_p, C.c_int], C.c_int),
    "klab_engine_probe_read": ([C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_double)], C.c_int),
    "klab_engine_loss_ptr": ([C.c_void_p], C.c_void_p),
    "klab_engine_set_loss_out": ([C.c_void_p, C.c_void_p], C.c_int),
    "klab_engine_err_ptr": ([C.c_void_p], C.c_void_p),
    "klab_engine_rng_ptr": ([C.c_void_p], C.c_void_p),
    "klab_engine_buffer": ([C.c_void_p, C.c_char_p, C.POINTER(C.c_long), C.POINTER(C.c_long), C.POINTER(C.c_int)], C.c_void_p),
    "klab_gelu_fwd": ([C.c_void_p, C.c_void_p, C.c_int, C.c_long, C.c_void_p], C.c_int),
    "klab_swin_cpb_bias_bwd": ([C.c_void_p] * 11 + [C.c_int] * 4 + [C.c_void_p], C.c_int),
    "klab_swin_cpb_bias_bwd_pz": ([C.c_void_p] * 11 + [C.c_int] * 5 + [C.c_void_p], C.c_int),
}
_sigs_installed = False


def lib():
    global _sigs_installed
    l = L.load()
    if not _sigs_installed:
        for name, (argt, rest) in ENGINE_SIGS.items():
            fn = getattr(l, name)
            fn.argtypes = argt
            fn.restype = rest
        _sigs_installed = True
    return l


def _c_t5(c: T5Config) -> CT5Cfg:
    if c.feed_forward_proj != "relu":
        raise NotImplementedError("only the v1.0 ReLU feed-forward is reachable from the reference (ref/modules/config.py:8-9)")
    if c.decoder_start_token_id is None:
        raise ValueError("self.model.config.decoder_start_token_id has to be defined.")  # HF/t5:622-626
    if c.pad_token_id is None:
        raise ValueError("self.model.config.pad_token_id has to be defined.")  # HF/t5:632-633
    return CT5Cfg(c.vocab_size, c.d_model, c.d_kv, c.num_heads, c.d_ff, c.num_layers, c.num_decoder_layers,
                  c.relative_attention_num_buckets, c.relative_attention_max_distance, float(c.dropout_rate),
                  float(c.layer_norm_epsilon), int(c.decoder_start_token_id), int(c.pad_token_id), int(c.scale_decoder_outputs))


def _c_swin(c: SwinConfig) -> CSwinCfg:
    s = CSwinCfg()
    s.image_size, s.patch, s.in_ch, s.embed_dim, s.n_stages = c.image_size, c.patch_size, c.num_channels, c.embed_dim, len(c.depths)
    if len(c.depths) > 8:
        raise NotImplementedError("more than 8 Swin stages")
    for i, (dd, hh) in enumerate(zip(c.depths, c.num_heads)):
        s.depths[i], s.heads[i] = dd, hh
    for i, pw in enumerate(list(c.pretrained_window_sizes)[:len(c.depths)]):
        s.pretrained_window[i] = pw
    s.window = c.window_size
    if float(c.mlp_ratio) != int(c.mlp_ratio):
        raise NotImplementedError("fractional mlp_ratio")
    s.mlp_ratio, s.qkv_bias, s.ln_eps = int(c.mlp_ratio), int(c.qkv_bias), float(c.layer_norm_eps)
    return s


# ---- input-independent tables, computed with the reference's arithmetic ---------------------------
def t5_bucket_table(Lq: int, Lk: int, bidirectional: bool, num_buckets: int, max_distance: int) -> torch.Tensor:
    """T5Attention._relative_position_bucket over (memory - context) (HF/t5:216-262, 264-272).
    Kept in torch on the host so the float-log truncation is bit-identical to the reference's."""
    ctx = torch.arange(Lq, dtype=torch.long)[:, None]
    mem = torch.arange(Lk, dtype=torch.long)[None, :]
    rel = mem - ctx
    out = torch.zeros_like(rel)
    nb = num_buckets
    if bidirectional:
        nb //= 2
        out = out + (rel > 0).to(torch.long) * nb
        rel = torch.abs(rel)
    else:
        rel = -torch.min(rel, torch.zeros_like(rel))
    max_exact = nb // 2
    is_small = rel < max_exact
    large = max_exact + (torch.log(rel.float() / max_exact) / math.log(max_distance / max_exact) * (nb - max_exact)).to(torch.long)
    large = torch.min(large, torch.full_like(large, nb - 1))
    return (out + torch.where(is_small, rel, large)).to(torch.int32).contiguous()


def swin_cpb_tables(w: int, pretrained_w: int):
    """relative_coords_table [(2w-1)^2, 2] f32 and relative_position_index [w^2*w^2] i32 (HF/swinv2:457-492)."""
    rc = torch.arange(-(w - 1), w, dtype=torch.int64).float()
    table = torch.stack(torch.meshgrid([rc, rc], indexing="ij")).permute(1, 2, 0).contiguous().unsqueeze(0)
    if pretrained_w > 0:
        table = table / (pretrained_w - 1)
    elif w > 1:
        table = table / (w - 1)
    table = table * 8
    table = torch.sign(table) * torch.log2(torch.abs(table) + 1.0) / math.log2(8)
    c = torch.arange(w)
    coords = torch.stack(torch.meshgrid([c, c], indexing="ij")).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += w - 1
    rel[:, :, 1] += w - 1
    rel[:, :, 0] *= 2 * w - 1
    return table.view(-1, 2).float().contiguous(), rel.sum(-1).view(-1).to(torch.int32).contiguous()


@dataclass
class ParamSpec:
    name: str
    shape: tuple
    grad_off: int


class Engine:
    """One native plan per (configs, dtype, train_swin).  bind() attaches storage for a batch shape."""

    MODELS = ("swin", "lang", "main")

    def __init__(self, swin: SwinConfig, lang: T5Config, main: T5Config, dtype=torch.bfloat16, train_swin=False):
        self.swin_cfg, self.lang_cfg, self.main_cfg = swin, lang, main
        self.dtype = dtype
        self.train_swin = bool(train_swin)
        if swin.hidden_size != main.d_model:
            # the reference fails at its torch.cat (ref/models/model.py:23) with this RuntimeError
            raise RuntimeError(f"Sizes of tensors must match except in dimension 1. Expected size {swin.hidden_size} "
                               f"but got size {main.d_model} for tensor number 1 in the list.")
        if lang.d_model != main.d_model:
            raise RuntimeError(f"Sizes of tensors must match except in dimension 1. Expected size {swin.hidden_size} "
                               f"but got size {lang.d_model} for tensor number 1 in the list.")
        self._cfg = CModelCfg(_c_swin(swin), _c_t5(lang), _c_t5(main), L.dtype_code(dtype), int(self.train_swin))
        self._lib = lib()
        self._h = self._lib.klab_engine_create(C.byref(self._cfg))
        if not self._h:
            raise ValueError("klab_engine_create rejected the configuration")
        self.params = {m: self._param_table(i) for i, m in enumerate(self.MODELS)}
        self.grad_elems = {m: int(self._lib.klab_engine_grad_elems(self._h, i)) for i, m in enumerate(self.MODELS)}
        self.segments = []
        for s in range(3):
            mi, off, ln = C.c_int(), C.c_long(), C.c_long()
            L.check(self._lib.klab_engine_segment(self._h, s, C.byref(mi), C.byref(off), C.byref(ln)), "klab_engine_segment")
            self.segments.append((self.MODELS[mi.value], off.value, ln.value))
        self.buckets = []  # per segment: [(offset, length)] of the layer buckets, in the order backward finishes them
        for sg in range(3):
            out = []
            for i in range(max(0, self._lib.klab_engine_num_buckets(self._h, sg))):
                off, ln = C.c_long(), C.c_long()
                L.check(self._lib.klab_engine_bucket(self._h, sg, i, C.byref(off), C.byref(ln)), "klab_engine_bucket")
                out.append((off.value, ln.value))
            self.buckets.append(out)
        self._keep = None
        self.shape = None

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                self._lib.klab_engine_destroy(h)
            except Exception:
                pass

    def _param_table(self, mi) -> List[ParamSpec]:
        n = self._lib.klab_engine_num_params(self._h, mi)
        out = []
        buf = C.create_string_buffer(256)
        shape = (C.c_long * 4)()
        nd, go = C.c_int(), C.c_long()
        for i in range(n):
            L.check(self._lib.klab_engine_param_info(self._h, mi, i, buf, 256, shape, C.byref(nd), C.byref(go)), "param_info")
            out.append(ParamSpec(buf.value.decode(), tuple(shape[k] for k in range(nd.value)), go.value))
        return out

    def workspace_bytes(self, B, Ls, Lt) -> int:
        return int(self._lib.klab_engine_workspace_bytes(self._h, B, Ls, Lt))

    def bind(self, B, Ls, Lt, tensors, main_grads, swin_grads, device):
        """tensors: {model: [fp32 device tensors in table order]}."""
        nbytes = self.workspace_bytes(B, Ls, Lt)
        if nbytes == 0:
            raise ValueError("bad batch shape")
        ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        m = self.main_cfg
        lang_b = t5_bucket_table(Ls, Ls, True, self.lang_cfg.relative_attention_num_buckets,
                                 self.lang_cfg.relative_attention_max_distance).to(device)
        Le = self.n_img + Ls
        enc_b = t5_bucket_table(Le, Le, True, m.relative_attention_num_buckets, m.relative_attention_max_distance).to(device)
        dec_b = t5_bucket_table(Lt, Lt, False, m.relative_attention_num_buckets, m.relative_attention_max_distance).to(device)
        s = self.swin_cfg
        R0 = s.image_size // s.patch_size
        coords, index = [], []
        for st in range(len(s.depths)):
            R = R0 >> st
            w = min(R, s.window_size)
            pw = list(s.pretrained_window_sizes)[st] if st < len(s.pretrained_window_sizes) else 0
            ct, ix = swin_cpb_tables(w, pw)
            coords.append(ct.to(device))
            index.append(ix.to(device))
        arrs = {}
        for mname in self.MODELS:
            ts = tensors[mname]
            assert len(ts) == len(self.params[mname])
            for t, spec in zip(ts, self.params[mname]):
                if t.dtype != torch.float32 or not t.is_contiguous() or tuple(t.shape) != spec.shape or t.device != ws.device:
                    raise ValueError(f"parameter {mname}.{spec.name}: expected contiguous fp32 {spec.shape} on {ws.device}, "
                                     f"got {t.dtype} {tuple(t.shape)} on {t.device}")
            arrs[mname] = (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        carr = (C.c_void_p * len(coords))(*[t.data_ptr() for t in coords])
        iarr = (C.c_void_p * len(index))(*[t.data_ptr() for t in index])
        rc = self._lib.klab_engine_bind(self._h, B, Ls, Lt, ws.data_ptr(), nbytes, arrs["swin"], arrs["lang"], arrs["main"],
                                        main_grads.data_ptr(), swin_grads.data_ptr() if swin_grads is not None else None,
                                        lang_b.data_ptr(), enc_b.data_ptr(), dec_b.data_ptr(), carr, iarr, L.stream_ptr())
        L.check(rc, "klab_engine_bind")
        self._keep = (ws, lang_b, enc_b, dec_b, coords, index, tensors, main_grads, swin_grads)
        self.shape = (B, Ls, Lt)
        self.workspace = ws
        lp = self._lib.klab_engine_loss_ptr(self._h)
        off = lp - ws.data_ptr()
        self.loss_view = ws[off:off + 4].view(torch.float32)
        ep = self._lib.klab_engine_err_ptr(self._h)
        eo = ep - ws.data_ptr()
        self.err_view = ws[eo:eo + 4].view(torch.int32)
        ro = self._lib.klab_engine_rng_ptr(self._h) - ws.data_ptr()
        self.rng_view = ws[ro:ro + 12].view(torch.int32)  # {step seed, base seed, forwards since seeding}

    @property
    def n_img(self):
        s = self.swin_cfg
        Rl = (s.image_size // s.patch_size) >> (len(s.depths) - 1)
        return Rl * Rl

    def forward(self, pixels, src_ids, tgt_ids, training, seed, want_grad=True):
        L.check(self._lib.klab_engine_forward(self._h, pixels.data_ptr(), src_ids.data_ptr(), tgt_ids.data_ptr(), int(training),
                                              int(seed) & 0xFFFFFFFF, int(want_grad), L.stream_ptr()), "klab_engine_forward")

    def set_loss_out(self, out):
        """the next forward writes its loss into `out` (a 1-element fp32 device tensor) instead of loss_view; False under graph replay"""
        rc = self._lib.klab_engine_set_loss_out(self._h, out.data_ptr() if out is not None else None)
        if rc < 0:
            L.check(rc, "klab_engine_set_loss_out")
        return rc == 1

    def decode_step(self, t, prev_tokens):
        """decoder over position t (>= 1) only, self-attention K/V from the binding's cache; logits -> buffer("logits_step")"""
        L.check(self._lib.klab_engine_decode_step(self._h, int(t), prev_tokens.data_ptr(), L.stream_ptr()), "klab_engine_decode_step")

    def backward(self, segment, dloss=None):
        L.check(self._lib.klab_engine_backward(self._h, segment, dloss.data_ptr() if dloss is not None else None, L.stream_ptr()),
                "klab_engine_backward")

    def adam_step(self, m, v, lr, beta1, beta2, eps, weight_decay, bias_corr1, bias_corr2, segment=None):
        """torch.optim.Adam's update for every parameter of the trainable T5 in one kernel (+ refreshed bf16 copies);
        segment = 0 / 1: only the tensors of that backward segment (both = one full step)."""
        if segment is not None:
            L.check(self._lib.klab_engine_adam_step_segment(self._h, int(segment), m.data_ptr(), v.data_ptr(), lr, beta1, beta2, eps,
                                                            weight_decay, bias_corr1, bias_corr2, L.stream_ptr()), "klab_engine_adam_step_segment")
            return
        L.check(self._lib.klab_engine_adam_step(self._h, m.data_ptr(), v.data_ptr(), lr, beta1, beta2, eps, weight_decay, bias_corr1,
                                                bias_corr2, L.stream_ptr()), "klab_engine_adam_step")

    def set_bucket_events(self, on=True):
        """record the per-layer bucket events during backward (a data-parallel reducer is attached)"""
        L.check(self._lib.klab_engine_set_bucket_events(self._h, int(on)), "klab_engine_set_bucket_events")

    def bucket_wait(self, segment, i, stream):
        """`stream` (torch.cuda.Stream) waits until layer bucket i of the last backward of `segment` is final; False when the
        engine has no per-layer events for it (graph replay): the caller then waits for the whole segment."""
        rc = self._lib.klab_engine_bucket_wait(self._h, int(segment), int(i), stream.cuda_stream)
        if rc == L.ERR_UNSUPPORTED:
            return False
        L.check(rc, "klab_engine_bucket_wait")
        return True

    def get_rng(self):
        """(base seed, forwards since seeding) of the device-side dropout RNG; synchronises the current stream"""
        b, n = C.c_uint32(), C.c_uint32()
        L.check(self._lib.klab_engine_get_rng(self._h, C.byref(b), C.byref(n), L.stream_ptr()), "klab_engine_get_rng")
        return int(b.value), int(n.value)

    def set_rng(self, base, counter):
        L.check(self._lib.klab_engine_set_rng(self._h, int(base) & 0xFFFFFFFF, int(counter) & 0xFFFFFFFF, L.stream_ptr()), "klab_engine_set_rng")

    def set_graph(self, on=True):
        """replay the launch sequences as hipGraphs (inputs are staged, so any input tensors may be passed)."""
        L.check(self._lib.klab_engine_set_graph(self._h, int(on)), "klab_engine_set_graph")

    def probe_enable(self, on=True):
        L.check(self._lib.klab_engine_probe_enable(self._h, int(on)), "klab_engine_probe_enable")

    def probe_read(self, channel=0):
        """(launches, total_ms, total_flops) of a probe channel since probe_enable() -- 0: LM-head logits GEMM, 1: grouped
        weight-gradient launches; call after a synchronize."""
        n, ms, fl = C.c_int(), C.c_float(), C.c_double()
        L.check(self._lib.klab_engine_probe_read(self._h, int(channel), C.byref(n), C.byref(ms), C.byref(fl)), "klab_engine_probe_read")
        return n.value, ms.value, fl.value

    def buffer(self, name):
        rows, cols, dt = C.c_long(), C.c_long(), C.c_int()
        p = self._lib.klab_engine_buffer(self._h, name.encode(), C.byref(rows), C.byref(cols), C.byref(dt))
        if not p:
            raise KeyError(name)
        tdt = torch.float32 if dt.value == L.F32 else torch.bfloat16
        es = 4 if dt.value == L.F32 else 2
        off = p - self.workspace.data_ptr()
        return self.workspace[off:off + rows.value * cols.value * es].view(tdt).view(rows.value, cols.value)
