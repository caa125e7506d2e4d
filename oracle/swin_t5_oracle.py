"""CPU oracle for the Swin-V2 -> T5 caption-training hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch (CPU, fp32 or fp64) restatement of the arithmetic the reference
executes through ``ref/models/model.py:19-26``; that arithmetic lives in the un-vendored,
un-pinned third-party package ``transformers`` (``ref/requirements.txt:3``; de-facto pin: 5.15.0 as
installed in the build container).  Every function cites the file:line it follows, using the
prefixes of SURVEY.md (``ref/`` = /root/reference, ``HF/swinv2`` = transformers/models/swinv2/
modeling_swinv2.py, ``HF/t5`` = transformers/models/t5/modeling_t5.py).

PARITY PINNING: the reference has no tests or golden vectors of its own, so this
oracle is pinned against outputs of the reference itself: ``tests/golden/*.npz`` were produced by
``tests/golden/make_goldens.py`` importing ``/root/reference/models/model.py`` in the build
container; ``tests/test_oracle_golden.py`` checks loss, activations and every gradient.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module; the product (``klab_multimodalmodel_amd``) never does.  It imports neither ``transformers``
nor anything under ``/root/reference``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# ----------------------------------------------------------------------------------------------
# configs (field names follow HF/swinv2cfg:56-73 and HF/t5cfg:44-62)
# ----------------------------------------------------------------------------------------------
@dataclass
class SwinCfg:
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
        return cls(**{k: d[k] for k in keys if k in d})


@dataclass
class T5Cfg:
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
    decoder_start_token_id: int = 0
    pad_token_id: int = 0
    scale_decoder_outputs: bool = True  # HF/t5cfg:82-83 (tied embeddings => scale by d^-0.5)

    def __post_init__(self):
        if self.num_decoder_layers is None:
            self.num_decoder_layers = self.num_layers

    @classmethod
    def from_dict(cls, d):
        keys = cls.__dataclass_fields__.keys()
        c = cls(**{k: d[k] for k in keys if k in d and d[k] is not None})
        if d.get("tie_word_embeddings", None) is False:
            c.scale_decoder_outputs = False
        return c


# ----------------------------------------------------------------------------------------------
# Swin-V2
# ----------------------------------------------------------------------------------------------
def swin_patch_embed(sd: SD, cfg: SwinCfg, pixel_values: Tensor) -> Tuple[Tensor, int]:
    """conv4x4 stride 4 + flatten + LayerNorm (HF/swinv2:281,293-302,242)."""
    w = sd["embeddings.patch_embeddings.projection.weight"]
    b = sd["embeddings.patch_embeddings.projection.bias"]
    x = F.conv2d(pixel_values, w, b, stride=cfg.patch_size)
    R = x.shape[-1]
    x = x.flatten(2).transpose(1, 2)
    x = F.layer_norm(x, (x.shape[-1],), sd["embeddings.norm.weight"], sd["embeddings.norm.bias"],
                     cfg.layer_norm_eps)
    return x, R


def swin_window_shift(R: int, window: int, block_index: int) -> Tuple[int, int]:
    """window/shift clamp (HF/swinv2:615-618,737)."""
    w = min(R, window)
    shift = 0 if (block_index % 2 == 0) else window // 2
    if R <= w:
        shift = 0
    return w, shift


def swin_coords_table_and_index(w: int, pretrained_w: int, dtype) -> Tuple[Tensor, Tensor]:
    """log-spaced relative coords table + pairwise index (HF/swinv2:457-492)."""
    rc = torch.arange(-(w - 1), w, dtype=torch.int64).float()
    table = torch.stack(torch.meshgrid([rc, rc], indexing="ij")).permute(1, 2, 0).contiguous().unsqueeze(0)
    if pretrained_w > 0:
        table = table / (pretrained_w - 1)
    elif w > 1:
        table = table / (w - 1)
    table = table * 8
    table = torch.sign(table) * torch.log2(torch.abs(table) + 1.0) / math.log2(8)
    table = table.to(dtype)
    c = torch.arange(w)
    coords = torch.stack(torch.meshgrid([c, c], indexing="ij")).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += w - 1
    rel[:, :, 1] += w - 1
    rel[:, :, 0] *= 2 * w - 1
    return table, rel.sum(-1)


def swin_cpb_bias(sd: SD, prefix: str, w: int, pretrained_w: int, heads: int, dtype) -> Tensor:
    """16*sigmoid(MLP(coords))[index] -> [h, n, n] (HF/swinv2:376-378,418-428)."""
    table, index = swin_coords_table_and_index(w, pretrained_w, dtype)
    h1 = F.relu(F.linear(table, sd[prefix + "continuous_position_bias_mlp.0.weight"],
                         sd[prefix + "continuous_position_bias_mlp.0.bias"]))
    t = F.linear(h1, sd[prefix + "continuous_position_bias_mlp.2.weight"]).view(-1, heads)
    n = w * w
    bias = t[index.view(-1)].view(n, n, -1).permute(2, 0, 1).contiguous()
    return 16 * torch.sigmoid(bias)


def swin_shift_mask(R: int, w: int, shift: int, dtype) -> Optional[Tensor]:
    """9-region cyclic-shift mask, 0/-100 (HF/swinv2:620-643)."""
    if shift <= 0:
        return None
    idx = torch.arange(R)
    reg = (idx >= R - w).long() + (idx >= R - shift).long()
    img = (reg[None, :, None, None] * 3 + reg[None, None, :, None]).to(dtype)
    mw = _window_partition(img, w).view(-1, w * w)
    m = mw.unsqueeze(1) - mw.unsqueeze(2)
    return m.masked_fill(m != 0, -100.0).masked_fill(m == 0, 0.0)


def _window_partition(x: Tensor, w: int) -> Tensor:
    B, H, W, C = x.shape  # HF/swinv2:146-155
    x = x.view(B, H // w, w, W // w, w, C)
    return x.transpose(2, 3).contiguous().view(-1, w, w, C)


def _window_reverse(win: Tensor, w: int, H: int, W: int) -> Tensor:
    C = win.shape[-1]  # HF/swinv2:158-166
    x = win.view(-1, H // w, W // w, w, w, C)
    return x.transpose(2, 3).contiguous().view(-1, H, W, C)


def swin_window_attention(sd: SD, prefix: str, xw: Tensor, heads: int, w: int, pretrained_w: int,
                          mask: Optional[Tensor]) -> Tensor:
    """Swinv2SelfAttention.forward on partitioned windows [nW*B, n, C] (HF/swinv2:389-455)."""
    Bw, n, C = xw.shape
    hd = C // heads
    q = F.linear(xw, sd[prefix + "query.weight"], sd.get(prefix + "query.bias")).view(Bw, n, heads, hd).transpose(1, 2)
    k = F.linear(xw, sd[prefix + "key.weight"]).view(Bw, n, heads, hd).transpose(1, 2)
    v = F.linear(xw, sd[prefix + "value.weight"], sd.get(prefix + "value.bias")).view(Bw, n, heads, hd).transpose(1, 2)
    s = F.normalize(q, dim=-1) @ F.normalize(k, dim=-1).transpose(-2, -1)  # eps 1e-12 (:413-415)
    scale = torch.clamp(sd[prefix + "logit_scale"], max=math.log(1.0 / 0.01)).exp()  # (:416)
    s = s * scale
    s = s + swin_cpb_bias(sd, prefix, w, pretrained_w, heads, xw.dtype).unsqueeze(0)
    if mask is not None:
        nW = mask.shape[0]
        s = s.view(Bw // nW, nW, heads, n, n) + mask.unsqueeze(1).unsqueeze(0)
        s = s + mask.unsqueeze(1).unsqueeze(0)  # the pinned transformers adds it twice (:433-436)
        s = s.view(-1, heads, n, n)
    p = F.softmax(s, dim=-1)
    ctx = (p @ v).permute(0, 2, 1, 3).contiguous().view(Bw, n, C)
    return ctx


def swin_block(sd: SD, prefix: str, x: Tensor, R: int, heads: int, w: int, shift: int, pretrained_w: int,
               eps: float) -> Tensor:
    """Swinv2Layer.forward, eval mode (HF/swinv2:652-705); res-post-norm."""
    B, T, C = x.shape
    shortcut = x
    h = x.view(B, R, R, C)
    # pad to a multiple of the window on the right / bottom with ZERO rows (maybe_pad, HF/swinv2:645-650): the padded tokens
    # go through the attention like any other (k = 0, v = value bias) and are cropped afterwards (HF/swinv2:688-690)
    pad = (w - R % w) % w
    Rp = R + pad
    if pad:
        h = F.pad(h, (0, 0, 0, pad, 0, pad))
    if shift > 0:
        h = torch.roll(h, shifts=(-shift, -shift), dims=(1, 2))
    xw = _window_partition(h, w).view(-1, w * w, C)
    mask = swin_shift_mask(Rp, w, shift, x.dtype)  # regions of the PADDED grid (HF/swinv2:675)
    ctx = swin_window_attention(sd, prefix + "attention.self.", xw, heads, w, pretrained_w, mask)
    ao = F.linear(ctx, sd[prefix + "attention.output.dense.weight"], sd[prefix + "attention.output.dense.bias"])
    h = _window_reverse(ao.view(-1, w, w, C), w, Rp, Rp)
    if shift > 0:
        h = torch.roll(h, shifts=(shift, shift), dims=(1, 2))
    if pad:
        h = h[:, :R, :R, :].contiguous()
    h = h.view(B, T, C)
    h = F.layer_norm(h, (C,), sd[prefix + "layernorm_before.weight"], sd[prefix + "layernorm_before.bias"], eps)
    x = shortcut + h
    m = F.gelu(F.linear(x, sd[prefix + "intermediate.dense.weight"], sd[prefix + "intermediate.dense.bias"]))
    m = F.linear(m, sd[prefix + "output.dense.weight"], sd[prefix + "output.dense.bias"])
    m = F.layer_norm(m, (C,), sd[prefix + "layernorm_after.weight"], sd[prefix + "layernorm_after.bias"], eps)
    return x + m


def swin_patch_merge(sd: SD, prefix: str, x: Tensor, R: int, eps: float) -> Tensor:
    """2x2 gather (0,0),(1,0),(0,1),(1,1) -> Linear(4C,2C,no bias) -> LN(2C) (HF/swinv2:333-356)."""
    B, T, C = x.shape
    assert R % 2 == 0
    f = x.view(B, R, R, C)
    f = torch.cat([f[:, 0::2, 0::2], f[:, 1::2, 0::2], f[:, 0::2, 1::2], f[:, 1::2, 1::2]], -1)
    f = f.view(B, -1, 4 * C)
    f = F.linear(f, sd[prefix + "reduction.weight"])
    return F.layer_norm(f, (2 * C,), sd[prefix + "norm.weight"], sd[prefix + "norm.bias"], eps)


def swin_forward(sd: SD, cfg: SwinCfg, pixel_values: Tensor) -> Tensor:
    """Swinv2Model.forward(...).last_hidden_state, eval mode (HF/swinv2:917-958); pooler skipped
    because ref/models/model.py:22 discards it."""
    x, R = swin_patch_embed(sd, cfg, pixel_values)
    nstage = len(cfg.depths)
    for s in range(nstage):
        for b in range(cfg.depths[s]):
            w, shift = swin_window_shift(R, cfg.window_size, b)
            x = swin_block(sd, f"encoder.layers.{s}.blocks.{b}.", x, R, cfg.num_heads[s], w, shift,
                           cfg.pretrained_window_sizes[s] if s < len(cfg.pretrained_window_sizes) else 0,
                           cfg.layer_norm_eps)
        if s < nstage - 1:
            x = swin_patch_merge(sd, f"encoder.layers.{s}.downsample.", x, R, cfg.layer_norm_eps)
            R //= 2
    C = x.shape[-1]
    return F.layer_norm(x, (C,), sd["layernorm.weight"], sd["layernorm.bias"], cfg.layer_norm_eps)


# ----------------------------------------------------------------------------------------------
# T5
# ----------------------------------------------------------------------------------------------
def t5_rmsnorm(x: Tensor, w: Tensor, eps: float) -> Tensor:
    """T5LayerNorm (HF/t5:59-72): no mean subtraction, no bias, fp32 accumulation."""
    var = x.to(torch.float32 if x.dtype != torch.float64 else torch.float64).pow(2).mean(-1, keepdim=True)
    return w * (x * torch.rsqrt(var + eps)).to(x.dtype)


def t5_relative_position_bucket(rel: Tensor, bidirectional: bool, num_buckets: int, max_distance: int) -> Tensor:
    """HF/t5:216-262 (float-log truncation kept exactly: fp32 log, then .to(long))."""
    out = torch.zeros_like(rel)
    if bidirectional:
        num_buckets //= 2
        out = out + (rel > 0).to(torch.long) * num_buckets
        rel = torch.abs(rel)
    else:
        rel = -torch.min(rel, torch.zeros_like(rel))
    max_exact = num_buckets // 2
    is_small = rel < max_exact
    large = max_exact + (torch.log(rel.float() / max_exact) / math.log(max_distance / max_exact)
                         * (num_buckets - max_exact)).to(torch.long)
    large = torch.min(large, torch.full_like(large, num_buckets - 1))
    return out + torch.where(is_small, rel, large)


def t5_position_bias(table: Tensor, Lq: int, Lk: int, bidirectional: bool, cfg: T5Cfg) -> Tensor:
    """compute_bias -> [1, h, Lq, Lk] (HF/t5:264-279)."""
    ctx = torch.arange(Lq, dtype=torch.long)[:, None]
    mem = torch.arange(Lk, dtype=torch.long)[None, :]
    b = t5_relative_position_bucket(mem - ctx, bidirectional, cfg.relative_attention_num_buckets,
                                    cfg.relative_attention_max_distance)
    return table[b].permute(2, 0, 1).unsqueeze(0)


def t5_attention(sd: SD, prefix: str, x: Tensor, kv: Tensor, bias: Optional[Tensor], causal: bool, cfg: T5Cfg,
                 p_drop: float) -> Tensor:
    """T5Attention.forward + eager_attention_forward (HF/t5:281-369,144-173): unscaled scores,
    additive position bias, causal mask in the decoder (HF/t5:697-711), prob dropout."""
    B, Lq, _ = x.shape
    Lk = kv.shape[1]
    h, dk = cfg.num_heads, cfg.d_kv
    q = F.linear(x, sd[prefix + "q.weight"]).view(B, Lq, h, dk).transpose(1, 2)
    k = F.linear(kv, sd[prefix + "k.weight"]).view(B, Lk, h, dk).transpose(1, 2)
    v = F.linear(kv, sd[prefix + "v.weight"]).view(B, Lk, h, dk).transpose(1, 2)
    s = q @ k.transpose(2, 3)  # scaling = 1.0 (HF/t5:196-197)
    if bias is not None:
        s = s + bias
    if causal:
        keep = torch.tril(torch.ones(Lq, Lk, dtype=torch.bool))
        s = s.masked_fill(~keep, float("-inf"))
    p = F.softmax(s, dim=-1)
    if p_drop > 0:
        p = F.dropout(p, p_drop, True)
    o = (p @ v).transpose(1, 2).reshape(B, Lq, h * dk)
    return F.linear(o, sd[prefix + "o.weight"])


def _drop(x: Tensor, p: float) -> Tensor:
    return F.dropout(x, p, True) if p > 0 else x


def t5_stack(sd: SD, prefix: str, cfg: T5Cfg, x: Tensor, is_decoder: bool, enc_out: Optional[Tensor],
             p_drop: float) -> Tensor:
    """T5Stack.forward from inputs_embeds (HF/t5:663-750); no padding mask is ever supplied by the
    reference (ref/models/model.py:21,26)."""
    nl = cfg.num_decoder_layers if is_decoder else cfg.num_layers
    eps = cfg.layer_norm_epsilon
    L = x.shape[1]
    x = _drop(x, p_drop)
    bias = t5_position_bias(sd[prefix + "block.0.layer.0.SelfAttention.relative_attention_bias.weight"], L, L,
                            not is_decoder, cfg).to(x.dtype)
    for i in range(nl):
        bp = f"{prefix}block.{i}."
        n = t5_rmsnorm(x, sd[bp + "layer.0.layer_norm.weight"], eps)
        x = x + _drop(t5_attention(sd, bp + "layer.0.SelfAttention.", n, n, bias, is_decoder, cfg, p_drop), p_drop)
        li = 1
        if is_decoder:
            n = t5_rmsnorm(x, sd[bp + "layer.1.layer_norm.weight"], eps)
            x = x + _drop(t5_attention(sd, bp + "layer.1.EncDecAttention.", n, enc_out, None, False, cfg, p_drop), p_drop)
            li = 2
        n = t5_rmsnorm(x, sd[bp + f"layer.{li}.layer_norm.weight"], eps)
        hmid = _drop(F.relu(F.linear(n, sd[bp + f"layer.{li}.DenseReluDense.wi.weight"])), p_drop)
        x = x + _drop(F.linear(hmid, sd[bp + f"layer.{li}.DenseReluDense.wo.weight"]), p_drop)
    x = t5_rmsnorm(x, sd[prefix + "final_layer_norm.weight"], eps)
    return _drop(x, p_drop)


def t5_shift_right(labels: Tensor, cfg: T5Cfg) -> Tensor:
    """HF/t5:618-637."""
    out = labels.new_zeros(labels.shape)
    out[..., 1:] = labels[..., :-1].clone()
    out[..., 0] = cfg.decoder_start_token_id
    return out.masked_fill(out == -100, cfg.pad_token_id)


def t5_encoder_model(sd: SD, cfg: T5Cfg, input_ids: Tensor) -> Tensor:
    """T5EncoderModel.forward(input_ids).last_hidden_state in eval mode (HF/t5:1128-1135)."""
    emb = sd["shared.weight"] if "shared.weight" in sd else sd["encoder.embed_tokens.weight"]
    return t5_stack(sd, "encoder.", cfg, F.embedding(input_ids, emb), False, None, 0.0)


def t5_seq2seq(sd: SD, cfg: T5Cfg, inputs_embeds: Tensor, labels: Tensor, training: bool):
    """T5ForConditionalGeneration.forward(inputs_embeds=..., labels=...) (HF/t5:1009-1054).
    Returns (loss, encoder_out, decoder_out, logits)."""
    p = cfg.dropout_rate if training else 0.0
    enc = t5_stack(sd, "encoder.", cfg, inputs_embeds, False, None, p)
    dec_in = F.embedding(t5_shift_right(labels, cfg), sd["shared.weight"])
    dec = t5_stack(sd, "decoder.", cfg, dec_in, True, enc, p)
    seq = dec * (cfg.d_model ** -0.5) if cfg.scale_decoder_outputs else dec
    logits = F.linear(seq, sd["shared.weight"])  # tied lm_head (HF/t5:902-906)
    loss = F.cross_entropy(logits.view(-1, logits.shape[-1]), labels.view(-1), ignore_index=-100)
    return loss, enc, dec, logits


# ----------------------------------------------------------------------------------------------
# MyModel.forward (ref/models/model.py:19-26)
# ----------------------------------------------------------------------------------------------
def mymodel_forward(swin_sd: SD, lang_sd: SD, main_sd: SD, swin_cfg: SwinCfg, lang_cfg: T5Cfg, main_cfg: T5Cfg,
                    pixel_values: Tensor, src_ids: Tensor, tgt_ids: Tensor, training: bool = False,
                    image_model_train: bool = False, return_parts: bool = False):
    with torch.no_grad():  # ref/models/model.py:20-21
        lang = t5_encoder_model(lang_sd, lang_cfg, src_ids)
    if image_model_train:
        img = swin_forward(swin_sd, swin_cfg, pixel_values)  # ref/models/model.py:22 (Swin is eval: SURVEY §0.4)
    else:
        with torch.no_grad():
            img = swin_forward(swin_sd, swin_cfg, pixel_values)
    cat = torch.cat((img, lang), dim=1)  # ref/models/model.py:23
    loss, enc, dec, logits = t5_seq2seq(main_sd, main_cfg, cat, tgt_ids, training)  # :26
    if return_parts:
        return loss, dict(image_embeddings=img, language_embeddings=lang, encoder_out=enc, decoder_out=dec,
                          logits=logits)
    return loss


# ----------------------------------------------------------------------------------------------
# helpers shared by tests / bench cpu_baseline
# ----------------------------------------------------------------------------------------------
def expand_tied(sd: SD) -> SD:
    """fixtures store `shared.weight` once; nothing else is needed by this oracle."""
    return sd


def hf_like_init_swin(cfg: SwinCfg, gen: torch.Generator, dtype=torch.float32) -> SD:
    """Random Swin-V2 weights with the shapes of HF's state dict and the statistics of
    HF `_init_weights` (HF/swinv2:873-886: trunc-normal-ish N(0, .02) linears, unit LN, log(10) scale)."""
    sd: SD = {}
    C = cfg.embed_dim

    def lin(name, o, i, bias=True):
        sd[name + ".weight"] = torch.randn(o, i, generator=gen, dtype=dtype) * 0.02
        if bias:
            sd[name + ".bias"] = torch.zeros(o, dtype=dtype)

    def ln(name, c):
        sd[name + ".weight"] = torch.ones(c, dtype=dtype)
        sd[name + ".bias"] = torch.zeros(c, dtype=dtype)

    sd["embeddings.patch_embeddings.projection.weight"] = torch.randn(C, cfg.num_channels, cfg.patch_size, cfg.patch_size,
                                                                      generator=gen, dtype=dtype) * 0.02
    sd["embeddings.patch_embeddings.projection.bias"] = torch.zeros(C, dtype=dtype)
    ln("embeddings.norm", C)
    ns = len(cfg.depths)
    for s in range(ns):
        Cs = C * 2 ** s
        for b in range(cfg.depths[s]):
            p = f"encoder.layers.{s}.blocks.{b}."
            sd[p + "attention.self.logit_scale"] = torch.log(10 * torch.ones(cfg.num_heads[s], 1, 1, dtype=dtype))
            lin(p + "attention.self.continuous_position_bias_mlp.0", 512, 2)
            lin(p + "attention.self.continuous_position_bias_mlp.2", cfg.num_heads[s], 512, bias=False)
            lin(p + "attention.self.query", Cs, Cs, cfg.qkv_bias)
            lin(p + "attention.self.key", Cs, Cs, False)
            lin(p + "attention.self.value", Cs, Cs, cfg.qkv_bias)
            lin(p + "attention.output.dense", Cs, Cs)
            ln(p + "layernorm_before", Cs)
            lin(p + "intermediate.dense", int(cfg.mlp_ratio * Cs), Cs)
            lin(p + "output.dense", Cs, int(cfg.mlp_ratio * Cs))
            ln(p + "layernorm_after", Cs)
        if s < ns - 1:
            lin(f"encoder.layers.{s}.downsample.reduction", 2 * Cs, 4 * Cs, False)
            ln(f"encoder.layers.{s}.downsample.norm", 2 * Cs)
    ln("layernorm", C * 2 ** (ns - 1))
    return sd


def hf_like_init_t5(cfg: T5Cfg, gen: torch.Generator, encoder_only: bool = False, dtype=torch.float32) -> SD:
    """Random T5 weights with HF's state-dict shapes and `_init_weights` statistics (HF/t5:562-616)."""
    sd: SD = {}
    d, dk, h, ff = cfg.d_model, cfg.d_kv, cfg.num_heads, cfg.d_ff
    inner = h * dk
    sd["shared.weight"] = torch.randn(cfg.vocab_size, d, generator=gen, dtype=dtype)
    stacks = [("encoder.", cfg.num_layers, False)]
    if not encoder_only:
        stacks.append(("decoder.", cfg.num_decoder_layers, True))
    for prefix, nl, dec in stacks:
        for i in range(nl):
            bp = f"{prefix}block.{i}."

            def attn(ap):
                sd[ap + "q.weight"] = torch.randn(inner, d, generator=gen, dtype=dtype) * (d * dk) ** -0.5
                sd[ap + "k.weight"] = torch.randn(inner, d, generator=gen, dtype=dtype) * d ** -0.5
                sd[ap + "v.weight"] = torch.randn(inner, d, generator=gen, dtype=dtype) * d ** -0.5
                sd[ap + "o.weight"] = torch.randn(d, inner, generator=gen, dtype=dtype) * inner ** -0.5

            attn(bp + "layer.0.SelfAttention.")
            if i == 0:
                sd[bp + "layer.0.SelfAttention.relative_attention_bias.weight"] = \
                    torch.randn(cfg.relative_attention_num_buckets, h, generator=gen, dtype=dtype) * d ** -0.5
            sd[bp + "layer.0.layer_norm.weight"] = torch.ones(d, dtype=dtype)
            li = 1
            if dec:
                attn(bp + "layer.1.EncDecAttention.")
                sd[bp + "layer.1.layer_norm.weight"] = torch.ones(d, dtype=dtype)
                li = 2
            sd[bp + f"layer.{li}.DenseReluDense.wi.weight"] = torch.randn(ff, d, generator=gen, dtype=dtype) * d ** -0.5
            sd[bp + f"layer.{li}.DenseReluDense.wo.weight"] = torch.randn(d, ff, generator=gen, dtype=dtype) * ff ** -0.5
            sd[bp + f"layer.{li}.layer_norm.weight"] = torch.ones(d, dtype=dtype)
        sd[prefix + "final_layer_norm.weight"] = torch.ones(d, dtype=dtype)
    return sd
