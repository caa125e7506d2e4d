#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE itself (build container only).

This script imports ``/root/reference/models/model.py`` (ref/models/model.py:8-42) against
config-constructed, seeded random-init HuggingFace weights saved to a scratch directory
(SURVEY.md §8c), runs ``MyModel.forward`` + ``loss.backward()`` in eval mode (dropout off, the
reference's own parity-capable mode) and stores inputs / weights / outputs / grads as ``.npz``
fixtures next to this file.  Only the *outputs* of this script travel to the GPU box; the
reference's sources never do.  It also records the argparse defaults (ref/modules/config.py:6-22).  The span-mask loader's
vectors (ref/modules/loader.py:56-77) are produced by ``make_spanmask_goldens.py`` next to this file, which runs the
reference's own ``RedCapsDatasetLoader.__getitem__`` (45 ``(seed, caption) -> (src, tgt)`` pairs in ``spanmask.json``).

Run:  python tests/golden/make_goldens.py      (needs /root/reference + transformers; CPU only)
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

CONFIGS = {
    # 4 stages, window 4 on 64x64: windows/shifts (4,0),(4,2)/(4,0),(4,2)/(4,0),(4,0)/(2,0),(2,0)
    # => covers shift mask, R<=w clamp, three merges; T5 with inner_dim != d_model.
    "tiny_a": dict(
        swin=dict(image_size=64, patch_size=4, embed_dim=16, depths=[2, 2, 2, 2], num_heads=[1, 2, 4, 8],
                  window_size=4),
        t5=dict(vocab_size=384, d_model=128, d_kv=16, num_heads=4, d_ff=256, num_layers=2,
                num_decoder_layers=2),
        B=3, Ls=5, Lt=7, pad_tail=True, seed=11,
    ),
    # 2 stages, the real window (7 => n=49, not a power of two) and real head dim (32);
    # longer sequences exercise the log-spaced relative-position buckets; asymmetric decoder depth.
    "tiny_b": dict(
        swin=dict(image_size=56, patch_size=4, embed_dim=32, depths=[2, 2], num_heads=[1, 2],
                  window_size=7),
        t5=dict(vocab_size=512, d_model=64, d_kv=32, num_heads=2, d_ff=128, num_layers=2,
                num_decoder_layers=3),
        B=2, Ls=20, Lt=40, pad_tail=True, seed=23,
    ),
    # window PADDING (HF/swinv2:645-650, 688-690): 80 px / patch 4 -> 20x20 tokens with window 6 -> padded to 24x24 (16 windows,
    # shift 3), then 10x10 -> padded to 12x12; padded tokens are zero rows that still act as keys (k = 0, v = value bias).
    "tiny_c": dict(
        swin=dict(image_size=80, patch_size=4, embed_dim=32, depths=[2, 2], num_heads=[1, 2], window_size=6),
        t5=dict(vocab_size=256, d_model=64, d_kv=32, num_heads=2, d_ff=128, num_layers=2, num_decoder_layers=2),
        B=2, Ls=6, Lt=9, pad_tail=True, seed=31,
    ),
    # padding together with windows of more than 64 tokens: 96 px -> 24x24 tokens, window 10 (n = 100) -> padded to 30x30 (shift
    # 5), then 12x12 -> padded to 20x20; pretrained_window_sizes set (log-spaced coordinate table normalised by them)
    "tiny_d": dict(
        swin=dict(image_size=96, patch_size=4, embed_dim=32, depths=[2, 2], num_heads=[1, 2], window_size=10,
                  pretrained_window_sizes=[5, 5]),
        t5=dict(vocab_size=256, d_model=64, d_kv=32, num_heads=2, d_ff=128, num_layers=2, num_decoder_layers=2),
        B=2, Ls=6, Lt=9, pad_tail=True, seed=41,
    ),
}


def build_dirs(cfg, root):
    from transformers import Swinv2Config, Swinv2Model, T5Config, T5EncoderModel, T5ForConditionalGeneration
    torch.manual_seed(cfg["seed"])
    swin = Swinv2Model(Swinv2Config(**cfg["swin"]))
    t5cfg = T5Config(**cfg["t5"], decoder_start_token_id=0)
    lang = T5EncoderModel(t5cfg)
    main = T5ForConditionalGeneration(t5cfg)
    # HF init leaves logit_scale / norm weights at constants; perturb every tensor a little so
    # that parity cannot pass by accident on identity-like parameters.
    g = torch.Generator().manual_seed(cfg["seed"] + 1)
    with torch.no_grad():
        for m in (swin, lang, main):
            for n, p in m.named_parameters():
                p.add_(torch.randn(p.shape, generator=g) * 0.05 * (p.abs().mean() + 0.02))
    dirs = {}
    for name, m in (("swin", swin), ("lang", lang), ("main", main)):
        d = os.path.join(root, name)
        m.save_pretrained(d)
        dirs[name] = d
    return dirs


def make_inputs(cfg):
    g = torch.Generator().manual_seed(cfg["seed"] + 1234)
    B, Ls, Lt = cfg["B"], cfg["Ls"], cfg["Lt"]
    V = cfg["t5"]["vocab_size"]
    H = cfg["swin"]["image_size"]
    pix = torch.randn(B, 3, H, H, generator=g)
    src = torch.randint(2, V, (B, Ls), generator=g)
    tgt = torch.randint(2, V, (B, Lt), generator=g)
    src[:, -1] = 1
    tgt[:, -1] = 1
    if cfg["pad_tail"]:  # trailing pads (id 0) on the last row: "pads are scored" 
        tgt[-1, -max(2, Lt // 4):] = 0
        tgt[-1, -max(2, Lt // 4) - 1] = 1
        src[-1, -2:] = 0
        src[-1, -3] = 1
    return pix, src, tgt


def run_reference(cfg, dirs, train_swin):
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    from models.model import MyModel  # the reference's own class
    args = types.SimpleNamespace(result_dir="/tmp", language_model_name=dirs["lang"],
                                 image_model_name=dirs["swin"], image_model_train=train_swin,
                                 transformer_model_name=dirs["main"])
    model = MyModel(args)
    model.transformer.eval()  # dropout off; Swin/lang are eval already
    pix, src, tgt = make_inputs(cfg)
    caps = {}

    def cap(name):
        def hook(_m, _i, out):
            caps[name] = out.last_hidden_state.detach().clone() if hasattr(out, "last_hidden_state") else out
        return hook

    model.image_model.register_forward_hook(cap("image_embeddings"))
    model.language_model.register_forward_hook(cap("language_embeddings"))
    model.transformer.encoder.register_forward_hook(cap("encoder_out"))
    model.transformer.decoder.register_forward_hook(cap("decoder_out"))
    loss = model({"pixel_values": pix}, {"input_ids": src}, {"input_ids": tgt})
    loss.backward()
    out = dict(pixel_values=pix, src_ids=src, tgt_ids=tgt, loss=loss.detach())
    for k, v in caps.items():
        out["act." + k] = v
    for prefix, m in (("main.", model.transformer), ("swin.", model.image_model), ("lang.", model.language_model)):
        for n, p in m.state_dict().items():
            # the four tied T5 tables are one tensor (HF/t5:902-906): store `shared.weight` only
            if n in ("encoder.embed_tokens.weight", "decoder.embed_tokens.weight", "lm_head.weight"):
                assert torch.equal(p, m.state_dict()["shared.weight"])
                continue
            out["w." + prefix + n] = p.detach()
        for n, p in m.named_parameters():
            if p.grad is not None:
                out["g." + prefix + n] = p.grad.detach()
    return out


def config_goldens():
    # load the one file directly: the package __init__ pulls in torchvision/pycocotools, which
    # this image lacks (an ordinary ModuleNotFoundError), and nothing is stubbed to get around it.
    import importlib.util
    spec = importlib.util.spec_from_file_location("_ref_config", os.path.join(REF, "modules", "config.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    parse_arguments = mod.parse_arguments
    argv = sys.argv
    sys.argv = ["train.py"]
    try:
        ns = parse_arguments()
    finally:
        sys.argv = argv
    return vars(ns)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    torch.set_num_threads(4)
    for name, cfg in CONFIGS.items():
        if a.only and a.only != name:
            continue
        root = tempfile.mkdtemp(prefix="klab_gold_")
        dirs = build_dirs(cfg, root)
        out = run_reference(cfg, dirs, train_swin=True)
        out_frozen = run_reference(cfg, dirs, train_swin=False)
        assert torch.equal(out["loss"], out_frozen["loss"])
        assert not any(k.startswith("g.swin.") for k in out_frozen)
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"),
                            **{k: v.numpy() for k, v in out.items()})
        cfg_json = {k: v for k, v in cfg.items()}
        cfg_json["swin_config"] = json.load(open(os.path.join(dirs["swin"], "config.json")))
        cfg_json["t5_config"] = json.load(open(os.path.join(dirs["main"], "config.json")))
        json.dump(cfg_json, open(os.path.join(HERE, f"{name}.json"), "w"), indent=1, sort_keys=True)
        print(name, "loss", float(out["loss"]), "tensors", len(out))
        shutil.rmtree(root)
    if not a.only:
        json.dump(config_goldens(), open(os.path.join(HERE, "argparse_defaults.json"), "w"), indent=1, sort_keys=True)
        print("config goldens written")


if __name__ == "__main__":
    main()
