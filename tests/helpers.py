"""Shared fixtures loading for tests (goldens made by tests/golden/make_goldens.py)."""
import json
import os

import numpy as np
import torch

from oracle import swin_t5_oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name, dtype=torch.float32):
    z = np.load(os.path.join(GOLD, f"{name}.npz"))
    meta = json.load(open(os.path.join(GOLD, f"{name}.json")))
    sds = {"swin": {}, "lang": {}, "main": {}}
    grads = {"swin": {}, "main": {}}
    acts = {}
    for k in z.files:
        t = torch.from_numpy(z[k])
        if k.startswith("w."):
            _, m, n = k.split(".", 2)
            sds[m][n] = t.to(dtype) if t.is_floating_point() else t
        elif k.startswith("g."):
            _, m, n = k.split(".", 2)
            grads[m][n] = t
        elif k.startswith("act."):
            acts[k[4:]] = t
    swin_cfg = O.SwinCfg.from_dict(meta["swin_config"])
    t5_cfg = O.T5Cfg.from_dict(meta["t5_config"])
    inputs = dict(pixel_values=torch.from_numpy(z["pixel_values"]).to(dtype),
                  src_ids=torch.from_numpy(z["src_ids"]), tgt_ids=torch.from_numpy(z["tgt_ids"]))
    return dict(sds=sds, grads=grads, acts=acts, swin_cfg=swin_cfg, t5_cfg=t5_cfg, inputs=inputs,
                loss=float(z["loss"]), meta=meta)


def rel_l2(a, b):
    a = a.double().flatten()
    b = b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def cosine(a, b):
    a = a.double().flatten()
    b = b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-30))


def fp8_rows(x):
    """per-row OCP e4m3 quantisation exactly as klab_quant_fp8_rows does it: scale = amax / 448 (1 for a zero row), values / scale
    rounded to nearest-even e4m3.  Returns (dequantised float tensor, scales)."""
    import torch
    x = x.float()
    amax = x.abs().amax(dim=-1, keepdim=True)
    sc = torch.where(amax > 0, amax * (1.0 / 448.0), torch.ones_like(amax))
    q = (x * (1.0 / sc)).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float()
    return q, sc
