"""`from_pretrained`-shaped loading without `transformers`: config.json + model.safetensors /
pytorch_model.bin from a local directory (what ref/models/model.py:14-17 passes when the name is a
path), plus the public architecture hyper-parameters of the hub names the reference's CLI allows
(ref/modules/config.py:6-9) for random-init benchmarking when no weights are on disk."""
import json
import os

import torch

from .engine import SwinConfig, T5Config

# public hyper-parameters of the checkpoints the reference's argparse accepts
KNOWN_T5 = {
    "t5-small": dict(d_model=512, d_kv=64, d_ff=2048, num_layers=6, num_heads=8),
    "t5-base": dict(d_model=768, d_kv=64, d_ff=3072, num_layers=12, num_heads=12),
    "t5-large": dict(d_model=1024, d_kv=64, d_ff=4096, num_layers=24, num_heads=16),
    "t5-3b": dict(d_model=1024, d_kv=128, d_ff=16384, num_layers=24, num_heads=32),
    "t5-11b": dict(d_model=1024, d_kv=128, d_ff=65536, num_layers=24, num_heads=128),
}
KNOWN_SWIN = {
    "microsoft/swinv2-tiny-patch4-window8-256": dict(image_size=256, embed_dim=96, depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24), window_size=8),
    "microsoft/swinv2-small-patch4-window8-256": dict(image_size=256, embed_dim=96, depths=(2, 2, 18, 2), num_heads=(3, 6, 12, 24), window_size=8),
    "microsoft/swinv2-base-patch4-window8-256": dict(image_size=256, embed_dim=128, depths=(2, 2, 18, 2), num_heads=(4, 8, 16, 32), window_size=8),
    "microsoft/swinv2-base-patch4-window16-256": dict(image_size=256, embed_dim=128, depths=(2, 2, 18, 2), num_heads=(4, 8, 16, 32), window_size=16),
}


def _read_state_dict(path):
    st = os.path.join(path, "model.safetensors")
    if os.path.exists(st):
        from safetensors.torch import load_file
        return load_file(st)
    pb = os.path.join(path, "pytorch_model.bin")
    if os.path.exists(pb):
        return torch.load(pb, map_location="cpu", weights_only=True)
    raise OSError(f"Error no file named model.safetensors or pytorch_model.bin found in directory {path}.")


def resolve(name, kind):
    """-> (config, state_dict | None).  kind in {"swin", "t5"}."""
    if os.path.isdir(name):
        cfg = json.load(open(os.path.join(name, "config.json")))
        conf = SwinConfig.from_dict(cfg) if kind == "swin" else T5Config.from_dict(cfg)
        return conf, _read_state_dict(name)
    table = KNOWN_SWIN if kind == "swin" else KNOWN_T5
    if name in table and os.environ.get("KLAB_ALLOW_RANDOM_INIT", "0") == "1":
        conf = SwinConfig(**table[name]) if kind == "swin" else T5Config(**table[name])
        return conf, None
    raise OSError(f"{name} is not a local folder with config.json + weights, and this build has no hub access "
                  f"(offline). Pass a local directory, or set KLAB_ALLOW_RANDOM_INIT=1 to build the named "
                  f"architecture with random weights.")
