"""Generates tests/golden/image_pre.npz from the real thing: Pillow's Image.resize (ref/modules/loader.py:15) and the installed
transformers ViTImageProcessor (ref/train.py:55) on seeded synthetic images.  Run from the repo root:
    python tests/golden/make_image_pre_golden.py
"""
import os

import numpy as np
import torch
from PIL import Image
from transformers import ViTImageProcessor

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = [  # (height, width, loader size, processor size)
    (61, 83, 32, 24), (40, 40, 32, 24), (97, 50, 32, 28), (20, 25, 32, 24), (120, 100, 256, 224)]


def synth(h, w, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([127 + 120 * np.sin(xx / 7.0 + yy / 5.0), 127 + 120 * np.cos(xx / 3.0), yy * 255.0 / max(h - 1, 1)], -1)
    return np.clip(base + rng.normal(0, 20, base.shape), 0, 255).astype(np.uint8)


def main():
    out = {}
    for i, (h, w, mid, osz) in enumerate(CASES):
        img = synth(h, w, 100 + i)
        loader = Image.fromarray(img).convert("RGB").resize((mid, mid))                     # loader.py:15
        t = torch.from_numpy(np.asarray(loader).transpose(2, 0, 1).copy()).float().div(255)  # ToTensor, loader.py:16
        proc = ViTImageProcessor(size={"height": osz, "width": osz})
        pv = proc([t], return_tensors="pt")["pixel_values"][0].numpy()                       # train.py:55
        out[f"img{i}"] = img
        out[f"mid{i}"] = np.asarray(loader)
        if osz <= 32:
            out[f"pv{i}"] = pv
        else:  # keep the fixture small: the uint8 image the floats are an affine map of, plus a checksum of the floats
            u8 = np.rint((pv * 0.5 + 0.5) * 255 * 255).astype(np.uint8).transpose(1, 2, 0)
            assert np.abs(((u8.astype(np.float64) / 255 / 255 - 0.5) / 0.5).transpose(2, 0, 1) - pv).max() < 1e-7
            out[f"u8_{i}"] = u8
            out[f"pvsum{i}"] = np.array([pv.astype(np.float64).sum(), np.abs(pv.astype(np.float64)).max()])
    out["cases"] = np.array(CASES)
    np.savez_compressed(os.path.join(HERE, "image_pre.npz"), **out)


if __name__ == "__main__":
    main()
