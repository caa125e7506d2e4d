"""JPEG files for the decoder tests, produced with Pillow's encoder: every sampling mode, grey, odd sizes (partial MCUs), widths
of 1-5 pixels (the non-"fancy" upsampling branch), restart intervals, optimised Huffman tables, extreme quality settings,
progressive files."""
import io

import numpy as np


def _img(rng, h, w, kind):
    yy, xx = np.mgrid[0:h, 0:w]
    if kind == "noise":
        a = rng.integers(0, 256, (h, w, 3))
    elif kind == "smooth":
        a = np.stack([127 + 120 * np.sin(xx / 7.0 + yy / 11.0), 127 + 120 * np.cos(xx / 5.0), 255.0 * yy / max(h - 1, 1) + 0 * xx], axis=2)
    else:  # edges: saturated blocks (drives the IDCT output far outside [0, 255] -> range limiting)
        a = np.where(((xx // 3 + yy // 5) % 2)[..., None] > 0, [255, 0, 255], [0, 255, 0]) + rng.integers(-3, 4, (h, w, 3))
    return np.clip(a, 0, 255).astype(np.uint8)


def jpeg_cases():
    from PIL import Image
    rng = np.random.default_rng(11)
    out = []

    def add(name, arr, mode="RGB", **kw):
        im = Image.fromarray(arr if mode == "RGB" else arr[..., 0], mode if mode != "RGB" else None)
        buf = io.BytesIO()
        im.save(buf, "JPEG", **kw)
        out.append((name, buf.getvalue()))

    for sub, sname in ((0, "444"), (1, "422"), (2, "420")):
        for (h, w) in ((64, 64), (37, 53), (17, 100), (120, 9), (8, 8), (1, 1), (33, 3), (5, 4), (16, 5), (3, 2)):
            for kind in ("noise", "smooth", "edges"):
                add(f"{sname}_{h}x{w}_{kind}", _img(rng, h, w, kind), quality=int(rng.integers(30, 96)), subsampling=sub)
    for q in (1, 5, 50, 100):
        add(f"420_q{q}", _img(rng, 61, 77, "smooth"), quality=q, subsampling=2)
        add(f"444_q{q}", _img(rng, 40, 40, "noise"), quality=q, subsampling=0)
    add("420_optimize", _img(rng, 90, 131, "smooth"), quality=85, subsampling=2, optimize=True)
    add("422_optimize", _img(rng, 90, 131, "noise"), quality=70, subsampling=1, optimize=True)
    for ri in (1, 2, 7):
        try:
            add(f"420_restart{ri}", _img(rng, 75, 99, "smooth"), quality=80, subsampling=2, restart_marker_blocks=ri)
            add(f"444_restart_rows{ri}", _img(rng, 50, 70, "noise"), quality=60, subsampling=0, restart_marker_rows=ri)
        except TypeError:
            pass
    # progressive files (spectral selection + successive approximation, one-component AC scans over the component's own block grid)
    for sub, sname in ((0, "444"), (1, "422"), (2, "420")):
        for (h, w) in ((64, 64), (37, 53), (120, 9), (8, 8), (1, 1), (33, 3), (200, 301)):
            for kind in ("noise", "smooth", "edges"):
                add(f"prog_{sname}_{h}x{w}_{kind}", _img(rng, h, w, kind), quality=int(rng.integers(20, 96)), subsampling=sub, progressive=True)
    add("prog_420_optimize_q100", _img(rng, 90, 131, "noise"), quality=100, subsampling=2, progressive=True, optimize=True)
    add("prog_grey", _img(rng, 77, 45, "smooth"), mode="L", quality=60, progressive=True)
    for ri in (1, 3):
        try:
            add(f"prog_420_restart{ri}", _img(rng, 75, 99, "smooth"), quality=80, subsampling=2, progressive=True, restart_marker_blocks=ri)
        except TypeError:
            pass
    add("grey_50x70", _img(rng, 50, 70, "smooth"), mode="L", quality=75)
    add("grey_9x9", _img(rng, 9, 9, "noise"), mode="L", quality=90)
    add("big_420", _img(rng, 480, 640, "smooth"), quality=90, subsampling=2)
    return out
