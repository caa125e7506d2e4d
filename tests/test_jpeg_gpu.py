"""Row f-1, JPEG decoding on the GPU: csrc/jpeg.hip (through the C ABI) against Pillow's decode of the same files -- the
reference's `Image.open(path).convert('RGB')` (ref/modules/loader.py:15) -- and against oracle/jpeg_oracle.py; byte-exact."""
import io
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.jpeg_cases import jpeg_cases  # noqa: E402

pytestmark = pytest.mark.gpu


def _pil(data):
    from PIL import Image
    return np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))


def test_device_decode_is_pillow_exact():
    from klab_multimodalmodel_amd import ops
    cases = jpeg_cases()
    for pipelined in (False, True):  # one batch in one go / cut into chunks whose host and device halves overlap
        outs = ops.jpeg_decode([d for _n, d in cases], pipelined=pipelined)  # every sampling mode / size side by side in the same launches
        torch.cuda.synchronize()
        for (name, data), got in zip(cases, outs):
            want = _pil(data)
            g = got.cpu().numpy()
            assert g.shape == want.shape, (name, g.shape, want.shape)
            bad = int((g != want).sum())
            assert bad == 0, (pipelined, name, bad, int(np.abs(g.astype(int) - want.astype(int)).max()))


def test_device_decode_matches_oracle_on_synthetic_coefficients():
    """coefficients no encoder would write (full int16-safe range, dense high frequencies): the range-limit wrap and the 64-bit
    intermediate arithmetic, device vs oracle"""
    from klab_multimodalmodel_amd import ops
    from oracle import jpeg_oracle
    from PIL import Image
    rng = np.random.default_rng(3)
    buf = io.BytesIO()
    Image.fromarray(rng.integers(0, 255, (40, 56, 3), dtype=np.uint8)).save(buf, "JPEG", quality=90, subsampling=2)
    coefs_t, qt, items, rgb_bytes = ops.jpeg_entropy_decode_batch([buf.getvalue()])
    coefs = coefs_t.numpy()
    coefs[:] = rng.integers(-1023, 1024, coefs.shape)
    qt[:] = rng.integers(1, 64, qt.shape)
    rgb, _d = ops.jpeg_decode_device(coefs_t, qt, items, rgb_bytes)
    f = items[0].info
    want = jpeg_oracle.reconstruct(coefs, qt[0], f)
    got = rgb[:f.height * f.width * 3].view(f.height, f.width, 3).cpu().numpy()
    assert (got == want).all(), int((got != want).sum())


def test_from_jpeg_equals_from_decoded():
    """the whole input pipeline from the file bytes: identical pixel_values to PIL decode + from_decoded (which is itself
    Pillow-exact, tests/test_image_pre.py)"""
    from PIL import Image
    from klab_multimodalmodel_amd.modules.image_pipeline import GpuImageProcessor
    rng = np.random.default_rng(9)
    datas = []
    for (h, w, sub) in ((480, 640, 2), (333, 500, 2), (500, 375, 1), (640, 427, 0), (224, 224, 2), (97, 131, 2)):
        yy, xx = np.mgrid[0:h, 0:w]
        a = np.stack([127 + 100 * np.sin(xx / 17.0) * np.cos(yy / 23.0), 255.0 * xx / w + 0 * yy, 255.0 * yy / h + 0 * xx], axis=2)
        a = np.clip(a + rng.normal(0, 6, a.shape), 0, 255).astype(np.uint8)
        buf = io.BytesIO()
        Image.fromarray(a).save(buf, "JPEG", quality=88, subsampling=sub)
        datas.append(buf.getvalue())
    buf = io.BytesIO()
    Image.fromarray(rng.integers(0, 255, (300, 200), dtype=np.uint8), "L").save(buf, "JPEG", quality=70)
    datas.append(buf.getvalue())
    buf = io.BytesIO()
    Image.fromarray(a).save(buf, "JPEG", quality=85, progressive=True)
    datas.append(buf.getvalue())
    proc = GpuImageProcessor.from_pretrained("microsoft/swinv2-base-patch4-window8-256")
    a = proc.from_jpeg(datas)["pixel_values"]
    b = proc.from_decoded([Image.open(io.BytesIO(d)).convert("RGB") for d in datas])["pixel_values"]
    torch.cuda.synchronize()
    assert a.shape == b.shape == (len(datas), 3, 256, 256)
    assert torch.equal(a, b)
    cmyk = io.BytesIO()
    Image.fromarray(rng.integers(0, 255, (32, 32, 3), dtype=np.uint8)).convert("CMYK").save(cmyk, "JPEG")
    with pytest.raises(NotImplementedError):
        proc.from_jpeg([cmyk.getvalue()])
    # the explicit host route for exactly the files the device decoder does not take (here: CMYK JPEG, PNG), in batch order
    png = io.BytesIO()
    Image.fromarray(rng.integers(0, 255, (50, 70, 3), dtype=np.uint8)).save(png, "PNG")
    mixed = [datas[0], cmyk.getvalue(), datas[2], png.getvalue()]
    c = proc.from_jpeg(mixed, other_formats="host")["pixel_values"]
    d = proc.from_decoded([Image.open(io.BytesIO(x)).convert("RGB") for x in mixed])["pixel_values"]
    torch.cuda.synchronize()
    assert torch.equal(c, d)
