"""Row f-1, JPEG decoding, CPU side: the product's host Huffman decoder (csrc/jpeg_host.cpp, through the C ABI) + the numpy
restatement of the device stages (oracle/jpeg_oracle.py) must reproduce Pillow's decode -- the reference's decoder
(`Image.open(path).convert('RGB')`, ref/modules/loader.py:15) -- byte for byte.  This pins the oracle the GPU tests use."""
import io
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.jpeg_cases import jpeg_cases  # noqa: E402


def _pil(data):
    from PIL import Image
    return np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))


def test_host_entropy_decoder_and_oracle_match_pillow():
    from klab_multimodalmodel_amd import ops
    from oracle import jpeg_oracle
    cases = jpeg_cases()
    datas = [d for _n, d in cases]
    coefs_t, qt, items, _rgb = ops.jpeg_entropy_decode_batch(datas, n_threads=4)
    coefs = coefs_t.numpy()
    for i, (name, data) in enumerate(cases):
        f = items[i].info
        got = jpeg_oracle.reconstruct(coefs[items[i].coef_block0:items[i].coef_block0 + f.coef_blocks], qt[i], f)
        want = _pil(data)
        assert got.shape == want.shape, (name, got.shape, want.shape)
        bad = int((got != want).sum())
        assert bad == 0, (name, bad, int(np.abs(got.astype(int) - want.astype(int)).max()))


def test_header_info_and_unsupported_files():
    from PIL import Image
    from klab_multimodalmodel_amd import ops
    rng = np.random.default_rng(5)
    img = Image.fromarray(rng.integers(0, 255, (37, 53, 3), dtype=np.uint8))
    buf = io.BytesIO()
    img.save(buf, "JPEG", quality=80, subsampling=2)
    f = ops.jpeg_read_info(buf.getvalue())
    assert (f.width, f.height, f.ncomp, f.hmax, f.vmax, f.supported, f.progressive) == (53, 37, 3, 2, 2, 1, 0)
    assert (f.mcus_x, f.mcus_y, list(f.bw), list(f.bh)) == (4, 3, [8, 4, 4], [6, 3, 3])
    buf = io.BytesIO()
    img.save(buf, "JPEG", quality=80, progressive=True)
    f = ops.jpeg_read_info(buf.getvalue())
    assert f.progressive == 1 and f.supported == 1
    buf = io.BytesIO()
    img.convert("CMYK").save(buf, "JPEG")
    assert ops.jpeg_read_info(buf.getvalue()).supported == 0
    with pytest.raises(NotImplementedError):
        ops.jpeg_entropy_decode_batch([buf.getvalue()])
    with pytest.raises(ValueError):
        ops.jpeg_read_info(b"not a jpeg at all")
    # truncated entropy data decodes (zeros are fed, as libjpeg does) rather than reading out of bounds
    buf = io.BytesIO()
    img.save(buf, "JPEG", quality=80)
    cut = buf.getvalue()[:len(buf.getvalue()) // 2]
    ops.jpeg_entropy_decode_batch([cut])
