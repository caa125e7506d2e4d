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
    # a decompression bomb: ~200 bytes declaring 65535 x 65535 must be refused from the header, before anything is sized from it
    # (the reference's Image.open raises DecompressionBombError above 178,956,970 pixels)
    buf = io.BytesIO()
    img.save(buf, "JPEG", quality=80)
    d = bytearray(buf.getvalue())
    i = bytes(d).find(b"\xff\xc0")
    d[i + 5:i + 9] = bytes([0xFF, 0xFF, 0xFF, 0xFF])
    with pytest.raises((NotImplementedError, ValueError, RuntimeError)):
        ops.jpeg_read_info(bytes(d))
    with pytest.raises((NotImplementedError, ValueError, RuntimeError)):
        ops.jpeg_entropy_decode_batch([bytes(d)])
    # truncated entropy data decodes (zeros are fed, as libjpeg does) rather than reading out of bounds
    buf = io.BytesIO()
    img.save(buf, "JPEG", quality=80)
    cut = buf.getvalue()[:len(buf.getvalue()) // 2]
    ops.jpeg_entropy_decode_batch([cut])


def test_host_decoder_survives_mutated_files(tmp_path):
    """the host stage parses bytes from disk: an AddressSanitizer + UBSan build of csrc/jpeg_host.cpp is run over ~1500 corrupted
    variants of valid files (bit flips, truncations, spliced segments, inflated dimensions); it may reject them, never touch memory
    outside its buffers"""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no host compiler")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "jpeg_fuzz"
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", f"-I{root}/include",
           f"{root}/klab_multimodalmodel_amd/csrc/jpeg_host.cpp", f"{root}/tests/jpeg_fuzz_main.cpp", "-o", str(exe), "-lpthread"]
    subprocess.run(cmd, check=True, capture_output=True)
    rng = np.random.default_rng(1234)
    seeds = [d for n, d in jpeg_cases() if any(k in n for k in ("37x53", "8x8", "restart", "grey_9x9", "prog_420_64x64", "prog_grey", "optimize"))]
    files = []
    k = 0
    for d in seeds:
        arr = np.frombuffer(d, np.uint8)
        for _ in range(40):
            m = arr.copy()
            kind = rng.integers(0, 5)
            if kind == 0:    # a few random byte substitutions anywhere
                for _j in range(int(rng.integers(1, 6))):
                    m[rng.integers(0, len(m))] = rng.integers(0, 256)
            elif kind == 1:  # truncation
                m = m[:int(rng.integers(2, len(m)))]
            elif kind == 2:  # damage concentrated in the headers (markers, lengths, table definitions, frame / scan headers)
                for _j in range(int(rng.integers(1, 4))):
                    m[rng.integers(2, min(len(m), 700))] = rng.integers(0, 256)
            elif kind == 3:  # a slice of the file spliced in somewhere else
                a, b = sorted(rng.integers(0, len(m), 2))
                at = int(rng.integers(0, len(m)))
                m = np.concatenate([m[:at], m[a:b], m[at:]])
            else:            # a second frame header with other dimensions / sampling pasted behind the first
                i = d.find(b"\xff\xc0") if b"\xff\xc0" in d else d.find(b"\xff\xc2")
                ln = (d[i + 2] << 8) | d[i + 3]
                sof = bytearray(d[i:i + 2 + ln])
                sof[5:9] = bytes([0x03, 0xe8, 0x03, 0xe8])  # 1000 x 1000
                at = i + 2 + ln
                m = np.concatenate([m[:at], np.frombuffer(bytes(sof), np.uint8), m[at:]])
            f = tmp_path / f"m{k}.jpg"
            f.write_bytes(m.tobytes())
            files.append(str(f))
            k += 1
    # deterministic: every seed cut right behind the length field of each of its marker segments (a header parser that reads a
    # segment's first byte before checking its length walks one byte past the buffer exactly there)
    for d in seeds:
        i = 2
        while i + 4 <= len(d) and d[i] == 0xFF:
            mk = d[i + 1]
            f = tmp_path / f"m{k}.jpg"
            f.write_bytes(d[:i + 4])
            files.append(str(f))
            k += 1
            if mk == 0xDA:
                break
            i += 2 + ((d[i + 2] << 8) | d[i + 3])
    assert len(files) > 1000
    for i in range(0, len(files), 400):
        r = subprocess.run([str(exe)] + files[i:i + 400], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-3000:])
