"""where the host side of GpuImageProcessor.from_jpeg spends its time (per B=64 batch of 640x480 files)"""
import io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import torch
from PIL import Image
from klab_multimodalmodel_amd import _lib as L, ops
from tests.golden.make_image_pre_golden import synth

B = 64
datas = []
for i in range(B):
    b = io.BytesIO(); Image.fromarray(synth(480, 640, i)).save(b, "JPEG", quality=90, subsampling=2); datas.append(b.getvalue())
lib = L.load()
def t(f, n=5):
    f(); t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e3
items = (L.JpegItem * B)()
def infos():
    for i, d in enumerate(datas):
        lib.klab_jpeg_read_info(C.cast(C.c_char_p(d), C.c_void_p), len(d), C.byref(items[i].info))
print("read_info x64: %.2f ms" % t(infos))
blocks = sum(items[i].info.coef_blocks for i in range(B))
print("pinned torch.empty: %.2f ms" % t(lambda: torch.empty((blocks, 64), dtype=torch.int16, pin_memory=True)))
coefs_t = torch.empty((blocks, 64), dtype=torch.int16, pin_memory=True)
coefs = coefs_t.numpy(); qt = np.zeros((B, 3, 64), np.uint16)
off = np.concatenate([[0], np.cumsum([items[i].info.coef_blocks for i in range(B)])])
ptrs = (C.c_void_p * B)(*[C.cast(C.c_char_p(d), C.c_void_p) for d in datas]); sizes = (C.c_size_t * B)(*[len(d) for d in datas])
cps = (C.c_void_p * B)(*[coefs.ctypes.data + int(off[i]) * 128 for i in range(B)]); rcs = (C.c_int * B)()
for thr in (1, 4, 8, 16, 32):
    ms = t(lambda: lib.klab_jpeg_entropy_decode_batch(C.cast(ptrs, C.c_void_p), C.cast(sizes, C.c_void_p), B, C.cast(cps, C.c_void_p), qt.ctypes.data, None, C.cast(rcs, C.c_void_p), thr))
    print("C decode, %2d threads: %.2f ms (%.2f ms per image per thread)" % (thr, ms, ms * thr / B))
print("whole ops.jpeg_entropy_decode_batch: %.2f ms" % t(lambda: ops.jpeg_entropy_decode_batch(datas, 16)))
print("H2D of the coefficients (%.0f MB): %.2f ms" % (coefs_t.numel() * 2 / 1e6, t(lambda: (coefs_t.cuda(non_blocking=True), torch.cuda.synchronize()))))
