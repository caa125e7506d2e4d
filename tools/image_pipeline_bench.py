"""f-1 data point: images/s of the GPU input pipeline (both Pillow resizes + rescale + normalise, `csrc/image_pre.hip`) against
the reference's host path (PIL resize + ToTensor + ViTImageProcessor, ref/modules/loader.py:15-16 + ref/train.py:55) on one
core, for a B=64 batch of 640x480 RGB images that are already decoded.  `python tools/image_pipeline_bench.py [B]`"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    from klab_multimodalmodel_amd.modules.image_pipeline import GpuImageProcessor
    from tests.golden.make_image_pre_golden import synth
    imgs = [synth(480, 640, i) for i in range(B)]
    proc = GpuImageProcessor()
    # (a) kernels only, input bytes resident in HBM
    from klab_multimodalmodel_amd import ops
    sizes = [a.size for a in imgs]
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    src = torch.from_numpy(np.concatenate([a.reshape(-1) for a in imgs])).cuda()
    desc = torch.from_numpy(np.stack([offs[:-1], np.full(B, 480 | (640 << 32), np.int64)], 1).copy()).cuda()
    pv = torch.empty(B, 3, 224, 224, device="cuda")
    for _ in range(3):
        ops.image_preprocess(src, desc, B, 480, 640, pv)
    torch.cuda.synchronize()
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        ops.image_preprocess(src, desc, B, 480, 640, pv)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    nbytes = B * (480 * 640 * 3 + 480 * 256 * 3 * 2 + 256 * 256 * 3 * 2 + 3 * 224 * 224 * 4)
    print(f"kernels only, B={B}: {dt * 1e3:.3f} ms/batch => {B / dt:.0f} images/s ({nbytes / dt / 1e9:.0f} GB/s of algorithmic bytes)")
    # (b) from host arrays (pinned staging + H2D included)
    for _ in range(2):
        proc.from_decoded(imgs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        proc.from_decoded(imgs)
    torch.cuda.synchronize()
    dt2 = (time.perf_counter() - t0) / 5
    print(f"from host uint8 arrays (pack + H2D + kernels): {dt2 * 1e3:.2f} ms/batch => {B / dt2:.0f} images/s")
    # (d) from JPEG file bytes: host Huffman decoding (C++ threads) + device reconstruction + the resizes
    import io
    from PIL import Image
    datas = []
    for a in imgs:
        buf = io.BytesIO()
        Image.fromarray(a).save(buf, "JPEG", quality=90, subsampling=2)
        datas.append(buf.getvalue())
    jbytes = sum(len(d) for d in datas)
    thr = min(16, os.cpu_count() or 1)
    for _ in range(2):
        proc.from_jpeg(datas, n_threads=thr)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        proc.from_jpeg(datas, n_threads=thr)
    torch.cuda.synchronize()
    dtj = (time.perf_counter() - t0) / 5
    print(f"from JPEG bytes ({jbytes / B / 1e3:.0f} KB/image, {thr} host threads): {dtj * 1e3:.2f} ms/batch => {B / dtj:.0f} images/s")
    t0 = time.perf_counter()
    for _ in range(5):
        coefs_t, qt, items, rgb_bytes = ops.jpeg_entropy_decode_batch(datas, thr)
    dth = (time.perf_counter() - t0) / 5
    print(f"  host entropy decoding alone: {dth * 1e3:.2f} ms/batch => {B / dth:.0f} images/s ({jbytes / dth / 1e6:.0f} MB/s of JPEG)")
    # device reconstruction alone (coefficients resident): through the C ABI with preallocated buffers
    import ctypes as C
    from klab_multimodalmodel_amd import _lib as L
    lib = L.load()
    coefs_dev = coefs_t.cuda(); qt_dev = torch.from_numpy(qt.view(np.int16)).cuda()
    items_dev = torch.from_numpy(np.frombuffer(bytes(items), dtype=np.uint8).copy()).cuda()
    nws = lib.klab_jpeg_decode_ws_bytes(C.cast(items, C.c_void_p), B)
    ws = torch.empty(nws, dtype=torch.uint8, device="cuda"); rgb = torch.empty(rgb_bytes, dtype=torch.uint8, device="cuda")
    def dev():
        L.check(lib.klab_jpeg_decode_device(coefs_dev.data_ptr(), qt_dev.data_ptr(), C.cast(items, C.c_void_p), items_dev.data_ptr(), B,
                                            rgb.data_ptr(), ws.data_ptr(), nws, L.stream_ptr()), "jpeg")
    for _ in range(3):
        dev()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        dev()
    torch.cuda.synchronize()
    dtd = (time.perf_counter() - t0) / 20
    alg = coefs_t.numel() * 2 + 2 * coefs_t.numel() + rgb_bytes  # coefficients in, sample planes out and in again, RGB out
    print(f"  device IDCT + upsampling + colour alone: {dtd * 1e3:.3f} ms/batch => {B / dtd:.0f} images/s ({alg / dtd / 1e9:.0f} GB/s of algorithmic bytes)")
    t0 = time.perf_counter()
    for d in datas[:16]:
        np.asarray(Image.open(io.BytesIO(d)).convert("RGB"))
    dtp = (time.perf_counter() - t0) / 16
    print(f"  PIL decode (libjpeg-turbo, 1 core): {dtp * 1e3:.2f} ms/image => {1 / dtp:.0f} images/s")
    # (c) the reference's host path on one core
    try:
        from PIL import Image
        from transformers import ViTImageProcessor
        hp = ViTImageProcessor()
        m = min(B, 16)
        t0 = time.perf_counter()
        ts = [torch.from_numpy(np.asarray(Image.fromarray(a).convert('RGB').resize((256, 256))).transpose(2, 0, 1).copy()).float().div(255)
              for a in imgs[:m]]
        ref = hp(ts, return_tensors="pt")["pixel_values"]
        dt3 = (time.perf_counter() - t0) / m
        print(f"reference host path (PIL + ViTImageProcessor, 1 core, decode excluded): {dt3 * 1e3:.2f} ms/image => {1 / dt3:.0f} images/s")
        print("max |gpu - host| =", float((pv[:m].cpu() - ref).abs().max()))
    except Exception as e:  # PIL / transformers missing: only the GPU numbers
        print("host path not timed:", e)


if __name__ == "__main__":
    main()
