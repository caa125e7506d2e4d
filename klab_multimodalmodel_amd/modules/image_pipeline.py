"""GPU input pipeline: the reference's per-image host work as three HIP kernels.

Reference (main process, one image at a time):
    ref/modules/loader.py:15-16   Image.open(path).convert('RGB').resize((256, 256)) ; ToTensor()
    ref/train.py:55               images = image_processor(images, return_tensors="pt").to(device_id)
Everything after the file read runs on the GPU and is bit-identical to Pillow at every uint8 stage: JPEG reconstruction
(`csrc/jpeg.hip`: dequantisation, inverse DCT, chroma upsampling, colour transform; only the serial Huffman decoding stays on the
host, in C++ threads, `csrc/jpeg_host.cpp`), both Pillow resizes, the float conversion, the processor's rescale (twice, as the
reference effectively does) and normalisation (`csrc/image_pre.hip`).  Entry points:

    GpuImageProcessor()(images)                 drop-in for `image_processor(images, return_tensors="pt")`: `images` is the
                                                DataLoader's [B, 3, 256, 256] float batch in [0, 1] (or a list of such CHW tensors)
    GpuImageProcessor().from_decoded(arrays)    list of decoded HWC uint8 RGB images of ANY size (numpy / PIL.Image / tensor):
                                                replaces loader.py:15-16 as well (use `DatasetLoader(decode_only=True)`)
    GpuImageProcessor().from_jpeg(files)        list of JPEG files (paths or bytes): replaces `Image.open(path).convert('RGB')` too.
                                                Baseline and progressive Huffman files; a CMYK, 12-bit or arithmetic-coded file
                                                raises NotImplementedError unless `other_formats="host"` is passed (then exactly
                                                those files are decoded with PIL and resized on the device like the rest)

Both return {"pixel_values": cuda float32 [B, 3, 224, 224]} -- what `MyModel.forward` takes.  No CPU fallback: without the HIP
library the call raises.
"""
import numpy as np
import torch

from .. import ops

BILINEAR, BICUBIC = 2, 3


class BatchFeature(dict):
    """the `.to(device)`-able mapping `image_processor(...)` returns (ref/train.py:55 calls `.to(device_id)` on it)"""

    def to(self, device):
        return BatchFeature({k: v.to(device) for k, v in self.items()})

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


class GpuImageProcessor:
    def __init__(self, size=224, resample=BILINEAR, rescale_factor=1 / 255, image_mean=(0.5, 0.5, 0.5), image_std=(0.5, 0.5, 0.5),
                 loader_size=256, loader_resample=BICUBIC, device="cuda"):
        """defaults = `ViTImageProcessor()` (HF/vitproc:20-27) behind the reference loader's 256x256 BICUBIC resize"""
        if isinstance(size, dict):
            if size["height"] != size["width"]:
                raise NotImplementedError("square outputs only")
            size = size["height"]
        self.size, self.resample = int(size), int(resample)
        self.mean, self.std = tuple(float(m) for m in image_mean), tuple(float(s) for s in image_std)
        self.loader_size, self.loader_resample = int(loader_size), int(loader_resample)
        # the processor turns [0,1] floats back into uint8 for Pillow, scales the result by 1/255 and THEN applies do_rescale
        self.rescale = (1.0 / 255.0) * float(rescale_factor)
        self.device = torch.device(device)

    # `preprocessor_config.json` of the hub checkpoints the reference's CLI accepts (ref/modules/config.py:6-7), restated from
    # the public model cards: ViTImageProcessor, 256x256, BICUBIC, ImageNet mean / std.  Used when the NAME is given and no
    # local directory exists (no hub access here); anything else must be a local directory or explicit keyword arguments.
    KNOWN = {name: dict(size=256, resample=BICUBIC, rescale_factor=1 / 255, image_mean=(0.485, 0.456, 0.406),
                        image_std=(0.229, 0.224, 0.225))
             for name in ("microsoft/swinv2-tiny-patch4-window8-256", "microsoft/swinv2-small-patch4-window8-256",
                          "microsoft/swinv2-base-patch4-window8-256", "microsoft/swinv2-base-patch4-window16-256")}

    @classmethod
    def from_pretrained(cls, path=None, **kw):
        """`AutoImageProcessor.from_pretrained(args.image_model_name)` (ref/train.py:39): a local directory with
        `preprocessor_config.json`, or one of the KNOWN hub names.  Anything else raises OSError like the reference does
        offline -- it never falls back to bare ViTImageProcessor defaults (a silent 224 / bilinear / 0.5-mean pipeline for a
        256 / bicubic / ImageNet checkpoint).  Explicit keyword arguments alone (path=None) build a processor directly."""
        import json
        import os
        keys = ("size", "resample", "rescale_factor", "image_mean", "image_std")
        if path is None:
            if not kw:
                raise OSError("GpuImageProcessor.from_pretrained needs a local directory, a known checkpoint name or explicit settings")
            return cls(**kw)
        cfg_file = os.path.join(path, "preprocessor_config.json") if os.path.isdir(path) else None
        if cfg_file and os.path.exists(cfg_file):
            with open(cfg_file) as f:
                cfg = json.load(f)
            for flag in ("do_resize", "do_rescale", "do_normalize"):
                if cfg.get(flag, True) is False:
                    raise NotImplementedError(f"preprocessor_config.json sets {flag}=false: outside the reference's pipeline")
            return cls(**{**{k: cfg[k] for k in keys if k in cfg}, **kw})
        if path in cls.KNOWN:
            return cls(**{**cls.KNOWN[path], **kw})
        raise OSError(f"Can't load image processor for '{path}': not a local directory containing preprocessor_config.json and not "
                      f"one of {sorted(cls.KNOWN)} (this build has no hub access).")

    def _out(self, n):
        return torch.empty(n, 3, self.size, self.size, dtype=torch.float32, device=self.device)

    def __call__(self, images, return_tensors="pt"):
        if return_tensors != "pt":
            raise ValueError("return_tensors must be 'pt'")
        if isinstance(images, (list, tuple)):
            images = torch.stack([torch.as_tensor(i) for i in images])
        if images.dim() == 3:
            images = images.unsqueeze(0)
        if images.dtype == torch.uint8:
            raise TypeError("uint8 images: use from_decoded() (HWC, any size)")
        B, C, H, W = images.shape
        if C != 3 or H != W:
            raise ValueError("expected [B, 3, S, S] float images in [0, 1] (the reference DataLoader's batches)")
        # [0,1] floats -> the uint8 image Pillow sees (HF to_pil_image: x * 255 -> uint8, truncating; exact for k / 255)
        u8 = (images.to(self.device, non_blocking=True) * 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()
        pv = self._out(B)
        ops.image_preprocess(u8, None, B, H, W, pv, mid=H, out=self.size, filter_a=0, filter_b=self.resample, rescale=self.rescale,
                             mean=self.mean, std=self.std)
        return BatchFeature(pixel_values=pv)

    def from_decoded(self, images):
        """list of decoded RGB images (HWC uint8 numpy arrays, PIL images or tensors) of any size"""
        arrs = [np.asarray(im.convert("RGB") if hasattr(im, "convert") else im, dtype=np.uint8) for im in images]
        for a in arrs:
            if a.ndim != 3 or a.shape[2] != 3:
                raise ValueError("expected HWC RGB uint8 images")
        n = len(arrs)
        sizes = [a.shape[0] * a.shape[1] * 3 for a in arrs]
        offs = np.concatenate([[0], np.cumsum([(s + 15) // 16 * 16 for s in sizes])]).astype(np.int64)
        total = int(offs[-1])
        # two pinned staging buffers, grown on demand and alternated: the async H2D copy of batch i may still be reading one
        # while batch i+1 is packed into the other (pinning per call costs more than the whole pipeline)
        self._flip = 1 - getattr(self, "_flip", 0)
        stage = getattr(self, "_stage", None) or [None, None]
        if stage[self._flip] is None or stage[self._flip].numel() < total:
            stage[self._flip] = torch.empty(max(total, 1), dtype=torch.uint8).pin_memory()
        self._stage = stage
        ev = getattr(self, "_stage_ev", None) or [None, None]
        if ev[self._flip] is not None:
            ev[self._flip].synchronize()  # the copy that last used this buffer has finished
        host = stage[self._flip][:total]
        hb = host.numpy()
        desc = np.zeros((n, 2), np.int64)
        for i, a in enumerate(arrs):
            hb[offs[i]:offs[i] + sizes[i]] = a.reshape(-1)
            desc[i, 0] = offs[i]
            desc[i, 1] = a.shape[0] | (a.shape[1] << 32)  # {int height, width} little-endian
        src = host.to(self.device, non_blocking=True)
        ev[self._flip] = torch.cuda.Event()
        ev[self._flip].record()
        self._stage_ev = ev
        dd = torch.from_numpy(desc).to(self.device)
        pv = self._out(n)
        ops.image_preprocess(src, dd, n, max(a.shape[0] for a in arrs), max(a.shape[1] for a in arrs), pv, mid=self.loader_size,
                             out=self.size, filter_a=self.loader_resample, filter_b=self.resample, rescale=self.rescale, mean=self.mean,
                             std=self.std)
        return BatchFeature(pixel_values=pv)

    def from_jpeg(self, files, n_threads=None, other_formats="raise"):
        """list of image files (paths, bytes or binary file objects) -> pixel_values; `Image.open(f).convert('RGB')` + from_decoded.
        Files the device decoder does not take (CMYK / 12-bit / arithmetic-coded JPEG, PNG, ...): other_formats="raise" (default)
        raises NotImplementedError naming them; other_formats="host" -- an explicit request, never a silent fallback -- decodes
        exactly those files with PIL on the host and sends them through the same device resize (`from_decoded`)."""
        if other_formats not in ("raise", "host"):
            raise ValueError("other_formats must be 'raise' or 'host'")
        datas = []
        for f in files:
            if isinstance(f, (bytes, bytearray, memoryview)):
                datas.append(bytes(f))
            elif hasattr(f, "read"):
                datas.append(f.read())
            else:
                with open(f, "rb") as fh:
                    datas.append(fh.read())
        if n_threads is None:
            import os
            n_threads = min(16, os.cpu_count() or 1)
        n = len(datas)
        dev_idx, host_idx = list(range(n)), []
        if other_formats == "host":
            dev_idx = []
            for i, d in enumerate(datas):
                try:
                    ok = bool(ops.jpeg_read_info(d).supported)
                except ValueError:  # not a JPEG stream at all
                    ok = False
                (dev_idx if ok else host_idx).append(i)
        pv = self._out(n)
        if dev_idx:
            sub = [datas[i] for i in dev_idx]
            rgb, desc, items = ops.jpeg_decode_pipelined(sub, self.device, n_threads)
            out = pv if not host_idx else self._out(len(sub))
            ops.image_preprocess(rgb, desc, len(sub), max(it.info.height for it in items), max(it.info.width for it in items), out,
                                 mid=self.loader_size, out=self.size, filter_a=self.loader_resample, filter_b=self.resample,
                                 rescale=self.rescale, mean=self.mean, std=self.std)
            if host_idx:
                pv[torch.tensor(dev_idx, device=self.device)] = out
        if host_idx:
            import io
            from PIL import Image
            dec = self.from_decoded([Image.open(io.BytesIO(datas[i])).convert("RGB") for i in host_idx])["pixel_values"]
            pv[torch.tensor(host_idx, device=self.device)] = dec
        return BatchFeature(pixel_values=pv)
