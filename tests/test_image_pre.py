"""image preprocessing.  CPU: the oracle (oracle/image_pre.py) against Pillow itself, the installed
ViTImageProcessor and the committed golden vectors.  GPU: the HIP kernels through the C ABI against the oracle, bit-exact at
every uint8 stage (the float output is an affine map of the last one; tolerance 1e-6 absolute on it)."""
import os

import numpy as np
import pytest
import torch

from oracle import image_pre as O
from tests.golden.make_image_pre_golden import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "image_pre.npz")
SIZES = [(480, 640), (427, 640), (375, 500), (100, 80), (256, 256), (1000, 37), (31, 700), (256, 300), (1, 1), (2, 513)]


def test_oracle_matches_golden_vectors():
    g = np.load(GOLD)
    for i, (h, w, mid, osz) in enumerate(g["cases"]):
        img = g[f"img{i}"]
        assert img.shape == (h, w, 3)
        assert np.array_equal(O.resize_u8(img, mid, mid, O.BICUBIC), g[f"mid{i}"])
        pv, u8 = O.preprocess(img, mid=int(mid), out=int(osz))
        if f"pv{i}" in g:
            assert np.array_equal(pv, g[f"pv{i}"])
        else:
            assert np.array_equal(u8, g[f"u8_{i}"])
            assert abs(pv.astype(np.float64).sum() - g[f"pvsum{i}"][0]) < 1e-6


@pytest.mark.parametrize("h,w", SIZES)
def test_oracle_resize_equals_pillow(h, w):
    Image = pytest.importorskip("PIL.Image")
    img = synth(h, w, h * 1000 + w)
    for f, osz in ((O.BICUBIC, 256), (O.BILINEAR, 224), (O.BILINEAR, 256), (O.BICUBIC, 224)):
        ref = np.asarray(Image.fromarray(img).resize((osz, osz), resample=f))
        assert np.array_equal(O.resize_u8(img, osz, osz, f), ref), (f, osz)
    assert np.array_equal(np.asarray(Image.fromarray(img).resize((256, 256))), O.resize_u8(img, 256, 256, O.BICUBIC))  # PIL default


def test_oracle_equals_reference_chain_with_installed_processor():
    """loader.py:15-16 + train.py:55 executed with Pillow + transformers vs the oracle: identical floats"""
    Image = pytest.importorskip("PIL.Image")
    tr = pytest.importorskip("transformers")
    img = synth(333, 500, 7)
    loader = Image.fromarray(img).convert("RGB").resize((256, 256))
    t = torch.from_numpy(np.asarray(loader).transpose(2, 0, 1).copy()).float().div(255)
    ref = tr.ViTImageProcessor()([t], return_tensors="pt")["pixel_values"][0].numpy()
    pv, _ = O.preprocess(img)
    assert np.abs(pv - ref).max() <= 1e-7
    assert ref.max() < -0.99  # the reference's double rescale: everything lands in [-1, -0.992]


def test_uint8_roundtrip_of_totensor_values():
    v = torch.arange(256, dtype=torch.float32) / 255  # ToTensor
    assert torch.equal((v * 255).to(torch.uint8), torch.arange(256, dtype=torch.uint8))  # to_pil_image's truncating cast


# ------------------------------------------------------------------------------------------------------------- GPU
@pytest.fixture(scope="module")
def proc():
    from klab_multimodalmodel_amd.modules.image_pipeline import GpuImageProcessor
    return GpuImageProcessor()


def _u8_of(pv):
    return np.rint((pv.astype(np.float64) * 0.5 + 0.5) * 255 * 255).astype(np.int64)


@pytest.mark.gpu
def test_gpu_pipeline_from_decoded_is_bit_exact(proc):
    imgs = [synth(h, w, 31 * h + w) for h, w in SIZES + [(640, 480), (1200, 1600)]]
    pv = proc.from_decoded(imgs)["pixel_values"].cpu().numpy()
    assert pv.shape == (len(imgs), 3, 224, 224)
    for i, im in enumerate(imgs):
        ref, u8 = O.preprocess(im)
        assert np.array_equal(_u8_of(pv[i]).transpose(1, 2, 0), u8.astype(np.int64)), SIZES[i] if i < len(SIZES) else i
        assert np.abs(pv[i] - ref).max() <= 1e-6


@pytest.mark.gpu
def test_gpu_processor_drop_in_for_dataloader_batches(proc):
    """`image_processor(images, return_tensors="pt")` on the DataLoader's [B,3,256,256] floats (ref/train.py:55)"""
    mids = [O.resize_u8(synth(300 + 17 * i, 400 - 9 * i, i), 256, 256, O.BICUBIC) for i in range(5)]
    batch = torch.stack([torch.from_numpy(m.transpose(2, 0, 1).copy()).float().div(255) for m in mids])
    out = proc(batch, return_tensors="pt")
    pv = out["pixel_values"].cpu().numpy()
    assert out.pixel_values.is_cuda and out.to("cuda:0")["pixel_values"].shape == (5, 3, 224, 224)
    for i, m in enumerate(mids):
        ref, _ = O.preprocess(m, filter_a=None)
        assert np.abs(pv[i] - ref).max() <= 1e-6
        assert np.array_equal(_u8_of(pv[i]), _u8_of(ref))


@pytest.mark.gpu
def test_gpu_pipeline_golden_and_small_sizes():
    from klab_multimodalmodel_amd.modules.image_pipeline import GpuImageProcessor
    g = np.load(GOLD)
    for i, (h, w, mid, osz) in enumerate(g["cases"]):
        p = GpuImageProcessor(size=int(osz), loader_size=int(mid))
        pv = p.from_decoded([g[f"img{i}"]])["pixel_values"][0].cpu().numpy()
        if f"pv{i}" in g:
            assert np.abs(pv - g[f"pv{i}"]).max() <= 1e-6
            assert np.array_equal(_u8_of(pv), _u8_of(g[f"pv{i}"]))
        else:
            assert np.array_equal(_u8_of(pv).transpose(1, 2, 0), g[f"u8_{i}"].astype(np.int64))


@pytest.mark.gpu
def test_gpu_pipeline_properties_at_full_batch(proc):
    """size-independent checks on a B=64 batch of COCO-sized images: constant images stay constant (weights sum to 1 after
    rounding only approximately -- Pillow's own behaviour, so compare with the oracle on one of them), batch entries are
    independent of their neighbours, and a second call reproduces the first bit for bit."""
    rng = np.random.default_rng(0)
    imgs = [synth(int(rng.integers(300, 641)), int(rng.integers(300, 641)), 1000 + i) for i in range(64)]
    a = proc.from_decoded(imgs)["pixel_values"]
    b = proc.from_decoded(imgs)["pixel_values"]
    assert torch.equal(a, b)
    c = proc.from_decoded(imgs[5:9])["pixel_values"]
    assert torch.equal(a[5:9], c)
    ref, _ = O.preprocess(imgs[63])
    assert np.abs(a[63].cpu().numpy() - ref).max() <= 1e-6
    flat = np.full((480, 640, 3), 200, np.uint8)
    pv = proc.from_decoded([flat])["pixel_values"][0].cpu().numpy()
    assert np.array_equal(_u8_of(pv), _u8_of(O.preprocess(flat)[0]))


@pytest.mark.gpu
def test_gpu_pipeline_rejects_what_it_cannot_do(proc):
    with pytest.raises(NotImplementedError):
        proc.from_decoded([np.zeros((4, 9000, 3), np.uint8)])  # > 30x reduction: weight table exceeds LDS
    with pytest.raises(ValueError):
        proc.from_decoded([np.zeros((4, 4), np.uint8)])
    with pytest.raises(TypeError):
        proc(torch.zeros(1, 3, 256, 256, dtype=torch.uint8))
