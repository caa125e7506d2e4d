"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol declared in
include/klab_mm.h (no compute without a GPU), the host mirror keeps the reference's surfaces
(state-dict schema, argparse defaults, span-mask text prep, from_pretrained-style loading)."""
import json
import os
import re
import sys
import types

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def built():
    from klab_multimodalmodel_amd.build import build_library
    build_library(verbose=False)
    from klab_multimodalmodel_amd import _lib
    return _lib.load()


def test_library_exports_every_declared_symbol(built):
    hdr = open(os.path.join(ROOT, "include", "klab_mm.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(klab_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 35
    for n in sorted(names):
        assert hasattr(built, n), f"{n} declared in klab_mm.h but not exported by libklab_mm.so"
    assert built.klab_version() >= 1
    from klab_multimodalmodel_amd import engine
    engine.lib()  # installs the engine signatures: every one must resolve
    from klab_multimodalmodel_amd import _lib
    assert set(_lib.SIGNATURES) | set(engine.ENGINE_SIGS) >= names


def test_missing_library_fails_loudly(monkeypatch):
    from klab_multimodalmodel_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libklab_mm.so")
    with pytest.raises(_lib.KlabError, match="no CPU fallback"):
        _lib.load()


def _args(train=False):
    return types.SimpleNamespace(result_dir="/tmp", language_model_name="-", image_model_name="-", image_model_train=train,
                                 transformer_model_name="-")


def _tiny():
    from klab_multimodalmodel_amd.engine import SwinConfig, T5Config
    return (SwinConfig(image_size=64, embed_dim=16, depths=[2, 2, 2, 2], num_heads=[1, 2, 4, 8], window_size=4),
            T5Config(vocab_size=384, d_model=128, d_kv=16, num_heads=4, d_ff=256, num_layers=2))


def test_state_dict_schema_matches_the_reference_goldens(built):
    """key names and shapes of transformer / image_model / language_model == the HF state dicts the reference saves."""
    import numpy as np
    from klab_multimodalmodel_amd.models.model import MyModel
    sw, t5 = _tiny()
    m = MyModel(_args(True), _configs=(sw, t5, t5))
    z = np.load(os.path.join(GOLD, "tiny_a.npz"))
    for prefix, tree in (("w.main.", m.transformer), ("w.swin.", m.image_model), ("w.lang.", m.language_model)):
        sd = tree.state_dict()
        gold = {k[len(prefix):]: z[k].shape for k in z.files if k.startswith(prefix)}
        for k, shp in gold.items():
            assert k in sd and tuple(sd[k].shape) == tuple(shp), (prefix, k)
        extra = set(sd) - set(gold)
        assert extra <= {"encoder.embed_tokens.weight", "decoder.embed_tokens.weight", "lm_head.weight"}, extra
    sd = m.transformer.state_dict()
    for k in ("encoder.embed_tokens.weight", "decoder.embed_tokens.weight", "lm_head.weight"):
        assert sd[k].data_ptr() == sd["shared.weight"].data_ptr()  # tied (HF/t5:902-906)
    # reference semantics of the three towers (ref/models/model.py:14-17)
    assert not any(p.requires_grad for p in m.language_model.parameters())
    assert all(p.requires_grad for p in m.image_model.parameters())
    assert not m.transformer.training and not m.image_model.training
    assert len(list(m.transformer.parameters())) == len({p.data_ptr() for p in m.transformer.parameters()})


def test_save_load_roundtrip_and_checkpoint_keys(built, tmp_path):
    from klab_multimodalmodel_amd.models.model import MyModel
    sw, t5 = _tiny()
    a = _args(True)
    a.result_dir = str(tmp_path)
    m = MyModel(a, _configs=(sw, t5, t5), _seed=1)
    m.save("x.pth")
    ck = torch.load(tmp_path / "x.pth")
    assert set(ck) == {"transformer", "image_model"}  # ref/models/model.py:32-35
    m2 = MyModel(a, _configs=(sw, t5, t5), _seed=2)
    assert not torch.equal(m2.transformer.get_parameter("shared.weight"), m.transformer.get_parameter("shared.weight"))
    m2.load("x.pth")
    for (k, p), (_, q) in zip(m.transformer.named_parameters(), m2.transformer.named_parameters()):
        assert torch.equal(p, q), k
    a2 = _args(False)
    a2.result_dir = str(tmp_path)
    MyModel(a2, _configs=(sw, t5, t5)).save("y.pth")
    assert set(torch.load(tmp_path / "y.pth")) == {"transformer"}


def test_from_pretrained_style_directories(built, tmp_path):
    """MyModel(args) with local directories (config.json + model.safetensors), the only form usable offline."""
    import numpy as np
    from safetensors.torch import save_file
    from klab_multimodalmodel_amd.models.model import MyModel
    z = np.load(os.path.join(GOLD, "tiny_b.npz"))
    meta = json.load(open(os.path.join(GOLD, "tiny_b.json")))
    for name, prefix, cfg in (("swin", "w.swin.", meta["swin_config"]), ("lang", "w.lang.", meta["t5_config"]), ("main", "w.main.", meta["t5_config"])):
        d = tmp_path / name
        d.mkdir()
        json.dump(cfg, open(d / "config.json", "w"))
        save_file({k[len(prefix):]: torch.from_numpy(z[k]).contiguous() for k in z.files if k.startswith(prefix)}, str(d / "model.safetensors"))
    a = types.SimpleNamespace(result_dir=str(tmp_path), language_model_name=str(tmp_path / "lang"), image_model_name=str(tmp_path / "swin"),
                              image_model_train=False, transformer_model_name=str(tmp_path / "main"))
    m = MyModel(a)
    assert torch.equal(m.transformer.get_parameter("shared.weight"), torch.from_numpy(z["w.main.shared.weight"]))
    assert torch.equal(m.image_model.get_parameter("layernorm.weight"), torch.from_numpy(z["w.swin.layernorm.weight"]))
    with pytest.raises(OSError):
        MyModel(types.SimpleNamespace(result_dir="/tmp", language_model_name="t5-small", image_model_name="x", image_model_train=False,
                                      transformer_model_name="t5-small"))


def test_cpu_forward_is_refused(built):
    from klab_multimodalmodel_amd.models.model import MyModel
    sw, t5 = _tiny()
    m = MyModel(_args(), _configs=(sw, t5, t5))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m({"pixel_values": torch.zeros(1, 3, 64, 64)}, {"input_ids": torch.ones(1, 3, dtype=torch.long)}, {"input_ids": torch.ones(1, 3, dtype=torch.long)})


def test_argparse_defaults_equal_the_reference(monkeypatch):
    from klab_multimodalmodel_amd.modules import parse_arguments
    monkeypatch.setattr(sys, "argv", ["train.py"])
    got = vars(parse_arguments())
    want = json.load(open(os.path.join(GOLD, "argparse_defaults.json")))
    assert got == want
    monkeypatch.setattr(sys, "argv", ["train.py", "--language_model_name", "nope"])
    with pytest.raises(SystemExit):
        parse_arguments()


def test_span_mask_matches_recorded_reference_output():
    from klab_multimodalmodel_amd.modules.loader import span_mask
    """45 (seed, caption) pairs produced by the reference's own RedCapsDatasetLoader.__getitem__
    (tests/golden/make_spanmask_goldens.py): punctuation splitting, the 15 % + 1 count, sentinel numbering, RNG consumption,
    one-word / empty / whitespace-heavy captions."""
    vecs = json.load(open(os.path.join(GOLD, "spanmask.json")))["vectors"]
    assert len(vecs) >= 20
    for g in vecs:
        torch.manual_seed(g["seed"])
        src, tgt = span_mask(g["caption"])
        assert (src, tgt) == (g["src"], g["tgt"]), g


def test_span_mask_properties():
    from hypothesis import given, settings, strategies as st
    from klab_multimodalmodel_amd.modules.loader import span_mask

    @settings(max_examples=60, deadline=None)
    @given(st.lists(st.sampled_from(["cat", "dog", "a", "the", "runs", "fast", "blue", "sky", "on", "mat"]), min_size=1, max_size=30),
           st.integers(0, 1000))
    def prop(words, seed):
        torch.manual_seed(seed)
        src, tgt = span_mask(" ".join(words))
        s, t = src.split(), tgt.split()
        k = int(len(words) * 0.15) + 1
        sent_src = [w for w in s if w.startswith("<extra_id_")]
        assert sent_src == [f"<extra_id_{i}>" for i in range(k)]  # numbered in order of appearance
        assert t[0] == "<extra_id_0>" and t[-1] == f"<extra_id_{k}>" and len(t) == 2 * k + 1
        # substituting the target words back reconstructs the caption
        fill = {f"<extra_id_{i}>": t[2 * i + 1] for i in range(k)}
        assert [fill.get(w, w) for w in s] == words
    prop()


def test_dataset_item_matches_totensor_semantics(tmp_path):
    from PIL import Image
    import numpy as np
    from klab_multimodalmodel_amd.modules.loader import DatasetLoader
    arr = (np.arange(12 * 10 * 3) % 251).astype(np.uint8).reshape(12, 10, 3)
    Image.fromarray(arr).save(tmp_path / "x.png")
    ds = DatasetLoader()
    ds.images, ds.src_texts, ds.tgt_texts = [str(tmp_path / "x.png")], ["s"], ["t"]
    img, s, t = ds[0]
    assert img.shape == (3, 256, 256) and img.dtype == torch.float32 and 0.0 <= float(img.min()) and float(img.max()) <= 1.0
    ref = torch.from_numpy(np.asarray(Image.fromarray(arr).convert("RGB").resize((256, 256))).transpose(2, 0, 1).copy()).float() / 255
    assert torch.equal(img, ref) and (s, t) == ("s", "t") and len(ds) == 1
    # klab extensions: the decoded image at its own size / the file's bytes (for GpuImageProcessor.from_decoded / .from_jpeg)
    from klab_multimodalmodel_amd.modules.loader import collate_decoded
    Image.fromarray(arr).save(tmp_path / "y.jpg", quality=90)
    ds.images = [str(tmp_path / "y.jpg")]
    ds.decode_only = True
    img, _s, _t = ds[0]
    assert img.dtype == torch.uint8 and tuple(img.shape) == (12, 10, 3)
    assert np.array_equal(img.numpy(), np.asarray(Image.open(tmp_path / "y.jpg").convert("RGB")))
    ds.file_bytes = True
    raw, _s, _t = ds[0]
    assert isinstance(raw, bytes) and raw == (tmp_path / "y.jpg").read_bytes()
    imgs, ss, ts = collate_decoded([ds[0], ds[0]])
    assert imgs == [raw, raw] and ss == ["s", "s"] and ts == ["t", "t"]


def test_host_tables_match_oracle():
    from klab_multimodalmodel_amd.engine import swin_cpb_tables, t5_bucket_table
    from oracle import swin_t5_oracle as O
    for Lq, Lk, bi in ((7, 7, False), (69, 69, True), (160, 160, True), (128, 128, False)):
        ctx = torch.arange(Lq)[:, None]
        mem = torch.arange(Lk)[None, :]
        assert torch.equal(t5_bucket_table(Lq, Lk, bi, 32, 128).long(), O.t5_relative_position_bucket(mem - ctx, bi, 32, 128))
    for w, pw in ((7, 0), (4, 0), (2, 0), (8, 6)):
        ct, ix = swin_cpb_tables(w, pw)
        rt, ri = O.swin_coords_table_and_index(w, pw, torch.float32)
        assert torch.equal(ct, rt.view(-1, 2)) and torch.equal(ix.long(), ri.view(-1))


def test_fused_adam_falls_back_to_torch_adam_for_foreign_parameters():
    """optim.FusedAdam is always safe to use: parameters that do not belong to a klab MyModel get torch.optim.Adam's update."""
    import torch
    from klab_multimodalmodel_amd.optim import FusedAdam
    torch.manual_seed(0)
    w0 = torch.randn(7, 5)
    a, b = torch.nn.Parameter(w0.clone()), torch.nn.Parameter(w0.clone())
    oa, ob = torch.optim.Adam([a], lr=1e-2, weight_decay=0.1), FusedAdam([b], lr=1e-2, weight_decay=0.1)
    for _ in range(3):
        for p, o in ((a, oa), (b, ob)):
            (p ** 2).sum().backward()
            o.step()
            o.zero_grad()
    assert ob._fallback is not None
    assert torch.equal(a.detach(), b.detach())


def test_async_checkpointer_roundtrip_and_reference_schema(built, tmp_path):
    """AsyncCheckpointer writes the reference's {'transformer', 'image_model'} schema (loadable by MyModel.load,
    ref/models/model.py:36-42) plus optimizer / scheduler / step; load_checkpoint restores all of it."""
    from klab_multimodalmodel_amd.checkpoint import AsyncCheckpointer, load_checkpoint
    from klab_multimodalmodel_amd.models.model import MyModel
    sw, t5 = _tiny()
    a = _args(True)
    a.result_dir = str(tmp_path)
    m = MyModel(a, _configs=(sw, t5, t5), _seed=1)
    opt = torch.optim.Adam(m.transformer.parameters(), lr=1e-3)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=1, gamma=0.5)
    for p in m.transformer.parameters():
        p.grad = torch.ones_like(p)
    opt.step(); sched.step()
    ck = AsyncCheckpointer(str(tmp_path))
    path = ck.save(m, opt, sched, step=7, name="ck.pth")
    ck.wait()
    assert os.path.exists(path) and not os.path.exists(path + ".tmp")
    raw = torch.load(path, weights_only=False)
    assert set(raw) >= {"transformer", "image_model", "optimizer", "scheduler", "step"} and raw["step"] == 7
    m2 = MyModel(a, _configs=(sw, t5, t5), _seed=2)
    m2.load("ck.pth")  # the reference-schema loader reads the same file
    for (k, x), (_, y) in zip(m.transformer.state_dict().items(), m2.transformer.state_dict().items()):
        assert torch.equal(x, y), k
    m3 = MyModel(a, _configs=(sw, t5, t5), _seed=3)
    opt3 = torch.optim.Adam(m3.transformer.parameters(), lr=1e-3)
    sched3 = torch.optim.lr_scheduler.StepLR(opt3, step_size=1, gamma=0.5)
    assert load_checkpoint(path, m3, opt3, sched3) == 7
    assert opt3.param_groups[0]["lr"] == opt.param_groups[0]["lr"] == 5e-4
    s, s3 = opt.state_dict()["state"], opt3.state_dict()["state"]
    assert s.keys() == s3.keys() and all(torch.equal(s[k]["exp_avg"], s3[k]["exp_avg"]) for k in s)
    for (k, x), (_, y) in zip(m.image_model.state_dict().items(), m3.image_model.state_dict().items()):
        assert torch.equal(x, y), k


def test_image_processor_from_pretrained_never_silently_defaults(tmp_path):
    """ref/train.py:39 `AutoImageProcessor.from_pretrained(args.image_model_name)`: a hub name without a local config must not
    yield bare ViTImageProcessor defaults (224 / bilinear / mean 0.5) for a 256 / bicubic / ImageNet checkpoint."""
    from klab_multimodalmodel_amd.modules.image_pipeline import BICUBIC, GpuImageProcessor
    with pytest.raises(OSError):
        GpuImageProcessor.from_pretrained("someone/unknown-checkpoint", device="cpu")
    with pytest.raises(OSError):
        GpuImageProcessor.from_pretrained(str(tmp_path), device="cpu")  # a directory without preprocessor_config.json
    with pytest.raises(OSError):
        GpuImageProcessor.from_pretrained()
    p = GpuImageProcessor.from_pretrained("microsoft/swinv2-base-patch4-window8-256", device="cpu")  # the reference's default name
    assert p.size == 256 and p.resample == BICUBIC and abs(p.mean[0] - 0.485) < 1e-9 and abs(p.std[2] - 0.225) < 1e-9
    with open(os.path.join(str(tmp_path), "preprocessor_config.json"), "w") as f:
        json.dump({"size": {"height": 192, "width": 192}, "resample": 3, "image_mean": [0.1, 0.2, 0.3], "image_std": [0.5, 0.5, 0.5],
                   "rescale_factor": 0.00392156862745098, "image_processor_type": "ViTImageProcessor"}, f)
    q = GpuImageProcessor.from_pretrained(str(tmp_path), device="cpu")
    assert q.size == 192 and q.resample == 3 and q.mean == (0.1, 0.2, 0.3)
    r = GpuImageProcessor.from_pretrained(size=224, device="cpu")  # explicit settings only
    assert r.size == 224


def test_bench_workloads_follow_the_survey_table():
    """bench.py's workloads = BASELINE.json configs[1..4] as SURVEY §8(d) resolves them (shapes, per-GPU batch, fwd+bwd
    GFLOP/sample), and the span-mask batch has the grammar of the reference's RedCaps loader after tokenisation
    (ref/modules/loader.py:56-72): sentinels 32099 - k in increasing k in the source, <extra_id_0> w.. <extra_id_k> </s> in the
    target, zero padding, </s> closing every source row."""
    import bench
    w = bench.WORKLOADS
    assert (w["caption"]["B"], w["caption"]["Ls"], w["caption"]["Lt"], w["caption"]["gflop"]) == (64, 9, 64, 27.26)
    assert (w["cfg3"]["B"], w["cfg3"]["gflop"], w["cfg3"]["train_swin"]) == (32, 137.27, True)
    assert (w["spanmask"]["Ls"], w["spanmask"]["Lt"], w["spanmask"]["gflop"]) == (32, 16, 118.90)
    assert w["cfg5"]["swin"]["window_size"] == 24 and w["cfg5"]["swin"]["image_size"] == 384 and w["cfg5"]["gflop"] == 817.26
    for name in w:
        sw, t5 = bench.workload_configs(name)
        assert sw.hidden_size == t5.d_model, name  # no projection between the towers (ref/models/model.py:23)
    pix, src, tgt = bench.synth_spanmask_batch(8, 32, 16, 8, 32128, "cpu", seed=3)
    assert pix.shape == (8, 3, 8, 8) and src.shape == (8, 32) and tgt.shape == (8, 16)
    for b in range(8):
        s_sent = [int(x) for x in src[b] if int(x) >= 32000]
        t_sent = [int(x) for x in tgt[b] if int(x) >= 32000]
        assert s_sent == [32099 - k for k in range(4)] and t_sent == [32099 - k for k in range(5)]
        row = src[b].tolist()
        end = row.index(1)
        assert all(x == 0 for x in row[end + 1:]) and int(tgt[b, 0]) == 32099
        trow = tgt[b].tolist()
        assert 1 in trow and all(x == 0 for x in trow[trow.index(1) + 1:])


def test_bench_refuses_to_measure_fewer_ranks_than_asked_for():
    """`python bench.py --gpus 8` without a launcher and without 8 visible GPUs must fail, never print an n_gpus: 1 line."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "KLAB_BENCH_DEVICE")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"], cwd=root, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
