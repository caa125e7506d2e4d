"""Pin the CPU oracle against outputs of the reference itself (tests/golden/*.npz)."""
import pytest
import torch

from oracle import swin_t5_oracle as O
from tests.helpers import load_golden, rel_l2


@pytest.mark.parametrize("name", ["tiny_a", "tiny_b", "tiny_c", "tiny_d"])  # c, d: window padding (HF/swinv2:645-650), d with n = 100
def test_oracle_matches_reference_loss_acts_grads(name):
    g = load_golden(name)
    sds = {m: {k: v.clone().requires_grad_(v.is_floating_point() and m != "lang") for k, v in sd.items()}
           for m, sd in g["sds"].items()}
    loss, parts = O.mymodel_forward(sds["swin"], sds["lang"], sds["main"], g["swin_cfg"], g["t5_cfg"], g["t5_cfg"],
                                    training=False, image_model_train=True, return_parts=True, **g["inputs"])
    assert abs(float(loss.detach()) - g["loss"]) <= 1e-5 * abs(g["loss"])
    for k in ("image_embeddings", "language_embeddings", "encoder_out", "decoder_out"):
        assert rel_l2(parts[k].detach(), g["acts"][k]) < 2e-6, k
    loss.backward()
    worst = 0.0
    for m in ("main", "swin"):
        assert set(g["grads"][m]) <= set(sds[m]) | {"encoder.embed_tokens.weight", "decoder.embed_tokens.weight", "lm_head.weight"}
        for k, ref in g["grads"][m].items():
            if k in ("encoder.embed_tokens.weight", "decoder.embed_tokens.weight", "lm_head.weight"):
                continue
            got = sds[m][k].grad
            assert got is not None, (m, k)
            if float(ref.abs().max()) == 0.0:
                assert float(got.abs().max()) < 1e-12, (m, k)
                continue
            e = rel_l2(got, ref)
            worst = max(worst, e)
            assert e < 1e-4, (m, k, e)
    print(name, "worst grad rel-L2", worst)


def test_oracle_fp64_close_to_fp32():
    g = load_golden("tiny_a", dtype=torch.float64)
    loss = O.mymodel_forward(g["sds"]["swin"], g["sds"]["lang"], g["sds"]["main"], g["swin_cfg"], g["t5_cfg"],
                             g["t5_cfg"], **g["inputs"])
    assert abs(float(loss) - g["loss"]) < 1e-5


def test_rel_bucket_known_values():
    # spot values of HF/t5:216-262 (bidirectional, 32 buckets, max 128)
    rel = torch.tensor([[-200, -20, -8, -1, 0, 1, 7, 8, 20, 127, 128, 500]])
    b = O.t5_relative_position_bucket(rel, True, 32, 128)
    assert b.tolist() == [[15, 10, 8, 1, 0, 17, 23, 24, 26, 31, 31, 31]]
    u = O.t5_relative_position_bucket(rel, False, 32, 128)
    assert u.tolist() == [[31, 17, 8, 1, 0, 0, 0, 0, 0, 0, 0, 0]]


def test_shift_right():
    cfg = O.T5Cfg()
    lab = torch.tensor([[5, 6, -100, 1]])
    assert O.t5_shift_right(lab, cfg).tolist() == [[0, 5, 6, 0]]
