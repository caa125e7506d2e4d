"""Greedy-decoding throughput of MyModel.generate at BASELINE configs[1] shapes (B=64, max_length 20): captions/s."""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench


def main():
    dev = torch.device("cuda:0")
    from klab_multimodalmodel_amd.models.model import MyModel
    sw, t5 = bench.cfg2_configs()
    args = types.SimpleNamespace(result_dir="/tmp", language_model_name="t5-small", image_model_name="swinv2-C64-224-w7",
                                 image_model_train=False, transformer_model_name="t5-small")
    model = MyModel(args, _configs=(sw, t5, t5), _seed=0, dtype="bf16").to(dev)
    pix, src, _tgt = bench.synth_batch(64, 9, 64, 224, 32128, dev, seed=1)
    images, se = {"pixel_values": pix}, {"input_ids": src}
    for _ in range(2):
        out = model(images, se, None, return_loss=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        out = model(images, se, None, return_loss=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"generate: {tuple(out.shape)} tokens in {dt * 1e3:.1f} ms => {64 / dt:.0f} captions/s")


if __name__ == "__main__":
    main()
