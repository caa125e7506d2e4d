"""One-GPU data point for BASELINE configs[2]/[3] as SURVEY §8(d) resolves them: Swin-V2 C=96 (2,2,18,2) heads (3,6,12,24)
224 w7, UNFROZEN, + T5-base; per-GPU batch 32, Ls=9, Lt=64, bf16, fwd+bwd+FusedAdam.  Not the headline metric (bench.py)."""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    dev = torch.device("cuda:0")
    from klab_multimodalmodel_amd.engine import SwinConfig, T5Config
    from klab_multimodalmodel_amd.models.model import MyModel
    from klab_multimodalmodel_amd.optim import FusedAdam
    sw = SwinConfig(image_size=224, embed_dim=96, depths=(2, 2, 18, 2), num_heads=(3, 6, 12, 24), window_size=7)
    t5 = T5Config(d_model=768, d_ff=3072, num_heads=12, num_layers=12, num_decoder_layers=12)
    args = types.SimpleNamespace(result_dir="/tmp", language_model_name="-", image_model_name="-", image_model_train=True,
                                 transformer_model_name="-")
    model = MyModel(args, _configs=(sw, t5, t5), _seed=0, dtype="bf16").to(dev)
    model._direct_grads = True
    opt = FusedAdam(model.transformer.parameters(), lr=1e-4)
    model.transformer.train()
    pix, src, tgt = bench.synth_batch(B, 9, 64, 224, 32128, dev, seed=1)
    images, se, te = {"pixel_values": pix}, {"input_ids": src}, {"input_ids": tgt}

    def step():
        loss = model(images, se, te)
        loss.backward()
        opt.step()
        opt.zero_grad()
        return loss

    for _ in range(4):
        step()
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"config-3 architecture, B={B}: {dt * 1e3:.1f} ms/step => {B / dt:.0f} samples/s (137.27 GFLOP/sample => "
          f"{B * 137.27 / dt / 1e3:.0f} TFLOP/s), loss {float(loss):.3f}")


if __name__ == "__main__":
    main()
