"""Host-side cost of one training step, phase by phase (no device waits inside a phase): shows whether the Python /
launch path or the GPU bounds the step.  python tools/host_phase_bench.py [steps]"""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    dev = torch.device("cuda:0")
    from klab_multimodalmodel_amd.models.model import MyModel
    sw, t5 = bench.cfg2_configs()
    args = types.SimpleNamespace(result_dir="/tmp", language_model_name="t5-small", image_model_name="swinv2-C64-224-w7",
                                 image_model_train=False, transformer_model_name="t5-small")
    model = MyModel(args, _configs=(sw, t5, t5), _seed=0, dtype="bf16").to(dev)
    model._direct_grads = True
    opt = torch.optim.Adam(model.transformer.parameters(), lr=1e-3, fused=True)
    model.transformer.train()
    pix, src, tgt = bench.synth_batch(64, 9, 64, 224, 32128, dev, seed=1)
    images, se, te = {"pixel_values": pix}, {"input_ids": src}, {"input_ids": tgt}
    acc = {"fwd": 0.0, "bwd": 0.0, "adam": 0.0, "zero": 0.0}
    for it in range(steps + 5):
        if it == 5:
            torch.cuda.synchronize()
            acc = {k: 0.0 for k in acc}
            t_all = time.perf_counter()
        t = time.perf_counter(); loss = model(images, se, te); acc["fwd"] += time.perf_counter() - t
        t = time.perf_counter(); loss.backward(); acc["bwd"] += time.perf_counter() - t
        t = time.perf_counter(); opt.step(); acc["adam"] += time.perf_counter() - t
        t = time.perf_counter(); opt.zero_grad(); acc["zero"] += time.perf_counter() - t
    host = time.perf_counter() - t_all
    torch.cuda.synchronize()
    tot = time.perf_counter() - t_all
    print({k: round(v / steps * 1e3, 3) for k, v in acc.items()}, "host ms/step", round(host / steps * 1e3, 3), "total ms/step", round(tot / steps * 1e3, 3))


if __name__ == "__main__":
    main()
