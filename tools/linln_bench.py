#!/usr/bin/env python3
"""klab_swin_linear_ln_fused against klab_gemm + klab_layernorm_fwd (Swin stage 2 of the caption tower: 12 544 rows, C = 256):
device time per call from HIP events."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from klab_multimodalmodel_amd import ops  # noqa: E402


def timeit(fn, n=100):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    dt = torch.bfloat16
    Cc = 256
    for M, K in ((12544, 256), (12544, 1024), (50176, 256), (50176, 1024)):
        x = torch.randn(M, K, device="cuda").to(dt)
        w = (torch.randn(Cc, K, device="cuda") * K ** -0.5).to(dt)
        bias, gamma, beta = torch.randn(Cc, device="cuda"), torch.ones(Cc, device="cuda"), torch.zeros(Cc, device="cuda")
        sc = torch.randn(M, Cc, device="cuda")
        y = torch.empty(M, Cc, device="cuda", dtype=dt)
        out = torch.empty(M, Cc, device="cuda")
        outt = torch.empty(M, Cc, device="cuda", dtype=dt)
        mean, rstd = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")

        def two():
            ops.gemm(x, w, y, M=M, N=Cc, K=K, bias=bias)
            ops.layernorm_fwd(y, gamma, beta, shortcut=sc, out=out, outt=outt, mean=mean, rstd=rstd)

        def gemm_only():
            ops.gemm(x, w, y, M=M, N=Cc, K=K, bias=bias)

        def fused():
            ops.swin_linear_ln_fused(x, sc, w, bias, gamma, beta, out, outt)

        res = {"two": [], "gemm": [], "fused": []}
        for _ in range(3):
            res["two"].append(timeit(two))
            res["gemm"].append(timeit(gemm_only))
            res["fused"].append(timeit(fused))
        med = {k: sorted(v)[1] for k, v in res.items()}
        alg = (M * K * 2 + Cc * K * 2 + M * Cc * (4 + 4 + 2)) / 1e3  # KB: x, W, shortcut in, out f32 + bf16
        print(f"M={M:6d} K={K:5d}  gemm+LN {med['two']:7.1f} us  (gemm alone {med['gemm']:6.1f})   fused {med['fused']:7.1f} us"
              f"   = {alg / med['fused'] / 1e3:5.2f} TB/s algorithmic, {2.0 * M * K * Cc / med['fused'] / 1e6:6.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
