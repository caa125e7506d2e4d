#!/usr/bin/env python3
"""klab_gemm_fp8 on the forward Linear shapes of BASELINE configs[4] (T5-large widths, M = 32 x 153 encoder tokens): us and TFLOP/s
per launch, next to klab_gemm (bf16) on the same shapes.  KLAB_FP8_SCALED=0 selects the non-scaled fp8 kernels (A/B)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from klab_multimodalmodel_amd import ops  # noqa: E402

SHAPES = [("L enc qkv", 4896, 3072, 1024), ("L enc o", 4896, 1024, 1024), ("L enc wi", 4896, 4096, 1024), ("L enc wo", 4896, 1024, 4096),
          ("L dec wi", 2048, 4096, 1024), ("L lm head", 2048, 32128, 1024), ("swin2 fc1", 18432, 2048, 512), ("swin2 fc2", 18432, 512, 2048),
          ("sq 4096 k4096", 4096, 4096, 4096)]


def timeit(fn, n=20):
    for _ in range(3):
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
    for name, M, N, K in SHAPES:
        A = torch.randn(M, K, device="cuda").bfloat16()
        B = torch.randn(N, K, device="cuda").bfloat16()
        C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        A8, sa = ops.quant_fp8_rows(A)
        B8, sb = ops.quant_fp8_rows(B)
        t8 = timeit(lambda: ops.gemm_fp8(A8, sa, B8, sb, C))
        t16 = timeit(lambda: ops.gemm(A, B, C, M=M, N=N, K=K))
        fl = 2.0 * M * N * K
        print(f"{name:14s} M={M:6d} N={N:6d} K={K:5d}  fp8 {t8:8.1f} us {fl / t8 / 1e6:7.1f} TF/s   bf16 {t16:8.1f} us {fl / t16 / 1e6:7.1f} TF/s", flush=True)


if __name__ == "__main__":
    main()
