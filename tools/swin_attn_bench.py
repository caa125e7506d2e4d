#!/usr/bin/env python3
"""Timing of the large-window Swin attention kernels at the BASELINE configs[4] stage shapes (B = 8):
forward / backward, matrix-core streaming form vs vector-ALU tiled form, with and without the bias-table gradient."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from klab_multimodalmodel_amd import ops  # noqa: E402


def timeit(fn, n=10):
    for _ in range(2):
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
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    for R, w, shift, H, C in ((96, 24, 12, 4, 128), (48, 24, 12, 8, 256), (24, 24, 0, 16, 512), (12, 12, 0, 32, 1024)):
        n, ntab, nW = w * w, (2 * w - 1) ** 2, (R // w) ** 2
        M = B * R * R
        qkv = torch.randn(M, 3 * C, device="cuda").bfloat16()
        btab = (16 * torch.sigmoid(torch.randn(ntab, H))).cuda()
        ls = torch.full((H,), 2.3, device="cuda")
        ctx = torch.empty(M, C, device="cuda", dtype=torch.bfloat16)
        lse = torch.empty(B * nW * H * n, device="cuda")
        dctx = torch.randn(M, C, device="cuda").bfloat16()
        dqkv = torch.empty_like(qkv)
        dls = torch.zeros(H, device="cuda")
        dtab = torch.zeros(ntab, H, device="cuda")
        kw = dict(B=B, R=R, w=w, shift=shift, H=H, C=C)
        flops = 4.0 * n * n * (C // H) * B * nW * H
        for mf in (True, False):
            tf = timeit(lambda: ops.swin_attn_fwd(qkv, ctx, None, ls, lse, bias_table=btab, mfma=mf, **kw))
            tb = timeit(lambda: ops.swin_attn_bwd(qkv, ctx, None, ls, lse, dctx, dqkv, None, dls, bias_table=btab, dbias_table=dtab, mfma=mf, **kw))
            tb0 = timeit(lambda: ops.swin_attn_bwd(qkv, ctx, None, ls, lse, dctx, dqkv, None, dls, bias_table=btab, dbias_table=None, mfma=mf, **kw))
            print(f"R={R:3d} w={w} H={H:2d} C={C:4d} n={n} {'mfma' if mf else 'valu'}: fwd {tf:8.1f} us ({flops / tf / 1e6:6.1f} TF/s)  "
                  f"bwd {tb:8.1f} us  bwd without d(table) {tb0:8.1f} us")


if __name__ == "__main__":
    main()
