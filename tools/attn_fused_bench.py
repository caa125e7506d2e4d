#!/usr/bin/env python3
"""klab_t5_attn_fused_fwd against the three launches it replaces (T5-small, B = 64): device time per call from HIP events."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from klab_multimodalmodel_amd import ops  # noqa: E402


def timeit(fn, n=50):
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
    B, H, dk, d = 64, 8, 64, 512
    inner = H * dk
    for name, Lq, Lk, cross, causal in (("dec self", 64, 64, False, True), ("enc self", 58, 58, False, False), ("dec cross", 64, 58, True, False)):
        x = torch.randn(B * Lq, d, device="cuda")
        gamma = torch.ones(d, device="cuda")
        nproj = inner if cross else 3 * inner
        w = (torch.randn(nproj, d, device="cuda") * d ** -0.5).to(dt)
        bias = None if cross else torch.randn(H, Lq, Lk, device="cuda")
        kv = torch.randn(B * Lk, 2 * inner, device="cuda").to(dt) if cross else None
        sd = torch.tensor([5], dtype=torch.int32, device="cuda")
        xn = torch.empty(B * Lq, d, device="cuda", dtype=dt)
        rs = torch.empty(B * Lq, device="cuda")
        pr = torch.empty(B * Lq, nproj, device="cuda", dtype=dt)
        ctx = torch.empty(B * Lq, inner, device="cuda", dtype=dt)
        lse = torch.empty(B, H, Lq, device="cuda")
        kw = dict(B=B, H=H, Lq=Lq, Lk=Lk, dk=dk, bias=bias, causal=causal, drop_p=0.1, seed=sd, tag=3)

        def three():
            ops.rmsnorm_fwd(x, gamma, y=xn, rstd=rs)
            ops.gemm(xn, w, pr, M=B * Lq, N=nproj, K=d)
            if cross:
                ops.t5_attn_fwd(pr, kv[:, :inner], kv[:, inner:], ctx, lse, ldq=inner, ldk=2 * inner, ldv=2 * inner, **kw)
            else:
                ops.t5_attn_fwd(pr, pr[:, inner:], pr[:, 2 * inner:], ctx, lse, ldq=3 * inner, ldk=3 * inner, ldv=3 * inner, **kw)

        def fused():
            ops.t5_attn_fused_fwd(x, gamma, w, xn, rs, pr, ctx, lse, cross=cross, k=kv[:, :inner] if cross else None,
                                  v=kv[:, inner:] if cross else None, ldk=2 * inner if cross else None, ldv=2 * inner if cross else None, **kw)

        res = {"three": [], "fused": []}
        for _ in range(3):
            res["three"].append(timeit(three))
            res["fused"].append(timeit(fused))
        print(f"{name:10s} three launches {sorted(res['three'])[1]:7.1f} us   fused {sorted(res['fused'])[1]:7.1f} us", flush=True)


if __name__ == "__main__":
    main()
