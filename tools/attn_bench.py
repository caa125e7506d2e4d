#!/usr/bin/env python3
"""Timing of the T5 / Swin attention kernels at BASELINE configs[1] shapes (B=64)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from klab_multimodalmodel_amd import ops  # noqa: E402


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
    dt = torch.bfloat16
    B, H, dk = 64, 8, 64
    inner = H * dk
    for name, Lq, Lk, causal, use_bias in (("enc self", 58, 58, False, True), ("dec self", 64, 64, True, True), ("cross", 64, 58, False, False)):
        q = torch.randn(B * Lq, 3 * inner, device="cuda").to(dt)
        kv = torch.randn(B * Lk, 2 * inner, device="cuda").to(dt)
        bias = torch.randn(H, Lq, Lk, device="cuda") if use_bias else None
        ctx = torch.empty(B * Lq, inner, device="cuda", dtype=dt)
        lse = torch.empty(B, H, Lq, device="cuda")
        dctx = torch.randn(B * Lq, inner, device="cuda").to(dt)
        dq = torch.empty(B * Lq, 3 * inner, device="cuda", dtype=dt)
        dkv = torch.empty(B * Lk, 2 * inner, device="cuda", dtype=dt)
        dbias = torch.zeros(H, Lq, Lk, device="cuda") if use_bias else None
        sd = torch.tensor([1], dtype=torch.int32, device="cuda")
        kw = dict(B=B, H=H, Lq=Lq, Lk=Lk, dk=dk, bias=bias, causal=causal, drop_p=0.1, seed=sd, tag=3)
        lds = dict(ldq=3 * inner, ldk=2 * inner, ldv=2 * inner)
        f = timeit(lambda: ops.t5_attn_fwd(q, kv[:, :inner], kv[:, inner:], ctx, lse, **lds, **kw))
        b1 = timeit(lambda: ops.t5_attn_bwd(q, kv[:, :inner], kv[:, inner:], ctx, lse, dctx, dq, dkv[:, :inner], dkv[:, inner:], dbias=dbias,
                                            lddq=3 * inner, lddk=2 * inner, lddv=2 * inner, **lds, **kw))
        b0 = timeit(lambda: ops.t5_attn_bwd(q, kv[:, :inner], kv[:, inner:], ctx, lse, dctx, dq, dkv[:, :inner], dkv[:, inner:], dbias=None,
                                            lddq=3 * inner, lddk=2 * inner, lddv=2 * inner, **lds, **kw))
        print(f"t5 {name:9s} fwd {f:7.1f} us   bwd {b1:7.1f} us   bwd(no dbias) {b0:7.1f} us")
    for name, R, C, Hh in (("swin st0", 56, 64, 2), ("swin st1", 28, 128, 4), ("swin st2", 14, 256, 8), ("swin st3", 7, 512, 16)):
        w = 7
        n = w * w
        qkv = torch.randn(B * R * R, 3 * C, device="cuda").to(dt)
        ctx = torch.empty(B * R * R, C, device="cuda", dtype=dt)
        bias = torch.randn(Hh, n, n, device="cuda")
        ls = torch.full((Hh,), 2.3, device="cuda")
        for shift in (0, 3 if R > w else 0):
            t = timeit(lambda: ops.swin_attn_fwd(qkv, ctx, bias, ls, None, B=B, R=R, w=w, shift=shift, H=Hh, C=C))
            print(f"{name} shift {shift}: fwd {t:7.1f} us")


if __name__ == "__main__":
    main()
