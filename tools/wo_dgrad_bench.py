#!/usr/bin/env python3
"""The FFN's wo input gradient (dh = (dy @ Wo) * relu-mask, M x 2048 outputs over K = 512) with and without its mask epilogue, next to
the wi forward of the same size: device time per call from HIP events."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from klab_multimodalmodel_amd import ops, _lib as L  # noqa: E402


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
    M, ff, d = 4096, 2048, 512
    dy = torch.randn(M, d, device="cuda").to(dt)
    wo = (torch.randn(d, ff, device="cuda") * 0.03).to(dt)   # [d, ff]: the dgrad contracts over its rows (m-major B)
    wi = (torch.randn(ff, d, device="cuda") * 0.03).to(dt)   # [ff, d]: K-major B of the forward
    x = torch.randn(M, d, device="cuda").to(dt)
    h = torch.relu(torch.randn(M, ff, device="cuda")).to(dt)
    out = torch.empty(M, ff, device="cuda", dtype=dt)
    r = {}
    r["wi fwd (K-major B, relu)"] = lambda: ops.gemm(x, wi, out, M=M, N=ff, K=d, act=L.ACT_RELU)
    r["wi fwd plain"] = lambda: ops.gemm(x, wi, out, M=M, N=ff, K=d)
    r["wo dgrad plain (m-major B)"] = lambda: ops.gemm(dy, wo, out, M=M, N=ff, K=d, b_kmajor=False)
    r["wo dgrad + relu mask (aux)"] = lambda: ops.gemm(dy, wo, out, M=M, N=ff, K=d, b_kmajor=False, aux=h, aux_mode=L.AUX_NONZERO, aux_scale=1.0 / 0.9)
    for _ in range(2):
        for k, f in r.items():
            print(f"{k:34s} {timeit(f):7.1f} us", flush=True)


if __name__ == "__main__":
    main()
