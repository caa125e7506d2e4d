#!/usr/bin/env python3
"""Prototype pricing of the chunked LM head + cross-entropy (HF/t5:1044-1054): per chunk of R token rows
    logits = h W^T (bf16)  ->  CE in place (d logits)  ->  dX rows = d logits @ W  ->  dW += d logits^T h
so that a chunk's logits (R x 32128 bf16 = 33 MB at R = 512) are produced and consumed while they are still in the 256 MiB
Infinity Cache, against the whole-tensor form the engine runs (263 MB of logits written and re-read three times).
Device time of the four launches per chunk from HIP events, same process, interleaved; configs[1] sizes (B*Lt = 4096, d = 512)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from klab_multimodalmodel_amd import ops  # noqa: E402

M, V, D = 4096, 32128, 512


def run(chunk, h, W, labels, logits, dX, dW, inv_n, loss_row, loss):
    for r0 in range(0, M, chunk):
        r1 = min(M, r0 + chunk)
        lg = logits[r0:r1]
        ops.gemm(h[r0:r1], W, lg, M=r1 - r0, N=V, K=D, alpha=D ** -0.5, name_tag=1 if chunk == M else 0)
        ops.ce_fwd(lg, labels[r0:r1], inv_n, loss_row[r0:r1], loss, write_grad=True)
        ops.gemm(lg, W, dX[r0:r1], M=r1 - r0, N=D, K=V, b_kmajor=False, alpha=D ** -0.5, accumulate=True, atomic_ok=True)
        ops.gemm(lg, h[r0:r1], dW, M=V, N=D, K=r1 - r0, a_kmajor=False, b_kmajor=False, alpha=D ** -0.5, accumulate=True, atomic_ok=True)


def main():
    g = torch.Generator().manual_seed(0)
    h = torch.randn(M, D, generator=g).cuda().bfloat16()
    W = (torch.randn(V, D, generator=g) * 0.05).cuda().bfloat16()
    labels = torch.randint(2, 32000, (M,), generator=g).cuda()
    logits = torch.empty(M, V, device="cuda", dtype=torch.bfloat16)
    dX = torch.zeros(M, D, device="cuda")
    dW = torch.zeros(V, D, device="cuda")
    inv_n = torch.full((1,), 1.0 / M, device="cuda")
    loss_row = torch.zeros(M, device="cuda")
    loss = torch.zeros(1, device="cuda")
    res = {}
    for rnd in range(3):
        for chunk in (M, 2048, 1024, 512, 256):
            dX.zero_(); dW.zero_()
            for _ in range(2):
                run(chunk, h, W, labels, logits, dX, dW, inv_n, loss_row, loss)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 5
            e0.record()
            for _ in range(n):
                run(chunk, h, W, labels, logits, dX, dW, inv_n, loss_row, loss)
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(chunk, []).append(e0.elapsed_time(e1) / n * 1e3)
    for chunk, v in res.items():
        v.sort()
        print(f"rows per chunk {chunk:5d} ({M // chunk:2d} x 4 launches): LM head fwd + CE + dgrad + wgrad = {v[len(v) // 2]:8.1f} us (min {v[0]:.1f})", flush=True)


if __name__ == "__main__":
    main()
