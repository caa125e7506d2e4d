#!/usr/bin/env python3
"""Per-shape timing of klab_gemm on the shapes of BASELINE configs[1] (B=64): TFLOP/s per launch."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from klab_multimodalmodel_amd import ops  # noqa: E402

SHAPES = [  # name, M, N, K, a_kmajor, b_kmajor, atomic
    ("lmhead fwd", 4096, 32128, 512, True, True, False),
    ("lmhead dgrad", 4096, 512, 32128, True, False, True),
    ("lmhead wgrad", 32128, 512, 4096, False, False, True),
    ("qkv fwd", 4096, 1536, 512, True, True, False),
    ("qkv fwd enc", 3712, 1536, 512, True, True, False),
    ("o fwd enc", 3712, 512, 512, True, True, False),
    ("wi fwd enc", 3712, 2048, 512, True, True, False),
    ("wo fwd enc", 3712, 512, 2048, True, True, False),
    ("o dgrad", 4096, 512, 512, True, False, False),
    ("o fwd", 4096, 512, 512, True, True, False),
    ("wi fwd", 4096, 2048, 512, True, True, False),
    ("wo fwd", 4096, 512, 2048, True, True, False),
    ("kv_all fwd", 3712, 6144, 512, True, True, False),
    ("qkv dgrad", 4096, 512, 1536, True, False, False),
    ("wi dgrad", 4096, 512, 2048, True, False, False),
    ("wo dgrad", 4096, 2048, 512, True, False, False),
    ("o wgrad", 512, 512, 4096, False, False, True),
    ("qkv wgrad", 1536, 512, 4096, False, False, True),
    ("wi wgrad", 2048, 512, 4096, False, False, True),
    ("wo wgrad", 512, 2048, 4096, False, False, True),
    ("swin0 qkv", 200704, 192, 64, True, True, False),
    ("swin0 proj", 200704, 64, 64, True, True, False),
    ("swin0 fc1", 200704, 256, 64, True, True, False),
    ("swin0 fc2", 200704, 64, 256, True, True, False),
    ("swin2 qkv", 12544, 768, 256, True, True, False),
    ("swin2 fc1", 12544, 1024, 256, True, True, False),
    ("swin2 fc2", 12544, 256, 1024, True, True, False),
]


LARGE = [  # BASELINE configs[4] (T5-large d=1024, ff=4096, inner=1024; encoder 32 x 153 tokens, decoder 32 x 64)
    ("L enc qkv fwd", 4896, 3072, 1024, True, True, False),
    ("L enc o fwd", 4896, 1024, 1024, True, True, False),
    ("L enc wi fwd", 4896, 4096, 1024, True, True, False),
    ("L enc wo fwd", 4896, 1024, 4096, True, True, False),
    ("L enc qkv dgrad", 4896, 1024, 3072, True, False, False),
    ("L enc o dgrad", 4896, 1024, 1024, True, False, False),
    ("L enc wi dgrad", 4896, 1024, 4096, True, False, False),
    ("L enc wo dgrad", 4896, 4096, 1024, True, False, False),
    ("L dec wi fwd", 2048, 4096, 1024, True, True, False),
    ("L dec wi dgrad", 2048, 1024, 4096, True, False, False),
    ("L dec wo dgrad", 2048, 4096, 1024, True, False, False),
    ("L enc wi wgrad", 4096, 1024, 4896, False, False, True),
    ("L enc wo wgrad", 1024, 4096, 4896, False, False, True),
    ("L swin0 qkv fwd", 294912, 384, 128, True, True, False),
    ("L swin0 qkv dgrad", 294912, 128, 384, True, False, False),
    ("L swin0 fc1 fwd", 294912, 512, 128, True, True, False),
    ("L swin0 fc2 dgrad", 294912, 512, 128, True, False, False),
    ("L swin2 qkv fwd", 18432, 1536, 512, True, True, False),
    ("L swin2 fc1 fwd", 18432, 2048, 512, True, True, False),
    ("L swin2 fc1 dgrad", 18432, 512, 2048, True, False, False),
    ("L swin2 fc2 dgrad", 18432, 2048, 512, True, False, False),
]


CFG3 = [  # BASELINE configs[2] trainable Swin-V2 (C=96, B=32): stage 0 = 100352 tokens, stage 1 = 25088, stage 2 = 6272
    ("S0 qkv fwd", 100352, 288, 96, True, True, False),
    ("S0 proj fwd", 100352, 96, 96, True, True, False),
    ("S0 fc1 fwd", 100352, 384, 96, True, True, False),
    ("S0 fc2 fwd", 100352, 96, 384, True, True, False),
    ("S0 qkv dgrad", 100352, 96, 288, True, False, False),
    ("S0 proj dgrad", 100352, 96, 96, True, False, False),
    ("S0 fc1 dgrad", 100352, 96, 384, True, False, False),
    ("S0 fc2 dgrad", 100352, 384, 96, True, False, False),
    ("S1 qkv fwd", 25088, 576, 192, True, True, False),
    ("S1 fc1 fwd", 25088, 768, 192, True, True, False),
    ("S1 fc2 fwd", 25088, 192, 768, True, True, False),
    ("S1 fc1 dgrad", 25088, 192, 768, True, False, False),
    ("S1 fc2 dgrad", 25088, 768, 192, True, False, False),
    ("S2 qkv fwd", 6272, 1152, 384, True, True, False),
    ("S2 proj fwd", 6272, 384, 384, True, True, False),
    ("S2 fc1 fwd", 6272, 1536, 384, True, True, False),
    ("S2 fc2 fwd", 6272, 384, 1536, True, True, False),
    ("S2 qkv dgrad", 6272, 384, 1152, True, False, False),
    ("S2 fc1 dgrad", 6272, 384, 1536, True, False, False),
    ("S2 fc2 dgrad", 6272, 1536, 384, True, False, False),
    ("S3 fc1 fwd", 1568, 3072, 768, True, True, False),
    ("S3 fc2 dgrad", 1568, 3072, 768, True, False, False),
]


def main():
    dt = torch.bfloat16
    tot = 0.0
    flt = sys.argv[1:]
    shapes = SHAPES
    if flt and flt[0] == "--large":
        shapes, flt = LARGE, flt[1:]
    elif flt and flt[0] == "--cfg3":
        shapes, flt = CFG3, flt[1:]
    elif flt and flt[0] == "--sq":  # 4096 x 4096 outputs (256 tiles of 256 x 256) over K: per-k-tile slope and fixed cost of a kernel
        shapes, flt = [(f"sq k{K} {n}", 4096, 4096, K, ak, bk, False) for ak, bk, n in ((True, True, "tt"), (True, False, "tf"), (False, False, "ff"))
                       for K in (128, 512, 1024, 2048, 4096, 8192)], flt[1:]
    elif flt and flt[0] == "--ksweep":  # fixed cost of a launch: the same output tile grid at K = 64 ... 4096
        shapes, flt = [(f"k{K} {n}", 4096, N, K, True, bk, False) for N in (512, 2048) for bk, n in ((True, f"N{N} fwd"), (False, f"N{N} dgrad"))
                       for K in (64, 128, 256, 512, 1024, 2048, 4096)], flt[1:]
    for name, M, N, K, ak, bk, atomic in shapes:
        if flt and not any(f in name for f in flt):
            continue
        A = torch.randn((M, K) if ak else (K, M), device="cuda").to(dt)
        B = torch.randn((N, K) if bk else (K, N), device="cuda").to(dt)
        C = torch.zeros(M, N, device="cuda", dtype=torch.float32 if atomic else dt)
        # KLAB_BENCH_AB=1: every shape twice in this process -- the four-wave ring (name_tag 3) and the 256 x 256 eight-wave kernel
        # wherever it is legal (name_tag 2) -- interleaved rounds, median of the per-round times
        tags = [3, 2] if os.environ.get("KLAB_BENCH_AB") == "1" else [int(os.environ["KLAB_BENCH_TAG"])] if "KLAB_BENCH_TAG" in os.environ else [1 if name == "lmhead fwd" else 0]
        res = {}
        for rnd_ in range(3 if len(tags) > 1 else 1):
            for tag in tags:
                kw = dict(M=M, N=N, K=K, a_kmajor=ak, b_kmajor=bk, accumulate=atomic, atomic_ok=atomic, name_tag=tag)
                for _ in range(3):
                    ops.gemm(A, B, C, **kw)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                n = 20
                e0.record()
                for _ in range(n):
                    ops.gemm(A, B, C, **kw)
                e1.record()
                torch.cuda.synchronize()
                res.setdefault(tag, []).append(e0.elapsed_time(e1) / n * 1e3)
        line = f"{name:16s} M={M:6d} N={N:6d} K={K:6d}"
        for tag in tags:
            us = sorted(res[tag])[len(res[tag]) // 2]
            tf = 2.0 * M * N * K / (us * 1e-6) / 1e12
            line += f"  [tag {tag}] {us:8.1f} us {tf:7.1f} TF/s"
        tot += sorted(res[tags[-1]])[len(res[tags[-1]]) // 2]
        print(line, flush=True)
    print("sum us", tot)


if __name__ == "__main__":
    main()
