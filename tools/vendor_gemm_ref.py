"""Vendor reference point: torch.matmul (hipBLASLt) on the LM-head / T5 GEMM shapes of BASELINE configs[1], to compare with tools/gemm_bench.py."""
import torch
torch.manual_seed(0)
for (M,N,K,name) in [(4096,32128,512,"lmhead fwd"),(4096,512,32128,"lmhead dgrad"),(32128,512,4096,"lmhead wgrad"),(4096,2048,512,"wi fwd"),(4096,512,2048,"wo fwd")]:
    a=torch.randn(M,K,device="cuda",dtype=torch.bfloat16); b=torch.randn(N,K,device="cuda",dtype=torch.bfloat16)
    for _ in range(5): c=a@b.t()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): c=a@b.t()
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)/20*1e3
    print(f"torch.matmul (hipBLASLt) {name:14s} M={M} N={N} K={K}: {us:8.1f} us  {2*M*N*K/us/1e6:7.1f} TF/s")
# TN form (both operands token-major, as the LM-head weight gradient reads them): dW[N_vocab, d] = dlogits[tokens, vocab]^T @ X[tokens, d]
for (M,N,K,name) in [(32128,512,4096,"lmhead wgrad TN"),(2048,512,4096,"wi wgrad TN"),(512,2048,4096,"wo wgrad TN")]:
    a=torch.randn(K,M,device="cuda",dtype=torch.bfloat16); b=torch.randn(K,N,device="cuda",dtype=torch.bfloat16)
    for _ in range(5): c=a.t()@b
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): c=a.t()@b
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)/20*1e3
    print(f"torch.matmul (hipBLASLt) {name:16s} M={M} N={N} K={K}: {us:8.1f} us  {2*M*N*K/us/1e6:7.1f} TF/s (bf16 out)")
# BASELINE configs[4] shapes (T5-large d = 1024, ff = 4096; encoder 32 x 153 tokens, decoder 32 x 64; Swin-V2 384 / window 24):
# forward (K-major x K-major) and input-gradient (K-major x m-major) forms, as tools/gemm_bench.py --large runs them
import sys
if "--large" in sys.argv:
    L = [("L enc qkv fwd", 4896, 3072, 1024, True), ("L enc o fwd", 4896, 1024, 1024, True), ("L enc wi fwd", 4896, 4096, 1024, True),
         ("L enc wo fwd", 4896, 1024, 4096, True), ("L enc qkv dgrad", 4896, 1024, 3072, False), ("L enc wi dgrad", 4896, 1024, 4096, False),
         ("L enc wo dgrad", 4896, 4096, 1024, False), ("L dec wi fwd", 2048, 4096, 1024, True), ("L dec wi dgrad", 2048, 1024, 4096, False),
         ("swin2 qkv fwd", 18432, 1536, 512, True), ("swin2 fc1 fwd", 18432, 2048, 512, True), ("swin2 fc1 dgrad", 18432, 512, 2048, False),
         ("sq k4096", 4096, 4096, 4096, True), ("sq k8192", 4096, 4096, 8192, True)]
    for name, M, N, K, bk in L:
        a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
        b = torch.randn((N, K) if bk else (K, N), device="cuda", dtype=torch.bfloat16)
        f = (lambda: a @ b.t()) if bk else (lambda: a @ b)
        for _ in range(5): c = f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): c = f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(f"torch.matmul (hipBLASLt) {name:16s} M={M:6d} N={N:6d} K={K:6d}: {us:8.1f} us  {2*M*N*K/us/1e6:7.1f} TF/s")
