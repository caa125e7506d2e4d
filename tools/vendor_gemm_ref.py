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
