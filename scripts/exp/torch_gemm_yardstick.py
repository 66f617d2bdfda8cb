"""Yardstick only (not used by the product): what torch.matmul (hipBLASLt / rocBLAS) reaches on the
cross-encoder's GEMM shapes in fp32, bf16 and fp16, next to the hand-written kernels' numbers."""
import torch
torch.backends.cuda.matmul.allow_tf32 = False
M = 178405
shapes = [("qkv (MiniLM)", 1152, 384), ("attn out", 384, 384), ("ffn1", 1536, 384), ("ffn2", 384, 1536),
          ("qkv (base)", 2304, 768), ("ffn1 (base)", 3072, 768), ("ffn2 (base)", 768, 3072)]
for dt in (torch.float32, torch.bfloat16, torch.float16):
    for name, N, K in shapes:
        a = torch.randn(M, K, device="cuda", dtype=dt); w = torch.randn(N, K, device="cuda", dtype=dt)
        for _ in range(3): (a @ w.T)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): c = a @ w.T
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"{str(dt):16s} {name:14s} N={N:5d} K={K:5d}: {ms:7.3f} ms  {2.0*M*N*K/ms/1e9:7.1f} TF/s", flush=True)
