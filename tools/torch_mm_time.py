#!/usr/bin/env python3
"""Library fp32 GEMM (torch.mm -> rocBLAS / hipBLASLt) at the decoder's shapes, for comparison with tools/gemm_bench.cpp."""
import sys, torch
torch.backends.cuda.matmul.allow_tf32 = False
for M, N, K in [(17408, 1536, 512), (17408, 512, 512), (17408, 1024, 512), (17408, 512, 1024), (9728, 512, 512)]:
    a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda")
    for _ in range(10): torch.mm(a, w.t())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): torch.mm(a, w.t())
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / 50
    print(f"torch.mm M={M} N={N} K={K}: {us:.1f} us  {2.0*M*N*K/us*1e-6:.1f} TFLOP/s ({2.0*M*N*K/us*1e-6/157.3:.3f})")
