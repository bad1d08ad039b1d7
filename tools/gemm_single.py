#!/usr/bin/env python3
"""One GEMM shape, a few launches -- target for rocprofv3 --pmc passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from snerf_amd import ops
M, N, K = 524288, 1024, 1024
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 1
kind = sys.argv[2] if len(sys.argv) > 2 else "nt"
A = (torch.rand(M, K, device="cuda") * 2 - 1).bfloat16()
W = ((torch.rand(N, K, device="cuda") * 2 - 1) / K ** 0.5).bfloat16()
b = torch.rand(N, device="cuda")
Y = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
dW = torch.zeros(N, K, device="cuda")
for _ in range(4):
    if kind == "nt":
        ops.linear_fwd(A, W, b, Y, K, N, ops.ACT_RELU, ops.BF16, variant=variant)
    else:
        ops.linear_wgrad(Y, A, dW, N, K, ops.BF16, variant=variant)
torch.cuda.synchronize()
