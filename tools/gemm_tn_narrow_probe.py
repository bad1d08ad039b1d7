#!/usr/bin/env python3
"""The narrow weight-gradient launches of the step (128 x 128 kernel): staging depth (double buffer vs a ring of four slots, variant
bit 128) x workgroup target (default 1024; variant bits 16 / 32 = 512 / 2048), with the XCD-aware slice placement of round 3 in both; checked against
torch's fp32 matmul of the same bf16 operands."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from snerf_amd import ops

torch.manual_seed(0)
shapes = ((524288, 1024, 96, 128, 1024), (524288, 128, 1051, 1088, 128), (524288, 128, 128, 128, 128), (524288, 64, 1024, 1024, 64),
          (262144, 256, 96, 128, 256), (524288, 64, 128, 128, 64))
for (M, N, K, ldx, ldz) in shapes:
    dZ = (torch.randn(M, ldz, device="cuda") * (torch.rand(M, ldz, device="cuda") > 0.5)).bfloat16()
    X = torch.relu(torch.randn(M, ldx, device="cuda")).bfloat16()
    nv = 1 if N == 64 else N                                          # N = 64: the N = 1 / 3 heads (one valid column block)
    ref = (dZ[:, :nv].float().t() @ X[:, :K].float())
    line = f"M={M} N={nv:4d} K={K:4d}:"
    for bits, name in ((0, "dbuf/1024"), (16, "dbuf/512"), (128, "ring4/1024"), (128 | 16, "ring4/512"), (32, "dbuf/2048")):
        dW = torch.zeros(N, ldx, device="cuda")
        for _ in range(3):
            ops.linear_wgrad(dZ[:, :N], X, dW, nv, K, ops.BF16, variant=3 | bits)
        dW.zero_()
        ops.linear_wgrad(dZ[:, :N], X, dW, nv, K, ops.BF16, variant=3 | bits)
        err = float((dW[:nv, :K] - ref).abs().max() / ref.abs().max())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.linear_wgrad(dZ[:, :N], X, dW, nv, K, ops.BF16, variant=3 | bits)
        e1.record(); torch.cuda.synchronize()
        line += f"  {name} {e0.elapsed_time(e1) * 100:6.1f} us ({err:.0e})"
    print(line, flush=True)
