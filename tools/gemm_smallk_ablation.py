#!/usr/bin/env python3
"""Where the time of the persistent NT kernel goes on the K = 128 launches of the step (N = 1024: first layer forward, colour-head data
gradient): ablations of a `make PROBE=1` build -- variant bit 16 = no staging loads, bit 64 = no emit path (slab read-back + global stores), bit 8192 = only the store instructions removed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from snerf_amd import ops

M = 524288
for N, K in ((1024, 128), (1024, 256), (1024, 1024)):
    A = (torch.rand(M, K, device="cuda") * 2 - 1).bfloat16()
    W = ((torch.rand(N, K, device="cuda") * 2 - 1) / K ** 0.5).bfloat16()
    b = torch.rand(N, device="cuda")
    Y = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    line = f"M={M} N={N} K={K}:"
    for name, v in (("full", 8), ("no loads", 8 | 16), ("no store instructions", 8 | (1 << 13)), ("no emit path", 8 | 64), ("neither", 8 | 16 | 64)):
        for _ in range(2):
            ops.linear_fwd(A, W, b, Y, K, N, ops.ACT_RELU, ops.BF16, variant=v)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.linear_fwd(A, W, b, Y, K, N, ops.ACT_RELU, ops.BF16, variant=v)
        e1.record(); torch.cuda.synchronize()
        line += f"  {name} {e0.elapsed_time(e1) * 100:7.1f} us"
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        Y.zero_()
    e1.record(); torch.cuda.synchronize()
    print(line + f"  | plain fill of Y {e0.elapsed_time(e1) * 100:7.1f} us", flush=True)
