#!/usr/bin/env python3
"""Persistent NT kernel with staggered workgroup starts (variant bits 9..12 = offset per slot in units of KT x 64 clocks; 8 slots per XCD):
do the chip-wide store bursts at the tile boundaries cost time?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from snerf_amd import ops

M = 524288
for N, K in ((1024, 1024), (1024, 128), (256, 256)):
    A = (torch.rand(M, K, device="cuda") * 2 - 1).bfloat16()
    W = ((torch.rand(N, K, device="cuda") * 2 - 1) / K ** 0.5).bfloat16()
    b = torch.rand(N, device="cuda")
    Y = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    ref = None
    line = f"M={M} N={N} K={K}:"
    for st in (0, 1, 2, 4, 6, 0):
        v = 8 | (st << 9)
        for _ in range(3):
            ops.linear_fwd(A, W, b, Y, K, N, ops.ACT_RELU, ops.BF16, variant=v)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.linear_fwd(A, W, b, Y, K, N, ops.ACT_RELU, ops.BF16, variant=v)
        e1.record(); torch.cuda.synchronize()
        if ref is None:
            ref = Y.clone()
        assert torch.equal(Y, ref)
        line += f"  stagger {st}: {e0.elapsed_time(e1) * 50:7.1f} us"
    print(line, flush=True)
