#!/usr/bin/env python3
"""Weight-gradient launches of the step that do NOT run on the 256 x 256 8-phase kernel (N or K too narrow): time vs the number of M
slices (workgroups per 128 x 128 output tile).  Each slice ends with one fp32 atomic per output element, so fewer, longer slices trade
atomics for parallelism.  Default: aim at 512 workgroups; variant bits 16 / 32 / 64 = 1024 / 2048 / 4096 (round 1's choice).
Measured (MI355X, round 2), 4096 / 2048 / 1024 / 512 workgroups: N=1024 K=96: 355 / 331 / 330 / 322 us; N=128 K=1051: 378 / 352 / 389 / 338;
N=1024 K=128: 393 / 340 / 336 / 302; the 60-70 us launches do not care."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from snerf_amd import ops

torch.manual_seed(0)
for (M, N, K, ldx) in ((524288, 1024, 96, 128), (524288, 128, 1051, 1088), (524288, 128, 128, 128), (262144, 256, 96, 128), (524288, 1024, 128, 128)):
    dZ = (torch.randn(M, N, device="cuda") * (torch.rand(M, N, device="cuda") > 0.5)).bfloat16()
    X = torch.relu(torch.randn(M, ldx, device="cuda")).bfloat16()
    line = f"M={M} N={N} K={K}:"
    ref = None
    for bits in (0, 16, 32, 64):
        dW = torch.zeros(N, ldx, device="cuda")
        for _ in range(3):
            ops.linear_wgrad(dZ, X, dW, N, K, ops.BF16, variant=3 | bits)
        dW.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.linear_wgrad(dZ, X, dW, N, K, ops.BF16, variant=3 | bits)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        if ref is None:
            ref = dW.clone()
        err = float((dW - ref).abs().max() / ref.abs().max())
        line += f"  target {512 << (0 if bits == 0 else (1 if bits == 16 else 2 if bits == 32 else 3))}: {us:7.1f} us (rel diff {err:.1e})"
    print(line, flush=True)
