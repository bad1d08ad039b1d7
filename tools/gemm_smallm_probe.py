#!/usr/bin/env python3
"""Small-M launches (the 512..1024 rays-per-GPU steps): which NT variant is best when the 256x256 tiling leaves compute units idle?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from snerf_amd import ops

g = torch.Generator(device="cuda").manual_seed(0)


def run(M, N, K, variant):
    A = torch.relu(torch.randn(M, K, device="cuda", generator=g)).bfloat16()
    W = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
    b = torch.randn(N, device="cuda", generator=g) * 0.1
    Y = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    f = lambda: ops.linear_fwd(A, W, b, Y, K, N, ops.ACT_RELU, ops.BF16, variant=variant)
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20


for M in (16384, 32768, 65536, 131072):
    for N, K in ((256, 256), (1024, 1024), (1024, 128)):
        row = [run(M, N, K, v) for v in (0, 1, 4, 8)]
        print(f"M={M:7d} N={N:5d} K={K:5d} variants 0/1/4/8 us: " + " ".join(f"{x * 1e3:8.1f}" for x in row) + f"   TF/s@best {2.0 * M * N * K / min(row) / 1e9:7.1f}", flush=True)
