#!/usr/bin/env python3
"""Which NT kernel variant serves the small-K launches of the step best?  (M = 524 288, N = 1024, K = 128: first layer forward, and the
data gradient of the first cond layer into the bottleneck.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from snerf_amd import ops

M = 524288
g = torch.Generator(device="cuda").manual_seed(0)


def run(N, K, act, variant, colsum=False):
    A = torch.relu(torch.randn(M, K, device="cuda", generator=g)).bfloat16()
    W = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
    b = torch.randn(N, device="cuda", generator=g) * 0.1
    Y = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    cs = torch.zeros(N, device="cuda") if colsum else None
    aux = None
    if act in (ops.ACT_RELU_BITS, ops.ACT_MASK_BITS):
        if not ops.relu_bits_ok(A, W, Y, K, N, ops.BF16, variant):
            return None
        aux = torch.zeros(ops.mask_bits_words(M, N), dtype=torch.int32, device="cuda")
    f = lambda: ops.linear_fwd(A, W, None if act == ops.ACT_MASK_BITS else b, Y, K, N, act, ops.BF16, aux=aux, colsum=cs, variant=variant)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10


for N, K in ((1024, 128), (1024, 256), (128, 1024), (128, 128), (256, 128), (256, 256)):
    for act, cs, name in ((ops.ACT_RELU, False, "relu"), (ops.ACT_RELU_BITS, False, "relu+bits"), (ops.ACT_MASK_BITS, True, "mask+colsum"), (0, True, "colsum")):
        row = []
        for v in (0, 1, 4, 8):
            ms = run(N, K, act, v, cs)
            row.append("   n/a " if ms is None else f"{ms:7.3f}")
        print(f"N={N:5d} K={K:5d} {name:12s} variants 0/1/4/8 ms: " + " ".join(row), flush=True)
