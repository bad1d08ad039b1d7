#!/usr/bin/env python3
"""PROBE build: streaming (nt) hints on the staging loads of the 8-phase weight-gradient kernel (SNERF_TN8_DBG: 1 = dZ, 2 = X), on the
path-A shape (M = 524 288, N = K = 1024) and the path-B shape (M = 6 291 456, N = K = 256); 6 interleaved rounds, medians."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from snerf_amd import ops

for M, N, K in ((524288, 1024, 1024), (6291456, 256, 256)):
    dZ = (torch.randn(M, N, device="cuda") * (torch.rand(M, N, device="cuda") > 0.5)).bfloat16()
    X = torch.relu(torch.randn(M, K, device="cuda")).bfloat16()
    dW = torch.zeros(N, K, device="cuda")
    cases = [("default", 0), ("dZ nt", 1), ("X nt", 2), ("both nt", 3)]
    t = {n: [] for n, _ in cases}
    for rep in range(6):
        for name, bits in cases:
            os.environ["SNERF_TN8_DBG"] = str(bits)
            f = lambda: ops.linear_wgrad(dZ, X, dW, N, K, ops.BF16, variant=3)
            f(); f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                f()
            e1.record(); torch.cuda.synchronize()
            t[name].append(e0.elapsed_time(e1) * 100)
    for name, _ in cases:
        v = sorted(t[name])
        print(f"M={M} N={N} K={K}  {name:10s} median {statistics.median(v):7.1f} us  min {v[0]:7.1f}  max {v[-1]:7.1f}", flush=True)
