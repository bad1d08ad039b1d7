#!/usr/bin/env python3
"""What the cross-slice reduction of the 256 x 256 weight-gradient kernel costs (round 4): gemm_tn8_kernel at N = K = 1024 with
(a) fp32 atomics into dW (default), (b) partial tiles + fixed-order fold (deterministic mode), (c) one valid output row only (the
epilogue all but skipped: the MFMA loop alone).  M = the fine / coarse row counts of the 512-ray and the 4096-ray step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from snerf_amd import ops

N = K = 1024
for M in (32768, 98304, 262144, 786432):
    Z = (torch.rand(M, N, device="cuda") * 2 - 1).bfloat16()
    X = (torch.rand(M, K, device="cuda") * 2 - 1).bfloat16()
    dW = torch.zeros(N, K, device="cuda")
    res = {}
    for name, kw, nv in (("atomics", {}, N), ("partials + fold", dict(deterministic=True), N), ("MFMA loop only", {}, 1)):
        ops.WGRAD_FOLD = False                   # (the library's default policy would pick the fold for these shapes)
        for _ in range(5):
            ops.linear_wgrad(Z, X, dW, nv, K, ops.BF16, variant=2, **kw)
        torch.cuda.synchronize()
        reps = 40
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(reps):
            ops.linear_wgrad(Z, X, dW, nv, K, ops.BF16, variant=2, **kw)
        t1.record()
        torch.cuda.synchronize()
        res[name] = t0.elapsed_time(t1) / reps * 1e3
    fl = 2.0 * M * N * K
    print(f"M = {M:7d}: " + ", ".join(f"{k} {v:7.1f} us ({fl / v / 1e6:6.0f} TFLOP/s)" for k, v in res.items()), flush=True)
