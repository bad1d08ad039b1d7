#!/usr/bin/env python3
"""(WITHDRAWN: the kernel lives in tools/probes/gemm_nt4_kernel.hip since round 4; kept for the record of profiles/r3_w_*)  A/B of the two-workgroups-per-CU NT kernel (gemm_nt4_kernel: 128 x 256 tiles, 4 waves, BK = 32; variant bit 14) against the shipped
persistent 8-phase kernel on the train step's large forward shapes; checks the result against an fp64 matmul first."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from snerf_amd import ops

NT4 = 8 | (1 << 14)


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


torch.manual_seed(0)
# correctness on a ragged shape
M, N, K = 1000, 512, 1152
A = (torch.rand(M, K, device="cuda") * 2 - 1).bfloat16()
W = ((torch.rand(N, K, device="cuda") * 2 - 1) / K ** 0.5).bfloat16()
b = torch.rand(N, device="cuda")
ref = torch.relu(A.double() @ W.double().t() + b.double())
for v in (8, NT4):
    Y = torch.full((M, N), 7.0, dtype=torch.bfloat16, device="cuda")
    ops.linear_fwd(A, W, b, Y, K, N, ops.ACT_RELU, ops.BF16, variant=v)
    err = float((Y.double() - ref).abs().max() / ref.abs().max())
    print(f"variant {v:6d}: max err / max |ref| {err:.3e}")
    assert err < 1e-2
for (M, N, K) in ((524288, 1024, 1024), (524288, 1024, 1152), (65536, 1024, 1024)):
    A = (torch.rand(M, K, device="cuda") * 2 - 1).bfloat16()
    A = torch.relu(A)                                     # post-ReLU operand like the real step
    W = ((torch.rand(N, K, device="cuda") * 2 - 1) / K ** 0.5).bfloat16()
    b = torch.rand(N, device="cuda")
    Y = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    out = []
    for rep in range(2):
        for name, v in (("nt8p", 8), ("nt4", NT4)):
            for act, an in ((ops.ACT_RELU, "relu"), (ops.ACT_NONE, "none")):
                ms = timeit(lambda: ops.linear_fwd(A, W, b, Y, K, N, act, ops.BF16, variant=v))
                out.append(f"{name}/{an} {ms * 1e3:7.1f} us {2.0 * M * N * K / ms / 1e9:7.1f} TF")
    print(f"M={M} N={N} K={K}: " + " | ".join(out))
