#!/usr/bin/env python3
"""Split-bf16 NT GEMM flavours at the step's large shape, one launch each (M = 524288, N = K = 1024): forward with bit masks, data gradient
with bit masks, with and without the bias-gradient column sums; against the plain bf16 launches on the same box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from snerf_amd import ops

M, N, K = 524288, 1024, 1024
dev = torch.device("cuda")
torch.manual_seed(0)


def t(fn, n=5):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for dt, km, kw in ((ops.BF16, 1, 1), (ops.BF16X3, 2, 3)):
    A = torch.randn(M, K * km, device=dev).bfloat16()
    W = (torch.randn(N, K * kw, device=dev) * 0.03).bfloat16()
    b = torch.zeros(N, device=dev)
    Y = torch.empty(M, N * km, dtype=torch.bfloat16, device=dev)
    words = torch.empty(ops.mask_bits_words(M, N), dtype=torch.int32, device=dev)
    cs = torch.zeros(N, device=dev)
    f = t(lambda: ops.linear_fwd(A, W, b, Y, K, N, ops.ACT_RELU_BITS, dt, aux=words, variant=8))
    d0 = t(lambda: ops.linear_fwd(A, W, None, Y, K, N, ops.ACT_MASK_BITS, dt, aux=words, variant=8))
    d1 = t(lambda: ops.linear_fwd(A, W, None, Y, K, N, ops.ACT_MASK_BITS, dt, aux=words, colsum=cs, variant=8))
    r = t(lambda: ops.linear_fwd(A, W, b, Y, K, N, ops.ACT_RELU, dt, variant=8))
    print(f"dt {dt}: fwd relu+bits {f:8.1f} us, relu {r:8.1f} us, dgrad mask-bits {d0:8.1f} us, + colsum {d1:8.1f} us  ({2.0 * M * N * K * kw / f / 1e6:.0f} TFLOP/s executed fwd)", flush=True)
