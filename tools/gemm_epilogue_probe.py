#!/usr/bin/env python3
"""What do the epilogue flavours of the persistent NT kernel cost?  One shape (M = 524 288, N = K = 1024, bf16), activation-like
operands (half of A zero, as after a ReLU), HIP-event timing of each flavour."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from snerf_amd import ops

M, N, K = 524288, 1024, 1024
g = torch.Generator(device="cuda").manual_seed(0)
A = torch.relu(torch.randn(M, K, device="cuda", generator=g)).bfloat16()
W = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
b = torch.randn(N, device="cuda", generator=g) * 0.1
Y = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
bits = torch.zeros(ops.mask_bits_words(M, N), dtype=torch.int32, device="cuda")
cs = torch.zeros(N, device="cuda")


def t(name, act, aux=None, colsum=None, bias=b):
    f = lambda: ops.linear_fwd(A, W, bias, Y, K, N, act, ops.BF16, aux=aux, colsum=colsum, variant=8)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{name:34s} {ms:7.3f} ms  {2.0 * M * N * K / ms / 1e9:7.1f} TFLOP/s", flush=True)


t("plain (bias, no activation)", 0)
t("no bias", 0, bias=None)
t("ReLU", ops.ACT_RELU)
t("ReLU + bit mask out", ops.ACT_RELU_BITS, aux=bits)
t("bit mask in", ops.ACT_MASK_BITS, aux=bits, bias=None)
t("column sums", 0, colsum=cs, bias=None)
t("bit mask in + column sums", ops.ACT_MASK_BITS, aux=bits, colsum=cs, bias=None)
