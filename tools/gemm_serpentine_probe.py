#!/usr/bin/env python3
"""A chain of NT launches (layer l reads what layer l - 1 wrote; M = 524 288, N = K = 1024, ReLU + bit masks like the training forward):
all launches walking the row panels first-to-last, against alternating directions (variant bit 15 on every second launch) so that a
launch starts on the rows its producer wrote last -- still in the 256 MB Infinity Cache."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from snerf_amd import ops

M, N, K, L = 524288, 1024, 1024, 8
X = torch.relu(torch.randn(M, K, device="cuda")).bfloat16()
Ws = [(torch.randn(N, K, device="cuda") * (2.0 / K) ** 0.5).bfloat16() for _ in range(L)]
b = torch.zeros(N, device="cuda")
bufs = [torch.empty(M, N, dtype=torch.bfloat16, device="cuda") for _ in range(L)]
bits = [torch.zeros(ops.mask_bits_words(M, N), dtype=torch.int32, device="cuda") for _ in range(L)]


def chain(serp):
    a = X
    for l in range(L):
        v = 8 | ((1 << 15) if (serp and (l & 1)) else 0)
        ops.linear_fwd(a, Ws[l], b, bufs[l], K, N, ops.ACT_RELU_BITS, ops.BF16, aux=bits[l], variant=v)
        a = bufs[l]


ref = None
for rep in range(3):
    for serp in (False, True):
        chain(serp); chain(serp)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            chain(serp)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        if ref is None:
            ref = (bufs[-1].clone(), bits[-1].clone())
        same = torch.equal(bufs[-1], ref[0]) and torch.equal(bits[-1], ref[1])
        print(f"{'alternating' if serp else 'one direction':14s} {ms:7.3f} ms per {L}-layer chain   {ms / L * 1e3:7.1f} us per launch   {'same result' if same else 'DIFFERENT'}", flush=True)
