#!/usr/bin/env python3
"""Reproducer of the K = 128 race of gemm_nt8p_kernel (round 4): with two k-tiles per output tile the DMA of the NEXT tile's bias vector was
not ordered before the epilogue unit that reads it.  Alone, or after a launch that leaves the same bias in the CU's LDS, the kernel still
produced the right values; right after ANOTHER persistent GEMM (another layer's bias in LDS) about one launch in 150 took a stale bias into
one 32 x 64 output block.  Found through the early-ray-termination leg (two renders of one model differed by more than the bound).

    python tools/probes/gemm_k128_bias_race.py      # expects 0 events after the fix (s_waitcnt in P1 of the last k-tile when KT == 2)
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from snerf_amd import ops

dev = torch.device("cuda", 0)
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.rand(*s, device=dev, generator=g) * 2 - 1
M, H, K = 524288, 1024, 128
buf = rnd(M, H + K).bfloat16()
A = buf[:, H:]
W0, b0 = (rnd(H, K) / K ** 0.5).bfloat16(), rnd(H)
W1, b1 = (rnd(H, H) / H ** 0.5).bfloat16(), rnd(H)
Y, Z = torch.empty(M, H, dtype=torch.bfloat16, device=dev), torch.empty(M, H, dtype=torch.bfloat16, device=dev)
bits = torch.empty(ops.mask_bits_words(M, H), dtype=torch.int32, device=dev)
for name, act, aux in (("ReLU", ops.ACT_RELU, None), ("ReLU + bit masks", ops.ACT_RELU_BITS, bits)):
    ops.linear_fwd(A, W0, b0, Y, K, H, act, ops.BF16, aux=aux, variant=8)
    ref = Y.clone()
    events, t0 = [], time.time()
    for r in range(2000):
        ops.linear_fwd(Y, W1, b1, Z, H, H, ops.ACT_RELU, ops.BF16, variant=8)          # another layer: its bias stays in the CUs' LDS
        ops.linear_fwd(A, W0, b0, Y, K, H, act, ops.BF16, aux=aux, variant=8)
        if not torch.equal(Y, ref):
            ne = Y != ref
            rows = ne.any(-1).nonzero().flatten()
            events.append((r, int(ne.sum()), int(rows[0]) % 256, int(rows[-1]) - int(rows[0]) + 1))
    print(f"K = 128 layer ({name}) after a K = 1024 layer: {len(events)} events in 2000 launches ({time.time() - t0:.1f} s) {events[:4]}")
