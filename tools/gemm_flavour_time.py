#!/usr/bin/env python3
"""Times the two big flavours of the persistent NT kernel on the step's shape (M = 524 288, N = K = 1024): forward (ReLU + bit masks out)
and data gradient (bit masks in + column sums), activation-like operands.  One line; target of tools/gemm_policy_ab.sh."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from snerf_amd import ops

M, N, K = 524288, 1024, 1024
A = torch.relu(torch.randn(M, K, device="cuda")).bfloat16()
W = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
b = torch.randn(N, device="cuda") * 0.1
Y = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
bits = torch.zeros(ops.mask_bits_words(M, N), dtype=torch.int32, device="cuda")
cs = torch.zeros(N, device="cuda")
out = []
for f in (lambda: ops.linear_fwd(A, W, b, Y, K, N, ops.ACT_RELU_BITS, ops.BF16, aux=bits, variant=8),
          lambda: ops.linear_fwd(A, W, None, Y, K, N, ops.ACT_MASK_BITS, ops.BF16, aux=bits, colsum=cs, variant=8)):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40):
        f()
    e1.record(); torch.cuda.synchronize()
    out.append(e0.elapsed_time(e1) * 25)
print(f"{sys.argv[1] if len(sys.argv) > 1 else '':28s} fwd ReLU+bits {out[0]:7.1f} us   dgrad bits+colsum {out[1]:7.1f} us", flush=True)
