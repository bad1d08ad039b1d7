#!/usr/bin/env python3
"""Reference point for the dominant kernel: the vendor library (torch.mm -> hipBLASLt / rocBLAS) on the SAME GEMM shapes of the
path-A train step, next to snerf_linear_fwd / snerf_linear_wgrad.  Says whether the ~0.37 of the dense MFMA peak is this kernel's
schedule or what the part gives on these shapes (M = 524 288 rows, N = K = 1024: every operand streams once, intensity 511 FLOP/B,
package at its power cap -- profiles/r2_c_power_clock_trace.txt)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from snerf_amd import ops


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    dev = "cuda"
    out = {}
    for (M, N, K) in ((524288, 1024, 1024), (262144, 1024, 1024), (6291456, 256, 256)):
        A = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
        W = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.03
        b = torch.zeros(N, device=dev)
        Y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        dZ = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
        dW = torch.zeros(N, K, device=dev)
        fl = 2.0 * M * N * K
        r = {}
        r["nt_vendor_ms"] = timeit(lambda: torch.mm(A, W.t(), out=Y))
        r["nt_vendor_bias_relu_ms"] = timeit(lambda: torch.relu_(torch.addmm(b.to(torch.bfloat16), A, W.t(), out=Y)))
        r["nt_ours_bias_relu_ms"] = timeit(lambda: ops.linear_fwd(A, W, b, Y, K, N, ops.ACT_RELU, ops.BF16, variant=8))
        if N % 256 == 0:
            words = torch.empty(ops.mask_bits_words(M, N), dtype=torch.int32, device=dev)
            cs = torch.zeros(N, device=dev)
            r["nt_ours_relu_bits_ms"] = timeit(lambda: ops.linear_fwd(A, W, b, Y, K, N, ops.ACT_RELU_BITS, ops.BF16, aux=words, variant=8))
            r["nt_ours_mask_bits_colsum_ms"] = timeit(lambda: ops.linear_fwd(dZ, W, None, Y, K, N, ops.ACT_MASK_BITS, ops.BF16, aux=words, colsum=cs, variant=8))
            r["nt_ours_mask_colsum_ms"] = timeit(lambda: ops.linear_fwd(dZ, W, None, Y, K, N, ops.ACT_MASK, ops.BF16, aux=A, colsum=cs, variant=8))
        dWb = torch.empty(N, K, device=dev, dtype=torch.bfloat16)
        r["tn_vendor_ms"] = timeit(lambda: torch.mm(dZ.t(), A, out=dWb))
        r["tn_ours_ms"] = timeit(lambda: ops.linear_wgrad(dZ, A, dW, N, K, ops.BF16, variant=3))
        for k in list(r):
            r[k.replace("_ms", "_TFLOPs")] = round(fl / r[k] / 1e9, 1)
            r[k] = round(r[k], 4)
        out[f"M{M}_N{N}_K{K}"] = r
        del A, W, Y, dZ, dW, dWb
    print(json.dumps(out))


if __name__ == "__main__":
    main()
