#!/usr/bin/env python3
"""Race screen + A/B timing of the 8-phase NT kernel (variant 4) against the block-issue 256x256 kernel (variant 1).
Both accumulate k in the same order, so their outputs must be bit-identical on every run."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from snerf_amd import ops
from gemm_probe import timeit


def main():
    dt = ops.BF16
    bad = 0
    for (M, N, K) in ((4096, 256, 128), (300000, 256, 128), (4096, 512, 192), (65536, 1024, 1024), (65536 + 77, 1024, 1152), (524288, 1024, 1024)):
        A = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16)
        W = ((torch.rand(N, K, device="cuda") * 2 - 1) / K ** 0.5).to(torch.bfloat16)
        b = torch.rand(N, device="cuda")
        Y1 = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        Y4 = torch.empty_like(Y1)
        ops.linear_fwd(A, W, b, Y1, K, N, ops.ACT_RELU, dt, variant=1)
        for rep in range(12 if M < 500000 else 6):
            Y4.fill_(-3.0)
            ops.linear_fwd(A, W, b, Y4, K, N, ops.ACT_RELU, dt, variant=4 if rep % 2 else 8)
            torch.cuda.synchronize()
            nd = int((Y1.view(torch.int16) != Y4.view(torch.int16)).sum())
            if nd:
                bad += 1
                rows = (Y1.view(torch.int16) != Y4.view(torch.int16)).any(1).nonzero().flatten()
                print(f"MISMATCH M={M} N={N} K={K} rep {rep}: {nd} elements, first rows {rows[:8].tolist()}", flush=True)
        print(f"screen M={M} N={N} K={K}: done", flush=True)
    print("RACE SCREEN", "FAILED" if bad else "clean")
    M, N, K = 524288, 1024, 1024
    A = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    W = ((torch.rand(N, K, device="cuda") * 2 - 1) / K ** 0.5).to(torch.bfloat16)
    b = torch.rand(N, device="cuda")
    Y = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    cs = torch.zeros(N, device="cuda")
    for rnd in range(2):
        for v in (1, 4, 8, 8 + 64):
            ms = timeit(lambda: ops.linear_fwd(A, W, b, Y, K, N, ops.ACT_RELU, dt, variant=v), reps=10)
            print(f"NT fwd   variant={v:3d}: {ms:7.3f} ms {2.0 * M * N * K / ms / 1e9:8.1f} TF/s", flush=True)
        bits = torch.zeros(ops.mask_bits_words(M, N), dtype=torch.int32, device="cuda")
        ms = timeit(lambda: ops.linear_fwd(A, W, b, Y, K, N, ops.ACT_RELU_BITS, dt, aux=bits, variant=8), reps=10)
        print(f"NT fwd+bits  variant=  8: {ms:7.3f} ms {2.0 * M * N * K / ms / 1e9:8.1f} TF/s", flush=True)
        ms = timeit(lambda: ops.linear_fwd(A, W, None, Y, K, N, ops.ACT_MASK_BITS, dt, aux=bits, colsum=cs, variant=8), reps=10)
        print(f"NT dgrad bits variant=  8: {ms:7.3f} ms {2.0 * M * N * K / ms / 1e9:8.1f} TF/s", flush=True)
        for v in (1, 4, 8):
            ms = timeit(lambda: ops.linear_fwd(A, W, None, Y, K, N, ops.ACT_MASK, dt, aux=A, colsum=cs, variant=v), reps=10)
            print(f"NT dgrad variant={v:3d}: {ms:7.3f} ms {2.0 * M * N * K / ms / 1e9:8.1f} TF/s", flush=True)
    A2 = (torch.rand(M, 1152, device="cuda") * 2 - 1).to(torch.bfloat16)
    W2 = ((torch.rand(N, 1152, device="cuda") * 2 - 1) / 34).to(torch.bfloat16)
    for v in (1, 4, 8):
        ms = timeit(lambda: ops.linear_fwd(A2, W2, b, Y, 1152, N, ops.ACT_RELU, dt, variant=v), reps=10)
        print(f"NT fwd K=1152 variant={v:3d}: {ms:7.3f} ms {2.0 * M * N * 1152 / ms / 1e9:8.1f} TF/s", flush=True)


def wgrad():
    dt = ops.BF16
    for (M, N, K) in ((524288, 1024, 1024), (524288, 1024, 1152), (262144, 256, 256), (528384, 1024, 1088)):
        dZ = (torch.rand(M, N, device="cuda") * 2 - 1).to(torch.bfloat16)
        X = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16)
        ref = None
        for v in (1, 2):
            dW = torch.zeros(N, K, device="cuda")
            ops.linear_wgrad(dZ, X, dW, N, K, dt, variant=v)
            torch.cuda.synchronize()
            if ref is None:
                ref = dW.clone()
            else:
                err = float((dW - ref).abs().max() / ref.abs().max())
                print(f"TN M={M} N={N} K={K}: variant 2 vs 1 max rel diff {err:.2e}", flush=True)
            ms = timeit(lambda: ops.linear_wgrad(dZ, X, dW, N, K, dt, variant=v), reps=10)
            print(f"TN M={M} N={N} K={K} variant={v}: {ms:7.3f} ms {2.0 * M * N * K / ms / 1e9:8.1f} TF/s", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "wgrad":
        wgrad()
        sys.exit(0)
    main()
