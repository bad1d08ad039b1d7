#!/usr/bin/env python3
"""Micro-benchmark / ablation of the MFMA GEMM kernels at the bench shapes (run on the GPU box).
variant bits: low nibble = tile variant, bit4 = no staging loads, bit5 = no LDS fragment reads, bit6 = no stores."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from snerf_amd import ops


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 524288
    for dt, name in ((ops.BF16, "bf16"), (ops.F32, "f32")):
        tdt = ops.torch_dtype(dt)
        Mx = M if dt == ops.BF16 else M // 8
        for N, K in ((1024, 1024), (1024, 1152) if dt == ops.BF16 else (1024, 1120), (256, 256), (128, 1088) if dt == ops.BF16 else (128, 1056)):
            A = (torch.rand(Mx, K, device="cuda") * 2 - 1).to(tdt)
            W = ((torch.rand(N, K, device="cuda") * 2 - 1) / K ** 0.5).to(tdt)
            b = torch.rand(N, device="cuda")
            Y = torch.empty(Mx, N, dtype=tdt, device="cuda")
            variants = [0, 16, 64] + ([1, 65, 4, 8] if dt == ops.BF16 and N % 256 == 0 else [])
            for v in variants:
                ms = timeit(lambda: ops.linear_fwd(A, W, b, Y, K, N, ops.ACT_RELU, dt, variant=v))
                print(f"NT {name} M={Mx} N={N} K={K} variant={v:3d}: {ms:8.3f} ms  {2.0 * Mx * N * K / ms / 1e9:8.1f} TF/s", flush=True)
            cs = torch.zeros(N, device="cuda")
            for v in ([0] + ([1, 4, 8] if dt == ops.BF16 and N % 256 == 0 else [])) if K == N else []:
                ms = timeit(lambda: ops.linear_fwd(A, W, None, Y, K, N, ops.ACT_MASK, dt, aux=A, colsum=cs, variant=v))
                print(f"NT {name} dgrad(mask+colsum) M={Mx} N={N} K={K} variant={v:3d}: {ms:8.3f} ms  {2.0 * Mx * N * K / ms / 1e9:8.1f} TF/s", flush=True)
            dZ = (torch.rand(Mx, N, device="cuda") * 2 - 1).to(tdt)
            dW = torch.zeros(N, K, device="cuda")
            for v in (0, 1):
                ms = timeit(lambda: ops.linear_wgrad(dZ, A, dW, N, K, dt, variant=v))
                print(f"TN {name} M={Mx} N={N} K={K} variant={v}  : {ms:8.3f} ms  {2.0 * Mx * N * K / ms / 1e9:8.1f} TF/s", flush=True)
            # library reference point (hipBLASLt through torch) on the same operands
            ms = timeit(lambda: torch.relu_(torch.addmm(b.to(tdt), A, W.t())))
            print(f"torch addmm+relu {name} M={Mx} N={N} K={K}  : {ms:8.3f} ms  {2.0 * Mx * N * K / ms / 1e9:8.1f} TF/s", flush=True)


if __name__ == "__main__":
    main()
