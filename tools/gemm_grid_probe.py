#!/usr/bin/env python3
"""PROBE build (make PROBE=1): the persistent NT kernel on G = 256 ... 16 workgroups, 32 tiles each (M scaled with G), with and without
the store instructions of its epilogue units (ablation bit 8 = variant bit 13): is what the row stores cost a property of the CU or of the
whole chip storing at once?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from snerf_amd import ops

N = K = 1024
for G in (256, 128, 64, 32, 16):
    os.environ["SNERF_NT8P_GRID"] = str(G)
    M = G * 32 * 256 // 4
    A = torch.relu(torch.randn(M, K, device="cuda")).bfloat16()
    W = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
    b = torch.randn(N, device="cuda") * 0.1
    Y = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    line = f"G={G:3d} M={M:6d}:"
    res = []
    for rep in range(2):
        for name, v in (("stores", 8), ("no store instr", 8 | (1 << 13))):
            for _ in range(3):
                ops.linear_fwd(A, W, b, Y, K, N, ops.ACT_RELU, ops.BF16, variant=v)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.linear_fwd(A, W, b, Y, K, N, ops.ACT_RELU, ops.BF16, variant=v)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 50
            res.append(us)
            line += f"  {name} {us:7.1f} us ({2.0 * M * N * K / us / 1e6 / G * 256:6.0f} TFLOP/s x256/G)"
    line += f"   store cost per tile {(res[0] + res[2] - res[1] - res[3]) / 2 / 32 * 1e3:6.0f} ns"
    print(line, flush=True)
