#!/usr/bin/env python3
"""Times the fused data-gradient chains (snerf_fchain_bwd) alone: classic NeRF 8 x 256 at the fine pass's rows (32768 rays x 192) and
the proposal MLP at the path-A step's rows (4096 x 64), against the HBM bytes they must write (every layer's gradient, bf16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from snerf_amd import ops


def run(net, M, frags, widths, nbits, dcols, iters=10):
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    stream = (torch.randn(frags, 512, device=dev, generator=g) * 0.05).to(torch.bfloat16)
    bits = [torch.randint(-2 ** 31, 2 ** 31 - 1, (ops.mask_bits_words(M, 128 if i == 8 else 256),), device=dev, dtype=torch.int32, generator=g) for i in range(nbits)]
    d_raw = torch.randn(M, dcols, device=dev, generator=g)
    dz = [torch.empty(M, w, dtype=torch.bfloat16, device=dev) for w in widths]
    gb = [torch.zeros(w, device=dev) for w in widths]
    for _ in range(2):
        ops.fchain_bwd(net, d_raw, stream, bits, dz, gb)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters):
        ops.fchain_bwd(net, d_raw, stream, bits, dz, gb)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    by = M * sum(widths) * 2
    fl = 2.0 * M * frags * 512
    print(f"net {net} M={M}: {us:9.1f} us  stores {by / 1e9:.2f} GB -> {by / us / 1e6:.2f} TB/s   MFMA {fl / us / 1e6:.0f} TFLOP/s")


if __name__ == "__main__":
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 32768 * 192
    run(ops.CHAIN_CLASSIC, M, 1104, [128] + [256] * 9, 9, 4)
    run(ops.CHAIN_PROPOSAL, 4096 * 64, 400, [256] * 4, 4, 1)
    run(ops.CHAIN_PROPOSAL, M, 400, [256] * 4, 4, 1)
