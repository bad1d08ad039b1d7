#!/usr/bin/env python3
"""Fused colour head (snerf_fcolour_fwd / _bwd) against the per-layer GEMM launches it replaces, same box, same operands:
inference forward, training forward (stores + bit masks), the data-gradient chain; the forward at both input read-ahead depths."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from snerf_amd import ops
from snerf_amd.mlp import MipNerfNet, ParamArena

M, H = 524288, 1024
dev = torch.device("cuda")
torch.manual_seed(0)
shapes = [("mlp." + n, s) for n, s in MipNerfNet.param_shapes(H, 8, 4, 96, 27, 3, 128)]
arena = ParamArena(shapes, dev)
for n in arena.names:
    p = arena.p[n]
    (torch.nn.init.xavier_uniform_(p if p.dim() == 2 else p.view(1, -1)) if n.endswith("weight") else p.uniform_(-0.05, 0.05))
net = MipNerfNet(arena, "mlp.", ops.BF16, H)
x = torch.relu(torch.randn(M, H, device=dev)).bfloat16()
CB = net.buf(M, H + net.Cw); CB.zero_()
CB[:, H:H + 27] = (torch.rand(M, 27, device=dev) * 2 - 1).bfloat16()
d_rgb = torch.randn(M, 3, device=dev) * 1e-3


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for fused in (True, False):
    net.fused_colour = fused
    net.ensure_packed(True)
    net.fwd("bottleneck", x, H, CB[:, :H], H)                      # the producer: writes CB[:, :H] and its ReLU bit mask
    bb = net._bits[(CB.data_ptr(), M)]
    raw = net.buf(M, 3, f32=True)

    def fwd(keep, variant=0):
        net._bits = {(CB.data_ptr(), M): bb} if keep else None
        if fused:
            net._colour_streams(keep)
            cys = cbits = None
            if keep:
                cys = [net.buf(M, 128) for _ in range(3)]
                cbits = [torch.empty(ops.mask_bits_words(M, 128), dtype=torch.int32, device=dev) for _ in range(3)]
            ops.fcolour_fwd(CB, net._cfwd[0], net._cfwd[1], raw, cys, cbits, variant=variant)
            return cys, cbits
        cx, ck, ys = CB, H + net.Cw, []
        for j in range(3):
            cy = net.buf(M, 128)
            net.fwd(f"cond_layers.{j}.layers.0", cx, ck, cy, 128)
            ys.append(cy); cx, ck = cy, 128
        net.fwd("rgb", cx, 128, raw, 3, ops.ACT_NONE, out_f32=True)
        return ys, None
    t_inf = timeit(lambda: fwd(False))
    t_trn = timeit(lambda: fwd(True))
    line = f"{'fused' if fused else 'per-layer'}: forward inference {t_inf:7.1f} us, training {t_trn:7.1f} us"
    if fused:
        line += f"; read-ahead 5 lines: inference {timeit(lambda: fwd(False, 1)):7.1f} us, training {timeit(lambda: fwd(True, 1)):7.1f} us"
    net._bits = {(CB.data_ptr(), M): bb}
    ys, cbits = fwd(True)
    DB = net.buf(M, H + 64)
    arena.grad.zero_()
    if fused:
        dCs = [net.buf(M, 128) for _ in range(3)]
        gb = [net.gB("cond_layers.2.layers.0"), net.gB("cond_layers.1.layers.0"), net.gB("cond_layers.0.layers.0"), net.gB("bottleneck_layer.layers.0")]
        t_b = timeit(lambda: ops.fcolour_bwd(d_rgb, net._cbwd[0], [cbits[2], cbits[1], cbits[0], bb[0]], dCs, DB[:, :H], gb))
    else:
        def bwd():
            dz = net.head_grad(d_rgb, 3)
            dC = net.buf(M, 128)
            net.dgrad("rgb", dz, dz.shape[1], dC, 128, mask=ys[2], colsum=net.gB("cond_layers.2.layers.0"))
            for j in (2, 1):
                dX = net.buf(M, 128)
                net.dgrad(f"cond_layers.{j}.layers.0", dC, 128, dX, 128, mask=ys[j - 1], colsum=net.gB(f"cond_layers.{j - 1}.layers.0"))
                dC = dX
            net.dgrad("cond_layers.0.layers.0", dC, 128, DB, H, mask=CB, colsum=net.gB("bottleneck_layer.layers.0"))
        t_b = timeit(bwd)
    print(line + f"; data-gradient chain {t_b:7.1f} us", flush=True)
print(f"(M = {M}: forward reads {M * 2176 / 1e9:.2f} GB, training forward also writes {3 * M * 256 / 1e9:.2f} GB; backward writes {M * (2048 + 768) / 1e9:.2f} GB)")
