#!/usr/bin/env python3
"""The fused classic-network kernel alone (csrc/fmlp.hip), a few launches -- target for rocprofv3 --kernel-trace / --pmc passes.
Prints its own HIP-event timing too."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from snerf_amd import classic, ops
M = int(sys.argv[1]) if len(sys.argv) > 1 else 32768 * 192
torch.manual_seed(0)
net = classic.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=4, skips=[4], use_viewdirs=True, compute="bf16")
net.net._fused_ready()
E = (torch.rand(M, 64, device="cuda") * 2 - 1).bfloat16()
VE = (torch.rand(M, 64, device="cuda") * 2 - 1).bfloat16()
out = torch.empty(M, 4, device="cuda")
for _ in range(2):
    ops.fmlp_classic_fwd(E, VE, net.net.fstream, net.net.fbias, out)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    ops.fmlp_classic_fwd(E, VE, net.net.fstream, net.net.fbias, out)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f"fmlp_classic M={M}: {ms:.3f} ms  {M * 2 * 593408 / ms / 1e9:.1f} TFLOP/s algorithmic ({M * 1212416 / ms / 1e9:.1f} executed)")
S = 192
pts = torch.randn(M, 3, device="cuda")
vd = torch.nn.functional.normalize(torch.randn(M // S, 3, device="cuda"), dim=-1)
for _ in range(2):
    ops.fmlp_classic_pts_fwd(pts, vd, S, net.net.fstream, net.net.fbias, out)
e0.record()
for _ in range(5):
    ops.fmlp_classic_pts_fwd(pts, vd, S, net.net.fstream, net.net.fbias, out)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f"fmlp_classic_pts (in-kernel embedding) M={M}: {ms:.3f} ms  {M * 2 * 593408 / ms / 1e9:.1f} TFLOP/s algorithmic")
# training forward: the same launch + 4864 B / row of activation stores
acts = [torch.empty(M, 256, device="cuda", dtype=torch.bfloat16) for _ in range(9)] + [torch.empty(M, 128, device="cuda", dtype=torch.bfloat16)]
bits = [torch.empty(ops.mask_bits_words(M, 256), device="cuda", dtype=torch.int32) for _ in range(8)]
bits.append(torch.empty(ops.mask_bits_words(M, 128), device="cuda", dtype=torch.int32))       # views_linears.0
for _ in range(2):
    ops.fmlp_classic_train_fwd(E, VE, net.net.fstream, net.net.fbias, out, acts, bits)
e0.record()
for _ in range(5):
    ops.fmlp_classic_train_fwd(E, VE, net.net.fstream, net.net.fbias, out, acts, bits)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f"fmlp_classic_train M={M}: {ms:.3f} ms  {M * 2 * 593408 / ms / 1e9:.1f} TFLOP/s algorithmic, stores {M * 4864 / ms / 1e9:.2f} TB/s")
# what the part does with the same bytes as a plain write stream
e0.record()
for _ in range(5):
    for a in acts:
        a.zero_()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f"plain fill of the same buffers: {ms:.3f} ms  {M * 4864 / ms / 1e9:.2f} TB/s")
