#!/usr/bin/env python3
"""Determinism screen of every fused-MLP kernel variant: repeated launches on identical inputs must give bit-identical outputs (and
stored activations / bit masks for the training variants)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from snerf_amd import classic, ops, mlp
from snerf_amd.mlp import ParamArena

ITERS = int(os.environ.get("ITERS", "3000"))
torch.manual_seed(0)
net = classic.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=4, skips=[4], use_viewdirs=True, compute="bf16", device="cuda")
n = net.net
n._fused_ready()
shapes = mlp.MipProposalNet.param_shapes(256, 4, 96)
arena = ParamArena(shapes, torch.device("cuda"))
arena.load({k: torch.randn(s) * (1.4 / s[-1] ** 0.5 if len(s) == 2 else 0.1) for k, s in shapes})
prop = mlp.MipProposalNet(arena, "", ops.BF16, 256, 4, 96)
prop._fused_ready()
M, S = 4096, 8
pts = torch.rand(M, 3, device="cuda") * 4 - 2
vd = torch.nn.functional.normalize(torch.randn(M // S, 3, device="cuda"), dim=-1)
E, VE = n.buf(M, 64), n.buf(M, 64)
ops.classic_embed(pts, vd, S, 10, 4, E, None, 64, VE, 64, n.dt)
EP = torch.zeros(M, prop.Ew, dtype=torch.bfloat16, device="cuda"); EP[:, :96] = torch.randn(M, 96, device="cuda") * 0.5
big = torch.randn(2048, 2048, device="cuda").bfloat16()


def classic_plain():
    o = n.buf(M, 4, f32=True); ops.fmlp_classic_fwd(E, VE, n.fstream, n.fbias, o); return [o]
def classic_pts():
    o = n.buf(M, 4, f32=True); ops.fmlp_classic_pts_fwd(pts, vd, S, n.fstream, n.fbias, o); return [o]
def classic_train():
    o = n.buf(M, 4, f32=True)
    acts = [n.buf(M, 256) for _ in range(9)] + [n.buf(M, 128)]
    bits = [torch.zeros(ops.mask_bits_words(M, 256), dtype=torch.int32, device="cuda") for _ in range(8)]
    ops.fmlp_classic_train_fwd(E, VE, n.fstream, n.fbias, o, acts, bits); return [o] + acts + bits
def prop_plain():
    o = prop.buf(M, 1, f32=True); ops.fmlp_proposal_fwd(EP, prop.fstream, prop.fbias, o); return [o]
def prop_train():
    o = prop.buf(M, 1, f32=True)
    acts = [prop.buf(M, 256) for _ in range(4)]
    bits = [torch.zeros(ops.mask_bits_words(M, 256), dtype=torch.int32, device="cuda") for _ in range(4)]
    ops.fmlp_proposal_train_fwd(EP, prop.fstream, prop.fbias, o, acts, bits); return [o] + acts + bits

total_bad = 0
for name, fn in (("classic_pts (EMBED)", classic_pts), ("classic_plain", classic_plain), ("classic_train", classic_train), ("proposal_plain", prop_plain), ("proposal_train", prop_train)):
    ref, bad = None, 0
    for it in range(ITERS):
        out = fn()
        if it % 3 == 0:
            torch.mm(big, big)
        if ref is None:
            ref = [t.clone() for t in out]
        elif not all(torch.equal(a, b) for a, b in zip(out, ref)):
            bad += 1
    print(f"{name:22s} {bad} of {ITERS - 1} repeats differ", flush=True)
    total_bad += bad
sys.exit(0 if total_bad == 0 else 1)
