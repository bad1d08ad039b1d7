#!/usr/bin/env python3
"""Seeded fuzz of oracle/callers.py (the rows either side of the path, SURVEY 8f-1 / 8f-2, S-NeRF side) against the IMPORTED reference:
ray generation (s-nerf/utils/sample_utils.py: sample_single_img for selected pixels, get_rays_single_img for whole frames) on random
cameras, poses and image sizes, and the loss terms (loss_factory.RgbLoss / DepthLoss with confidence / ProposalLoss) with their
gradients on random renderer outputs.  Sibling of oracle/fuzz_vs_reference.py; TEST INFRASTRUCTURE, build container only.

    python oracle/fuzz_callers_vs_reference.py --seeds 20 --log oracle/fuzz_callers_vs_reference.log
"""
import argparse
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True
REF = "/root/reference/s-nerf"

from oracle import callers as oc  # noqa: E402

WORST = {}


def note(stage, a, b, rtol, atol, exact=False):
    a = torch.as_tensor(a).detach().double(); b = torch.as_tensor(b).detach().double()
    assert a.shape == b.shape, (stage, a.shape, b.shape)
    err = (a - b).abs()
    worst_abs = float(err.max()) if err.numel() else 0.0
    excess = float((err - rtol * b.abs()).max()) if err.numel() else 0.0
    ok = (worst_abs == 0.0) if exact else (excess <= atol)
    w = WORST.setdefault(stage, {"n": 0, "abs": 0.0, "viol": 0, "bar": "bit-exact" if exact else f"rtol {rtol:g} atol {atol:g}"})
    w["n"] += 1; w["abs"] = max(w["abs"], worst_abs); w["viol"] += 0 if ok else 1


def import_reference():
    if not os.path.isdir(REF):
        raise SystemExit("reference tree not present")
    for name in ("turtle", "cv2", "imageio", "lpips", "kornia", "pyquaternion", "matplotlib", "matplotlib.pyplot", "nuscenes", "open3d", "skimage", "tqdm",
                 "torchvision", "torchvision.models", "torchvision.transforms", "scipy.spatial.transform"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                class _Stub(types.ModuleType):          # plotting / dataset packages the loss + ray code never calls
                    __path__ = []

                    def __getattr__(self, k):
                        if k.startswith("__"):
                            raise AttributeError(k)
                        return type(k, (), {})
                sys.modules[name] = _Stub(name)
    sys.path.insert(0, REF)
    import utils.sample_utils as su
    import model.loss_factory as lf
    return su, lf


def fuzz_seed(seed, su, lf):
    g = torch.Generator().manual_seed(30_000 + seed)
    R = lambda *s: torch.rand(*s, generator=g)
    args = types.SimpleNamespace(smooth_loss=False, no_ndc=True, N_rgb=100, encode_appearance=False, coarse_depth_mult=0.1 + 0.3 * float(R(1)),
                                 disparity_depth=bool(seed % 2 == 0), proposal_lambda=0.01 + 0.1 * float(R(1)))
    # ---- rays: a random small camera
    H, W = 20 + int(R(1) * 30), 24 + int(R(1) * 40)
    th, ph = float(R(1)) * 2 - 1, float(R(1)) * 0.4 - 0.2
    rot = torch.tensor([[np.cos(th), 0.0, np.sin(th)], [0.0, 1.0, 0.0], [-np.sin(th), 0.0, np.cos(th)]], dtype=torch.float64) @ \
        torch.tensor([[1.0, 0.0, 0.0], [0.0, np.cos(ph), -np.sin(ph)], [0.0, np.sin(ph), np.cos(ph)]], dtype=torch.float64)
    pose = torch.cat([rot, (torch.rand(3, 1, generator=g, dtype=torch.float64) * 4 - 2)], 1).float()
    fx, fy = 40 + 60 * float(R(1)), 40 + 60 * float(R(1))
    K = torch.tensor([[fx, 0.0, W * (0.4 + 0.2 * float(R(1)))], [0.0, fy, H * (0.4 + 0.2 * float(R(1)))], [0.0, 0.0, 1.0]], dtype=torch.float32)
    image = R(H, W, 3)
    depth = R(H, W) * 50 + 2
    depth[R(H, W) < 0.4] = 0
    n_sel = 50 + seed
    np.random.seed(100 + seed)
    rays, trgb, tdep, sel, _ = su.sample_single_img(args, image, depth, pose, K, near=2.0, far=100.0, near_far=False, batch_n=n_sel)
    full = su.get_rays_single_img(args, image, depth, pose, K, near=2.0, far=100.0, factor=1)
    for prefix, coords, ref in (("selected pixels", sel, rays),
                                ("whole frame", torch.stack(torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij"), -1).reshape(-1, 2), full)):
        near = float(ref.near.reshape(-1)[0]); far = float(ref.far.reshape(-1)[0])
        r = oc.pinhole_rays(coords, pose.numpy(), K.numpy(), H, near, far, training=prefix == "selected pixels", W=W)
        for k in ("origins", "directions", "viewdirs", "radii", "lossmult", "near", "far"):
            note(f"f-2 pinhole rays ({prefix}): {k}", r[k], getattr(ref, k).reshape(r[k].shape), 0, 0, exact=True)
    # ---- losses on random renderer outputs
    N, Sc, Pf = 40 + seed, 16 + seed % 20, 17 + seed % 13

    def fence(n, P):
        s = torch.sort(torch.rand(n, P, generator=g), dim=-1).values
        s[:, 0] = 0.0; s[:, -1] = 1.0
        return s
    s_c, s_f = fence(N, Sc + 1), fence(N, Pf)
    if Pf <= Sc + 1:
        s_f[:5] = s_c[:5, :Pf]; s_f[:5, -1] = 1.0; s_f[:5] = torch.sort(s_f[:5], -1).values       # ties between the two grids
    w_c = R(N, Sc) ** 4; w_c = (w_c / w_c.sum(-1, keepdim=True) * R(N, 1)).requires_grad_(True)
    w_f = R(N, Pf - 1) ** 6; w_f = w_f / w_f.sum(-1, keepdim=True) * R(N, 1)
    w_f[3] = 0.0
    pl = lf.ProposalLoss(args)(s_f, w_f, s_c, w_c)
    g_wc, = torch.autograd.grad(pl, w_c)
    w_c2 = w_c.detach().clone().requires_grad_(True)
    opl = oc.proposal_loss(s_f, w_f, s_c, w_c2, args.proposal_lambda)
    og, = torch.autograd.grad(opl, w_c2)
    note("f-1 ProposalLoss value", opl, pl, 1e-6, 1e-8); note("f-1 ProposalLoss d/d coarse weights", og, g_wc, 1e-5, 1e-8)
    rgb = R(N, 3).requires_grad_(True); tgt = R(N, 3)
    rl = lf.RgbLoss(args)(rgb, tgt); g_rgb, = torch.autograd.grad(rl, rgb)
    rgb2 = rgb.detach().clone().requires_grad_(True)
    orl = oc.rgb_loss(rgb2, tgt); og, = torch.autograd.grad(orl, rgb2)
    note("f-1 RgbLoss value", orl, rl, 0, 0, exact=True); note("f-1 RgbLoss gradient", og, g_rgb, 0, 0, exact=True)
    d1 = (R(N) * 60 + 2).requires_grad_(True); d0 = (R(N) * 60 + 2).requires_grad_(True)
    td = R(N) * 78 + 2; td[R(N) < 0.5] = 0
    conf = R(N)
    mask = td != 0
    dl = lf.DepthLoss(args)(d1.unsqueeze(-1)[mask.unsqueeze(-1)], d0.unsqueeze(-1)[mask.unsqueeze(-1)], td.unsqueeze(-1)[mask.unsqueeze(-1)])
    dl = (dl * conf[mask]).mean()
    g_d1, g_d0 = torch.autograd.grad(dl, [d1, d0])
    e1, e0 = d1.detach().clone().requires_grad_(True), d0.detach().clone().requires_grad_(True)
    odl = oc.depth_loss(e1, e0, td, conf, args.coarse_depth_mult, args.disparity_depth)
    o1, o0 = torch.autograd.grad(odl, [e1, e0])
    note("f-1 DepthLoss x confidence value", odl, dl, 1e-6, 1e-8)
    note("f-1 DepthLoss gradients (fine, coarse)", o1, g_d1, 1e-5, 1e-10); note("f-1 DepthLoss gradients (fine, coarse)", o0, g_d0, 1e-5, 1e-10)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=20)
    ap.add_argument("--log", default=None)
    args = ap.parse_args()
    su, lf = import_reference()
    for seed in range(args.seeds):
        fuzz_seed(seed, su, lf)
    lines = [f"# oracle/callers.py vs imported reference (s-nerf ray generation + loss modules), {args.seeds} seeds x random cameras / image sizes / renderer outputs "
             f"(python oracle/fuzz_callers_vs_reference.py --seeds {args.seeds})",
             f"# torch {torch.__version__}, numpy {np.__version__}; columns: stage | comparisons | worst |a - b| | bar | violations"]
    bad = 0
    for stage in sorted(WORST):
        w = WORST[stage]
        bad += w["viol"]
        lines.append(f"{stage:58s} | {w['n']:4d} | {w['abs']:.3e} | {w['bar']:24s} | {w['viol']}")
    lines.append(f"# total violations: {bad}")
    text = "\n".join(lines)
    print(text)
    if args.log:
        open(args.log, "w").write(text + "\n")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
