#!/usr/bin/env python3
"""Seeded fuzz of the zipnerf (path C) callers in oracle/callers.py against the IMPORTED reference -- rows 8f-1 / 8f-2 of the S-NeRF++ side:
camera_utils.pixels_to_rays on random cameras / pixels, and the whole loss tail of the training step (train_utils.compute_data_loss,
the depth and semantic terms of train.py:252-298, anti_interlevel_loss, distortion_loss) with its gradients w.r.t. the renderer
outputs, on random histograms of the waymo.gin level sizes.  The two regularisers are compared against the reference evaluated in
FLOAT64 (its fp32 cumsum of the +/- steps of a blurred histogram carries ~1e-4 of noise of its own, see tests/test_oracle_callers_golden.py)
and, loosely, against its fp32 evaluation.  Sibling of oracle/fuzz_callers_vs_reference.py; TEST INFRASTRUCTURE, build container only.

    python oracle/fuzz_zip_callers_vs_reference.py --seeds 12 --log oracle/fuzz_zip_callers_vs_reference.log
"""
import argparse
import importlib.util
import os
import sys
import types

os.environ["TORCHDYNAMO_DISABLE"] = "1"
sys.dont_write_bytecode = True
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)

from oracle import callers as oc  # noqa: E402

WORST = {}


def note(stage, a, b, rtol, atol):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (stage, a.shape, b.shape)
    err = np.abs(a - b)
    worst = float(err.max()) if err.size else 0.0
    excess = float((err - rtol * np.abs(b)).max()) if err.size else 0.0
    w = WORST.setdefault(stage, {"n": 0, "abs": 0.0, "viol": 0, "bar": f"rtol {rtol:g} atol {atol:g}"})
    w["n"] += 1; w["abs"] = max(w["abs"], worst); w["viol"] += 0 if excess <= atol else 1


def fuzz_seed(seed, ref, gzc):
    cu, tu, stepfun, rmath = ref
    g = torch.Generator().manual_seed(40_000 + seed)
    rng = np.random.default_rng(500 + seed)
    # ---- f-2: pixels_to_rays
    ncam, n = 1 + seed % 3, 100 + 10 * seed
    Ks = np.stack([np.array([[1500 + 900 * rng.random(), 0., 900 + 120 * rng.random()], [0., 1500 + 900 * rng.random(), 600 + 80 * rng.random()], [0., 0., 1.]]) for _ in range(ncam)])
    pixtocams = np.linalg.inv(Ks).astype(np.float32)
    c2w = []
    for _ in range(ncam):
        th = rng.random() * 6 - 3
        c2w.append([[np.cos(th), 0.04, np.sin(th), rng.random() - 0.5], [-0.04, 1.0, 0.02, rng.random() * 0.2 - 0.1], [-np.sin(th), -0.02, np.cos(th), rng.random() * 0.3]])
    c2w = np.asarray(c2w, np.float32)
    cam_idx = rng.integers(0, ncam, n)
    px = rng.integers(0, 1920, n).astype(np.int32); py = rng.integers(0, 1280, n).astype(np.int32)
    out = cu.pixels_to_rays(px, py, pixtocams[cam_idx], c2w[cam_idx])
    r = oc.zip_pixels_to_rays(px, py, cam_idx.astype(np.int32), pixtocams, c2w)
    for k, want in zip(("origins", "directions", "viewdirs", "radii", "imageplane", "base_x", "base_y"), out):
        note("f-2 pixels_to_rays " + k, r[k], np.asarray(want, np.float32), 2e-7, 1e-9)
    # ---- f-1: loss tail
    R, C = 40 + 4 * seed, 19
    cfg = types.SimpleNamespace(data_loss_type="charb", charb_padding=0.001, disable_multiscale_loss=False, compute_disp_metrics=False,
                                compute_normal_metrics=False, data_coarse_loss_mult=0.0, data_loss_mult=1.0, anti_interlevel_loss_mult=0.01,
                                pulse_width=[0.03, 0.003], distortion_loss_mult=0.005)
    s0, w0 = gzc.histogram(g, R, 64)
    s1, w1 = gzc.histogram(g, R, 64)
    s2, w2 = gzc.histogram(g, R, 32, peaky=True)
    s2[: R // 2] = 0.4 + 0.2 * s2[: R // 2]
    s2[: R // 2, 0], s2[: R // 2, -1] = 0.0, 1.0
    rgb = torch.rand(R, 3, generator=g); tgt = torch.rand(R, 3, generator=g)
    mask_rgb = torch.rand(R, generator=g) < 0.8
    depth = torch.rand(R, generator=g) * 60 + 1
    tdepth = torch.rand(R, generator=g) * 60 + 1
    tdepth[torch.rand(R, generator=g) < 0.4] = 0
    sem = torch.softmax(torch.randn(R, C, generator=g) * 2, -1) * torch.rand(R, 1, generator=g)
    labels = torch.randint(0, C, (R,), generator=g)
    leaves = [t.clone().requires_grad_(True) for t in (rgb, depth, sem, w0, w1, w2)]
    rgb_, depth_, sem_, w0_, w1_, w2_ = leaves
    hist = [dict(sdist=s0, weights=w0_), dict(sdist=s1, weights=w1_), dict(sdist=s2, weights=w2_)]
    data, stats = tu.compute_data_loss(dict(rgb=tgt, mask_rgb=mask_rgb), [dict(rgb=rgb_)], cfg)
    depth_mask = torch.logical_and(tdepth > 0, mask_rgb)
    dep_lam = 0.5
    l_depth = dep_lam * torch.abs(1 / (depth_[depth_mask] + 1e-5) - 1 / (1e-5 + tdepth[depth_mask])).mean()
    l_sem = torch.nn.NLLLoss()(torch.log(sem_[mask_rgb] + 1e-6), labels[mask_rgb].long()) * 0.04
    l_inter = tu.anti_interlevel_loss(hist, cfg)
    l_dist = tu.distortion_loss(hist, cfg)
    grads = torch.autograd.grad(data + l_depth + l_sem + l_inter + l_dist, leaves)
    dd = lambda t: t.double()
    leaves64 = [dd(t).clone().requires_grad_(True) for t in (w0, w1, w2)]
    hist64 = [dict(sdist=dd(s0), weights=leaves64[0]), dict(sdist=dd(s1), weights=leaves64[1]), dict(sdist=dd(s2), weights=leaves64[2])]
    li64, ld64 = tu.anti_interlevel_loss(hist64, cfg), tu.distortion_loss(hist64, cfg)
    g64 = torch.autograd.grad(li64 + ld64, leaves64)
    n_ = lambda t: t.detach().numpy()
    mask = n_(mask_rgb).astype(np.float64)
    dmask = mask * (n_(tdepth) > 0)
    args = (0.001, 1.0, dep_lam, 0.2, 0.04, [0.03, 0.003], 0.01, 0.005)
    L, G = oc.zip_loss_tail(n_(rgb), n_(tgt), mask, n_(depth), n_(tdepth), dmask, None, n_(sem), n_(labels), mask, [n_(s0), n_(s1), n_(s2)], [n_(w0), n_(w1), n_(w2)], *args)
    for k, want in (("data", data), ("mse", stats["mses"][0]), ("depth", l_depth), ("sem", l_sem)):
        note("f-1 loss value: " + k, L[k], float(torch.as_tensor(want).detach()), 2e-5, 1e-9)
    for k, want in (("interlevel", l_inter), ("distortion", l_dist)):
        note("f-1 loss value vs the reference's fp32 evaluation: " + k, L[k], float(want.detach()), 2e-4, 1e-9)
    for k, want in (("rgb", grads[0]), ("depth", grads[1]), ("semantic", grads[2])):
        sc = float(np.abs(n_(want)).max())
        note("f-1 gradient / its largest entry: d/d " + k, G[k] / sc, n_(want) / sc, 3e-5, 3e-5)
    for k, want in (("w0", grads[3]), ("w1", grads[4]), ("w2", grads[5])):
        sc = float(np.abs(n_(want)).max())
        note("f-1 gradient / its largest entry vs the reference's fp32 evaluation: d/d " + k, G[k] / sc, n_(want) / sc, 0, 2e-3)
    d = lambda t: n_(t).astype(np.float64)
    L, G = oc.zip_loss_tail(n_(rgb), n_(tgt), mask, None, None, None, None, None, None, None, [d(s0), d(s1), d(s2)], [d(w0), d(w1), d(w2)], *args)
    note("f-1 anti-interlevel loss vs the reference in float64", L["interlevel"], float(li64), 1e-10, 0)
    note("f-1 distortion loss vs the reference in float64", L["distortion"], float(ld64), 1e-10, 0)
    for i, k in enumerate(("w0", "w1", "w2")):
        note("f-1 regulariser gradients vs the reference in float64", G[k], n_(g64[i]), 1e-8, 1e-14)
    note("f-1 lossfun_distortion per ray", oc.lossfun_distortion(n_(s2), n_(w2)), n_(stepfun.lossfun_distortion(s2, w2)), 2e-5, 1e-8)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=12)
    ap.add_argument("--log", default=None)
    args = ap.parse_args()
    spec = importlib.util.spec_from_file_location("gen_golden_zip_callers", os.path.join(HERE, "gen_golden_zip_callers.py"))
    gzc = importlib.util.module_from_spec(spec); spec.loader.exec_module(gzc)
    ref = gzc.import_reference()
    for seed in range(args.seeds):
        fuzz_seed(seed, ref, gzc)
    lines = [f"# oracle/callers.py (zipnerf side) vs imported reference, {args.seeds} seeds x random cameras / pixels / renderer outputs "
             f"(python oracle/fuzz_zip_callers_vs_reference.py --seeds {args.seeds})",
             f"# torch {torch.__version__}, numpy {np.__version__}; columns: stage | comparisons | worst |a - b| | bar | violations"]
    bad = 0
    for stage in sorted(WORST):
        w = WORST[stage]
        bad += w["viol"]
        lines.append(f"{stage:84s} | {w['n']:4d} | {w['abs']:.3e} | {w['bar']:28s} | {w['viol']}")
    lines.append(f"# total violations: {bad}")
    text = "\n".join(lines)
    print(text)
    if args.log:
        open(args.log, "w").write(text + "\n")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
