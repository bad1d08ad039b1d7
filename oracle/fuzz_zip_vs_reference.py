#!/usr/bin/env python3
"""Seeded fuzz of oracle/zip.py (path C: everything around the hash grid) against the IMPORTED zipnerf reference, beyond the fixed
golden vectors -- the path-C sibling of oracle/fuzz_vs_reference.py.

TEST INFRASTRUCTURE, build container only (needs /root/reference; imported with the same stubs as oracle/gen_golden_zip.py, the
oracle's grid encoder injected for the un-runnable CUDA one, nothing copied).  Per seed: fresh inputs for every stage (ray warp,
dilation, interval sampling deterministic + single-jitter replay, helix multisampling deterministic + jitter replay, contraction,
alpha weights, volumetric rendering) and a whole `Model.forward` (3 levels, deterministic and randomized with the reference's RNG
draws replayed) on RANDOM hash tables and RANDOM network weights.

    python oracle/fuzz_zip_vs_reference.py --seeds 12 --log oracle/fuzz_zip_vs_reference.log
"""
import argparse
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True
os.environ.setdefault("TORCHDYNAMO_DISABLE", "1")

from oracle import zip as oz  # noqa: E402

WORST = {}


def note(stage, a, b, rtol, atol):
    a = a.detach().double(); b = b.detach().double()
    assert a.shape == b.shape, (stage, a.shape, b.shape)
    nan_ok = torch.equal(torch.isnan(a), torch.isnan(b))
    a = torch.nan_to_num(a, nan=0.0, posinf=3e38, neginf=-3e38); b = torch.nan_to_num(b, nan=0.0, posinf=3e38, neginf=-3e38)
    err = (a - b).abs()
    excess = err - rtol * b.abs()
    score = float(excess.max()) if excess.numel() else 0.0
    w = WORST.setdefault(stage, {"n": 0, "abs": 0.0, "worst": 0.0, "viol": 0, "bar": f"rtol {rtol:g} atol {atol:g}"})
    w["n"] += 1
    w["abs"] = max(w["abs"], float(err.max()) if err.numel() else 0.0)
    w["worst"] = max(w["worst"], score)
    w["viol"] += 0 if (nan_ok and score <= atol) else 1


def random_params(shapes, g):
    p = {}
    for k, s in shapes:
        if k.endswith("embeddings"):
            p[k] = (torch.rand(s, generator=g) - 0.5) * 0.8
        elif len(s) == 2:
            p[k] = torch.randn(s, generator=g) * (1.2 / s[1] ** 0.5)
        else:
            p[k] = torch.randn(s, generator=g) * 0.1
    return p


def fuzz_seed(seed, ref, gz):
    coord, rmath, models, render, stepfun = ref
    g = torch.Generator().manual_seed(20_000 + seed)
    R = lambda *s: torch.rand(*s, generator=g)
    nr = 5 + seed % 4
    near, far = torch.full((nr, 1), 0.05 + 0.2 * float(R(1))), torch.full((nr, 1), 5.0 + 20 * float(R(1)))
    _, s2t = coord.construct_ray_warps('power_transformation', near, far, -1.5)
    s = torch.sort(R(nr, 9), -1)[0]; s[:, 0] = 0; s[:, -1] = 1
    note("C1 ray warp s -> t", oz.s_to_t(s, near, far, -1.5), s2t(s), 1e-6, 1e-7)
    nt = 10 + seed % 5
    t = torch.sort(R(6, nt + 1), -1)[0]; t[:, 0] = 0; t[:, -1] = 1
    w = R(6, nt) ** 3; w[0] = 0; w[0, seed % nt] = 1; w = w / w.sum(-1, keepdim=True)
    dil = 0.002 + 0.02 * float(R(1))
    td, wd = stepfun.max_dilate_weights(t, w, dil, domain=(0., 1.), renormalize=True)
    otd, owd = oz.max_dilate_weights(t, w, dil, (0.0, 1.0))
    note("C2 max_dilate_weights t", otd, td, 0, 0); note("C2 max_dilate_weights w", owd, wd, 1e-6, 1e-8)
    logits = torch.where(t[..., 1:] > t[..., :-1], 0.7 * torch.log(w + 0.0), torch.full_like(w, -float("inf")))
    ns = 12 + seed % 7
    sd_det = stepfun.sample_intervals(None, t, logits, ns, single_jitter=True, domain=(0., 1.))
    torch.manual_seed(300 + seed)
    sd_rnd = stepfun.sample_intervals(True, t, logits, ns, single_jitter=True, domain=(0., 1.))
    torch.manual_seed(300 + seed)
    jit = torch.rand(6, 1)
    note("C3 sample_intervals (deterministic centres)", oz.sample_intervals(t, logits, oz.det_centers_u(ns))[0], sd_det, 1e-6, 1e-7)
    note("C3 sample_intervals (single jitter, RNG replayed)", oz.sample_intervals(t, logits, oz.rand_u(ns, jit))[0], sd_rnd, 1e-6, 1e-7)
    b = gz.make_batch(7, 500 + seed)
    tdist = torch.sort(R(7, 6) * 5 + 0.2, -1)[0]
    m0, s0 = render.cast_rays(tdist, b["origins"], b["directions"], b["radii"], rand=False, n=7, m=3, std_scale=0.35, batch=b)
    torch.manual_seed(400 + seed)
    m1, s1 = render.cast_rays(tdist, b["origins"], b["directions"], b["radii"], rand=True, n=7, m=3, std_scale=0.35, batch=b)
    torch.manual_seed(400 + seed)
    degj = torch.rand(7, 5, 7)
    m, sdd = oz.cast_rays(tdist, b["origins"], b["directions"], b["radii"], b["base_x"], b["base_y"], None)
    note("C4 cast_rays means (deterministic helix)", m, m0, 1e-6, 1e-6); note("C4 cast_rays stds", sdd, s0, 1e-6, 1e-9)
    m, sdd = oz.cast_rays(tdist, b["origins"], b["directions"], b["radii"], b["base_x"], b["base_y"], degj)
    note("C4 cast_rays means (rotation jitter replayed)", m, m1, 1e-6, 1e-6)
    x = torch.randn(40, 3, generator=g) * torch.tensor([0.3, 1.0, 4.0 + seed]); st = R(40) * 0.05
    zc, sc = coord.contract_mean_std(x, st)
    oz_z, oz_s = oz.contract_mean_std(x, st)
    note("C5 contract_mean_std z", oz_z, zc, 1e-6, 1e-7); note("C5 contract_mean_std std", oz_s, sc, 1e-6, 1e-9)
    dens = R(7, 5) * 3; dens[0] = 0
    for opaque in (True, False):
        wts = render.compute_alpha_weights(dens, tdist, b["directions"], opaque_background=opaque)[0]
        note("C9 compute_alpha_weights", oz.compute_alpha_weights(dens, tdist, b["directions"], opaque), wts, 1e-6, 1e-7)
        rgbs = R(7, 5, 3)
        bg = 1.0 if opaque else 0.5
        rend = render.volumetric_rendering(rgbs, wts, tdist, bg, b["far"], False)
        orend = oz.volumetric_rendering(rgbs, wts, tdist, bg)
        note("C9 volumetric_rendering rgb", orend["rgb"], rend["rgb"], 1e-6, 1e-6); note("C9 volumetric_rendering depth", orend["depth"], rend["depth"], 1e-6, 1e-6)

    # ---- C11 Model.forward: random tables and weights, deterministic + randomized (RNG replayed), + compute_extras
    specs = gz.small_specs()
    cfg = types.SimpleNamespace(use_semantic=False, vis_num_rays=8, zero_glo=True)
    torch.manual_seed(seed)
    model = models.Model(config=cfg, raydist_fn='power_transformation', opaque_background=True)
    model.nerf_mlp = models.NerfMLP(disable_density_normals=True, deg_view=1, grid_log2_hashmap_size=14, use_semantic=False)
    model.prop_mlp_0 = models.PropMLP(disable_density_normals=True, disable_rgb=True, grid_level_dim=1, grid_disired_resolution=512, grid_log2_hashmap_size=14)
    model.prop_mlp_1 = models.PropMLP(disable_density_normals=True, disable_rgb=True, grid_level_dim=1, grid_disired_resolution=2048, grid_log2_hashmap_size=14)
    shapes = oz.param_shapes(specs)
    p = random_params(shapes, g)
    model.load_state_dict(p, strict=False)
    n = 12
    batch = gz.make_batch(n, 900 + seed)
    frac = 0.2 + 0.7 * float(R(1))
    with torch.no_grad():
        rend, hist = model(None, dict(batch), train_frac=1.0, compute_extras=True)
        torch.manual_seed(600 + seed)
        rend_r, hist_r = model(True, dict(batch), train_frac=frac, compute_extras=False)
    torch.manual_seed(600 + seed)
    jit, degj = [], []
    for nsamp in (64, 64, 32):
        jit.append(torch.rand(n, 1)); degj.append(torch.rand(n, nsamp, 7))
    orend, ohist = oz.model_forward(p, specs, batch, train_frac=1.0, compute_extras=True, vis_num_rays=8)
    for lvl in range(3):
        note("C11 Model.forward deterministic: sdist", ohist[lvl]["sdist"], hist[lvl]["sdist"], 1e-5, 1e-6)
        note("C11 Model.forward deterministic: weights", ohist[lvl]["weights"], hist[lvl]["weights"], 1e-4, 1e-6)
        for k in ("acc", "distance_mean", "distance_percentile_5", "distance_median", "distance_percentile_95"):
            note("C11 compute_extras: acc / distance mean / percentiles", orend[lvl][k], rend[lvl][k], 2e-5, 2e-6)
    note("C11 Model.forward deterministic: rgb", orend[-1]["rgb"], rend[-1]["rgb"], 1e-5, 1e-5)
    note("C11 Model.forward deterministic: depth", orend[-1]["depth"], rend[-1]["depth"], 1e-5, 1e-5)
    orend, ohist = oz.model_forward(p, specs, batch, train_frac=frac, jitters=jit, deg_jitters=degj)
    for lvl in range(3):
        note("C11 Model.forward randomized (draws replayed): sdist", ohist[lvl]["sdist"], hist_r[lvl]["sdist"], 1e-5, 1e-6)
    note("C11 Model.forward randomized (draws replayed): rgb", orend[-1]["rgb"], rend_r[-1]["rgb"], 1e-5, 1e-5)
    note("C11 Model.forward randomized (draws replayed): depth", orend[-1]["depth"], rend_r[-1]["depth"], 1e-5, 1e-5)
    cfg.use_semantic = True
    model.config = cfg
    model.nerf_mlp.use_semantic = True
    with torch.no_grad():
        rend_s, _ = model(None, dict(batch), train_frac=1.0, compute_extras=False)
    osem, _ = oz.model_forward(p, specs, batch, train_frac=1.0, use_semantic=True)
    note("C8 semantic head (19 classes)", osem[-1]["semantic"], rend_s[-1]["semantic"], 1e-5, 1e-6)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=12)
    ap.add_argument("--log", default=None)
    args = ap.parse_args()
    spec = importlib.util.spec_from_file_location("gen_golden_zip", os.path.join(HERE, "gen_golden_zip.py"))
    gz = importlib.util.module_from_spec(spec); spec.loader.exec_module(gz)
    ref = gz.import_reference()
    for seed in range(args.seeds):
        fuzz_seed(seed, ref, gz)
    lines = [f"# oracle/zip.py vs imported zipnerf reference, {args.seeds} seeds x fresh inputs, random hash tables and weights (python oracle/fuzz_zip_vs_reference.py --seeds {args.seeds})",
             f"# torch {torch.__version__}, numpy {np.__version__}; columns: stage | comparisons | worst |a - b| | worst excess of |a - b| over rtol*|ref| (must stay <= atol) | bar | violations",
             "# (the grid lookup inside the reference Model is the oracle's own encoder, injected for the CUDA extension that cannot build here: this pins everything AROUND the grid)"]
    bad = 0
    for stage in sorted(WORST):
        w = WORST[stage]
        bad += w["viol"]
        lines.append(f"{stage:62s} | {w['n']:4d} | {w['abs']:.3e} | {w['worst']:.3e} | {w['bar']:24s} | {w['viol']}")
    lines.append(f"# total violations: {bad}")
    text = "\n".join(lines)
    print(text)
    if args.log:
        open(args.log, "w").write(text + "\n")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
