#!/usr/bin/env python3
"""Seeded fuzz of the CPU oracle against the IMPORTED reference (paths A and B), beyond the fixed golden vectors.

TEST INFRASTRUCTURE, build container only: needs /root/reference (imported exactly as oracle/gen_golden.py does, nothing copied) and
therefore never runs on the GPU box.  For every seed it draws fresh inputs AND fresh random network weights (the goldens use formula
weights), runs each stage through the reference and through the oracle's restatement, and records the worst deviation per stage:

    python oracle/fuzz_vs_reference.py --seeds 25            # prints one line per stage, exits 1 on a violation
    python oracle/fuzz_vs_reference.py --seeds 25 --log oracle/fuzz_vs_reference.log   # the committed summary

Bars: encodings / fence posts / sample indices bit-exact; everything else within the tolerance written next to the stage (the same
ones tests/test_oracle_golden.py uses for the goldens).
"""
import argparse
import importlib.util
import os
import sys
from collections import namedtuple

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True

from oracle import classic as oc  # noqa: E402
from oracle import common, mip as om  # noqa: E402

EPS32 = float(torch.finfo(torch.float32).eps)
WORST = {}


def note(stage, a, b, rtol, atol, exact=False):
    """records max(|a - b| - rtol |b|) relative to atol; returns nothing, violations are collected"""
    a = a.detach().double(); b = b.detach().double()
    assert a.shape == b.shape, (stage, a.shape, b.shape)
    nan_ok = torch.equal(torch.isnan(a), torch.isnan(b))
    a = torch.nan_to_num(a, nan=0.0, posinf=3e38, neginf=-3e38); b = torch.nan_to_num(b, nan=0.0, posinf=3e38, neginf=-3e38)
    err = (a - b).abs()
    worst_abs = float(err.max()) if err.numel() else 0.0
    if exact:
        ok = nan_ok and worst_abs == 0.0
        score = worst_abs
    else:
        excess = err - rtol * b.abs()
        score = float(excess.max()) if err.numel() else 0.0
        ok = nan_ok and score <= atol
    w = WORST.setdefault(stage, {"n": 0, "abs": 0.0, "worst": 0.0, "viol": 0, "bar": "bit-exact" if exact else f"rtol {rtol:g} atol {atol:g}"})
    w["n"] += 1
    w["abs"] = max(w["abs"], worst_abs)
    w["worst"] = max(w["worst"], score)
    w["viol"] += 0 if ok else 1


def rand_sd(shapes, g, w_gain=1.4, b_std=0.1, boost=()):
    sd = {}
    for k, s in shapes:
        if len(s) == 2:
            sd[k] = torch.randn(s, generator=g) * (w_gain / s[1] ** 0.5)
        else:
            sd[k] = torch.randn(s, generator=g) * b_std + (0.5 if k in boost else 0.0)
    return sd


def fuzz_seed(seed, ref):
    math_ops, mip, models, helpers, render = ref
    g = torch.Generator().manual_seed(10_000 + seed)
    R = lambda *s: torch.rand(*s, generator=g)
    RN = lambda *s: torch.randn(*s, generator=g)

    # ---- encodings (A10, B2): bit-exact
    x = R(40, 3) * 8 - 4
    note("A10 pos_enc (view dirs)", om.pos_enc(x, 0, 4, True), mip.pos_enc(x, 0, 4, True), 0, 0, exact=True)
    note("B2 Embedder 10 / 4", oc.embed(x, 10), helpers.get_embedder(10, 0)[0](x), 0, 0, exact=True)
    note("B2 Embedder 10 / 4", oc.embed(x, 4), helpers.get_embedder(4, 0)[0](x), 0, 0, exact=True)

    # ---- A2-A6: transform, cast_rays, sample2enc
    n, s = 24, 20
    o = RN(n, 3) * (0.3 if seed % 3 else 3.0)                  # every third seed: origins far out (contracted region)
    d = RN(n, 3); d = d / d.norm(dim=-1, keepdim=True) * (1 + 0.3 * R(n, 1))
    radii = 1e-3 + 4e-3 * R(n, 1)
    near, far = torch.full((n, 1), 1.8 + float(R(1))), torch.full((n, 1), 90.0 + 30 * float(R(1)))
    sv = torch.sort(R(n, s + 1), -1)[0]; sv[:, 0] = 0; sv[:, -1] = 1
    tv = mip.Transform_log(sv, near, far)
    note("A2 Transform_log", om.transform(sv, near, far, 0), tv, 0, 0, exact=True)
    for shape in ("cone", "cylinder"):
        m_r, c_r = mip.cast_rays(tv, o, d, radii, shape)
        m_o, c_o = om.cast_rays(tv, o, d, radii, shape)
        note("A3 cast_rays mean", m_o, m_r, 1e-6, 1e-6); note("A3 cast_rays cov", c_o, c_r, 1e-5, 1e-9)
    fm_r, fc_r = mip.sample2enc(sv, o, d, radii, "cone", near, far, s, 1, viewc=0., radius=3., transform_idx=0)
    fm_o, fc_o = om.sample2enc(sv, o, d, radii, near, far, "cone", 0, full_cov=True)
    note("A4-A6 sample2enc means", fm_o, fm_r, 1e-6, 1e-6); note("A4-A6 sample2enc covs", fc_o, fc_r, 2e-5, 1e-9)

    # ---- A7 IPE: bit-exact
    xm = RN(6, 12, 3) * (1.5 if seed % 2 else 40.0)            # large arguments: the safe_sin reduction branch
    cd = R(6, 12, 3) * 1e-3
    note("A7 integrated_pos_enc", om.integrated_pos_enc(xm, cd, 0, 16), mip.integrated_pos_enc((xm, cd), 0, 16, diag=True, device="cpu"), 0, 0, exact=True)

    # ---- A14 sorted_piecewise_constant_pdf: samples + (through them) the interval index
    nb = 24 + seed % 9
    bins = torch.sort(R(10, nb + 1), -1)[0]
    w = R(10, nb) ** 3
    w[0] = 0.0; w[1] = 0.0; w[1, seed % nb] = 1.0; w[2] = 0.25; w[3, : nb // 2] = 0.0
    for num in (nb, nb + 1):
        det = math_ops.sorted_piecewise_constant_pdf(bins, w, num, False)
        torch.manual_seed(100 + seed)
        rnd = math_ops.sorted_piecewise_constant_pdf(bins, w, num, True)
        torch.manual_seed(100 + seed)
        jit = torch.empty(10, num).uniform_(to=1 / num - EPS32)
        s_o, idx_d = om.sorted_piecewise_constant_pdf(bins, w, om.det_u(num), sum_mode="torch")
        note("A14 pdf samples (deterministic)", s_o, det, 0, 1e-7)
        s_o, idx_r = om.sorted_piecewise_constant_pdf(bins, w, om.rand_u(num, jit), sum_mode="torch")
        note("A14 pdf samples (randomized, RNG replayed)", s_o, rnd, 0, 1e-7)
        # the reference returns no index: the oracle's index is right iff the reference's sample lies inside that interval
        lo = torch.gather(bins, 1, idx_r.long()); hi = torch.gather(bins, 1, idx_r.long() + 1)
        inside = ((rnd >= lo - 1e-6) & (rnd <= hi + 1e-6)).double()
        note("A14 interval index consistent with the reference's samples", inside, torch.ones_like(inside), 0, 0, exact=True)
        # the canonical summation order (what the HIP kernel implements) picks the same intervals
        _, idx_c = om.sorted_piecewise_constant_pdf(bins, w, om.rand_u(num, jit))
        note("A14 canonical-order index == reference-order index", idx_c.double(), idx_r.double(), 0, 0, exact=True)

    # ---- A12 volumetric rendering
    n, s = 10, 18
    rgb = R(n, s, 3); dens = R(n, s, 1) * 2
    dens[0] = 0.0; dens[1] = 1e4
    sv6 = torch.sort(R(n, s + 1), -1)[0]; d6 = RN(n, 3)
    near6, far6 = torch.full((n, 1), 1.8), torch.full((n, 1), 110.0)
    sem = RN(n, s, 5)
    for wb in (False, True):
        r = mip.real_volumetric_rendering(rgb, dens, sv6, d6, sem, wb, near6, far6, 0)
        q = om.volumetric_rendering(rgb, dens, sv6, d6, near6, far6, wb, sem, 0)
        for name, a, b in zip(("rgb", "distance", "acc", "weights", "semantic"), q, r):
            note("A12 real_volumetric_rendering " + name, a, b, 1e-6, 1e-6)

    # ---- A8 / A9 networks + A15 the whole forward, random weights
    torch.manual_seed(seed)
    model = models.MipNerfModel(n_samples=16, N_fine=17, no_warp_sample=0, ray_shape="cone", fn=1, radius=3., transform_idx=0, real=True,
                                rgb_layer=3, hidden_layer=64, density_noise=0., max_deg_point=16, proposal_hidden_layer=64, proposal_loss=True)
    shapes = om.mipnerf_param_shapes(hidden=64, prop_hidden=64)
    assert [k for k, _ in shapes] == list(model.state_dict().keys())
    sd = rand_sd(shapes, g, boost=("mlp.density_layer.bias", "proposal.density_layer.bias"))
    model.load_state_dict(sd)
    enc = R(5, 4, 96) * 2 - 1; cond = R(5, 27) * 2 - 1
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    for prm in model.parameters():
        prm.grad = None
    rr, rd, _ = model.mlp(enc, cond); pr = model.proposal(enc)[1]
    ((rr ** 2).sum() + (rd * 0.5).sum() + (pr ** 2).sum()).backward()
    orr, ord_, _ = om.nerf_mlp(p, enc, cond); opr = om.proposal_mlp(p, enc)
    ((orr ** 2).sum() + (ord_ * 0.5).sum() + (opr ** 2).sum()).backward()
    note("A9 MLP raw_rgb / raw_density", orr, rr, 1e-5, 1e-5); note("A9 MLP raw_rgb / raw_density", ord_, rd, 1e-5, 1e-5)
    note("A8 proposal density", opr, pr, 1e-5, 1e-5)
    for k, v in model.named_parameters():
        if v.grad is not None:
            note("A8 / A9 parameter gradients (autograd through both)", p[k].grad, v.grad, 1e-4, 1e-4)
    rays = common.synthetic_rays(32, seed=50 + seed)
    Rays = namedtuple("Rays", ("origins", "directions", "viewdirs", "radii", "lossmult", "near", "far", "app"))
    with torch.no_grad():
        ret = model(Rays(**rays), False, False, 0.)
        torch.manual_seed(200 + seed)
        rnd_ret = model(Rays(**rays), True, False, 0.)
    torch.manual_seed(200 + seed)
    s_rand = torch.rand(32, 17); jit = torch.empty(32, 17).uniform_(to=1 / 17 - EPS32)
    for tag, rr_, oo in (("deterministic", ret, om.mipnerf_forward(sd, rays, 16, 17)),
                         ("randomized", rnd_ret, om.mipnerf_forward(sd, rays, 16, 17, s_rand=s_rand, u=om.rand_u(17, jit)))):
        note(f"A15 forward {tag}: level-0 fence posts", oo[0][3], rr_[0][3], 0, 0, exact=True)
        note(f"A15 forward {tag}: level-0 weights / distance / acc", oo[0][4], rr_[0][4], 1e-5, 1e-7)
        note(f"A15 forward {tag}: level-0 weights / distance / acc", oo[0][1], rr_[0][1], 1e-5, 1e-5)
        note(f"A15 forward {tag}: level-1 fence posts", oo[1][4], rr_[1][4], 1e-5, 1e-6)
        note(f"A15 forward {tag}: rgb / distance / acc / weights", oo[1][0], rr_[1][0], 1e-5, 1e-5)
        note(f"A15 forward {tag}: rgb / distance / acc / weights", oo[1][1], rr_[1][1], 1e-5, 1e-5)
        note(f"A15 forward {tag}: rgb / distance / acc / weights", oo[1][2], rr_[1][2], 1e-5, 1e-6)
        note(f"A15 forward {tag}: rgb / distance / acc / weights", oo[1][5], rr_[1][5], 1e-5, 1e-6)

    # ---- path B: raw2outputs, sample_pdf (+ indices), render_rays
    n, s = 9, 14
    raw = RN(n, s, 4); z = torch.sort(R(n, s) * 4 + 2, -1)[0]; rd9 = RN(n, 3)
    for wb in (False, True):
        r = helpers.raw2outputs(raw, z, rd9, 0, wb)
        q = oc.raw2outputs(raw, z, rd9, white_bkgd=wb)
        for name, a, b in zip(("rgb_map", "disp_map", "acc_map", "weights", "depth_map"), q, r):
            note("B5 raw2outputs " + name, a, b, 1e-6, 1e-6)
    bins9 = torch.sort(R(n, 15), -1)[0]; w9 = R(n, 14) ** 2; w9[0] = 0.0
    det9 = helpers.sample_pdf(bins9, w9, 24, det=True)
    rnd9 = helpers.sample_pdf(bins9, w9, 24, det=False, pytest=True)     # pytest=True: the reference seeds numpy with 0 itself
    np.random.seed(0); u9 = torch.Tensor(np.random.rand(n, 24))

    def ref_inds(bins, weights, u):          # the reference's own index expression (run_nerf_helpers.py:362), evaluated on its tensors
        ww = weights + 1e-5
        pdf = ww / torch.sum(ww, -1, keepdim=True)
        cdf = torch.cat([torch.zeros_like(pdf[..., :1]), torch.cumsum(pdf, -1)], -1)
        return torch.searchsorted(cdf, u.contiguous(), right=True)
    for tag, u, want in (("det", torch.linspace(0., 1., 24).expand(n, 24).contiguous(), det9), ("rand", u9, rnd9)):
        s_o, i_o = oc.sample_pdf(bins9, w9, u, sum_mode="torch")
        note(f"B6 sample_pdf samples ({tag})", s_o, want, 1e-6, 1e-6)
        note(f"B6 sample_pdf searchsorted indices ({tag})", i_o.double(), ref_inds(bins9, w9, u).double(), 0, 0, exact=True)
    torch.manual_seed(seed)
    coarse = helpers.NeRF(D=8, W=64, input_ch=63, input_ch_views=27, output_ch=4, skips=[4], use_viewdirs=True)
    fine = helpers.NeRF(D=8, W=64, input_ch=63, input_ch_views=27, output_ch=4, skips=[4], use_viewdirs=True)
    pc = rand_sd(oc.nerf_param_shapes(W=64), g, boost=("alpha_linear.bias",)); pf = rand_sd(oc.nerf_param_shapes(W=64), g, boost=("alpha_linear.bias",))
    coarse.load_state_dict(pc); fine.load_state_dict(pf)
    embed_fn, _ = helpers.get_embedder(10, 0); embeddirs_fn, _ = helpers.get_embedder(4, 0)
    nq = lambda inputs, viewdirs, network_fn: helpers.run_network(inputs, viewdirs, network_fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn, netchunk=1 << 16)
    nr = 20
    ro = RN(nr, 3) * 0.2
    rdir = RN(nr, 3); rdir = rdir / rdir.norm(dim=-1, keepdim=True) * (1 + 0.2 * R(nr, 1))
    vd = rdir / rdir.norm(dim=-1, keepdim=True)
    rb = torch.cat([ro, rdir, torch.full((nr, 1), 2.0), torch.full((nr, 1), 6.0), vd], -1)
    render._DEVICE = torch.device("cpu")
    with torch.no_grad():
        r0 = render.render_rays(rb, coarse, nq, N_samples=16, retraw=True, perturb=0., N_importance=0, white_bkgd=True)
        r2 = render.render_rays(rb, coarse, nq, N_samples=16, retraw=True, perturb=1., N_importance=32, network_fine=fine, white_bkgd=False, pytest=True)
    np.random.seed(0); t_rand = torch.Tensor(np.random.rand(nr, 16))      # (pytest=True re-seeds numpy with 0 before each draw)
    np.random.seed(0); u_rand = torch.Tensor(np.random.rand(nr, 32))
    q0 = oc.render_rays(rb, pc, None, 16, 0, white_bkgd=True, retraw=True)
    q2 = oc.render_rays(rb, pc, pf, 16, 32, t_rand=t_rand, u=u_rand, retraw=True, sum_mode="torch")
    for k in ("rgb_map", "disp_map", "acc_map", "depth_map", "z_vals_map", "weights", "raw"):
        note("B1 render_rays coarse only: " + k, q0[k], r0[k], 1e-4, 1e-4)
    for k in ("rgb_map", "disp_map", "acc_map", "depth_map", "z_vals_map", "weights", "rgb0", "disp0", "acc0", "z_std"):
        note("B1 render_rays hierarchical, numpy-seeded draws replayed: " + k, q2[k], r2[k], 1e-4, 1e-4)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=25)
    ap.add_argument("--log", default=None)
    args = ap.parse_args()
    spec = importlib.util.spec_from_file_location("gen_golden", os.path.join(HERE, "gen_golden.py"))
    gg = importlib.util.module_from_spec(spec); spec.loader.exec_module(gg)
    ref = gg._import_reference()
    torch.autograd.set_detect_anomaly(False)
    for seed in range(args.seeds):
        fuzz_seed(seed, ref)
    lines = [f"# oracle vs imported reference, {args.seeds} seeds x fresh inputs and random weights (python oracle/fuzz_vs_reference.py --seeds {args.seeds})",
             f"# torch {torch.__version__}, numpy {np.__version__}; columns: stage | comparisons | worst |a - b| | worst excess of |a - b| over rtol*|ref| (must stay <= atol) | bar | violations"]
    bad = 0
    for stage in sorted(WORST):
        w = WORST[stage]
        bad += w["viol"]
        lines.append(f"{stage:78s} | {w['n']:4d} | {w['abs']:.3e} | {w['worst']:.3e} | {w['bar']:24s} | {w['viol']}")
    lines.append(f"# total violations: {bad}")
    text = "\n".join(lines)
    print(text)
    if args.log:
        open(args.log, "w").write(text + "\n")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
