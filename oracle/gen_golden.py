#!/usr/bin/env python3
"""Capture golden vectors from the REFERENCE's own Python modules.

Runs only in the build container (needs /root/reference); the output
``tests/golden/*.npz`` files are data -- inputs and expected outputs -- and are
the only thing that travels.  The reference has no tests or fixtures for this
path, so these vectors are what pins the oracle.

    python oracle/gen_golden.py            # regenerate everything
    python oracle/gen_golden.py --check    # regenerate in memory and compare

Nothing of the reference's source is copied: its modules are imported from
where they lie, fed seeded inputs and formula weights (oracle/common.py), and
their outputs recorded.
"""
import argparse
import os
import sys
import types

os.environ.setdefault("TORCHDYNAMO_DISABLE", "1")
sys.dont_write_bytecode = True

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
OUT = os.path.join(REPO, "tests", "golden")
REF = "/root/reference/s-nerf"
sys.path.insert(0, REPO)

from oracle import common  # noqa: E402


def _import_reference():
    if not os.path.isdir(REF):
        raise SystemExit("reference tree not present; golden vectors can only be regenerated in the build container")
    for name in ("turtle", "cv2"):  # models.py:2 imports turtle; utils/render_utils.py:3 imports cv2
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.forward = lambda *a, **k: None
            sys.modules[name] = m
    sys.path.insert(0, REF)
    import model.math_ops as math_ops
    import model.mip as mip
    import model.models as models
    import model.run_nerf_helpers as helpers
    import model.render as render
    torch.autograd.set_detect_anomaly(False)  # run_nerf_helpers.py:2 switches it on
    return math_ops, mip, models, helpers, render


def t2n(d):
    out = {}
    for k, v in d.items():
        if v is None:
            continue
        if torch.is_tensor(v):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    return out


def gen_all():
    math_ops, mip, models, helpers, render = _import_reference()
    G = {}
    g = torch.Generator().manual_seed(1234)
    R = lambda *s: torch.rand(*s, generator=g)
    RN = lambda *s: torch.randn(*s, generator=g)

    # ---------------- G1: positional encodings (A10, B2) -----------------
    x = (R(32, 3) * 8 - 4)
    emb_pts, _ = helpers.get_embedder(10, 0)
    emb_dir, _ = helpers.get_embedder(4, 0)
    G["g1_posenc"] = dict(x=x, classic_pts=emb_pts(x), classic_dirs=emb_dir(x),
                          mip_view=mip.pos_enc(x, 0, 4, True))

    # ---------------- G2: cast_rays cone & cylinder (A3) ------------------
    n, s = 16, 16
    o = RN(n, 3) * 0.3
    d = RN(n, 3); d = d / d.norm(dim=-1, keepdim=True) * (1 + 0.3 * R(n, 1))
    radii = 1e-3 + 4e-3 * R(n, 1)
    near, far = torch.full((n, 1), 2.0), torch.full((n, 1), 100.0)
    sv = torch.sort(R(n, s + 1), -1)[0]
    sv[:, 0] = 0; sv[:, -1] = 1
    tv = mip.Transform_log(sv, near, far)
    m_c, c_c = mip.cast_rays(tv, o, d, radii, "cone")
    m_y, c_y = mip.cast_rays(tv, o, d, radii, "cylinder")
    G["g2_cast"] = dict(origins=o, directions=d, radii=radii, near=near, far=far, s_vals=sv, t_vals=tv,
                        cone_mean=m_c, cone_cov=c_c, cyl_mean=m_y, cyl_cov=c_y)

    # ---------------- G3: sample2enc incl. |x| <,=,> 3 (A2-A6) ------------
    o3 = o.clone(); d3 = d.clone()
    o3[0] = torch.tensor([0.0, 0.0, 0.0]); d3[0] = torch.tensor([0.0, 0.0, 1.0])   # passes through |x| = 3 region
    fm, fc = mip.sample2enc(sv, o3, d3, radii, "cone", near, far, s, 1, viewc=0., radius=3., transform_idx=0)
    G["g3_sample2enc"] = dict(origins=o3, directions=d3, radii=radii, near=near, far=far, s_vals=sv,
                              f_means=fm, f_covs=fc)
    # direct contraction / Jacobian probes at exactly-on-boundary points
    xb = torch.tensor([[[3.0, 0, 0], [0, 2.9999, 0], [0, 0, 3.0001], [1.0, 2.0, 2.0], [10.0, -20.0, 5.0], [0.1, 0.2, -0.3]]])
    fn, jac = mip.warp_fn(1, 0., far, 3.)
    G["g3b_contract"] = dict(x=xb, fx=fn(xb), J=jac(xb))

    # ---------------- G4: integrated_pos_enc diag & full (A7) -------------
    xm = RN(8, 16, 3) * 1.5
    cd = R(8, 16, 3) * 1e-3
    xm[0, 0] = torch.tensor([2.0, -1.99, 0.013])  # 2^15 * 2 >= 100 pi -> safe_sin mod branch
    cfull = torch.diag_embed(cd) + 1e-4 * (RN(8, 16, 3, 3))
    cfull = 0.5 * (cfull + cfull.transpose(-1, -2))
    enc_d = mip.integrated_pos_enc((xm, cd), 0, 16, diag=True, device="cpu")
    enc_f = mip.integrated_pos_enc((xm, cfull), 0, 16, diag=False)
    G["g4_ipe"] = dict(means=xm, cov_diag=cd, cov_full=cfull, enc_diag=enc_d, enc_full=enc_f)

    # ---------------- G5: sorted_piecewise_constant_pdf (A14) -------------
    nb = 32
    bins = torch.sort(R(12, nb + 1), -1)[0]
    w = R(12, nb) ** 3
    w[0] = 0.0                                  # all-zero weights (padding path)
    w[1] = 0.0; w[1, 7] = 1.0                   # one-hot
    w[2] = 0.25                                 # flat -> exact ties in cdf arithmetic
    w[3, :16] = 0.0                             # leading zeros -> repeated cdf values
    for num in (32, 33):
        torch.manual_seed(7)
        det = math_ops.sorted_piecewise_constant_pdf(bins, w, num, False)
        u_det = torch.linspace(0., 1. - torch.finfo(torch.float32).eps, num)
        torch.manual_seed(11)
        rnd = math_ops.sorted_piecewise_constant_pdf(bins, w, num, True)
        torch.manual_seed(11)  # replay the uniform_ draw the reference made (math_ops.py:52)
        jit = torch.empty(12, num).uniform_(to=1 / num - torch.finfo(torch.float32).eps)
        G[f"g5_pdf_{num}"] = dict(bins=bins, weights=w, det_samples=det, u_det=u_det, rand_samples=rnd, jitter=jit)

    # ---------------- G6: real_volumetric_rendering (A12) -----------------
    n, s = 12, 24
    rgb = R(n, s, 3)
    dens = R(n, s, 1) * 2
    dens[0] = 0.0                               # sigma = 0 row: distance 0 -> clipped to t_0
    dens[1] = 1e4                               # opaque at first sample
    sv6 = torch.sort(R(n, s + 1), -1)[0]
    d6 = RN(n, 3)
    near6, far6 = torch.full((n, 1), 1.8), torch.full((n, 1), 110.0)
    sem = RN(n, s, 5)
    for wb in (False, True):
        c, dist, acc, ww, se = mip.real_volumetric_rendering(rgb, dens, sv6, d6, sem, wb, near6, far6, 0)
        G[f"g6_volrend_white{int(wb)}"] = dict(rgb=rgb, density=dens, s_vals=sv6, dirs=d6, near=near6, far=far6,
                                               semantic_in=sem, comp_rgb=c, distance=dist, acc=acc, weights=ww, semantic=se)
    c, dist, acc, ww, se = mip.real_volumetric_rendering(None, dens, sv6, d6, None, False, near6, far6, 0)
    G["g6_volrend_norgb"] = dict(density=dens, s_vals=sv6, dirs=d6, near=near6, far=far6, distance=dist, acc=acc, weights=ww)

    # ---------------- G7: proposal / MLP forward + grads ------------------
    torch.manual_seed(0)
    model = models.MipNerfModel(n_samples=16, N_fine=17, no_warp_sample=0, ray_shape="cone", fn=1, radius=3.,
                                transform_idx=0, real=True, rgb_layer=3, hidden_layer=64, density_noise=0.,
                                max_deg_point=16, proposal_hidden_layer=64, proposal_loss=True)
    sd = common.fill_state_dict_(model.state_dict())
    model.load_state_dict(sd)
    enc = (R(6, 5, 96) * 2 - 1)
    cond = (R(6, 27) * 2 - 1)
    for prm in model.parameters():
        prm.grad = None
    rr, rd, _ = model.mlp(enc, cond)
    pr = model.proposal(enc)[1]
    loss = (rr ** 2).sum() + (rd * 0.5).sum() + (pr ** 2).sum()
    loss.backward()
    grads = {"grad." + k: v.grad.clone() for k, v in model.named_parameters() if v.grad is not None}
    G["g7_mlp_h64"] = dict(enc=enc, cond=cond, raw_rgb=rr, raw_density=rd, prop_density=pr, **grads)
    G["g7_mlp_h64"]["param_names"] = np.array(list(sd.keys()))

    # ---------------- G8: MipNerfModel.forward end to end -----------------
    rays = common.synthetic_rays(48, seed=3)
    from collections import namedtuple
    Rays = namedtuple("Rays", ("origins", "directions", "viewdirs", "radii", "lossmult", "near", "far", "app"))
    rt = Rays(**rays)
    with torch.no_grad():
        ret = model(rt, False, False, 0.)
    G["g8_mipnerf_det"] = dict(**{f"rays_{k}": v for k, v in rays.items()},
                               l0_distance=ret[0][1], l0_acc=ret[0][2], l0_s_vals=ret[0][3], l0_weights=ret[0][4],
                               l1_rgb=ret[1][0], l1_distance=ret[1][1], l1_acc=ret[1][2], l1_s_vals=ret[1][4], l1_weights=ret[1][5])
    # randomized: replay the three torch RNG draws (mip.py:283, math_ops.py:52)
    torch.manual_seed(5)
    with torch.no_grad():
        ret = model(rt, True, False, 0.)  # white_bg=True crashes the reference at level 0 (rgb is None, mip.py:188)
    torch.manual_seed(5)
    s_rand = torch.rand(48, 17)
    jit = torch.empty(48, 17).uniform_(to=1 / 17 - torch.finfo(torch.float32).eps)
    G["g8_mipnerf_rand"] = dict(s_rand=s_rand, jitter=jit,
                                l0_distance=ret[0][1], l0_acc=ret[0][2], l0_s_vals=ret[0][3], l0_weights=ret[0][4],
                                l1_rgb=ret[1][0], l1_distance=ret[1][1], l1_acc=ret[1][2], l1_s_vals=ret[1][4], l1_weights=ret[1][5])

    # one full-width (hidden 1024) 4-row case for the MLP alone
    torch.manual_seed(0)
    big = models.MLP(feature_dim=96, n_layers_condition=3, n_units=1024, cond_dim=27)
    sdb = common.fill_state_dict_(big.state_dict())
    big.load_state_dict(sdb)
    with torch.no_grad():
        rr, rd, _ = big(enc[:2, :2], cond[:2])
    G["g7_mlp_h1024"] = dict(enc=enc[:2, :2], cond=cond[:2], raw_rgb=rr, raw_density=rd)

    # ---------------- G9: classic path (B1-B6) ---------------------------
    n, s = 10, 16
    raw = RN(n, s, 4)
    z = torch.sort(R(n, s) * 4 + 2, -1)[0]
    rd9 = RN(n, 3)
    # the reference builds CPU tensors without device args; fine on CPU
    for wb in (False, True):
        out = helpers.raw2outputs(raw, z, rd9, 0, wb)
        G[f"g9_raw2outputs_white{int(wb)}"] = dict(raw=raw, z_vals=z, rays_d=rd9, rgb_map=out[0], disp_map=out[1],
                                                    acc_map=out[2], weights=out[3], depth_map=out[4])
    bins9 = torch.sort(R(n, 15), -1)[0]
    w9 = R(n, 14) ** 2
    w9[0] = 0.0
    det9 = helpers.sample_pdf(bins9, w9, 24, det=True)
    rnd9 = helpers.sample_pdf(bins9, w9, 24, det=False, pytest=True)
    np.random.seed(0)
    u9 = torch.Tensor(np.random.rand(n, 24))
    # indices as the reference computes them (run_nerf_helpers.py:362)
    def ref_inds(bins, weights, u):
        ww = weights + 1e-5
        pdf = ww / torch.sum(ww, -1, keepdim=True)
        cdf = torch.cat([torch.zeros_like(pdf[..., :1]), torch.cumsum(pdf, -1)], -1)
        return torch.searchsorted(cdf, u.contiguous(), right=True)
    G["g9_sample_pdf"] = dict(bins=bins9, weights=w9, det_samples=det9, rand_samples=rnd9, u_rand=u9,
                              det_inds=ref_inds(bins9, w9, torch.linspace(0., 1., 24).expand(n, 24)),
                              rand_inds=ref_inds(bins9, w9, u9))

    torch.manual_seed(0)
    coarse = helpers.NeRF(D=8, W=64, input_ch=63, input_ch_views=27, output_ch=4, skips=[4], use_viewdirs=True)
    fine = helpers.NeRF(D=8, W=64, input_ch=63, input_ch_views=27, output_ch=4, skips=[4], use_viewdirs=True)
    coarse.load_state_dict(common.fill_state_dict_(coarse.state_dict()))
    sdf = common.fill_state_dict_(fine.state_dict())
    sdf = {k: v.flip(0) for k, v in sdf.items()}  # make the fine net differ from the coarse one
    fine.load_state_dict(sdf)
    embed_fn, _ = helpers.get_embedder(10, 0)
    embeddirs_fn, _ = helpers.get_embedder(4, 0)
    nq = lambda inputs, viewdirs, network_fn: helpers.run_network(inputs, viewdirs, network_fn, embed_fn=embed_fn,
                                                                 embeddirs_fn=embeddirs_fn, netchunk=1 << 16)
    nr = 24
    ro = RN(nr, 3) * 0.2
    rdir = RN(nr, 3); rdir = rdir / rdir.norm(dim=-1, keepdim=True) * (1 + 0.2 * R(nr, 1))
    vd = rdir / rdir.norm(dim=-1, keepdim=True)
    rb = torch.cat([ro, rdir, torch.full((nr, 1), 2.0), torch.full((nr, 1), 6.0), vd], -1)
    render._DEVICE = torch.device("cpu")
    with torch.no_grad():
        run = helpers.run_network(ro[:, None, :] + rdir[:, None, :] * torch.linspace(2, 6, 8)[None, :, None], vd, coarse,
                                  embed_fn, embeddirs_fn)
        r0 = render.render_rays(rb, coarse, nq, N_samples=16, retraw=True, perturb=0., N_importance=0, white_bkgd=True)
        r1 = render.render_rays(rb, coarse, nq, N_samples=16, retraw=True, perturb=0., N_importance=32,
                                network_fine=fine, white_bkgd=False)
        r2 = render.render_rays(rb, coarse, nq, N_samples=16, retraw=True, perturb=1., N_importance=32,
                                network_fine=fine, white_bkgd=False, pytest=True)
    np.random.seed(0); t_rand = torch.Tensor(np.random.rand(nr, 16))
    np.random.seed(0); u_rand = torch.Tensor(np.random.rand(nr, 32))
    G["g9_render_rays"] = dict(ray_batch=rb, run_network_out=run, t_rand=t_rand, u_rand=u_rand,
                               **{f"c_{k}": v for k, v in r0.items()}, **{f"cf_{k}": v for k, v in r1.items()},
                               **{f"pt_{k}": v for k, v in r2.items()})
    return {k: t2n(v) for k, v in G.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    G = gen_all()
    os.makedirs(OUT, exist_ok=True)
    bad = 0
    for name, d in G.items():
        path = os.path.join(OUT, name + ".npz")
        if args.check:
            old = np.load(path, allow_pickle=False)
            for k, v in d.items():
                if v.dtype.kind in "US":
                    continue
                if not common.golden_equal(old[k], v):
                    print("MISMATCH", name, k, float(np.nanmax(np.abs(old[k].astype(np.float64) - v))))
                    bad += 1
        else:
            np.savez_compressed(path, **d)
            print("wrote", path, sum(v.nbytes for v in d.values()), "bytes")
    if args.check:
        print("check:", "OK" if bad == 0 else f"{bad} mismatches")
        sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
