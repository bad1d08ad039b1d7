#!/usr/bin/env python3
"""Golden vectors for the branches of the classic and mip paths a drop-in user can reach beyond the S-NeRF configuration (VERDICT r4
item 8), captured from the REFERENCE's own functions:

  g27_no_viewdirs    NeRF(use_viewdirs=False, output_ch=5) (run_nerf_helpers.py:100-124, create_nerf render.py:180-183: output_ch = 5 with
                     N_importance > 0): run_network output + parameter gradients, render_rays coarse + fine on an 8-column ray batch
  g28_identity_embed get_embedder(multires, i=-1) -> (nn.Identity(), 3) (run_nerf_helpers.py:55-57): NeRF(input_ch=3, input_ch_views=3),
                     run_network + render_rays
  g29_near_far       render(..., near=<tensor [N,1]>, far=<tensor [N,1]>) (render.py:74: near * ones_like(rays_d[..., :1]))
  g30_pose_fn0       MipNerfModel(fn=0) forward + gradients w.r.t. origins / directions / viewdirs (mip.py:323-341, 367-378, 381-395:
                     the view-centred warp under pose refinement)
  g31_no_integration MipNerfModel(disable_integration=1) (arg_parser.py:188 --disable_integration; models.py:132-133: the sample covariances
                     are replaced by zeros before integrated_pos_enc): outputs of both levels, parameter gradients, ray gradients

Runs only in the build container (needs /root/reference); the .npz files are data.
    python oracle/gen_golden_branches.py [--check]
"""
import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import common  # noqa: E402
from oracle.gen_golden import OUT, _import_reference, t2n  # noqa: E402


def gen_all():
    _, mip, models, helpers, render = _import_reference()
    render._DEVICE = torch.device("cpu")
    G = {}
    g = torch.Generator().manual_seed(27)
    R = lambda *s: torch.rand(*s, generator=g)
    RN = lambda *s: torch.randn(*s, generator=g)
    nr = 12
    ro = RN(nr, 3) * 0.2
    rdir = RN(nr, 3); rdir = rdir / rdir.norm(dim=-1, keepdim=True) * (1 + 0.2 * R(nr, 1))
    vd = rdir / rdir.norm(dim=-1, keepdim=True)
    pts = ro[:, None, :] + rdir[:, None, :] * torch.linspace(2, 6, 8)[None, :, None]

    # ---- use_viewdirs=False, output_ch=5
    embed_fn, ic = helpers.get_embedder(10, 0)
    mk = lambda: helpers.NeRF(D=8, W=64, input_ch=ic, input_ch_views=0, output_ch=5, skips=[4], use_viewdirs=False)
    coarse, fine = mk(), mk()
    coarse.load_state_dict(common.fill_state_dict_(coarse.state_dict()))
    fine.load_state_dict({k: v.flip(0) for k, v in common.fill_state_dict_(fine.state_dict()).items()})
    nq = lambda inputs, viewdirs, network_fn: helpers.run_network(inputs, viewdirs, network_fn, embed_fn=embed_fn, embeddirs_fn=None, netchunk=1 << 16)
    tgt = R(nr, 8, 5)
    run = helpers.run_network(pts, None, coarse, embed_fn, None)
    ((run - tgt) ** 2).sum().backward()
    grads = {"grad_" + k: p.grad for k, p in coarse.named_parameters() if p.grad is not None}
    rb8 = torch.cat([ro, rdir, torch.full((nr, 1), 2.0), torch.full((nr, 1), 6.0)], -1)
    with torch.no_grad():
        rr = render.render_rays(rb8, coarse, nq, N_samples=16, retraw=True, perturb=0., N_importance=32, network_fine=fine)
    G["g27_no_viewdirs"] = dict(ray_batch=rb8, pts=pts, target=tgt, run_network_out=run, param_names=np.array(list(coarse.state_dict().keys())),
                                **grads, **{f"rr_{k}": v for k, v in rr.items()})

    # ---- identity embedding
    e_id, ic3 = helpers.get_embedder(10, -1)
    ed_id, icv3 = helpers.get_embedder(4, -1)
    assert ic3 == 3 and icv3 == 3
    mk = lambda: helpers.NeRF(D=8, W=64, input_ch=3, input_ch_views=3, output_ch=4, skips=[4], use_viewdirs=True)
    c2, f2 = mk(), mk()
    c2.load_state_dict(common.fill_state_dict_(c2.state_dict()))
    f2.load_state_dict({k: v.flip(0) for k, v in common.fill_state_dict_(f2.state_dict()).items()})
    nq2 = lambda inputs, viewdirs, network_fn: helpers.run_network(inputs, viewdirs, network_fn, embed_fn=e_id, embeddirs_fn=ed_id, netchunk=1 << 16)
    tgt4 = R(nr, 8, 4)
    run2 = helpers.run_network(pts, vd, c2, e_id, ed_id)
    ((run2 - tgt4) ** 2).sum().backward()
    grads2 = {"grad_" + k: p.grad for k, p in c2.named_parameters() if p.grad is not None}
    rb11 = torch.cat([rb8, vd], -1)
    with torch.no_grad():
        rr2 = render.render_rays(rb11, c2, nq2, N_samples=16, retraw=True, perturb=0., N_importance=32, network_fine=f2)
    G["g28_identity_embed"] = dict(ray_batch=rb11, pts=pts, viewdirs=vd, target=tgt4, run_network_out=run2, **grads2,
                                   **{f"rr_{k}": v for k, v in rr2.items()})

    # ---- per-ray near / far tensors through render()
    emb, _ = helpers.get_embedder(10, 0)
    embd, _ = helpers.get_embedder(4, 0)
    mk = lambda: helpers.NeRF(D=8, W=64, input_ch=63, input_ch_views=27, output_ch=4, skips=[4], use_viewdirs=True)
    c3, f3 = mk(), mk()
    c3.load_state_dict(common.fill_state_dict_(c3.state_dict()))
    f3.load_state_dict({k: v.flip(0) for k, v in common.fill_state_dict_(f3.state_dict()).items()})
    nq3 = lambda inputs, viewdirs, network_fn: helpers.run_network(inputs, viewdirs, network_fn, embed_fn=emb, embeddirs_fn=embd, netchunk=1 << 16)
    near_t, far_t = 1.5 + R(nr, 1), 5.0 + 2 * R(nr, 1)
    kw = dict(network_fn=c3, network_query_fn=nq3, N_samples=16, N_importance=32, network_fine=f3, perturb=0., white_bkgd=False, raw_noise_std=0.)
    with torch.no_grad():
        out = render.render(6, 8, 7.5, chunk=5, rays=(ro, rdir), ndc=False, near=near_t, far=far_t, use_viewdirs=True, **kw)
    G["g29_near_far"] = dict(rays_o=ro, rays_d=rdir, near=near_t, far=far_t, **{f"{i}": v for i, v in enumerate(out[:4])},
                             **{f"x_{k}": v for k, v in out[4].items()})

    # ---- mip path: pose refinement through the view-centred warp fn = 0
    S0, P1, hidden, n = 16, 17, 64, 24
    m = models.MipNerfModel(n_samples=S0, N_fine=P1, no_warp_sample=0, ray_shape="cone", fn=0, radius=3., transform_idx=0, real=True, rgb_layer=3,
                            hidden_layer=hidden, density_noise=0., max_deg_point=16, proposal_hidden_layer=64, proposal_loss=True)
    m.load_state_dict(common.fill_state_dict_(m.state_dict()))
    rays = common.synthetic_rays(n, seed=5)
    rays["near"] = torch.full_like(rays["near"], 0.5)
    rays["far"] = torch.full_like(rays["far"], 30.0)
    leaves = {k: rays[k].clone().requires_grad_(True) for k in ("origins", "directions", "viewdirs")}
    Rays = models.Rays if hasattr(models, "Rays") else None
    import collections
    RT = collections.namedtuple("Rays", ("origins", "directions", "viewdirs", "radii", "lossmult", "near", "far", "app"))
    rr3 = RT(leaves["origins"], leaves["directions"], leaves["viewdirs"], rays["radii"], rays["lossmult"], rays["near"], rays["far"], rays["app"])
    viewc = torch.tensor([0.3, -0.2, 0.5])
    ret = m(rr3, False, False, viewc)
    w_rgb, w_d1, w_d0 = R(n, 3), R(n), R(n)
    loss = (ret[1][0] * w_rgb).sum() + 0.05 * (ret[1][1] * w_d1).sum() + 0.05 * (ret[0][1] * w_d0).sum()
    loss.backward()
    G["g30_pose_fn0"] = dict(**{k: v for k, v in rays.items()}, viewc=viewc, w_rgb=w_rgb, w_d1=w_d1, w_d0=w_d0, rgb=ret[1][0], dist1=ret[1][1], dist0=ret[0][1],
                             g_origins=leaves["origins"].grad, g_directions=leaves["directions"].grad, g_viewdirs=leaves["viewdirs"].grad,
                             S0=np.int64(S0), P1=np.int64(P1), hidden=np.int64(hidden))
    # ---- mip path: --disable_integration (contraction warp, pose refinement on)
    m = models.MipNerfModel(n_samples=S0, N_fine=P1, no_warp_sample=0, disable_integration=1, ray_shape="cone", fn=1, radius=3., transform_idx=0, real=True,
                            rgb_layer=3, hidden_layer=hidden, density_noise=0., max_deg_point=16, proposal_hidden_layer=64, proposal_loss=True)
    sd = common.fill_state_dict_(m.state_dict())
    m.load_state_dict(sd)
    rays = common.synthetic_rays(n, seed=31)
    leaves = {k: rays[k].clone().requires_grad_(True) for k in ("origins", "directions", "viewdirs")}
    rr4 = RT(leaves["origins"], leaves["directions"], leaves["viewdirs"], rays["radii"], rays["lossmult"], rays["near"], rays["far"], rays["app"])
    ret = m(rr4, False, False, 0.)
    w_rgb, w_d1, w_d0 = R(n, 3), R(n), R(n)
    loss = (ret[1][0] * w_rgb).sum() + 0.05 * (ret[1][1] * w_d1).sum() + 0.05 * (ret[0][1] * w_d0).sum()
    loss.backward()
    G["g31_no_integration"] = dict(**{k: v for k, v in rays.items()}, w_rgb=w_rgb, w_d1=w_d1, w_d0=w_d0, rgb=ret[1][0], dist1=ret[1][1], acc1=ret[1][2],
                                   dist0=ret[0][1], s1=ret[1][4], weights1=ret[1][5], loss=loss.detach(),
                                   g_origins=leaves["origins"].grad, g_directions=leaves["directions"].grad, g_viewdirs=leaves["viewdirs"].grad,
                                   param_names=np.array(list(sd.keys())), **{"grad." + k: p.grad for k, p in m.named_parameters()},
                                   S0=np.int64(S0), P1=np.int64(P1), hidden=np.int64(hidden))
    return {k: t2n(v) for k, v in G.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    G = gen_all()
    bad = 0
    for name, d in G.items():
        path = os.path.join(OUT, name + ".npz")
        if args.check:
            old = np.load(path, allow_pickle=False)
            for k, v in d.items():
                if not common.golden_equal(old[k], v):
                    print("MISMATCH", name, k)
                    bad += 1
        else:
            np.savez_compressed(path, **d)
            print("wrote", path, sum(v.nbytes for v in d.values()), "bytes")
    if args.check:
        print("check:", "OK" if bad == 0 else f"{bad} mismatches")
        sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
