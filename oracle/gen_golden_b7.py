#!/usr/bin/env python3
"""Golden vectors for SURVEY.md row B7 (the classic path's driver functions) and config 1 at its stated size (M1), captured from
the REFERENCE's own functions: get_rays / ndc_rays (s-nerf/model/run_nerf_helpers.py:247-258, 314-332), render / render_path
(s-nerf/model/render.py:22-135), NeRF_RGB (run_nerf_helpers.py:157-212) and the network_fn=None branches of render_rays
(render.py:358-371).  Runs only in the build container (needs /root/reference); the .npz files are data.

    python oracle/gen_golden_b7.py            # regenerate
    python oracle/gen_golden_b7.py --check    # regenerate in memory and compare
"""
import argparse
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import common  # noqa: E402
from oracle.gen_golden import OUT, _import_reference, t2n  # noqa: E402


def pose(yaw, pitch, t):
    """camera-to-world [3,4]: yaw about y, pitch about x, translation t"""
    cy, sy, cp, sp = math.cos(yaw), math.sin(yaw), math.cos(pitch), math.sin(pitch)
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
    return torch.tensor(np.concatenate([Ry @ Rx, np.asarray(t, dtype=np.float64)[:, None]], 1), dtype=torch.float32)


def gen_all():
    _, _, _, helpers, render = _import_reference()
    render._DEVICE = torch.device("cpu")
    G = {}
    g = torch.Generator().manual_seed(20)
    R = lambda *s: torch.rand(*s, generator=g)
    RN = lambda *s: torch.randn(*s, generator=g)

    # ---- get_rays / ndc_rays
    H, W, focal = 5, 7, 6.3
    c2w = pose(0.4, -0.2, [0.3, -0.1, 4.0])
    o, d = helpers.get_rays(H, W, focal, c2w)
    o2, d2 = helpers.get_rays(H, W, focal, c2w, ori_points=[3.1, 2.2])
    G["g20_get_rays"] = dict(H=np.int64(H), W=np.int64(W), focal=np.float64(focal), c2w=c2w, rays_o=o.contiguous(), rays_d=d,
                             ori_points=np.array([3.1, 2.2]), rays_o_ori=o2.contiguous(), rays_d_ori=d2)
    ro, rd = RN(33, 3) * 0.3, RN(33, 3)
    rd[:, 2] = -(0.5 + R(33))                                # forward-facing: d_z < 0
    no, nd = helpers.ndc_rays(9, 11, 8.7, 1.0, ro, rd)
    G["g20_ndc_rays"] = dict(H=np.int64(9), W=np.int64(11), focal=np.float64(8.7), near=np.float64(1.0), rays_o=ro, rays_d=rd,
                             ndc_o=no, ndc_d=nd)

    # ---- render(): formula networks, W = 64 (the size G9 uses)
    mk = lambda: helpers.NeRF(D=8, W=64, input_ch=63, input_ch_views=27, output_ch=4, skips=[4], use_viewdirs=True)
    coarse, fine = mk(), mk()
    coarse.load_state_dict(common.fill_state_dict_(coarse.state_dict()))
    fine.load_state_dict({k: v.flip(0) for k, v in common.fill_state_dict_(fine.state_dict()).items()})
    embed_fn, _ = helpers.get_embedder(10, 0)
    embeddirs_fn, _ = helpers.get_embedder(4, 0)
    nq = lambda inputs, viewdirs, network_fn: helpers.run_network(inputs, viewdirs, network_fn, embed_fn=embed_fn,
                                                                 embeddirs_fn=embeddirs_fn, netchunk=1 << 16)
    kw = dict(network_fn=coarse, network_query_fn=nq, N_samples=16, N_importance=32, network_fine=fine, perturb=0.,
              white_bkgd=False, raw_noise_std=0., retraw=True)
    H, W, focal = 6, 8, 7.5
    c2w = pose(0.3, 0.1, [0.2, 0.1, 4.0])
    c2w_s = pose(-0.2, 0.05, [0.0, 0.3, 4.2])
    flat = lambda out: {f"{i}": v for i, v in enumerate(out[:4])} | {f"x_{k}": v for k, v in out[4].items()}
    with torch.no_grad():
        a = render.render(H, W, focal, chunk=20, c2w=c2w, ndc=False, near=2., far=6., use_viewdirs=True, **kw)
        b = render.render(H, W, focal, chunk=20, c2w=c2w, ndc=True, near=0., far=1., use_viewdirs=True, **kw)
        c = render.render(H, W, focal, chunk=48, c2w=c2w, ndc=False, near=2., far=6., use_viewdirs=True, c2w_staticcam=c2w_s, **kw)
        gro, grd = helpers.get_rays(H, W, focal, c2w, ori_points=[4.4, 2.7])
        sel = torch.tensor([0, 5, 6, 17, 23, 30, 31, 40, 47])
        batch = (gro.reshape(-1, 3)[sel], grd.reshape(-1, 3)[sel])
        dep = R(9) * 3 + 2
        dd = render.render(H, W, focal, chunk=4, rays=batch, ndc=False, near=2., far=6., use_viewdirs=True, depths=dep, **kw)
    G["g20_render"] = dict(H=np.int64(H), W=np.int64(W), focal=np.float64(focal), c2w=c2w, c2w_static=c2w_s,
                           rays_o=batch[0].contiguous(), rays_d=batch[1].contiguous(), depths=dep,
                           **{f"a_{k}": v for k, v in flat(a).items()}, **{f"b_{k}": v for k, v in flat(b).items()},
                           **{f"c_{k}": v for k, v in flat(c).items()}, **{f"d_{k}": v for k, v in flat(dd).items()})

    # ---- render_path(): two poses at half resolution
    poses = torch.stack([torch.cat([pose(0.1 * i, 0.05, [0.1 * i, 0.0, 4.0]), torch.tensor([[0., 0., 0., 1.]])], 0) for i in range(2)])
    rk = dict(kw); rk.pop("retraw")
    rk.update(ndc=False, near=2., far=6., use_viewdirs=True)
    with torch.no_grad():
        rgbs, disps = render.render_path(poses, [12, 16, 15.0], 40, rk, render_factor=2)
    G["g20_render_path"] = dict(poses=poses, hwf=np.array([12, 16, 15.0]), rgbs=rgbs, disps=disps)

    # ---- NeRF_RGB + the network_fn=None branches
    alpha = mk()
    alpha.load_state_dict({k: v.flip(1) if v.dim() == 2 else v for k, v in common.fill_state_dict_(alpha.state_dict()).items()})
    rgbnet = helpers.NeRF_RGB(D=8, W=64, input_ch=63, input_ch_views=27, output_ch=4, skips=[4], use_viewdirs=True, alpha_model=alpha)
    own = {k: v for k, v in rgbnet.state_dict().items() if not k.startswith("alpha_model.")}
    sd = common.fill_state_dict_(own)
    sd.update({"alpha_model." + k: v for k, v in alpha.state_dict().items()})
    rgbnet.load_state_dict(sd)
    nr = 12
    ro = RN(nr, 3) * 0.2
    rdir = RN(nr, 3); rdir = rdir / rdir.norm(dim=-1, keepdim=True) * (1 + 0.2 * R(nr, 1))
    vd = rdir / rdir.norm(dim=-1, keepdim=True)
    rb = torch.cat([ro, rdir, torch.full((nr, 1), 2.0), torch.full((nr, 1), 6.0), vd], -1)
    pts = ro[:, None, :] + rdir[:, None, :] * torch.linspace(2, 6, 8)[None, :, None]
    tgt = R(nr, 8, 4)
    run = helpers.run_network(pts, vd, rgbnet, embed_fn, embeddirs_fn)
    ((run - tgt) ** 2).sum().backward()
    grads = {"grad_" + k: p.grad for k, p in rgbnet.named_parameters() if p.grad is not None}
    with torch.no_grad():
        r_none_rgb = render.render_rays(rb, None, nq, N_samples=16, retraw=True, perturb=0., N_importance=32, network_fine=rgbnet)
        plain = mk(); plain.load_state_dict(fine.state_dict()); plain.alpha_model = None
        r_none_plain = render.render_rays(rb, None, nq, N_samples=16, retraw=True, perturb=0., N_importance=32, network_fine=plain)
    G["g20_nerf_rgb"] = dict(ray_batch=rb, pts=pts, viewdirs=vd, target=tgt, run_network_out=run, **grads,
                             **{f"rgbnet_{k}": v for k, v in r_none_rgb.items()}, **{f"plain_{k}": v for k, v in r_none_plain.items()})

    # ---- config 1 at its stated size: Lego-style 400 x 400, focal 555.5, 4 m orbit looking at the origin,
    # near 2 / far 6, white background, 64 coarse samples, N_importance 0, NeRF 8 x 256 with formula weights.  Stored: every 97th pixel.
    H = W = 400
    focal = 555.5
    th, ph = 0.6, -0.5
    cam = np.array([4 * math.cos(ph) * math.sin(th), -4 * math.sin(ph), 4 * math.cos(ph) * math.cos(th)])
    z = cam / np.linalg.norm(cam)                           # camera looks down -z at the origin
    x = np.cross([0., 1., 0.], z); x /= np.linalg.norm(x)
    y = np.cross(z, x)
    c2w = torch.tensor(np.stack([x, y, z, cam], 1), dtype=torch.float32)
    net = helpers.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=4, skips=[4], use_viewdirs=True)
    net.load_state_dict(common.fill_state_dict_(net.state_dict()))
    idx = torch.arange(0, H * W, 97)
    with torch.no_grad():
        gro, grd = helpers.get_rays(H, W, focal, c2w)
        out = render.render(H, W, focal, chunk=1 << 15, rays=(gro.reshape(-1, 3)[idx], grd.reshape(-1, 3)[idx]), ndc=False, near=2., far=6.,
                            use_viewdirs=True, network_fn=net, network_query_fn=nq, N_samples=64, N_importance=0, network_fine=None,
                            perturb=0., white_bkgd=True, raw_noise_std=0.)
    G["g20_m1"] = dict(H=np.int64(H), W=np.int64(W), focal=np.float64(focal), c2w=c2w, idx=idx, rgb_map=out[0], disp_map=out[1],
                       acc_map=out[2], depth_map=out[3])
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
