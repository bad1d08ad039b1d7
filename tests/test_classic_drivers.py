"""the classic path's driver functions -- get_rays, ndc_rays, render, render_path, create_nerf,
NeRF_RGB and the network_fn=None branches of render_rays -- against golden vectors captured from the reference's own functions
(oracle/gen_golden_b7.py -> tests/golden/g20_*.npz).  Three layers, as everywhere in this suite: the oracle against the goldens
(CPU), the product's host logic on the oracle-based emulation of the kernels (CPU), the HIP kernels through the C-ABI (-m gpu)."""
import argparse
import os

import numpy as np
import pytest
import torch

from oracle import classic as oc
from oracle import common

from cpu_ops_emulation import emulate_ops
from test_paths import close, nerf_params

DEV = "cuda"


@pytest.fixture(params=[pytest.param("hip", marks=pytest.mark.gpu), "emulated"])
def backend(request):
    global DEV
    if request.param == "hip":
        DEV = "cuda"
        yield "hip"
    else:
        DEV = "cpu"
        with emulate_ops():
            yield "emulated"
    DEV = "cuda"


def nerf(W, compute, sd, cls=None, **kw):
    from snerf_amd import classic
    m = (cls or classic.NeRF)(D=8, W=W, input_ch=63, input_ch_views=27, output_ch=4, skips=[4], use_viewdirs=True, compute=compute, device=DEV, **kw)
    m.load_state_dict(sd)
    return m


def query_fn():
    from snerf_amd import classic
    e, _ = classic.get_embedder(10, 0); ev, _ = classic.get_embedder(4, 0)
    return classic.make_network_query_fn(e, ev), e, ev


def render_kw(coarse, fine):
    return dict(network_fn=coarse, network_query_fn=query_fn()[0], N_samples=16, N_importance=32, network_fine=fine, perturb=0.,
                white_bkgd=False, raw_noise_std=0., retraw=True)


# ------------------------------------------------------------------------------------------------ oracle vs reference goldens
def test_oracle_get_rays_and_ndc_rays_bit_exact(golden):
    g = golden("g20_get_rays")
    H, W, f = int(g["H"]), int(g["W"]), float(g["focal"])
    o, d = oc.get_rays(H, W, f, g["c2w"])
    assert torch.equal(d, g["rays_d"]) and torch.equal(o, g["rays_o"])
    o, d = oc.get_rays(H, W, f, g["c2w"], list(g["ori_points"].numpy()))
    assert torch.equal(d, g["rays_d_ori"]) and torch.equal(o, g["rays_o_ori"])
    g = golden("g20_ndc_rays")
    o, d = oc.ndc_rays(int(g["H"]), int(g["W"]), float(g["focal"]), float(g["near"]), g["rays_o"], g["rays_d"])
    assert torch.equal(o, g["ndc_o"]) and torch.equal(d, g["ndc_d"])


def _oracle_render_cases(g):
    H, W, f = int(g["H"]), int(g["W"]), float(g["focal"])
    kw = dict(N_samples=16, N_importance=32, retraw=True, sum_mode="torch")
    pc, pf = nerf_params(64), nerf_params(64, True)
    yield "a", oc.render(H, W, f, pc, pf, chunk=20, c2w=g["c2w"], ndc=False, near=2., far=6., use_viewdirs=True, **kw)
    yield "b", oc.render(H, W, f, pc, pf, chunk=20, c2w=g["c2w"], ndc=True, near=0., far=1., use_viewdirs=True, **kw)
    yield "c", oc.render(H, W, f, pc, pf, chunk=48, c2w=g["c2w"], ndc=False, near=2., far=6., use_viewdirs=True, c2w_staticcam=g["c2w_static"], **kw)
    yield "d", oc.render(H, W, f, pc, pf, chunk=4, rays=(g["rays_o"], g["rays_d"]), ndc=False, near=2., far=6., use_viewdirs=True,
                         depths=g["depths"], **kw)


def _check_render(tag, out, g, tol, skip_det_fine=True):
    """`out` = [rgb, disp, acc, depth, extras].  The deterministic hierarchical pass has u == cdf ties whose interval flips with the
    summation order, so value checks of the fine pass use a tolerance that admits a flipped interval; the
    coarse pass (rgb0 / weights / z_vals_map) is tight."""
    ex = out[4]
    for k in ("rgb0", "disp0", "acc0", "weights", "z_vals_map"):
        close(ex[k], g[f"{tag}_x_{k}"], tol, tol, f"{tag} {k}")
    for i, k in enumerate(("rgb_map", "disp_map", "acc_map", "depth_map")):
        assert out[i].shape == g[f"{tag}_{i}"].shape, (tag, k)
        close(out[i], g[f"{tag}_{i}"], 50 * tol, 50 * tol, f"{tag} {k}")


def test_oracle_render_vs_reference_golden(golden):
    g = golden("g20_render")
    with torch.no_grad():
        for tag, out in _oracle_render_cases(g):
            _check_render(tag, out, g, 2e-5)


def test_oracle_nerf_rgb_vs_reference_golden(golden):
    g = golden("g20_nerf_rgb")
    alpha = {k: (v.flip(1) if v.dim() == 2 else v) for k, v in nerf_params(64).items()}
    shapes = [(k, s) for k, s in oc.nerf_param_shapes(W=64) if not k.startswith("alpha_linear")]
    own = {k: v.clone().requires_grad_(True) for k, v in common.fill_state_dict_({k: torch.empty(s) for k, s in shapes}).items()}
    pts, vd = g["pts"], g["viewdirs"]
    e = torch.cat([oc.embed(pts.reshape(-1, 3), 10), oc.embed(vd[:, None].expand(pts.shape).reshape(-1, 3), 4)], -1)
    out = oc.nerf_rgb_mlp(own, alpha, e).reshape(pts.shape[0], pts.shape[1], 4)
    close(out, g["run_network_out"], 1e-5, 1e-5, "NeRF_RGB forward")
    ((out - g["target"]) ** 2).sum().backward()
    for k, p in own.items():
        close(p.grad, g["grad_" + k], 1e-4, 1e-4 * float(g["grad_" + k].abs().max()), "NeRF_RGB grad " + k)
    assert not any(k.startswith("grad_alpha_model") for k in g), "the reference's alpha model receives no gradient"


# ------------------------------------------------------------------------------------------------ product vs reference goldens
def test_get_rays_and_ndc_rays_bit_exact(backend, golden):
    from snerf_amd import classic
    g = golden("g20_get_rays")
    H, W, f = int(g["H"]), int(g["W"]), float(g["focal"])
    o, d = classic.get_rays(H, W, f, g["c2w"].to(DEV))
    assert o.shape == (H, W, 3) and torch.equal(d.cpu(), g["rays_d"]) and torch.equal(o.cpu(), g["rays_o"])
    o, d = classic.get_rays(H, W, f, g["c2w"].numpy(), ori_points=list(g["ori_points"].numpy()))
    assert torch.equal(d.cpu(), g["rays_d_ori"]) and torch.equal(o.cpu(), g["rays_o_ori"])
    g = golden("g20_ndc_rays")
    o, d = classic.ndc_rays(int(g["H"]), int(g["W"]), float(g["focal"]), float(g["near"]), g["rays_o"].to(DEV), g["rays_d"].to(DEV))
    assert torch.equal(o.cpu(), g["ndc_o"]) and torch.equal(d.cpu(), g["ndc_d"])


def test_render_vs_reference_golden(backend, golden):
    """render(): whole frame from c2w (ragged chunks), NDC, static camera, explicit rays + depth column."""
    from snerf_amd import classic
    g = golden("g20_render")
    H, W, f = int(g["H"]), int(g["W"]), float(g["focal"])
    coarse, fine = nerf(64, "f32", nerf_params(64)), nerf(64, "f32", nerf_params(64, True))
    kw = render_kw(coarse, fine)
    c2w = g["c2w"].to(DEV)
    with torch.no_grad():
        cases = {
            "a": classic.render(H, W, f, chunk=20, c2w=c2w, ndc=False, near=2., far=6., use_viewdirs=True, **kw),
            "b": classic.render(H, W, f, chunk=20, c2w=c2w, ndc=True, near=0., far=1., use_viewdirs=True, **kw),
            "c": classic.render(H, W, f, chunk=48, c2w=c2w, ndc=False, near=2., far=6., use_viewdirs=True, c2w_staticcam=g["c2w_static"].to(DEV), **kw),
            "d": classic.render(H, W, f, chunk=4, rays=(g["rays_o"].to(DEV), g["rays_d"].to(DEV)), ndc=False, near=2., far=6.,
                                use_viewdirs=True, depths=g["depths"].to(DEV), **kw),
        }
    for tag, out in cases.items():
        assert set(out[4].keys()) == {k[len(tag) + 3:] for k in g if k.startswith(tag + "_x_")}, "extras keys must equal the reference's"
        _check_render(tag, out, g, 1e-4)


def test_render_path_vs_reference_golden(backend, golden):
    from snerf_amd import classic
    g = golden("g20_render_path")
    coarse, fine = nerf(64, "f32", nerf_params(64)), nerf(64, "f32", nerf_params(64, True))
    rk = render_kw(coarse, fine); rk.pop("retraw")
    rk.update(ndc=False, near=2., far=6., use_viewdirs=True)
    with torch.no_grad():
        rgbs, disps = classic.render_path(g["poses"].to(DEV), list(g["hwf"].numpy()), 40, rk, render_factor=2)
    assert isinstance(rgbs, np.ndarray) and rgbs.shape == tuple(g["rgbs"].shape) and disps.shape == tuple(g["disps"].shape)
    close(torch.from_numpy(rgbs), g["rgbs"], 5e-3, 5e-3, "render_path rgbs")
    close(torch.from_numpy(disps), g["disps"], 5e-3, 5e-3, "render_path disps")


def test_nerf_rgb_and_alpha_model_branches(backend, golden):
    """NeRF_RGB forward + backward (no gradient into the frozen alpha model) and render_rays(network_fn=None, ...)."""
    from snerf_amd import classic
    g = golden("g20_nerf_rgb")
    alpha = nerf(64, "f32", {k: (v.flip(1) if v.dim() == 2 else v) for k, v in nerf_params(64).items()})
    shapes = [(k, s) for k, s in oc.nerf_param_shapes(W=64) if not k.startswith("alpha_linear")]
    own = common.fill_state_dict_({k: torch.empty(s) for k, s in shapes})
    sd = dict(own); sd.update({"alpha_model." + k: v for k, v in alpha.state_dict().items()})
    m = classic.NeRF_RGB(D=8, W=64, input_ch=63, input_ch_views=27, output_ch=4, skips=[4], use_viewdirs=True, alpha_model=alpha,
                         compute="f32", device=DEV)
    assert [k for k in m.state_dict() if not k.startswith("alpha_model.")] == list(own.keys())
    assert sorted(k for k in m.state_dict() if k.startswith("alpha_model.")) == sorted("alpha_model." + k for k in alpha.state_dict())
    m.load_state_dict(sd)
    nq, e, ev = query_fn()
    out = classic.run_network(g["pts"].to(DEV), g["viewdirs"].to(DEV), m, e, ev)
    close(out, g["run_network_out"], 1e-4, 1e-4, "NeRF_RGB run_network")
    ((out - g["target"].to(DEV)) ** 2).sum().backward()
    named = dict(m.named_parameters())
    for k in own:
        close(named[k].grad, g["grad_" + k], 1e-3, 1e-3 * float(g["grad_" + k].abs().max()), "NeRF_RGB grad " + k)
    assert all(p.grad is None for k, p in named.items() if k.startswith("alpha_model.")), "the alpha model is frozen"
    rb = g["ray_batch"].to(DEV)
    with torch.no_grad():
        r = classic.render_rays(rb, None, nq, N_samples=16, retraw=True, perturb=0., N_importance=32, network_fine=m)
        for k in ("rgb0", "disp0", "acc0", "weights", "z_vals_map"):
            close(r[k], g["rgbnet_" + k], 1e-4, 1e-4, "network_fn=None, NeRF_RGB fine: " + k)
        close(r["rgb_map"], g["rgbnet_rgb_map"], 5e-3, 5e-3, "network_fn=None, NeRF_RGB fine: rgb_map")
        plain = nerf(64, "f32", nerf_params(64, True))
        r = classic.render_rays(rb, None, nq, N_samples=16, retraw=True, perturb=0., N_importance=32, network_fine=plain)
        for k in ("rgb0", "disp0", "acc0", "weights"):
            close(r[k], g["plain_" + k], 1e-4, 1e-4, "network_fn=None, plain fine: " + k)
        close(r["rgb_map"], g["plain_rgb_map"], 5e-3, 5e-3, "network_fn=None, plain fine: rgb_map")


def test_module_prefixed_checkpoints_load(backend):
    """Checkpoints of the reference are saved from the DataParallel / DDP wrapper: every key starts with ``module.``
    (s-nerf/train.py:268; eval.py:72-74).  Both spellings load, mixed ones do not."""
    from snerf_amd import classic
    sd = nerf_params(64)
    m = nerf(64, "f32", {"module." + k: v for k, v in sd.items()})
    for k, v in m.state_dict().items():
        assert torch.equal(v.cpu(), sd[k])
    mixed = {("module." + k if i else k): v for i, (k, v) in enumerate(sd.items())}
    with pytest.raises(RuntimeError):
        m.load_state_dict(mixed)


def test_create_nerf_round_trip(backend, tmp_path):
    """create_nerf: models, optimiser and render kwargs of the reference's factory; reloads the newest checkpoint of basedir/expname,
    also one saved with the ``module.`` prefix."""
    from snerf_amd import classic
    args = argparse.Namespace(multires=10, multires_views=4, i_embed=0, use_viewdirs=True, N_importance=32, netdepth=8, netwidth=64,
                              netdepth_fine=8, netwidth_fine=64, alpha_model_path=None, no_coarse=False, weighted_loss=False,
                              netchunk=1 << 16, lrate=5e-4, basedir=str(tmp_path), expname="exp", ft_path=None, no_reload=False,
                              perturb=1., N_samples=16, white_bkgd=True, raw_noise_std=0., dataset_type="blender", no_ndc=False,
                              lindisp=False)
    tr, te, start, grad_vars, opt, conf = classic.create_nerf(args, compute="f32", device=DEV)
    assert start == 0 and conf is None and len(grad_vars) == 2 * 24 and isinstance(opt, torch.optim.Adam)
    assert set(tr) == {"network_query_fn", "perturb", "N_importance", "network_fine", "N_samples", "network_fn", "use_viewdirs", "white_bkgd",
                       "raw_noise_std", "ndc", "lindisp"} and tr["ndc"] is False
    assert te["perturb"] is False and te["raw_noise_std"] == 0. and te["network_fn"] is tr["network_fn"]
    os.makedirs(tmp_path / "exp")
    sdc, sdf = nerf_params(64), nerf_params(64, True)
    torch.save({"global_step": 7, "optimizer_state_dict": opt.state_dict(),
                "network_fn_state_dict": {"module." + k: v for k, v in sdc.items()}, "network_fine_state_dict": sdf},
               tmp_path / "exp" / "000007.tar")
    tr, te, start, _, _, _ = classic.create_nerf(args, compute="f32", device=DEV)
    assert start == 7
    assert all(torch.equal(v.cpu(), sdc[k]) for k, v in tr["network_fn"].state_dict().items())
    assert all(torch.equal(v.cpu(), sdf[k]) for k, v in tr["network_fine"].state_dict().items())
    # the kwargs drive render() as in the reference's training loop
    H, W, f = 4, 6, 5.0
    c2w = torch.tensor([[1., 0, 0, 0.1], [0, 1, 0, 0.2], [0, 0, 1, 4.0]])
    with torch.no_grad():
        rgb, disp, acc, depth, extras = classic.render(H, W, f, chunk=10, c2w=c2w.to(DEV), near=2., far=6., **te)
        ref = oc.render(H, W, f, sdc, sdf, chunk=10, c2w=c2w, ndc=False, near=2., far=6., use_viewdirs=True, N_samples=16, N_importance=32,
                        white_bkgd=True)
    assert rgb.shape == (H, W, 3) and depth.shape == (H, W)
    close(extras["rgb0"], ref[4]["rgb0"], 1e-4, 1e-4, "create_nerf -> render rgb0")


# ------------------------------------------------------------------------------------------------ config 1 at its stated size (M1)
@pytest.mark.gpu
@pytest.mark.parametrize("compute", ["f32", "bf16"])
def test_config1_m1_full_size(golden, compute):
    """BASELINE.json configs[0] at SURVEY.md section 8d's size M1: one 400 x 400 pinhole camera, focal 555.5, orbit pose, near 2 /
    far 6, white background, 64 coarse samples, N_importance 0, NeRF 8 x 256 -- the whole frame through render(); every 97th pixel
    against the reference's own render of that pixel."""
    from snerf_amd import classic
    g = golden("g20_m1")
    H, W, f = int(g["H"]), int(g["W"]), float(g["focal"])
    net = classic.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=4, skips=[4], use_viewdirs=True, compute=compute)
    net.load_state_dict(nerf_params(256))
    with torch.no_grad():
        rgb, disp, acc, depth, _ = classic.render(H, W, f, chunk=1 << 15, c2w=g["c2w"].cuda(), ndc=False, near=2., far=6., use_viewdirs=True,
                                                  network_fn=net, network_query_fn=query_fn()[0], N_samples=64, N_importance=0,
                                                  network_fine=None, perturb=0., white_bkgd=True, raw_noise_std=0.)
    assert rgb.shape == (H, W, 3) and disp.shape == (H, W)
    idx = g["idx"].cuda()
    pick = lambda t: t.reshape(H * W, -1)[idx].squeeze(-1).cpu()
    psnr = common.psnr(pick(rgb), g["rgb_map"])
    print(f"config 1 / M1 ({compute}): PSNR vs the reference's pixels {psnr:.1f} dB")
    if compute == "f32":
        close(pick(rgb), g["rgb_map"], 1e-4, 1e-4, "M1 rgb")
        close(pick(acc), g["acc_map"], 1e-4, 1e-4, "M1 acc")
        close(pick(depth), g["depth_map"], 1e-4, 1e-4, "M1 depth")
        assert psnr > 90
    else:
        assert psnr > 38, psnr                    # bf16 operands, fp32 accumulation, 64 samples: measured ~50 dB
        close(pick(acc), g["acc_map"], 3e-2, 3e-2, "M1 acc (bf16)")
    white = 1.0 - acc[..., None]
    assert float((rgb - white).min()) > -1e-3     # white background: rgb >= 1 - acc


# ---------------------------------------------------------------- branches beyond the S-NeRF configuration (golden g27 - g29) ----
def _fill(net, flip=False):
    sd = common.fill_state_dict_({k: torch.empty_like(v) for k, v in net.state_dict().items()})
    net.load_state_dict({k: v.flip(0) for k, v in sd.items()} if flip else sd)
    return net


def test_nerf_without_viewdirs_matches_the_reference(backend, golden):
    """NeRF(use_viewdirs=False, output_ch=5) (run_nerf_helpers.py:100-124; create_nerf's output_ch = 5, render.py:180-183): the
    reference's state_dict keys (views_linears.0 exists, unused), run_network on viewdirs=None, the parameter gradients, and
    render_rays on an 8-column ray batch -- against the reference's own outputs (g27)."""
    from snerf_amd import classic
    g = golden("g27_no_viewdirs")
    mk = lambda: classic.NeRF(D=8, W=64, input_ch=63, input_ch_views=0, output_ch=5, skips=[4], use_viewdirs=False, compute="f32", device=DEV)
    coarse, fine = _fill(mk()), _fill(mk(), True)
    assert list(coarse.state_dict().keys()) == [str(k) for k in g["param_names"]]
    embed_fn, ic = classic.get_embedder(10, 0)
    nq = classic.make_network_query_fn(embed_fn, None)
    run = classic.run_network(g["pts"].to(DEV), None, coarse, embed_fn, None)
    assert run.shape == (12, 8, 5)
    close(run, g["run_network_out"], 1e-4, 1e-4, "run_network (no viewdirs)")
    ((run - g["target"].to(DEV)) ** 2).sum().backward()
    named = dict(coarse.named_parameters())
    for k in named:
        if "grad_" + k in g:
            ref = g["grad_" + k]
            close(named[k].grad / (ref.abs().max() + 1e-12), ref / (ref.abs().max() + 1e-12), 0, 5e-4, "grad " + k)
        else:
            assert k.startswith("views_linears.0") and (named[k].grad is None or float(named[k].grad.abs().max()) == 0.0), k
    with pytest.raises(TypeError, match="use_viewdirs=False"):
        classic.run_network(g["pts"].to(DEV), g["ray_batch"][:, 3:6].to(DEV), coarse, embed_fn, None)
    with torch.no_grad():
        rr = classic.render_rays(g["ray_batch"].to(DEV), coarse, nq, N_samples=16, retraw=True, perturb=0., N_importance=32, network_fine=fine)
    for k in ("rgb_map", "disp_map", "acc_map", "depth_map", "z_vals_map", "weights", "rgb0", "disp0", "acc0"):
        close(rr[k], g["rr_" + k], 2e-4, 2e-4, "render_rays " + k)
    assert rr["raw"].shape == g["rr_raw"].shape


def test_identity_embedding_matches_the_reference(backend, golden):
    """get_embedder(multires, i=-1) = (identity, 3) (run_nerf_helpers.py:55-57): NeRF(input_ch=3, input_ch_views=3) sees the raw points
    and directions -- run_network, gradients, render_rays against the reference (g28)."""
    from snerf_amd import classic
    g = golden("g28_identity_embed")
    e, ic = classic.get_embedder(10, -1)
    ed, icv = classic.get_embedder(4, -1)
    assert ic == 3 and icv == 3
    x = torch.rand(5, 3)
    assert e(x) is x
    mk = lambda: classic.NeRF(D=8, W=64, input_ch=3, input_ch_views=3, output_ch=4, skips=[4], use_viewdirs=True, compute="f32", device=DEV)
    coarse, fine = _fill(mk()), _fill(mk(), True)
    nq = classic.make_network_query_fn(e, ed)
    run = classic.run_network(g["pts"].to(DEV), g["viewdirs"].to(DEV), coarse, e, ed)
    close(run, g["run_network_out"], 1e-4, 1e-4, "run_network (identity embedding)")
    ((run - g["target"].to(DEV)) ** 2).sum().backward()
    for k, p in coarse.named_parameters():
        ref = g["grad_" + k]
        close(p.grad / (ref.abs().max() + 1e-12), ref / (ref.abs().max() + 1e-12), 0, 5e-4, "grad " + k)
    with torch.no_grad():
        rr = classic.render_rays(g["ray_batch"].to(DEV), coarse, nq, N_samples=16, retraw=True, perturb=0., N_importance=32, network_fine=fine)
    for k in ("rgb_map", "disp_map", "acc_map", "depth_map", "weights", "rgb0", "disp0", "acc0"):
        close(rr[k], g["rr_" + k], 2e-4, 2e-4, "render_rays " + k)


def test_render_with_per_ray_near_far_tensors(backend, golden):
    """render(..., near=<[N,1] tensor>, far=<[N,1] tensor>) (render.py:74 broadcasts them against the rays) against the reference (g29)"""
    from snerf_amd import classic
    g = golden("g29_near_far")
    mk = lambda: classic.NeRF(D=8, W=64, input_ch=63, input_ch_views=27, output_ch=4, skips=[4], use_viewdirs=True, compute="f32", device=DEV)
    coarse, fine = _fill(mk()), _fill(mk(), True)
    e, _ = classic.get_embedder(10, 0)
    ed, _ = classic.get_embedder(4, 0)
    kw = dict(network_fn=coarse, network_query_fn=classic.make_network_query_fn(e, ed), N_samples=16, N_importance=32, network_fine=fine, perturb=0.,
              white_bkgd=False, raw_noise_std=0.)
    with torch.no_grad():
        out = classic.render(6, 8, 7.5, chunk=5, rays=(g["rays_o"].to(DEV), g["rays_d"].to(DEV)), ndc=False, near=g["near"].to(DEV), far=g["far"].to(DEV),
                             use_viewdirs=True, **kw)
    for i in range(4):
        close(out[i], g[str(i)], 2e-4, 2e-4, f"render output {i}")
    close(out[4]["z_vals_map"], g["x_z_vals_map"], 1e-6, 1e-6, "z_vals")


def test_fine_pass_early_termination_host_logic(backend):
    """render_rays(ert=(eps_t, G | schedule)) on both backends (the GPU-sized version with the fused kernels is tests/test_ert.py): without
    termination the grouped fine pass reproduces the plain render exactly; on an opaque medium fewer samples are evaluated and acc / rgb
    stay inside the exact bound eps_t; the evaluated samples' raw outputs are the plain render's."""
    from snerf_amd import classic
    torch.manual_seed(0)
    mk = lambda: classic.NeRF(D=8, W=64, input_ch=63, input_ch_views=27, output_ch=4, skips=[4], use_viewdirs=True, compute="f32", device=DEV)
    coarse, fine = mk(), mk()
    e, _ = classic.get_embedder(10, 0)
    ed, _ = classic.get_embedder(4, 0)
    q = classic.make_network_query_fn(e, ed)
    g = torch.Generator().manual_seed(1)
    N = 150
    d = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1)
    o = torch.randn(N, 3, generator=g) * 0.1 + torch.tensor([0.0, 0.0, 4.0])
    rays = torch.cat([o, -d, torch.full((N, 1), 2.0), torch.full((N, 1), 6.0), -d], -1).to(DEV)
    kw = dict(network_fn=coarse, network_query_fn=q, N_samples=64, perturb=0.0, N_importance=128, network_fine=fine, retraw=True)
    with torch.no_grad():
        full = classic.render_rays(rays, **kw)
        for G in (40, (96, 16)):
            same = classic.render_rays(rays, ert=(-1.0, G), **kw)
            for k in ("rgb_map", "acc_map", "depth_map", "raw"):
                assert torch.equal(full[k], same[k]), (G, k)
        for net in (coarse, fine):
            dict(net.named_parameters())["alpha_linear.bias"] += 40.0
            net.arena.bump()
        full = classic.render_rays(rays, **kw)
        fast = classic.render_rays(rays, ert=(1e-3, 16), **kw)
    ev, tot = (int(v) for v in fast["ert_evals"].sum(0))
    assert tot == N * 192 and 0 < ev < 0.85 * N * 192
    assert float((fast["acc_map"] - full["acc_map"]).abs().max()) <= 1e-3 * 1.01 + 1e-6
    assert float((fast["rgb_map"] - full["rgb_map"]).abs().max()) <= 1e-3 * 1.01 + 1e-6
    ev = (fast["raw"] != 0).any(-1)
    assert torch.equal(fast["raw"][ev], full["raw"][ev])
    with pytest.raises(NotImplementedError):
        classic.render_rays(rays, ert=(1e-3, 16), **kw)                      # (grad mode on)
    with torch.no_grad(), pytest.raises(ValueError):
        classic.render_rays(rays, ert=(1.0, 16), **kw)
