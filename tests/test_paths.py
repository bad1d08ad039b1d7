"""End-to-end parity of the drop-in operator API (render_rays / run_network / MipNerfModel) against the CPU
oracle and the golden vectors captured from the reference.  fp32 MFMA mode: rgb/depth within 1e-4 relative
(BASELINE.json north_star); bf16 mode: tolerance stated per check."""
import numpy as np
import pytest
import torch

from oracle import classic as oc
from oracle import common, mip as om

from cpu_ops_emulation import emulate_ops

import json
import os

_BOUNDS_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bf16_grad_bounds.json")
_BF16_BOUNDS = json.load(open(_BOUNDS_FILE)) if os.path.exists(_BOUNDS_FILE) else {}


@pytest.fixture(params=[pytest.param("hip", marks=pytest.mark.gpu), "emulated"])
def backend(request):
    """"hip": the real kernels through the C-ABI on the GPU box.  "emulated": snerf_amd.ops swapped for the CPU
    emulation (tests/cpu_ops_emulation.py) so the host logic above the C-ABI is checked without a GPU."""
    global DEV
    if request.param == "hip":
        DEV = "cuda"
        yield "hip"
    else:
        DEV = "cpu"
        with emulate_ops():
            yield "emulated"
    DEV = "cuda"


DEV = "cuda"


def D(t):
    return t.to(DEV)


def close(a, b, rtol, atol, what=""):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    assert torch.equal(torch.isnan(a), torch.isnan(b)), f"{what}: NaN pattern differs"
    a = torch.nan_to_num(a, nan=0.0); b = torch.nan_to_num(b, nan=0.0)
    err = (a - b).abs(); tol = atol + rtol * b.abs()
    bad = err > tol
    assert not bool(bad.any()), f"{what}: {int(bad.sum())}/{bad.numel()} out of tol, max err {err.max().item():.3e}, max ref {b.abs().max().item():.3e}"


def check_grads(named, ref_params, keys, compute, tol):
    """fp32: element-wise (relative to the parameter's largest gradient).  bf16: norm-wise per parameter -- a bf16
    rounding of a resampled depth re-phases the 2^10..2^15 encoding bands, so single high-band entries legitimately
    differ by O(1) while the gradient as a whole agrees."""
    for k in keys:
        gref = ref_params[k].grad
        got = named[k].grad.detach().cpu()
        if compute == "f32":
            scale = gref.abs().max().item() + 1e-12
            close(got / scale, gref / scale, 0, tol * 5, "grad " + k)
        elif compute == "bf16x3":
            # the arithmetic is fp32-class (loss / outputs held to the fp32 bounds), but a pre-activation within ~1e-5 of zero flips its
            # ReLU mask -- one element of these few hundred rows is O(1e-3) of a gradient's norm: norm-wise bound, 25x below bf16's
            rel = ((got - gref).norm() / (gref.norm() + 1e-12)).item()
            print(f"MEASURED split-bf16 grad {k}: rel L2 {rel:.3e}")
            assert rel < 6e-3, f"grad {k}: relative L2 error {rel:.3e}"
        else:
            # bf16: held PER PARAMETER to 1.3 x the error measured for it (tests/bf16_grad_bounds.json: per test and backend -- the HIP
            # kernels on an MI355X, the CPU emulation here), with a floor of 2e-3 for parameters whose error is at the noise level; a
            # parameter without an entry falls back to the old global bound (the global 0.113 hid a 30 % regression of the
            # first layer).  SNERF_DUMP_BF16_BOUNDS=<file>: record the measurements instead (tools/measure_bf16_grad_bounds.sh).
            rel = ((got - gref).norm() / (gref.norm() + 1e-12)).item()
            print(f"MEASURED bf16 grad {k}: rel L2 {rel:.3e}")
            test = os.environ.get("PYTEST_CURRENT_TEST", "?").split("::")[-1].split(" ")[0]
            be = "hip" if DEV == "cuda" else "emulated"
            dump = os.environ.get("SNERF_DUMP_BF16_BOUNDS")
            if dump:
                d = json.load(open(dump)) if os.path.exists(dump) else {}
                d.setdefault(test, {}).setdefault(be, {})[k] = max(rel, d.get(test, {}).get(be, {}).get(k, 0.0))
                json.dump(d, open(dump, "w"), indent=1, sort_keys=True)
                continue
            measured = _BF16_BOUNDS.get(test, {}).get(be, {}).get(k)
            bound = min(2 * tol, 0.113) if measured is None else max(1.3 * measured, 2e-3)
            assert rel < bound, f"grad {k}: relative L2 error {rel:.3e} (bound {bound:.3e}, measured {measured})"


def nerf_params(W, flip=False):
    """formula weights: the ones the golden vectors were captured with"""
    sd = common.fill_state_dict_({k: torch.empty(s) for k, s in oc.nerf_param_shapes(W=W)})
    return {k: v.flip(0) for k, v in sd.items()} if flip else sd


def random_params(shapes, seed, bias_boost=()):
    """seeded N(0, 1/in) weights (formula weights drive some widths to sigma <= 0 everywhere, i.e. zero gradients)"""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, s in shapes:
        if len(s) == 2:
            sd[k] = torch.randn(s, generator=g) * (1.4 / s[1] ** 0.5)
        else:
            sd[k] = torch.randn(s, generator=g) * 0.1 + (0.5 if k in bias_boost else 0.0)
    return sd


def make_nerf(W, compute, sd):
    from snerf_amd import classic
    m = classic.NeRF(D=8, W=W, input_ch=63, input_ch_views=27, output_ch=4, skips=[4], use_viewdirs=True, compute=compute, device=DEV)
    assert list(m.state_dict().keys()) == [k for k, _ in oc.nerf_param_shapes(W=W)], "state_dict keys must equal the reference module's"
    m.load_state_dict(sd)
    return m


def test_classic_render_rays_vs_reference_golden(backend, golden):
    """W=64 fp32: reproduces the reference's own render_rays outputs (captured in the build container)."""
    from snerf_amd import classic
    g = golden("g9_render_rays")
    coarse, fine = make_nerf(64, "f32", nerf_params(64)), make_nerf(64, "f32", nerf_params(64, True))
    embed_fn, _ = classic.get_embedder(10, 0); embeddirs_fn, _ = classic.get_embedder(4, 0)
    nq = classic.make_network_query_fn(embed_fn, embeddirs_fn)
    rb = g["ray_batch"].to(DEV)
    with torch.no_grad():
        pts = rb[:, None, 0:3] + rb[:, None, 3:6] * torch.linspace(2, 6, 8).to(DEV)[None, :, None]
        close(classic.run_network(pts, rb[:, -3:], coarse, embed_fn, embeddirs_fn), g["run_network_out"], 1e-4, 1e-4, "run_network")
        r0 = classic.render_rays(rb, coarse, nq, N_samples=16, retraw=True, perturb=0., N_importance=0, white_bkgd=True)
        for k in ("rgb_map", "disp_map", "acc_map", "depth_map", "z_vals_map", "weights", "raw"):
            close(r0[k], g["c_" + k], 1e-4, 1e-4, "coarse-only " + k)
        # randomized (numpy-seeded, pytest=True) hierarchical case: no exact u == cdf ties, so the fine pass matches too
        r2 = classic.render_rays(rb, coarse, nq, N_samples=16, retraw=True, perturb=1., N_importance=32, network_fine=fine, pytest=True)
        for k in ("rgb_map", "disp_map", "acc_map", "depth_map", "z_vals_map", "weights", "rgb0", "disp0", "acc0", "z_std"):
            close(r2[k], g["pt_" + k], 1e-4, 1e-4, "hierarchical " + k)
        close(r2["raw"], g["pt_raw"], 1e-3, 1e-3, "hierarchical raw")


@pytest.mark.parametrize("compute,W,tol", [("f32", 256, 1e-4), ("bf16", 256, 3e-2),
                                           # the fp32-class split modes on the classic network: the exact mode's bounds
                                           ("f16f8", 256, 1e-4), ("bf16x3", 256, 1e-4)])
def test_classic_render_rays_vs_oracle(backend, compute, W, tol):
    from snerf_amd import classic
    exact = compute in ("f32", "f16f8", "bf16x3")
    pc = random_params(oc.nerf_param_shapes(W=W), 11, ("alpha_linear.bias",))
    pf = random_params(oc.nerf_param_shapes(W=W), 12, ("alpha_linear.bias",))
    coarse, fine = make_nerf(W, compute, pc), make_nerf(W, compute, pf)
    embed_fn, _ = classic.get_embedder(10, 0); embeddirs_fn, _ = classic.get_embedder(4, 0)
    nq = classic.make_network_query_fn(embed_fn, embeddirs_fn)
    gg = torch.Generator().manual_seed(1)
    n = 150
    ro = torch.randn(n, 3, generator=gg) * 0.2
    rd = torch.nn.functional.normalize(torch.randn(n, 3, generator=gg), dim=-1) * (1 + 0.2 * torch.rand(n, 1, generator=gg))
    rb = torch.cat([ro, rd, torch.full((n, 1), 2.0), torch.full((n, 1), 6.0), torch.nn.functional.normalize(rd, dim=-1)], -1)
    t_rand, u = torch.rand(n, 64, generator=gg), torch.rand(n, 128, generator=gg)
    ref = oc.render_rays(rb, pc, pf, 64, 128, t_rand=t_rand, u=u, retraw=True)
    with torch.no_grad():
        out = classic.render_rays(rb.to(DEV), coarse, nq, N_samples=64, retraw=True, perturb=1., N_importance=128, network_fine=fine,
                                  t_rand=t_rand.to(DEV), u=u.to(DEV), return_inds=True)
    assert torch.equal(out["z_vals_map"].cpu(), ref["z_vals_map"]), "stratified z must be bit-exact"
    # raw2outputs gives the last sample an interval of 1e10, so alpha_last = 1 - exp(-relu(sigma_last) * 1e10) jumps from 0
    # to 1 at sigma_last = 0: rays whose last raw sigma is within rounding noise of 0 are excluded from value checks.
    ref_c = oc.render_rays(rb, pc, None, 64, 0, t_rand=t_rand, retraw=True)
    ok0 = ref_c["raw"][:, -1, 3].abs() > (1e-3 if exact else 5e-2)
    assert ok0.float().mean() > 0.8
    for k in ("rgb0", "acc0", "weights"):
        close(out[k][ok0], ref[k][ok0], tol, tol, k)
    # The inverse-CDF step divides by (c1 - c0), which is ~1e-5-small in empty intervals: a 1e-7 change of a coarse
    # weight moves a resampled depth by up to ~1e-3 there (a conditioning property of the reference algorithm, not of
    # an implementation).  So parity of the hierarchical pass is checked stage by stage on IDENTICAL inputs:
    #  (1) sampler: our z_samples/inds == oracle sampler applied to OUR coarse weights (bit-exact),
    w_ours = out["weights"].cpu()
    zs_ref, inds_ref = oc.sample_pdf(0.5 * (ref["z_vals_map"][:, 1:] + ref["z_vals_map"][:, :-1]), w_ours[:, 1:-1], u)
    assert torch.equal(out["inds"].cpu(), inds_ref), "interval indices must be bit-exact given identical weights"
    assert torch.equal(out["z_samples"].cpu(), zs_ref), "resampled depths must be bit-exact given identical weights"
    z_fine = out["z_vals_fine"].cpu()
    assert torch.equal(z_fine, torch.sort(torch.cat([ref["z_vals_map"], zs_ref], -1), -1)[0]), "merge-sort must be bit-exact"
    #  (2) fine network + compositing on those identical depths,
    pts = rb[:, None, 0:3] + rb[:, None, 3:6] * z_fine[..., None]
    raw_ref = oc.run_network(pts, rb[:, -3:], pf)
    fin = oc.raw2outputs(raw_ref, z_fine, rb[:, 3:6])
    rtol_raw = tol * 10
    close(out["raw"], raw_ref, rtol_raw, rtol_raw, "fine raw on identical depths")
    ok1 = raw_ref[:, -1, 3].abs() > (1e-3 if exact else 5e-2)
    assert ok1.float().mean() > 0.8
    for got, want, k in zip((out["rgb_map"], out["disp_map"], out["acc_map"], out["depth_map"]), (fin[0], fin[1], fin[2], fin[4]),
                            ("rgb_map", "disp_map", "acc_map", "depth_map")):
        close(got[ok1], want[ok1], tol, tol if k != "depth_map" or exact else 5e-2, k + " on identical depths")
    #  (3) and end to end against the all-oracle pipeline with the conditioning-aware tolerance.
    e2e = 2e-3 if exact else 5e-2
    ok = ok0 & ok1 & (ref["raw"][:, -1, 3].abs() > (1e-3 if exact else 5e-2))
    close(out["rgb_map"][ok], ref["rgb_map"][ok], e2e, e2e, "rgb_map end to end")
    close(out["z_std"], ref["z_std"], e2e, e2e, "z_std end to end")


@pytest.mark.parametrize("compute,W,tol", [("f32", 64, 2e-4), ("bf16", 128, 6e-2),
                                           # the split modes on the classic network (per-layer launches): fp32-class loss, gradients as the mode's backward
                                           ("bf16x3", 128, 3e-4), ("f16f8", 128, 3e-4), ("bf16x3_fwd", 128, 3e-4), ("fp16", 128, 2e-2)])
def test_classic_backward_vs_autograd(backend, compute, W, tol):
    from snerf_amd import classic
    pc = random_params(oc.nerf_param_shapes(W=W), 13, ("alpha_linear.bias",))
    net = make_nerf(W, compute, pc)
    embed_fn, _ = classic.get_embedder(10, 0); embeddirs_fn, _ = classic.get_embedder(4, 0)
    nq = classic.make_network_query_fn(embed_fn, embeddirs_fn)
    gg = torch.Generator().manual_seed(2)
    n, S = 40, 24
    ro = torch.randn(n, 3, generator=gg) * 0.2
    rd = torch.nn.functional.normalize(torch.randn(n, 3, generator=gg), dim=-1)
    rb = torch.cat([ro, rd, torch.full((n, 1), 2.0), torch.full((n, 1), 6.0), rd], -1)
    target = torch.rand(n, 3, generator=gg)
    pr = {k: v.clone().requires_grad_(True) for k, v in pc.items()}
    ref = oc.render_rays(rb, pr, None, S, 0)
    loss_ref = ((ref["rgb_map"] - target) ** 2).mean() + 0.1 * ref["depth_map"].mean() + 0.05 * (ref["weights"] ** 2).sum()
    loss_ref.backward()
    assert ref["acc_map"].mean().item() > 0.05 and pr["pts_linears.0.weight"].grad.norm().item() > 1e-4, "degenerate test scene"
    out = classic.render_rays(rb.to(DEV), net, nq, N_samples=S, perturb=0.)
    loss = ((out["rgb_map"] - target.to(DEV)) ** 2).mean() + 0.1 * out["depth_map"].mean() + 0.05 * (out["weights"] ** 2).sum()
    loss.backward()
    close(loss, loss_ref, tol, tol, "loss")
    if compute in ("f16f8", "bf16x3_fwd"):                      # fp32-class forward, 16-bit single-pass backward: norm-wise, well inside the bf16 mode's errors
        for k in pc:
            rel = ((dict(net.named_parameters())[k].grad.detach().cpu() - pr[k].grad).norm() / (pr[k].grad.norm() + 1e-12)).item()
            assert rel < 3e-2, f"{compute} grad {k}: rel L2 {rel:.3e}"
        return
    check_grads(dict(net.named_parameters()), pr, pc, compute, 8e-2 if compute == "fp16" else tol)


def mip_params(hidden, prop_hidden):
    return common.fill_state_dict_({k: torch.empty(s) for k, s in om.mipnerf_param_shapes(hidden=hidden, prop_hidden=prop_hidden)})


def make_mip(hidden, prop_hidden, n_samples, n_fine, compute, sd):
    from snerf_amd import mipnerf
    m = mipnerf.MipNerfModel(n_samples=n_samples, N_fine=n_fine, no_warp_sample=0, ray_shape="cone", fn=1, radius=3., transform_idx=0,
                             real=True, rgb_layer=3, hidden_layer=hidden, density_noise=0., max_deg_point=16,
                             proposal_hidden_layer=prop_hidden, proposal_loss=True, compute=compute, device=DEV)
    assert list(m.state_dict().keys()) == [k for k, _ in om.mipnerf_param_shapes(hidden=hidden, prop_hidden=prop_hidden)]
    m.load_state_dict(sd)
    return m


def test_mipnerf_forward_vs_reference_golden(backend, golden):
    """hidden 64 fp32: reproduces MipNerfModel.forward of the reference (deterministic and randomized draws)."""
    from snerf_amd import mipnerf
    g, r = golden("g8_mipnerf_det"), golden("g8_mipnerf_rand")
    rays = mipnerf.Rays(**{k[len("rays_"):]: v.to(DEV) for k, v in g.items() if k.startswith("rays_")})
    m = make_mip(64, 64, 16, 17, "f32", mip_params(64, 64))
    with torch.no_grad():
        ret = m(rays, False, False, 0.)
        assert ret[0][0] is None and ret[1][3] is None
        close(ret[0][3], g["l0_s_vals"], 0, 0, "s0"); close(ret[0][4], g["l0_weights"], 1e-4, 1e-6, "w0")
        close(ret[0][1], g["l0_distance"], 1e-4, 1e-4, "dist0"); close(ret[0][2], g["l0_acc"], 1e-4, 1e-5, "acc0")
        close(ret[1][4], g["l1_s_vals"], 1e-4, 1e-5, "s1")
        close(ret[1][0], g["l1_rgb"], 1e-4, 1e-4, "rgb"); close(ret[1][1], g["l1_distance"], 1e-4, 1e-4, "distance")
        close(ret[1][2], g["l1_acc"], 1e-4, 1e-5, "acc"); close(ret[1][5], g["l1_weights"], 1e-4, 1e-5, "w1")
        ret = m(rays, True, False, 0., s_rand=r["s_rand"].to(DEV), u=om.rand_u(17, r["jitter"]).to(DEV))
        close(ret[0][3], r["l0_s_vals"], 0, 0, "rand s0"); close(ret[0][4], r["l0_weights"], 1e-4, 1e-6, "rand w0")
        close(ret[1][4], r["l1_s_vals"], 1e-4, 1e-5, "rand s1"); close(ret[1][0], r["l1_rgb"], 1e-4, 1e-4, "rand rgb")
        close(ret[1][1], r["l1_distance"], 1e-4, 1e-4, "rand distance")


@pytest.mark.parametrize("compute,hidden,S0,P1,n,tol", [("f32", 1024, 64, 129, 96, 1e-4), ("bf16", 1024, 64, 129, 96, 4e-2),
                                                       ("f32", 128, 128, 128, 70, 1e-4),
                                                       # split-bf16 (three bf16 MFMA passes per product): the SAME 1e-4 bounds as exact fp32
                                                       ("bf16x3", 1024, 64, 129, 96, 1e-4), ("bf16x3", 128, 128, 128, 70, 1e-4),
                                                       ("f16f8", 1024, 64, 129, 96, 1e-4), ("fp16", 1024, 64, 129, 96, 1e-2)])
def test_mipnerf_forward_vs_oracle(backend, compute, hidden, S0, P1, n, tol):
    """Full-width network at the BASELINE shape (64 proposal + 128 fine intervals) and the shipped 128+127 shape."""
    from snerf_amd import mipnerf
    sd = mip_params(hidden, 256)
    rays_c = common.synthetic_rays(n, seed=5)
    gg = torch.Generator().manual_seed(6)
    s_rand = torch.rand(n, S0 + 1, generator=gg)
    u = om.rand_u(P1, torch.empty(n, P1).uniform_(0, 1 / P1 - om.EPS32, generator=gg))
    ref, aux = om.mipnerf_forward(sd, rays_c, S0, P1, s_rand=s_rand, u=u, return_aux=True)
    m = make_mip(hidden, 256, S0, P1, compute, sd)
    rays = mipnerf.Rays(**{k: v.to(DEV) for k, v in rays_c.items()})
    with torch.no_grad():
        ret = m(rays, True, False, 0., s_rand=s_rand.to(DEV), u=u.to(DEV))
    assert torch.equal(ret[0][3].cpu(), ref[0][3]), "level-0 fence posts must be bit-exact"
    close(ret[0][4], ref[0][4], tol, tol * 1e-2, "w0"); close(ret[0][1], ref[0][1], tol, tol, "dist0")
    if compute in ("f32", "bf16x3", "f16f8"):
        close(ret[1][4], ref[1][4], 1e-4, 1e-5, "s1")
        close(ret[1][0], ref[1][0], tol, tol, "rgb"); close(ret[1][1], ref[1][1], tol, tol, "distance"); close(ret[1][2], ref[1][2], tol, tol, "acc")
        psnr = common.psnr(ret[1][0].cpu(), ref[1][0])
        assert psnr > 80.0, f"PSNR vs oracle {psnr:.1f} dB"
    else:
        close(ret[1][0], ref[1][0], tol, tol, "rgb (bf16)"); close(ret[1][2], ref[1][2], tol, tol, "acc (bf16)")
        psnr = common.psnr(ret[1][0].cpu(), ref[1][0])
        print(f"MEASURED bf16 forward vs oracle: PSNR {psnr:.1f} dB, rgb max abs {float((ret[1][0].cpu() - ref[1][0]).abs().max()):.3e}, "
              f"acc max abs {float((ret[1][2].cpu() - ref[1][2]).abs().max()):.3e}, "
              f"distance rel max {float(((ret[1][1].cpu() - ref[1][1]).abs() / ref[1][1].abs()).max()):.3e}, w0 max abs {float((ret[0][4].cpu() - ref[0][4]).abs().max()):.3e}")
        # measured on MI355X (profiles/r2_a_bf16_bounds.txt): PSNR 95.2 dB, rgb max abs 3.8e-5, acc 4.2e-7, distance rel 2.5e-5, w0 8.3e-6;
        # the bounds sit within 10 dB / 3-5x of that, so a regression of the bf16 path cannot hide behind them
        if backend == "hip":
            assert psnr > 85.0, f"bf16 PSNR vs fp32 oracle {psnr:.1f} dB"
            assert float((ret[1][0].cpu() - ref[1][0]).abs().max()) < 2e-4 and float((ret[1][2].cpu() - ref[1][2]).abs().max()) < 3e-6
            assert float(((ret[1][1].cpu() - ref[1][1]).abs() / ref[1][1].abs()).max()) < 1.5e-4 and float((ret[0][4].cpu() - ref[0][4]).abs().max()) < 5e-5
        else:
            assert psnr > 35.0, f"bf16 PSNR vs fp32 oracle {psnr:.1f} dB (emulated kernels: torch bf16 matmul roundings)"


@pytest.mark.parametrize("compute,hidden,tol", [("f32", 64, 3e-4), ("bf16", 128, 8e-2), ("bf16x3", 64, 3e-4),
                                                # compute="fp16" on the mip path: fp16 operands on the f16 MFMA (11 bits instead of bf16's 8), the backward on
                                                # power-of-two scaled gradients (mlp._Net._scaled_backward); held to a quarter of the bf16 bound
                                                ("fp16", 128, 2e-2)])
def test_mipnerf_backward_vs_autograd(backend, compute, hidden, tol):
    from snerf_amd import mipnerf
    S0, P1, n = 24, 25, 36
    # seeded N(0, 1.4^2 / fan_in) weights: the formula weights of the goldens are rank 2 (sin(a i + b j + c)), which makes every
    # back-propagated signal a near-cancelling sum and amplifies bf16 rounding to O(1) relative errors in the early layers
    sd = random_params(om.mipnerf_param_shapes(hidden=hidden, prop_hidden=64), 21, ("mlp.density_layer.bias", "proposal.density_layer.bias"))
    rays_c = common.synthetic_rays(n, seed=7)
    gg = torch.Generator().manual_seed(8)
    target, tdepth = torch.rand(n, 3, generator=gg), torch.rand(n, generator=gg) * 50 + 5
    pr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    m = make_mip(hidden, 64, S0, P1, compute, sd)
    rays = mipnerf.Rays(**{k: v.to(DEV) for k, v in rays_c.items()})
    ret = m(rays, False, False, 0.)
    # bf16: a bf16 rounding of a proposal weight moves a resampled fence post, which re-phases the 2^10..2^15 bands of the level-1
    # encoding -- a conditioning property of the algorithm that would drown every level-1 gradient comparison.  The oracle is
    # therefore evaluated AT the fence posts the bf16 run resampled (they carry no gradient: stop_level_grad), and then EVERY
    # parameter gradient, the level-1 network's included, is held to the norm-wise bf16 bound.
    ref = om.mipnerf_forward(pr, rays_c, S0, P1, s1_override=None if compute in ("f32", "bf16x3") else ret[1][4].detach().cpu())

    def loss_fn(ret, tgt, td):  # RGB MSE + disparity-L1 depth on both levels (loss_factory.py:5-11, 26-37) + a weights term
        l = ((ret[1][0] - tgt) ** 2).mean()
        l = l + 0.2 * ((1 / ret[1][1] - 1 / td).abs()).mean() + 0.2 * 0.2 * ((1 / ret[0][1] - 1 / td).abs()).mean()
        return l + 0.01 * (ret[0][4] ** 2).sum() + 0.01 * ret[1][2].mean()
    loss_ref = loss_fn([[None, ref[0][1], ref[0][2], ref[0][3], ref[0][4]], [ref[1][0], ref[1][1], ref[1][2], None, ref[1][4], ref[1][5]]], target, tdepth)
    loss_ref.backward()
    loss = loss_fn(ret, target.to(DEV), tdepth.to(DEV))
    loss.backward()
    close(loss, loss_ref, tol, tol, "loss")
    named = dict(m.named_parameters())
    check_grads(named, pr, sd, compute, tol)


def test_mipnerf_split_forward_plain_backward(backend):
    """compute="bf16x3_fwd": the three-pass split-bf16 FORWARD (renders held to the fp32 contract) with a ONE-pass bf16 BACKWARD.
    (1) its outputs are bit-identical to compute="bf16x3" (the same forward launches); (2) loss and outputs against the oracle at the
    fp32 bounds; (3) every parameter gradient against the oracle's autograd norm-wise, and no worse than 1.5 x the plain bf16 mode's
    error on the same problem (the backward rounds its operands once, like that mode, but from exact forward values).  256-wide
    networks and 4800 rows: the persistent NT kernel with bit masks, the 8-phase weight gradient on the hi halves (M >= 4096), the
    128-wide colour head through the bf16-mask epilogue with a split-layout mask source."""
    from snerf_amd import mipnerf
    S0, P1, n, hidden = 24, 25, 200, 256
    sd = random_params(om.mipnerf_param_shapes(hidden=hidden, prop_hidden=256), 21, ("mlp.density_layer.bias", "proposal.density_layer.bias"))
    rays_c = common.synthetic_rays(n, seed=7)
    gg = torch.Generator().manual_seed(8)
    target, tdepth = torch.rand(n, 3, generator=gg), torch.rand(n, generator=gg) * 50 + 5
    pr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    rays = mipnerf.Rays(**{k: v.to(DEV) for k, v in rays_c.items()})

    def loss_fn(ret, tgt, td):
        l = ((ret[1][0] - tgt) ** 2).mean()
        l = l + 0.2 * ((1 / ret[1][1] - 1 / td).abs()).mean() + 0.2 * 0.2 * ((1 / ret[0][1] - 1 / td).abs()).mean()
        return l + 0.01 * (ret[0][4] ** 2).sum() + 0.01 * ret[1][2].mean()
    ref = om.mipnerf_forward(pr, rays_c, S0, P1)
    loss_ref = loss_fn([[None, ref[0][1], ref[0][2], ref[0][3], ref[0][4]], [ref[1][0], ref[1][1], ref[1][2], None, ref[1][4], ref[1][5]]], target, tdepth)
    loss_ref.backward()
    rets, errs = {}, {}
    for compute in ("bf16x3_fwd", "f16f8", "bf16x3", "bf16"):
        m = make_mip(hidden, 256, S0, P1, compute, sd)
        ret = m(rays, False, False, 0.)
        loss = loss_fn(ret, target.to(DEV), tdepth.to(DEV))
        loss.backward()
        rets[compute] = (loss.detach().cpu(), [t.detach().cpu() for t in (ret[1][0], ret[1][1], ret[1][2], ret[0][1], ret[0][4])])
        named = dict(m.named_parameters())
        errs[compute] = {k: ((named[k].grad.detach().cpu() - pr[k].grad).norm() / (pr[k].grad.norm() + 1e-12)).item() for k in sd}
    for a, b in zip(rets["bf16x3_fwd"][1], rets["bf16x3"][1]):
        assert torch.equal(a, b), "the forward of bf16x3_fwd must be the split-bf16 forward, bit for bit"
    assert torch.equal(rets["bf16x3_fwd"][0], rets["bf16x3"][0])
    worst = {c: max(e.values()) for c, e in errs.items()}
    print("MEASURED parameter-gradient rel L2 (max over parameters): " + ", ".join(f"{c} {v:.3e}" for c, v in worst.items()))
    # compute="f16f8": fp16 tiles + e4m3 correction tiles in the forward (two pass-equivalents), scaled fp16 backward -- the same bounds
    for mode in ("bf16x3_fwd", "f16f8"):
        close(rets[mode][0], loss_ref, 3e-4, 3e-4, mode + " loss")
        close(rets[mode][1][0], ref[1][0], 1e-4, 1e-4, mode + " rgb"); close(rets[mode][1][1], ref[1][1], 1e-4, 1e-4, mode + " distance")
        print(f"MEASURED {mode} vs oracle: rgb max abs {float((rets[mode][1][0] - ref[1][0].detach()).abs().max()):.3e}, "
              f"distance max rel {float(((rets[mode][1][1] - ref[1][1].detach()).abs() / ref[1][1].detach().abs()).max()):.3e}")
        for k in sd:
            e = errs[mode][k]
            assert e < max(1.5 * errs["bf16"][k], 4e-3), f"{mode} grad {k}: rel L2 {e:.3e} vs the plain bf16 mode's {errs['bf16'][k]:.3e}"
            assert e < 0.113, f"{mode} grad {k}: rel L2 {e:.3e}"


@pytest.mark.parametrize("compute,tol_out,tol_grad", [("f16f8", 1e-4, 2e-2), ("bf16x3_fwd", 1e-4, 3e-2), ("fp16", 5e-3, 6e-2)])
def test_mipnerf_new_modes_with_every_optional_branch(backend, compute, tol_out, tol_grad):
    """The round-6 compute modes through the branches the plain tests do not take: semantic head (its own trunk-side data gradient), appearance
    embedding (the condition block written from an fp32 image in the split layouts; its table gradient from the scaled backward's returned
    tensor), pose refinement (gradients w.r.t. origins / directions / viewdirs through input_grad -- scaled back by _scaled_backward).  Against
    the exact f32 mode of the same model on the same rays: outputs at `tol_out`, every gradient norm-wise at `tol_grad`."""
    from snerf_amd import mipnerf
    r = common.synthetic_rays(40, seed=3)
    r["app"] = torch.randint(0, 5, (40, 1), generator=torch.Generator().manual_seed(1)).float()
    res = {}
    for mode in ("f32", compute):
        torch.manual_seed(0)
        m = mipnerf.MipNerfModel(n_samples=16, N_fine=17, no_warp_sample=0, ray_shape="cone", fn=1, radius=3., transform_idx=0, real=True, rgb_layer=3,
                                 hidden_layer=256, density_noise=0., max_deg_point=16, proposal_hidden_layer=128, proposal_loss=True, semantic=True,
                                 semantic_class_num=7, encode_appearance=True, N_vocab=5, compute=mode, device=DEV)
        if "sd" in res:
            m.load_state_dict(res["sd"])
        else:
            res["sd"] = {k: v.detach().clone() for k, v in m.state_dict().items()}
        rays = mipnerf.Rays(**{k: v.to(DEV).clone().requires_grad_(k in ("origins", "directions", "viewdirs")) for k, v in r.items()})
        ret = m(rays, False, False, 0.)
        loss = (ret[1][0] ** 2).mean() + 1e-2 * ret[1][1].mean() + (ret[1][3] ** 2).mean() + 1e-2 * ret[0][1].mean()
        loss.backward()
        g = {k: p.grad.detach().cpu() for k, p in m.named_parameters()}
        g.update({"rays." + k: getattr(rays, k).grad.detach().cpu() for k in ("origins", "directions", "viewdirs")})
        res[mode] = ([t.detach().cpu() for t in (ret[1][0], ret[1][1], ret[1][3])], g)
    for a, b_, what in zip(res[compute][0], res["f32"][0], ("rgb", "distance", "semantic")):
        close(a, b_, tol_out, tol_out, f"{compute} {what}")
    worst = 0.0
    for k, gref in res["f32"][1].items():
        e = ((res[compute][1][k] - gref).norm() / (gref.norm() + 1e-12)).item()
        worst = max(worst, e)
        assert e < tol_grad, f"{compute} grad {k}: rel L2 {e:.3e}"
    print(f"MEASURED {compute} with semantic head + appearance embedding + ray gradients: worst gradient rel L2 {worst:.3e}")


def test_mipnerf_semantic_head_vs_reference_golden(backend, golden):
    """MipNerfModel(semantic=True): outputs and EVERY parameter gradient against the reference model's own (g14)."""
    g = golden("g14_mipnerf_semantic")
    from snerf_amd import mipnerf
    m = mipnerf.MipNerfModel(n_samples=16, N_fine=17, no_warp_sample=0, ray_shape="cone", fn=1, radius=3., transform_idx=0, real=True,
                             rgb_layer=3, hidden_layer=64, density_noise=0., max_deg_point=16, proposal_hidden_layer=64,
                             proposal_loss=True, semantic=True, semantic_class_num=7, compute="f32", device=DEV)
    names = [str(k) for k in g["param_names"]]
    assert list(m.state_dict().keys()) == names
    sd = common.fill_state_dict_({k: torch.empty(v.shape) for k, v in m.state_dict().items()})
    m.load_state_dict(sd)
    rays = mipnerf.Rays(**{k[5:]: torch.as_tensor(v).to(DEV) for k, v in g.items() if k.startswith("rays_")})
    ret = m(rays, False, False, 0.)
    sem = ret[1][3]
    assert sem is not None and tuple(sem.shape) == (24, 7)
    loss = ((ret[1][0] - g["target"].to(DEV)) ** 2).mean() + 0.01 * ret[0][1].mean() + 0.2 * (sem * g["semw"].to(DEV)).sum() / 24
    loss.backward()
    close(ret[1][0], g["l1_rgb"], 1e-4, 1e-5, "rgb"); close(ret[1][1], g["l1_distance"], 1e-4, 1e-4, "distance")
    close(sem, g["l1_semantic"], 1e-4, 1e-5, "semantic"); close(loss, g["loss"], 1e-4, 1e-6, "loss")
    named = dict(m.named_parameters())
    for k in names:
        got, ref = named[k].grad.detach().cpu(), g["grad." + k]
        rel = float((got - ref).norm() / (ref.norm() + 1e-20))
        assert rel < 5e-3, (k, rel)


@pytest.mark.parametrize("compute", ["f32", "bf16"])
def test_mipnerf_appearance_embedding_vs_reference_golden(backend, golden, compute):
    """MipNerfModel(encode_appearance=True) (models.py:57,63-64,153-159; arg_parser.py:222): the per-image embedding row rays.app selects
    is appended to the view condition -- state_dict keys incl. `emb.weight` in the reference's order, outputs and EVERY parameter
    gradient (the embedding table's included) against the reference model's own (g22); bf16: outputs / loss to bf16 tolerance."""
    g = golden("g22_mipnerf_appearance")
    from snerf_amd import mipnerf
    V = int(g["grad.emb.weight"].shape[0])
    m = mipnerf.MipNerfModel(n_samples=16, N_fine=17, no_warp_sample=0, ray_shape="cone", fn=1, radius=3., transform_idx=0, real=True,
                             rgb_layer=3, hidden_layer=64, density_noise=0., max_deg_point=16, proposal_hidden_layer=64,
                             proposal_loss=True, encode_appearance=True, N_vocab=V, compute=compute, device=DEV)
    names = [str(k) for k in g["param_names"]]
    assert list(m.state_dict().keys()) == names and names.index("emb.weight") == names.index("proposal.layers.0.layers.0.weight") - 1
    assert names == [k for k, _ in om.mipnerf_param_shapes(hidden=64, prop_hidden=64, cond_dim=75, n_vocab=V)]
    sd = common.fill_state_dict_({k: torch.empty(tuple(g["grad." + k].shape)) for k in names})
    m.load_state_dict(sd)
    rays = mipnerf.Rays(**{k[len("rays_"):]: v.to(DEV) for k, v in g.items() if k.startswith("rays_")})
    ret = m(rays, False, False, 0.)
    loss = ((ret[1][0] - g["target"].to(DEV)) ** 2).mean() + 0.01 * ret[0][1].mean() + 0.05 * (1.0 / ret[1][1]).mean()
    loss.backward()
    tol = 1e-4 if compute == "f32" else 3e-2
    close(ret[1][0], g["l1_rgb"], tol, tol * 0.1, "rgb"); close(ret[1][1], g["l1_distance"], tol, tol, "distance")
    close(ret[1][2], g["l1_acc"], tol, tol * 0.1, "acc"); close(loss, g["loss"], tol, tol * 1e-2, "loss")
    # the oracle restates the same branch
    ref = om.mipnerf_forward(sd, {k[len("rays_"):]: v for k, v in g.items() if k.startswith("rays_")}, 16, 17)
    close(ref[1][0], g["l1_rgb"], 1e-5, 1e-6, "oracle rgb"); close(ref[1][1], g["l1_distance"], 1e-5, 1e-5, "oracle distance")
    named = dict(m.named_parameters())
    for k in names:
        got, want = named[k].grad.detach().cpu(), g["grad." + k]
        rel = float((got - want).norm() / (want.norm() + 1e-20))
        # (bf16: the golden's formula weights are rank 2 -- every back-propagated signal is a near-cancelling sum that amplifies bf16
        # rounding to O(1) in the early layers, see test_mipnerf_backward_vs_autograd; the embedding's own gradient sits right below the
        # colour head and is held to the bf16 bound)
        if compute == "f32" or k == "emb.weight":
            assert rel < (5e-3 if compute == "f32" else 0.3), (k, rel)
    assert float(g["grad.emb.weight"].abs().sum()) > 0


@pytest.mark.parametrize("compute", ["f32", "bf16"])
def test_mipnerf_view_centred_warp_vs_reference_golden(backend, golden, compute):
    """MipNerfModel(fn=0): the view-centred warp (mip.py:367-378: fn1 + Jacobi_f, viewc = mean camera centre) instead of the
    contraction -- outputs and every parameter gradient against the reference model's own (g23)."""
    g = golden("g23_warp0")
    from snerf_amd import mipnerf
    m = mipnerf.MipNerfModel(n_samples=16, N_fine=17, no_warp_sample=0, ray_shape="cone", fn=0, radius=3., transform_idx=0, real=True,
                             rgb_layer=3, hidden_layer=64, density_noise=0., max_deg_point=16, proposal_hidden_layer=64,
                             proposal_loss=True, compute=compute, device=DEV)
    names = [str(k) for k in g["param_names"]]
    assert list(m.state_dict().keys()) == names
    sd = common.fill_state_dict_({k: torch.empty(tuple(g["grad." + k].shape)) for k in names})
    m.load_state_dict(sd)
    rays = mipnerf.Rays(**{k[len("rays_"):]: v.to(DEV) for k, v in g.items() if k.startswith("rays_")})
    ret = m(rays, False, False, g["viewc"])
    loss = ((ret[1][0] - g["target"].to(DEV)) ** 2).mean() + 0.01 * ret[0][1].mean() + 0.05 * (1.0 / ret[1][1]).mean()
    loss.backward()
    tol = 1e-4 if compute == "f32" else 3e-2
    close(ret[1][0], g["l1_rgb"], tol, tol * 0.1, "rgb"); close(ret[1][1], g["l1_distance"], tol, tol, "distance")
    close(ret[1][2], g["l1_acc"], tol, tol * 0.1, "acc"); close(loss, g["loss"], tol, tol * 1e-2, "loss")
    if compute == "f32":
        close(ret[1][4], g["l1_s_vals"], 1e-4, 1e-5, "fine fence posts")
        named = dict(m.named_parameters())
        for k in names:
            got, want = named[k].grad.detach().cpu(), g["grad." + k]
            rel = float((got - want).norm() / (want.norm() + 1e-20))
            assert rel < 5e-3, (k, rel)


def test_mipnerf_pose_refinement_through_the_view_centred_warp(backend, golden):
    """pose refinement of an fn = 0 model: d loss / d (origins, directions, viewdirs) through fn1 (mip.py:368-369) and Jacobi_f
    (:323-340) against the reference model's own autograd (g30; snerf_mip_encode_warp_bwd)."""
    from snerf_amd import mipnerf
    g = golden("g30_pose_fn0")
    S0, P1, hidden = int(g["S0"]), int(g["P1"]), int(g["hidden"])
    m = mipnerf.MipNerfModel(n_samples=S0, N_fine=P1, no_warp_sample=0, ray_shape="cone", fn=0, radius=3., transform_idx=0, real=True,
                             rgb_layer=3, hidden_layer=hidden, density_noise=0., max_deg_point=16, proposal_hidden_layer=64,
                             proposal_loss=True, compute="f32", device=DEV)
    m.load_state_dict(common.fill_state_dict_({k: torch.empty_like(v) for k, v in m.state_dict().items()}))
    rays = {k: g[k].to(DEV) for k in mipnerf.Rays._fields}
    for k in ("origins", "directions", "viewdirs"):
        rays[k] = rays[k].clone().requires_grad_(True)
    ret = m(mipnerf.Rays(**rays), False, False, g["viewc"])
    close(ret[1][0], g["rgb"], 1e-4, 1e-5, "rgb"); close(ret[1][1], g["dist1"], 1e-4, 1e-4, "distance")
    loss = (ret[1][0] * g["w_rgb"].to(DEV)).sum() + 0.05 * (ret[1][1] * g["w_d1"].to(DEV)).sum() + 0.05 * (ret[0][1] * g["w_d0"].to(DEV)).sum()
    loss.backward()
    for k in ("origins", "directions", "viewdirs"):
        ref_g = g["g_" + k]
        err = float((rays[k].grad.cpu() - ref_g).abs().max())
        assert float(ref_g.abs().max()) > 0 and err <= 3e-3 * float(ref_g.abs().max()), (k, err, float(ref_g.abs().max()))


def test_mipnerf_disable_integration(backend, golden):
    """MipNerfModel(disable_integration=1) -- the encoders zero the covariances (bit 1 of the kernels' `cone` argument), forward and
    pose-refinement backward -- against the reference model's own outputs, parameter gradients and ray gradients (g31).
    Without the exp(-var / 2) damping the 2^15 x features reach the network at full amplitude: one ulp of a contracted mean moves their phase
    by 2e-3 rad, and the REFERENCE's fp32 evaluation itself sits 1.5e-3 (distance) from the float64 evaluation of the same formulas.  So the
    yardstick is the float64 oracle: every output / gradient must be as close to it as the reference's own fp32 result is, within a factor
    of 5 (measured: 3.7 at most on the GPU -- the fine fence posts --, 1.0 emulated).  The encoder alone is held to 2e-6 on its own means by
    test_mip_encode_without_integration_fwd_and_bwd; the branch itself differs from the default path by 1e-3 .. 1e-1."""
    from snerf_amd import mipnerf
    g = golden("g31_no_integration")
    S0, P1, hidden = int(g["S0"]), int(g["P1"]), int(g["hidden"])
    m = mipnerf.MipNerfModel(n_samples=S0, N_fine=P1, no_warp_sample=0, disable_integration=1, ray_shape="cone", fn=1, radius=3., transform_idx=0, real=True,
                             rgb_layer=3, hidden_layer=hidden, density_noise=0., max_deg_point=16, proposal_hidden_layer=64,
                             proposal_loss=True, compute="f32", device=DEV)
    names = [str(k) for k in g["param_names"]]
    assert list(m.state_dict().keys()) == names
    sd = common.fill_state_dict_({k: torch.empty(tuple(g["grad." + k].shape)) for k in names})
    m.load_state_dict(sd)
    rays = {k: g[k].to(DEV) for k in mipnerf.Rays._fields}
    for k in ("origins", "directions", "viewdirs"):
        rays[k] = rays[k].clone().requires_grad_(True)
    ret = m(mipnerf.Rays(**rays), False, False, 0.)
    loss = (ret[1][0] * g["w_rgb"].to(DEV)).sum() + 0.05 * (ret[1][1] * g["w_d1"].to(DEV)).sum() + 0.05 * (ret[0][1] * g["w_d0"].to(DEV)).sum()
    loss.backward()
    # the float64 evaluation of the same formulas (oracle/mip.py)
    p64 = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    r64 = {k: g[k].double() for k in mipnerf.Rays._fields}
    for k in ("origins", "directions", "viewdirs"):
        r64[k].requires_grad_(True)
    ref = om.mipnerf_forward(p64, r64, S0, P1, disable_integration=True)
    ((ref[1][0] * g["w_rgb"].double()).sum() + 0.05 * (ref[1][1] * g["w_d1"].double()).sum() + 0.05 * (ref[0][1] * g["w_d0"].double()).sum()).backward()
    rel = lambda a, b: float(((a.detach().double().cpu() - b.detach()).abs() / b.detach().abs().clamp(min=1e-3)).max())
    for what, got, want64, gold in (("rgb", ret[1][0], ref[1][0], g["rgb"]), ("distance", ret[1][1], ref[1][1], g["dist1"]), ("acc", ret[1][2], ref[1][2], g["acc1"]),
                                    ("proposal distance", ret[0][1], ref[0][1], g["dist0"]), ("fine fence posts", ret[1][4], ref[1][4], g["s1"])):
        noise = rel(gold, want64)
        assert rel(got, want64) <= 5 * noise + 2e-5, (what, rel(got, want64), noise)
    plain = om.mipnerf_forward(sd, {k: g[k] for k in mipnerf.Rays._fields}, S0, P1)      # the default path: 3 % / 2 % / 4 % away in distance / proposal distance / fence posts
    assert rel(plain[1][1], ref[1][1]) > 3 * (5 * rel(g["dist1"], ref[1][1]) + 2e-5) and rel(plain[1][4], ref[1][4]) > 1e-2
    l2 = lambda a, b: float((a.detach().double().cpu() - b.detach()).norm() / (b.detach().norm() + 1e-30))
    named = dict(m.named_parameters())
    for k in names:
        noise = l2(g["grad." + k], p64[k].grad)
        assert l2(named[k].grad, p64[k].grad) <= 5 * noise + 2e-3, (k, l2(named[k].grad, p64[k].grad), noise)
    for k in ("origins", "directions", "viewdirs"):
        noise = l2(g["g_" + k], r64[k].grad)
        assert float(g["g_" + k].abs().max()) > 0 and l2(rays[k].grad, r64[k].grad) <= 5 * noise + 2e-3, (k, l2(rays[k].grad, r64[k].grad), noise)


def test_mip_trainer_passes_the_view_centre_of_the_fn0_warp(backend, golden):
    """MipTrainer on an fn = 0 model: the trainer calls model._run directly, so the warp centre the reference hands to every
    forward (train.py:36,112) has to come through step(viewc=...) / capture(viewc=...).  Without one the step refuses to run instead of
    warping around (0, 0, 0); with the golden's centre the trainer's forward reproduces the reference model's outputs (g23)."""
    g = golden("g23_warp0")
    from snerf_amd import mipnerf
    from snerf_amd.trainer import MipTrainer
    m = mipnerf.MipNerfModel(n_samples=16, N_fine=17, no_warp_sample=0, ray_shape="cone", fn=0, radius=3., transform_idx=0, real=True,
                             rgb_layer=3, hidden_layer=64, density_noise=0., max_deg_point=16, proposal_hidden_layer=64,
                             proposal_loss=True, compute="f32", device=DEV)
    names = [str(k) for k in g["param_names"]]
    m.load_state_dict(common.fill_state_dict_({k: torch.empty(tuple(g["grad." + k].shape)) for k in names}))
    rays = mipnerf.Rays(**{k[len("rays_"):]: v.to(DEV) for k, v in g.items() if k.startswith("rays_")})
    tr = MipTrainer(m, lr=0.0)
    tgt = g["target"].to(DEV)
    with pytest.raises(ValueError, match="view centre"):
        tr.step(rays, tgt, randomized=False)
    _, outs = tr.step(rays, tgt, randomized=False, viewc=g["viewc"])
    close(outs[4], g["l1_rgb"], 1e-4, 1e-5, "rgb through the trainer")
    close(outs[5], g["l1_distance"], 1e-4, 1e-4, "distance through the trainer")
    _, outs2 = tr.step(rays, tgt, randomized=False)              # the centre stays set
    assert torch.equal(outs2[4], outs[4])


def test_mipnerf_ray_gradients_vs_autograd(backend):
    """Pose refinement (configs/nuScenes_depth_6cams: pose_refine = True; utils/sample_utils.py:410-435): origins, directions and
    viewdirs are functions of a learnable camera pose.  d loss / d rays through both levels (encoders, contraction + Jacobian, lifted
    Gaussians, interval lengths, view encoding) against torch autograd through the oracle, fp32 kernels; the parameter gradients of the
    same backward are unchanged; a 6-dof pose in front of the rays receives the chained gradient."""
    from snerf_amd import mipnerf
    S0, P1, n, hidden = 24, 25, 36, 64
    sd = mip_params(hidden, 64)
    rays_c = common.synthetic_rays(n, seed=11)
    rays_c["near"], rays_c["far"] = torch.full_like(rays_c["near"], 0.5), torch.full_like(rays_c["far"], 30.0)   # samples on both sides of |x| = 3
    gg = torch.Generator().manual_seed(12)
    target, tdepth = torch.rand(n, 3, generator=gg), torch.rand(n, generator=gg) * 20 + 2

    def loss_fn(ret, tgt, td):
        l = ((ret[1][0] - tgt) ** 2).mean()
        l = l + 0.2 * ((1 / ret[1][1] - 1 / td).abs()).mean() + 0.04 * ((1 / ret[0][1] - 1 / td).abs()).mean()
        return l + 0.01 * (ret[0][4] ** 2).sum() + 0.01 * ret[1][2].mean()

    def posed(rays, rot, trans):         # sample_rays: directions / viewdirs rotated, origins shifted by the refined pose
        R = torch.eye(3, device=rot.device) + torch.stack([torch.stack([rot[0] * 0, -rot[2], rot[1]]), torch.stack([rot[2], rot[0] * 0, -rot[0]]),
                                                           torch.stack([-rot[1], rot[0], rot[0] * 0])])
        out = dict(rays)
        out["directions"] = (rays["directions"][:, None, :] * R).sum(-1)
        out["viewdirs"] = (rays["viewdirs"][:, None, :] * R).sum(-1)
        out["origins"] = rays["origins"] + trans
        return out
    # oracle
    pr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    rot_r, tr_r = torch.tensor([0.01, -0.02, 0.015], requires_grad=True), torch.tensor([0.05, -0.03, 0.02], requires_grad=True)
    rr = posed(rays_c, rot_r, tr_r)
    for k in ("origins", "directions", "viewdirs"):
        rr[k].retain_grad()
    ref = om.mipnerf_forward(pr, rr, S0, P1)
    loss_ref = loss_fn([[None, ref[0][1], ref[0][2], ref[0][3], ref[0][4]], [ref[1][0], ref[1][1], ref[1][2], None, ref[1][4], ref[1][5]]], target, tdepth)
    loss_ref.backward()
    # build
    m = make_mip(hidden, 64, S0, P1, "f32", sd)
    rot, tr = torch.tensor([0.01, -0.02, 0.015], device=DEV, requires_grad=True), torch.tensor([0.05, -0.03, 0.02], device=DEV, requires_grad=True)
    rb = posed({k: v.to(DEV) for k, v in rays_c.items()}, rot, tr)
    for k in ("origins", "directions", "viewdirs"):
        rb[k].retain_grad()
    ret = m(mipnerf.Rays(**rb), False, False, 0.)
    loss = loss_fn(ret, target.to(DEV), tdepth.to(DEV))
    loss.backward()
    close(loss, loss_ref, 3e-4, 3e-4, "loss")
    for k in ("origins", "directions", "viewdirs"):
        g, gr = rb[k].grad.cpu(), rr[k].grad
        assert float(gr.abs().max()) > 0
        err = (g - gr).abs().max().item()
        assert err <= 2e-3 * gr.abs().max().item(), (k, err, gr.abs().max().item())
    for g, gr, what in ((rot.grad.cpu(), rot_r.grad, "rotation"), (tr.grad.cpu(), tr_r.grad, "translation")):
        assert (g - gr).abs().max().item() <= 2e-3 * gr.abs().max().item(), (what, g, gr)
    check_grads(dict(m.named_parameters()), pr, sd, "f32", 3e-4)
    with pytest.raises(NotImplementedError):
        bad = dict(rb); bad["radii"] = rb["radii"].detach().clone().requires_grad_(True)
        m(mipnerf.Rays(**{k: (v.detach() if k != "radii" else v) for k, v in bad.items()}), False, False, 0.)


def test_mipnerf_ray_gradients_vs_reference_golden(backend, golden):
    """The same backward against the reference's own autograd (G19: reference MipNerfModel, rays requiring grad)."""
    from snerf_amd import mipnerf
    g = golden("g19_mipnerf_raygrad")
    sd = mip_params(64, 64)
    rays = {k[5:]: v.to(DEV) for k, v in g.items() if k.startswith("rays_")}
    for k in ("origins", "directions", "viewdirs"):
        rays[k] = rays[k].clone().requires_grad_(True)
    m = make_mip(64, 64, 16, 17, "f32", sd)
    ret = m(mipnerf.Rays(**rays), False, False, 0.)
    target, td = g["target"].to(DEV), g["target_depth"].to(DEV)
    loss = (((ret[1][0] - target) ** 2).mean() + 0.2 * ((1 / ret[1][1] - 1 / td).abs()).mean() + 0.04 * ((1 / ret[0][1] - 1 / td).abs()).mean()
            + 0.01 * (ret[0][4] ** 2).sum() + 0.01 * ret[1][2].mean())
    loss.backward()
    close(loss, g["loss"], 1e-4, 1e-5, "loss")
    close(ret[1][0], g["l1_rgb"], 1e-4, 1e-5, "rgb")
    for k in ("origins", "directions", "viewdirs"):
        ref_g = g["grad_" + k]
        err = float((rays[k].grad.cpu() - ref_g).abs().max())
        assert err <= 3e-3 * float(ref_g.abs().max()), (k, err, float(ref_g.abs().max()))


def test_render_image_vs_reference_golden(backend, golden):
    """chunked full-frame inference, ragged last chunk (35 rays in chunks of 8), against the reference's own
    render_image (models.py:328-360; g21) -- and against ONE forward over all rays (chunking must not change a value)."""
    from snerf_amd import mipnerf
    g = golden("g21_render_image")
    H, W, chunk = int(g["H"]), int(g["W"]), int(g["chunk"])
    m = mipnerf.MipNerfModel(n_samples=16, N_fine=17, no_warp_sample=0, ray_shape="cone", fn=1, radius=3., transform_idx=0, real=True,
                             rgb_layer=3, hidden_layer=64, density_noise=0., max_deg_point=16, proposal_hidden_layer=64,
                             proposal_loss=False, semantic=True, semantic_class_num=7, compute="f32", device=DEV)
    assert list(m.state_dict().keys()) == [str(k) for k in g["param_names"]]
    m.load_state_dict(common.fill_state_dict_({k: torch.empty(v.shape) for k, v in m.state_dict().items()}))
    flat = {k[5:]: torch.as_tensor(v).to(DEV) for k, v in g.items() if k.startswith("rays_")}
    rays = mipnerf.Rays(**{k: v.reshape(H, W, -1) for k, v in flat.items()})
    with torch.no_grad():
        rgb, dist, acc, sem = mipnerf.render_image(lambda r: m(r, False, False, 0.), rays, 0, chunk=chunk)
        one = m(mipnerf.Rays(**flat), False, False, 0.)[-1]
    assert rgb.shape == (H, W, 3) and dist.shape == (H, W) and acc.shape == (H, W) and sem.shape == (H, W, 7)
    close(rgb, g["rgb"], 1e-4, 1e-5, "render_image rgb"); close(dist, g["distance"], 1e-4, 1e-4, "render_image distance")
    close(acc, g["acc"], 1e-4, 1e-5, "render_image acc"); close(sem, g["semantic"], 1e-4, 1e-5, "render_image semantic")
    for got, want, k in zip((rgb, dist, acc, sem), one[:4], ("rgb", "distance", "acc", "semantic")):
        if backend == "hip":      # the HIP GEMMs reduce every row in a fixed order whatever the batch size; CPU BLAS (emulation) does not
            assert torch.equal(got.reshape(want.shape), want), f"chunking changed {k}"
        else:
            close(got.reshape(want.shape), want, 1e-5, 1e-6, f"chunked vs one forward: {k}")
    # without a semantic head (and with the proposal-loss extras in the level outputs) the fourth output is None
    m2 = make_mip(64, 64, 16, 17, "f32", mip_params(64, 64))
    with torch.no_grad():
        out = mipnerf.render_image(lambda r: m2(r, False, False, 0.), rays, 0, chunk=13)
        ref = om.mipnerf_forward(mip_params(64, 64), {k: v.cpu() for k, v in flat.items()}, 16, 17)
    assert out[3] is None and out[0].shape == (H, W, 3)
    close(out[0].reshape(-1, 3), ref[1][0], 1e-4, 1e-5, "render_image rgb vs oracle")
    close(out[1].reshape(-1), ref[1][1], 1e-4, 1e-4, "render_image distance vs oracle")
