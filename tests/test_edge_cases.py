"""Edge cases of the operator API, with the reference's behaviour as the bar (probed in the build container by importing the reference:
`MipNerfModel.forward` and `render_rays` both RAISE on an empty ray batch -- `reshape([0, -1])` is ambiguous, `torch.cat` of no chunks --
and both accept a single ray):
  * one ray, and ray counts that are multiples of nothing (1, 3, 33 rays x 16 / 17 samples: every kernel's tail path), against the oracle;
  * zero rays: an exception, as in the reference, not a crash of the process and not a silent empty result;
  * the same for path C where the domain allows (one ray)."""
import pytest
import torch

from oracle import classic as oc
from oracle import common, mip as om

import test_paths as tp
from test_paths import backend, close   # noqa: F401  (fixture)


@pytest.mark.parametrize("compute", ["f32", "f16f8", "bf16x3_fwd"])
@pytest.mark.parametrize("n", [1, 3, 33])
def test_mipnerf_tiny_ray_batches_vs_oracle(backend, n, compute):
    """(the 1e-4 modes alike: the split layouts' k-tiles, the e4m3 tiles and the scaled / plain backward on 16- to 561-row operands)"""
    from snerf_amd import mipnerf
    sd = tp.mip_params(64, 64)
    rays_c = common.synthetic_rays(n, seed=9)
    ref = om.mipnerf_forward(sd, rays_c, 16, 17)
    m = tp.make_mip(64, 64, 16, 17, compute, sd)
    rays = mipnerf.Rays(**{k: v.to(tp.DEV) for k, v in rays_c.items()})
    with torch.no_grad():
        ret = m(rays, False, False, 0.)
    assert torch.equal(ret[0][3].cpu(), ref[0][3]), "level-0 fence posts must be bit-exact"
    close(ret[0][4], ref[0][4], 1e-4, 1e-6, "w0")
    close(ret[1][0], ref[1][0], 1e-4, 1e-4, "rgb"); close(ret[1][1], ref[1][1], 1e-4, 1e-4, "distance"); close(ret[1][2], ref[1][2], 1e-4, 1e-5, "acc")
    # and the training direction runs on the same tails (finite gradients on every parameter)
    ret = m(rays, False, False, 0.)
    (ret[1][0].sum() + ret[1][1].sum() + ret[0][1].sum()).backward()
    for k, p in m.named_parameters():
        assert p.grad is not None and bool(torch.isfinite(p.grad).all()), k


@pytest.mark.parametrize("n", [1, 3, 33])
def test_classic_tiny_ray_batches_vs_oracle(backend, n):
    from snerf_amd import classic
    pc = tp.random_params(oc.nerf_param_shapes(W=64), 11, ("alpha_linear.bias",))
    pf = tp.random_params(oc.nerf_param_shapes(W=64), 12, ("alpha_linear.bias",))
    coarse, fine = tp.make_nerf(64, "f32", pc), tp.make_nerf(64, "f32", pf)
    embed_fn, _ = classic.get_embedder(10, 0); embeddirs_fn, _ = classic.get_embedder(4, 0)
    nq = classic.make_network_query_fn(embed_fn, embeddirs_fn)
    gg = torch.Generator().manual_seed(n)
    ro = torch.randn(n, 3, generator=gg) * 0.2
    rd = torch.nn.functional.normalize(torch.randn(n, 3, generator=gg), dim=-1)
    rb = torch.cat([ro, rd, torch.full((n, 1), 2.0), torch.full((n, 1), 6.0), rd], -1)
    ref = oc.render_rays(rb, pc, None, 16, 0, retraw=True)
    with torch.no_grad():
        out = classic.render_rays(rb.to(tp.DEV), coarse, nq, N_samples=16, retraw=True, perturb=0., N_importance=0)
        both = classic.render_rays(rb.to(tp.DEV), coarse, nq, N_samples=16, perturb=0., N_importance=8, network_fine=fine)
    assert torch.equal(out["z_vals_map"].cpu(), ref["z_vals_map"])
    close(out["raw"], ref["raw"], 1e-4, 1e-4, "raw")
    ok = ref["raw"][:, -1, 3].abs() > 1e-3          # (the last interval's alpha jumps 0 -> 1 at sigma = 0: see test_paths)
    for k in ("rgb_map", "acc_map", "weights"):
        close(out[k][ok], ref[k][ok], 1e-4, 1e-4, k)
    assert both["rgb_map"].shape == (n, 3) and both["z_vals_map"].shape == (n, 16) and both["rgb0"].shape == (n, 3) and bool(torch.isfinite(both["rgb_map"]).all())


def test_zero_rays_raise_like_the_reference(backend):
    """The reference raises on an empty batch (RuntimeError from an ambiguous reshape in MipNerfModel.forward, ValueError from torch.cat
    of no chunks in run_network); the drop-in raises too -- before any kernel is launched on a zero-sized grid."""
    from snerf_amd import classic, mipnerf
    sd = tp.mip_params(64, 64)
    m = tp.make_mip(64, 64, 16, 17, "f32", sd)
    rays_c = common.synthetic_rays(1, seed=9)
    rays = mipnerf.Rays(**{k: v[:0].to(tp.DEV) for k, v in rays_c.items()})
    with pytest.raises((RuntimeError, ValueError)):
        with torch.no_grad():
            m(rays, False, False, 0.)
    coarse = tp.make_nerf(64, "f32", tp.nerf_params(64))
    embed_fn, _ = classic.get_embedder(10, 0); embeddirs_fn, _ = classic.get_embedder(4, 0)
    nq = classic.make_network_query_fn(embed_fn, embeddirs_fn)
    with pytest.raises((RuntimeError, ValueError)):
        with torch.no_grad():
            classic.render_rays(torch.zeros(0, 11, device=tp.DEV), coarse, nq, N_samples=16, perturb=0., N_importance=0)
    # the model still works afterwards (no sticky error state in the library)
    rays1 = mipnerf.Rays(**{k: v.to(tp.DEV) for k, v in rays_c.items()})
    with torch.no_grad():
        ret = m(rays1, False, False, 0.)
    assert bool(torch.isfinite(ret[1][0]).all())


@pytest.mark.parametrize("n", [1, 3])
def test_zip_tiny_ray_batches_match_rows_of_the_full_batch(backend, golden, n):
    """Path C: rays are independent in the deterministic forward, so the first n rays evaluated alone must reproduce rows [:n] of the
    20-ray golden batch (which test_zip_paths pins against the reference) -- the tail paths of every kernel at R = 1 and R = 3."""
    import test_zip_paths as tz
    tz.DEV = tp.DEV
    g = golden("g11_zip_model")
    _, p = tz.zip_setup()
    batch = {k[2:]: v.to(tp.DEV) for k, v in g.items() if k.startswith("b_")}
    m = tz.make_model("f32", "f32", p)
    R = batch["origins"].shape[0]
    small = {k: (v[:n].contiguous() if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == R else v) for k, v in batch.items()}
    with torch.no_grad():
        rend, hist = m(None, small, 1.0, False)
    close(rend[-1]["rgb"], g["det_rgb"][:n], 2e-4, 2e-4, "rgb"); close(rend[-1]["depth"], g["det_depth"][:n], 2e-4, 2e-3, "depth")
    for lvl in range(3):
        close(hist[lvl]["sdist"], g[f"det_sdist{lvl}"][:n], 2e-4, 2e-4, f"sdist {lvl}")
    tz.DEV = "cuda"


@pytest.mark.parametrize("seed", range(6))
def test_random_shapes_and_weights_vs_oracle(backend, seed):
    """Seeded fuzz of the two torch-facing paths at random ray counts, sample counts and network weights (fp32 mode; the fixed cases of
    test_paths use round numbers): MipNerfModel.forward (deterministic and with the draws passed in) and render_rays vs the oracle."""
    from snerf_amd import classic, mipnerf
    g = torch.Generator().manual_seed(777 + seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    # ---- path A
    n, S0, P1 = ri(1, 150), ri(4, 40), ri(5, 41)
    shapes = om.mipnerf_param_shapes(hidden=64, prop_hidden=64)
    sd = tp.random_params(shapes, 100 + seed, ("mlp.density_layer.bias", "proposal.density_layer.bias"))
    rays_c = common.synthetic_rays(n, seed=300 + seed)
    s_rand = torch.rand(n, S0 + 1, generator=g)
    u = om.rand_u(P1, torch.empty(n, P1).uniform_(0, 1 / P1 - om.EPS32, generator=g))
    m = tp.make_mip(64, 64, S0, P1, "f32", sd)
    rays = mipnerf.Rays(**{k: v.to(tp.DEV) for k, v in rays_c.items()})
    for kw_o, kw_m, rnd in ((dict(), dict(), False), (dict(s_rand=s_rand, u=u), dict(s_rand=s_rand.to(tp.DEV), u=u.to(tp.DEV)), True)):
        ref = om.mipnerf_forward(sd, rays_c, S0, P1, **kw_o)
        with torch.no_grad():
            ret = m(rays, rnd, False, 0., **kw_m)
        what = f"seed {seed} n {n} S0 {S0} P1 {P1} rand {rnd}: "
        assert torch.equal(ret[0][3].cpu(), ref[0][3]), what + "level-0 fence posts must be bit-exact"
        close(ret[0][4], ref[0][4], 1e-4, 1e-6, what + "w0"); close(ret[1][4], ref[1][4], 1e-4, 1e-5, what + "s1")
        close(ret[1][0], ref[1][0], 1e-4, 1e-4, what + "rgb"); close(ret[1][1], ref[1][1], 1e-4, 1e-4, what + "distance")
        close(ret[1][2], ref[1][2], 1e-4, 1e-5, what + "acc")
    # ---- path B (coarse only: the hierarchical pass is conditioned on u == cdf ties, see test_paths)
    n, S = ri(1, 120), ri(2, 70)
    pc = tp.random_params(oc.nerf_param_shapes(W=64), 200 + seed, ("alpha_linear.bias",))
    coarse = tp.make_nerf(64, "f32", pc)
    embed_fn, _ = classic.get_embedder(10, 0); embeddirs_fn, _ = classic.get_embedder(4, 0)
    nq = classic.make_network_query_fn(embed_fn, embeddirs_fn)
    ro = torch.randn(n, 3, generator=g) * 0.2
    rd = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1) * (1 + 0.2 * torch.rand(n, 1, generator=g))
    rb = torch.cat([ro, rd, torch.full((n, 1), 2.0), torch.full((n, 1), 6.0), torch.nn.functional.normalize(rd, dim=-1)], -1)
    t_rand = torch.rand(n, S, generator=g)
    lindisp = bool(seed % 2)
    ref = oc.render_rays(rb, pc, None, S, 0, lindisp=lindisp, t_rand=t_rand, retraw=True, white_bkgd=bool(seed % 3 == 0))
    with torch.no_grad():
        out = classic.render_rays(rb.to(tp.DEV), coarse, nq, N_samples=S, retraw=True, lindisp=lindisp, perturb=1., N_importance=0,
                                  white_bkgd=bool(seed % 3 == 0), t_rand=t_rand.to(tp.DEV))
    what = f"seed {seed} n {n} S {S} lindisp {lindisp}: "
    assert torch.equal(out["z_vals_map"].cpu(), ref["z_vals_map"]), what + "stratified z must be bit-exact"
    close(out["raw"], ref["raw"], 1e-4, 1e-4, what + "raw")
    ok = ref["raw"][:, -1, 3].abs() > 1e-3
    for k in ("rgb_map", "acc_map", "weights", "depth_map"):
        close(out[k][ok], ref[k][ok], 1e-4, 1e-4, what + k)
