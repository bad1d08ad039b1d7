"""Rows 8f-1 / 8f-2: ray generation and the per-ray loss tail through the C-ABI, against the vectors captured from the
reference (tests/golden/g12_rays.npz, g13_losses.npz) and against oracle/callers.py on larger seeded inputs."""
import types

import numpy as np
import pytest
import torch

from oracle import callers as oc

gpu = pytest.mark.gpu


def _t(g, k, dev="cpu"):
    return torch.from_numpy(np.asarray(g[k])).to(dev)


@gpu
def test_pinhole_rays_vs_reference_golden(golden):
    from snerf_amd import ops, sample_utils as su
    g = golden("g12_rays")
    H, W = int(g["H"]), int(g["W"])
    K = g["intrinsic"]
    # training batch (sample_single_img): same selected pixels
    sel = _t(g, "sel_coords")
    r = su.rays_of_pixels(sel, g["pose"], K, H, W, float(g["sel_near"][0, 0]), float(g["sel_far"][0, 0]), training=True)
    # whole frame (get_rays_single_img, factor 1)
    args = types.SimpleNamespace(no_ndc=True)
    full = su.get_rays_single_img(args, torch.zeros(H, W, 3), None, g["pose"], K, near=2.0, far=100.0, factor=1, device="cuda")
    for prefix, rays, n in (("sel_", r, sel.shape[0]), ("full_", full, H * W)):
        for k in ("origins", "directions", "lossmult", "near", "far", "app"):
            got = getattr(rays, k).reshape(n, -1).cpu()
            assert torch.equal(got, _t(g, prefix + k)), (prefix, k, float((got - _t(g, prefix + k)).abs().max()))
        # norms / 3-term sums: the reference's vectorised CPU reductions associate differently in a few percent of the pixels
        for k in ("viewdirs", "radii"):
            got = getattr(rays, k).reshape(n, -1).cpu()
            assert torch.allclose(got, _t(g, prefix + k), rtol=3e-7, atol=0), (prefix, k, float((got - _t(g, prefix + k)).abs().max()))


@gpu
def test_sample_single_img_mirror(golden):
    """Same numpy RNG state -> same pixels, rays, targets as the reference's sample_single_img."""
    from snerf_amd import sample_utils as su
    g = golden("g12_rays")
    args = types.SimpleNamespace(no_ndc=True, smooth_loss=False, N_rgb=300)
    image, depth = _t(g, "image", "cuda"), _t(g, "depth", "cuda")
    np.random.seed(7)
    rays, trgb, tdep, sel, _ = su.sample_single_img(args, image, depth, g["pose"], g["intrinsic"], near=2.0, far=100.0, near_far=False, batch_n=300)
    assert torch.equal(sel.cpu(), _t(g, "sel_coords"))
    assert torch.equal(trgb.cpu(), _t(g, "target_rgb")) and torch.equal(tdep.cpu(), _t(g, "target_depth"))
    assert torch.equal(rays.directions.cpu(), _t(g, "sel_directions")) and torch.equal(rays.near.cpu(), _t(g, "sel_near"))


def _loss_inputs(N, Sc, Pf, seed):
    gen = torch.Generator().manual_seed(seed)

    def fence(P):
        s = torch.sort(torch.rand(N, P, generator=gen), dim=-1).values
        s[:, 0], s[:, -1] = 0.0, 1.0
        return s
    s_c, s_f = fence(Sc + 1), fence(Pf)
    k = min(Pf, Sc + 1)
    s_f[:5, :k] = s_c[:5, :k]
    s_f = torch.sort(s_f, dim=-1).values
    w_c = torch.rand(N, Sc, generator=gen) ** 4
    w_c = w_c / w_c.sum(-1, keepdim=True) * torch.rand(N, 1, generator=gen)
    w_f = torch.rand(N, Pf - 1, generator=gen) ** 6
    w_f = w_f / w_f.sum(-1, keepdim=True) * torch.rand(N, 1, generator=gen)
    w_f[3] = 0.0
    rgb, tgt = torch.rand(N, 3, generator=gen), torch.rand(N, 3, generator=gen)
    d1, d0 = torch.rand(N, generator=gen) * 60 + 2, torch.rand(N, generator=gen) * 60 + 2
    td = torch.rand(N, generator=gen) * 78 + 2
    td[torch.rand(N, generator=gen) < 0.5] = 0
    conf = torch.rand(N, generator=gen)
    return dict(s_c=s_c, s_f=s_f, w_c=w_c, w_f=w_f, rgb=rgb, tgt=tgt, d1=d1, d0=d0, td=td, conf=conf)


def _oracle_tail(x, lam_d, cm, lam_p, disparity=True):
    rgb, d1, d0, wc = (x[k].clone().requires_grad_(True) for k in ("rgb", "d1", "d0", "w_c"))
    lr = oc.rgb_loss(rgb, x["tgt"])
    ld = oc.depth_loss(d1, d0, x["td"], x["conf"], cm, disparity) * lam_d
    lp = oc.proposal_loss(x["s_f"], x["w_f"], x["s_c"], wc, lam_p)
    gs = torch.autograd.grad(lr + ld + lp, [rgb, d1, d0, wc])
    return (lr, ld, lp), gs


@gpu
@pytest.mark.parametrize("N,Sc,Pf", [(96, 128, 128), (1000, 128, 128), (333, 64, 33), (4096, 128, 128)])
def test_loss_tail_vs_oracle(golden, N, Sc, Pf):
    from snerf_amd import ops
    if N == 96:
        g = golden("g13_losses")
        x = {k: _t(g, k) for k in ("s_c", "s_f", "w_c", "w_f", "rgb", "tgt", "d1", "d0", "td", "conf")}
    else:
        x = _loss_inputs(N, Sc, Pf, N)
    (lr, ld, lp), (g_rgb, g_d1, g_d0, g_wc) = _oracle_tail(x, 0.2, 0.2, 0.05)
    c = {k: v.cuda().contiguous() for k, v in x.items()}
    out, h_rgb, h_d1, h_d0, h_wc = ops.mip_loss_tail(c["rgb"], c["tgt"], c["d1"], c["d0"], c["td"], c["conf"], c["s_f"], c["w_f"], c["s_c"], c["w_c"],
                                                     True, 0.2, 0.2, 0.05)
    out = out.cpu()
    assert int(out[0]) == int((x["td"] != 0).sum())
    for got, ref, name in ((out[1], lr, "rgb"), (out[2], ld, "depth"), (out[3], lp, "proposal")):
        assert abs(float(got) - float(ref)) <= 2e-6 * max(abs(float(ref)), 1e-3), (name, float(got), float(ref))
    if N == 96:                                                    # the reference's own numbers
        assert abs(float(out[3]) - float(g["proposal_loss"])) <= 2e-6 * float(g["proposal_loss"])
        # entries where +g / -g pairs cancel in the reverse prefix sum carry rounding noise of the row's magnitude
        assert torch.allclose(h_wc.cpu(), _t(g, "g_wc"), rtol=2e-5, atol=2e-6 * float(_t(g, "g_wc").abs().max()))
        assert torch.allclose(h_d1.cpu() / 0.2, _t(g, "g_d1"), rtol=2e-6, atol=0) and torch.allclose(h_rgb.cpu(), _t(g, "g_rgb"), rtol=1e-6, atol=0)
    assert torch.allclose(h_rgb.cpu(), g_rgb, rtol=1e-6, atol=0)
    assert torch.allclose(h_d1.cpu(), g_d1, rtol=2e-6, atol=0) and torch.allclose(h_d0.cpu(), g_d0, rtol=2e-6, atol=0)
    assert torch.allclose(h_wc.cpu(), g_wc, rtol=2e-5, atol=2e-6 * float(g_wc.abs().max())), float((h_wc.cpu() - g_wc).abs().max())
    # rgb-only call (no depth targets, no proposal loss)
    out2, h2, a, b, cgw = ops.mip_loss_tail(c["rgb"], c["tgt"], None, None, None, None, None, None, None, None, True, 0.2, 0.2, 0.05)
    assert a is None and b is None and cgw is None and torch.equal(h2, h_rgb) and float(out2[2]) == 0.0 and float(out2[3]) == 0.0


def test_trainer_loss_tail_host_logic():
    """MipTrainer.loss_and_grads on the CPU emulation == the oracle's loss tail (slots of the gradient tuple included)."""
    from tests.cpu_ops_emulation import emulate_ops
    from snerf_amd.trainer import MipTrainer
    x = _loss_inputs(64, 32, 33, 1)
    with emulate_ops():
        tr = MipTrainer.__new__(MipTrainer)
        tr.depth_lambda, tr.coarse_depth_mult, tr.proposal_loss, tr.proposal_lambda, tr.disparity_depth = 0.2, 0.2, True, 0.05, True
        outs = (x["d0"], None, x["s_c"], x["w_c"], x["rgb"], x["d1"], None, x["s_f"], x["w_f"])
        loss, g = tr.loss_and_grads(outs, x["tgt"], x["td"], x["conf"])
    (lr, ld, lp), (g_rgb, g_d1, g_d0, g_wc) = _oracle_tail(x, 0.2, 0.2, 0.05)
    assert abs(float(loss) - float(lr + ld + lp)) < 1e-6
    assert torch.allclose(g[0], g_d0) and g[1] is None and torch.allclose(g[2], g_wc) and torch.allclose(g[3], g_rgb) and torch.allclose(g[4], g_d1)
