"""oracle/callers.py (ray generation, loss tail) against vectors captured from the reference
(oracle/gen_golden_callers.py -> tests/golden/g12_rays.npz, g13_losses.npz)."""
import numpy as np
import torch

from oracle import callers as oc


def _t(g, k):
    return torch.from_numpy(np.asarray(g[k]))


def test_pinhole_rays_selected_and_full_frame(golden):
    g = golden("g12_rays")
    H, W = int(g["H"]), int(g["W"])
    for prefix, coords in (("sel_", _t(g, "sel_coords")),
                           ("full_", torch.stack(torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij"), -1).reshape(-1, 2))):
        near = float(g[prefix + "near"].reshape(-1)[0])
        far = float(g[prefix + "far"].reshape(-1)[0])
        r = oc.pinhole_rays(coords, g["pose"], g["intrinsic"], H, near, far, training=prefix == "sel_", W=W)
        for k in ("origins", "directions", "viewdirs", "radii", "lossmult", "near", "far", "app"):
            ref = _t(g, prefix + k)
            assert r[k].shape == ref.shape, (prefix, k, r[k].shape, ref.shape)
            assert torch.equal(r[k], ref), (prefix, k, float((r[k] - ref).abs().max()))


def test_loss_tail(golden):
    g = golden("g13_losses")
    w_c = _t(g, "w_c").requires_grad_(True)
    pl = oc.proposal_loss(_t(g, "s_f"), _t(g, "w_f"), _t(g, "s_c"), w_c, float(g["proposal_lambda"]))
    gw, = torch.autograd.grad(pl, w_c)
    assert abs(float(pl) - float(g["proposal_loss"])) <= 1e-7 * max(1.0, abs(float(pl)))
    assert torch.allclose(gw, _t(g, "g_wc"), rtol=1e-6, atol=1e-9)
    rgb = _t(g, "rgb").requires_grad_(True)
    rl = oc.rgb_loss(rgb, _t(g, "tgt"))
    gr, = torch.autograd.grad(rl, rgb)
    assert float(rl) == float(g["rgb_loss"]) and torch.equal(gr, _t(g, "g_rgb"))
    d1, d0 = _t(g, "d1").requires_grad_(True), _t(g, "d0").requires_grad_(True)
    dl = oc.depth_loss(d1, d0, _t(g, "td"), _t(g, "conf"), float(g["coarse_depth_mult"]), True)
    g1, g0 = torch.autograd.grad(dl, [d1, d0])
    assert abs(float(dl) - float(g["depth_loss"])) <= 1e-7
    assert torch.allclose(g1, _t(g, "g_d1"), rtol=1e-6, atol=0) and torch.allclose(g0, _t(g, "g_d0"), rtol=1e-6, atol=0)
