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
    assert abs(float(pl.detach()) - float(g["proposal_loss"])) <= 1e-7 * max(1.0, abs(float(pl.detach())))
    assert torch.allclose(gw, _t(g, "g_wc"), rtol=1e-6, atol=1e-9)
    rgb = _t(g, "rgb").requires_grad_(True)
    rl = oc.rgb_loss(rgb, _t(g, "tgt"))
    gr, = torch.autograd.grad(rl, rgb)
    assert float(rl.detach()) == float(g["rgb_loss"]) and torch.equal(gr, _t(g, "g_rgb"))
    d1, d0 = _t(g, "d1").requires_grad_(True), _t(g, "d0").requires_grad_(True)
    dl = oc.depth_loss(d1, d0, _t(g, "td"), _t(g, "conf"), float(g["coarse_depth_mult"]), True)
    g1, g0 = torch.autograd.grad(dl, [d1, d0])
    assert abs(float(dl.detach()) - float(g["depth_loss"])) <= 1e-7
    assert torch.allclose(g1, _t(g, "g_d1"), rtol=1e-6, atol=0) and torch.allclose(g0, _t(g, "g_d0"), rtol=1e-6, atol=0)


# ---------------------------------------------------------------- zipnerf (path C) callers ----
def _np(golden, name):
    return {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in golden(name).items()}


def test_zip_pixels_to_rays(golden):
    g = _np(golden, "g15_zip_rays")
    r = oc.zip_pixels_to_rays(g["pix_x"], g["pix_y"], g["cam_idx"], g["pixtocams"], g["camtoworlds"])
    for k in ("origins", "directions", "viewdirs", "radii", "imageplane", "base_x", "base_y"):
        assert r[k].shape == g[k].shape, (k, r[k].shape, g[k].shape)
        # fp64 arithmetic on both sides, rounded to fp32 once: a last-bit difference can only come from summation order
        np.testing.assert_allclose(r[k], g[k], rtol=2e-7, atol=1e-9, err_msg=k)


def test_zip_blur_and_quadratic_cdf_stages(golden):
    g = _np(golden, "g16_zip_losses")
    c, w = g["s2"].astype(np.float64), g["w2"].astype(np.float64)
    wn = w / (c[:, 1:] - c[:, :-1])
    for i, r in enumerate(g["pulse_width"]):
        xr32, _ = oc.blur_stepfun(g["s2"], wn, float(r))
        assert np.array_equal(xr32.astype(np.float32), g[f"blur_x{i}"])           # fp32 inputs: the reference's fp32 knots, bit for bit
        xr, yr = oc.blur_stepfun(c, wn, float(r))                                  # float64 inputs: the float64 evaluation
        scale = np.abs(g[f"blur_y{i}"]).max(-1, keepdims=True)
        np.testing.assert_allclose(yr, g[f"blur_y{i}_f64"], rtol=1e-9, atol=1e-9 * scale.max())   # the reference run in float64
        assert np.all(np.abs(yr - g[f"blur_y{i}"]) <= 5e-4 * scale)      # its fp32 cumsum of +/- steps: ~1e-4 of the row maximum
        area = 0.5 * (yr[:, 1:] + yr[:, :-1]) * (xr[:, 1:] - xr[:, :-1])
        cdf = np.concatenate([np.zeros_like(area[:, :1]), np.cumsum(area, -1)], -1)
        ci = oc.sorted_interp_quad(g[f"s{i}"].astype(np.float64), xr, yr, cdf)
        np.testing.assert_allclose(ci, g[f"cdf_interp{i}_f64"], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(ci, g[f"cdf_interp{i}"], rtol=5e-3, atol=2e-3)    # the reference's fp32 evaluation: noise of its own cumsums


def test_zip_loss_tail_values_and_gradients(golden):
    g = _np(golden, "g16_zip_losses")
    mask = g["mask_rgb"].astype(np.float64)
    dmask = mask * (g["target_depth"] > 0)
    args = (float(g["charb_padding"]), 1.0, float(g["depth_lambda"]), 0.2, float(g["sem_mult"]), [float(x) for x in g["pulse_width"]],
            float(g["anti_interlevel_mult"]), float(g["distortion_mult"]))
    L, G = oc.zip_loss_tail(g["rgb"], g["target_rgb"], mask, g["depth"], g["target_depth"], dmask, None, g["semantic"], g["labels"], mask,
                            [g["s0"], g["s1"], g["s2"]], [g["w0"], g["w1"], g["w2"]], *args)
    for k, ref in (("data", "loss_data"), ("mse", "mse"), ("depth", "loss_depth"), ("sem", "loss_sem"), ("interlevel", "loss_interlevel"),
                   ("distortion", "loss_distortion")):
        assert abs(L[k] - float(g[ref])) <= 2e-5 * abs(float(g[ref])) + 1e-9, (k, L[k], float(g[ref]))
    np.testing.assert_allclose(oc.lossfun_distortion(g["s2"], g["w2"]), g["distortion_per_ray"], rtol=2e-5, atol=1e-8)
    for k, ref in (("rgb", "g_rgb"), ("depth", "g_depth"), ("semantic", "g_sem"), ("w0", "g_w0"), ("w1", "g_w1"), ("w2", "g_w2")):
        tol = 3e-5 * np.abs(g[ref]).max()
        assert np.abs(G[k] - g[ref]).max() <= tol, (k, np.abs(G[k] - g[ref]).max(), tol)
    assert float(np.abs(g["g_w0"]).max()) > 0 and float(np.abs(g["g_w1"]).max()) > 0
    # against the reference evaluated in float64 (float64 inputs -> float64 knots on both sides): the algorithm itself, to rounding
    d = lambda k: g[k].astype(np.float64)
    L, G = oc.zip_loss_tail(g["rgb"], g["target_rgb"], mask, None, None, None, None, None, None, None, [d("s0"), d("s1"), d("s2")],
                            [d("w0"), d("w1"), d("w2")], *args)
    assert abs(L["interlevel"] - float(g["loss_interlevel_f64"])) <= 1e-10 * abs(L["interlevel"])
    assert abs(L["distortion"] - float(g["loss_distortion_f64"])) <= 1e-10 * abs(L["distortion"])
    for k in ("w0", "w1", "w2"):
        np.testing.assert_allclose(G[k], g[f"g_{k}_f64"], rtol=1e-8, atol=1e-14)
