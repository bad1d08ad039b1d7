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
@pytest.mark.parametrize("seed", range(5))
def test_pinhole_rays_random_cameras_vs_oracle(seed):
    """Seeded fuzz: random image sizes, intrinsics (principal point off-centre, fx != fy), rotations and pixel subsets -- the kernel
    against oracle/callers.py, which oracle/fuzz_callers_vs_reference.py pins bit for bit to the reference on the same kind of input."""
    from snerf_amd import sample_utils as su
    g = torch.Generator().manual_seed(4000 + seed)
    R = lambda *s: torch.rand(*s, generator=g)
    H, W = 20 + int(R(1) * 200), 24 + int(R(1) * 300)
    th, ph = float(R(1)) * 2 - 1, float(R(1)) * 0.4 - 0.2
    rot = torch.tensor([[np.cos(th), 0.0, np.sin(th)], [0.0, 1.0, 0.0], [-np.sin(th), 0.0, np.cos(th)]], dtype=torch.float64) @ \
        torch.tensor([[1.0, 0.0, 0.0], [0.0, np.cos(ph), -np.sin(ph)], [0.0, np.sin(ph), np.cos(ph)]], dtype=torch.float64)
    pose = torch.cat([rot, torch.rand(3, 1, generator=g, dtype=torch.float64) * 4 - 2], 1).float().numpy()
    K = np.array([[40 + 600 * float(R(1)), 0.0, W * (0.4 + 0.2 * float(R(1)))], [0.0, 40 + 600 * float(R(1)), H * (0.4 + 0.2 * float(R(1)))], [0.0, 0.0, 1.0]], dtype=np.float32)
    n = 1 + int(R(1) * 500)
    sel = torch.stack([torch.randint(0, H, (n,), generator=g), torch.randint(0, W, (n,), generator=g)], -1)
    sel[0] = torch.tensor([H - 1, W - 1]); sel[-1] = torch.tensor([0, 0])             # the last row / column conventions of the reference
    near, far = 0.5 + float(R(1)) * 2, 50 + float(R(1)) * 100
    for training in (True, False):
        want = oc.pinhole_rays(sel, pose, K, H, near, far, training=training, W=W)
        got = su.rays_of_pixels(sel, pose, K, H, W, near, far, training=training)
        for k in ("origins", "directions", "lossmult", "near", "far"):
            a = getattr(got, k).reshape(n, -1).cpu()
            assert torch.equal(a, want[k].reshape(n, -1)), (seed, training, k, float((a - want[k].reshape(n, -1)).abs().max()))
        for k in ("viewdirs", "radii"):       # norms / 3-term sums: see test_pinhole_rays_vs_reference_golden
            a = getattr(got, k).reshape(n, -1).cpu()
            assert torch.allclose(a, want[k].reshape(n, -1), rtol=3e-7, atol=0), (seed, training, k)


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


def test_smooth_loss_patches_and_loss_vs_reference_golden(golden):
    """--smooth_loss (g32, from the reference's sample_patches_pt / SmoothLoss): same numpy RNG state -> the same patch pixels (drawn BEFORE the random
    pixel batch), and the caller-side loss with its gradient on the reference's inputs.  Host logic only: runs without a GPU."""
    from snerf_amd import sample_utils as su
    g = golden("g32_smooth_patches")
    H, W, n_rgb, p, npatch = (int(g[k]) for k in ("H", "W", "N_rgb", "patch_sz", "N_patch"))
    np.random.seed(13)
    patches = su.sample_patches(H, W, p, npatch)
    sel = np.random.choice(H * W, size=[n_rgb], replace=False)
    want = g["sel_coords"].numpy() if torch.is_tensor(g["sel_coords"]) else np.asarray(g["sel_coords"])
    assert np.array_equal(patches, want[n_rgb:]) and np.array_equal(np.stack([sel // W, sel % W], -1), want[:n_rgb])
    assert np.array_equal(sel, np.asarray(g["sel_inds"]))
    dist = _t(g, "patch_distance").clone().requires_grad_(True)
    loss = su.smooth_loss(_t(g, "image"), _t(g, "skymask"), torch.as_tensor(want[n_rgb:]), dist, npatch, p, float(g["smooth_lambda"]))
    loss.backward()
    assert torch.allclose(loss.detach(), _t(g, "smooth_loss"), rtol=1e-6, atol=0), (float(loss), float(_t(g, "smooth_loss")))
    assert torch.allclose(dist.grad, _t(g, "g_patch_distance"), rtol=1e-5, atol=1e-9)
    with pytest.raises(ValueError):
        su.sample_patches(10, 10, 6, 2)


@gpu
def test_sample_single_img_with_smooth_loss_patches(golden):
    """the whole training batch of the --smooth_loss branch: random pixels + patches -> rays and targets as the reference's (g32)"""
    from snerf_amd import sample_utils as su
    g = golden("g32_smooth_patches")
    args = types.SimpleNamespace(no_ndc=True, smooth_loss=True, N_rgb=int(g["N_rgb"]), patch_sz=int(g["patch_sz"]), N_patch=int(g["N_patch"]))
    image, depth = _t(g, "image", "cuda"), _t(g, "depth", "cuda")
    np.random.seed(13)
    rays, trgb, tdep, sel, inds = su.sample_single_img(args, image, depth, g["pose"], g["intrinsic"], near=2.0, far=100.0, near_far=False)
    assert torch.equal(sel.cpu(), _t(g, "sel_coords")) and np.array_equal(np.asarray(inds), np.asarray(g["sel_inds"]))
    assert torch.equal(trgb.cpu(), _t(g, "target_rgb")) and torch.equal(tdep.cpu(), _t(g, "target_depth"))
    assert torch.equal(rays.directions.cpu(), _t(g, "sel_directions")) and torch.equal(rays.origins.cpu(), _t(g, "sel_origins"))
    assert torch.allclose(rays.radii.cpu(), _t(g, "sel_radii"), rtol=3e-7, atol=0) and torch.equal(rays.near.cpu(), _t(g, "sel_near"))


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
    assert abs(float(out[4]) - float(lr + ld + lp)) <= 4e-6 * float(lr + ld + lp)      # out[4]: the total, formed by the same launch
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
    assert abs(float(out2[4]) - float(out2[1])) <= 4e-6 * float(out2[1])


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
    assert abs(float(loss) - float((lr + ld + lp).detach())) < 1e-6
    assert torch.allclose(g[0], g_d0) and g[1] is None and torch.allclose(g[2], g_wc) and torch.allclose(g[3], g_rgb) and torch.allclose(g[4], g_d1)


# ---------------------------------------------------------------- zipnerf (path C) callers ----
def _np(golden, name):
    return {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in golden(name).items()}


@gpu
def test_zip_pixels_to_rays_vs_reference_golden_and_full_frame(golden):
    from snerf_amd import ops
    g = _np(golden, "g15_zip_rays")
    i32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).cuda()
    f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
    r = ops.zip_pixels_to_rays(i32(g["pix_x"]), i32(g["pix_y"]), i32(g["cam_idx"]), f32(g["pixtocams"]), f32(g["camtoworlds"]), want_imageplane=True)
    for k in ("origins", "directions", "viewdirs", "radii", "imageplane", "base_x", "base_y"):
        got = r[k].cpu().numpy()
        assert got.shape == g[k].shape, (k, got.shape, g[k].shape)
        np.testing.assert_allclose(got, g[k], rtol=2e-7, atol=1e-9, err_msg=k)     # float64 sums on both sides, one fp32 rounding
    # the whole 1920 x 1280 frame of camera 1 (BASELINE config 5) against the oracle; cam_idx = None means camera 0
    W, H = 1920, 1280
    yy, xx = np.meshgrid(np.arange(H, dtype=np.int32), np.arange(W, dtype=np.int32), indexing="ij")
    px, py = xx.reshape(-1), yy.reshape(-1)
    r = ops.zip_pixels_to_rays(i32(px), i32(py), None, f32(g["pixtocams"][1:]), f32(g["camtoworlds"][1:]))
    ref = oc.zip_pixels_to_rays(px, py, np.zeros(px.size, np.int64), g["pixtocams"][1:], g["camtoworlds"][1:])
    assert "imageplane" not in r
    for k in ("origins", "directions", "viewdirs", "radii", "base_x", "base_y"):
        np.testing.assert_allclose(r[k].cpu().numpy(), ref[k], rtol=2e-7, atol=1e-9, err_msg=k)
    assert float(r["viewdirs"].norm(dim=-1).sub(1).abs().max()) < 2e-7


def _zip_tail_inputs(g):
    mask = g["mask_rgb"].astype(np.float32)
    dmask = mask * (g["target_depth"] > 0)
    return mask, dmask


@gpu
def test_zip_loss_tail_vs_reference_golden(golden):
    from snerf_amd import ops
    g = _np(golden, "g16_zip_losses")
    c = lambda a, dt=np.float32: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).cuda()
    mask, dmask = _zip_tail_inputs(g)
    hist = [(c(g[f"s{i}"]), c(g[f"w{i}"])) for i in range(3)]
    out, G = ops.zip_loss_tail(c(g["rgb"]), c(g["target_rgb"]), c(mask), depth=c(g["depth"]), tdepth=c(g["target_depth"]), dmask=c(dmask),
                               sem=c(g["semantic"]), labels=c(g["labels"], np.int32), smask=c(mask), hist=hist,
                               charb_padding=float(g["charb_padding"]), depth_lambda=float(g["depth_lambda"]), sem_mult=float(g["sem_mult"]),
                               pulse_width=[float(x) for x in g["pulse_width"]], interlevel_mult=float(g["anti_interlevel_mult"]),
                               distortion_mult=float(g["distortion_mult"]))
    out = out.cpu().numpy().astype(np.float64)
    L = dict(zip(ops.ZIP_LOSS_NAMES, out[4:]))
    assert out[0] == 3 * mask.sum() and out[1] == dmask.sum() and out[2] == 0 and out[3] == mask.sum()
    # the reference's fp32 run: elementwise terms to fp32 rounding, the two regularisers within the noise of its fp32 cumsums
    for k, ref, tol in (("data", "loss_data", 2e-6), ("mse", "mse", 2e-6), ("depth", "loss_depth", 2e-6), ("sem", "loss_sem", 2e-6),
                        ("interlevel", "loss_interlevel", 2e-5), ("distortion", "loss_distortion", 2e-5)):
        assert abs(L[k] - float(g[ref])) <= tol * abs(float(g[ref])), (k, L[k], float(g[ref]))
    assert L["d_complete"] == 0
    for k, ref, tol in (("rgb", "g_rgb", 2e-6), ("depth", "g_depth", 2e-6), ("semantic", "g_sem", 2e-6), ("w0", "g_w0", 3e-5), ("w1", "g_w1", 3e-5),
                        ("w2", "g_w2", 3e-5)):
        got = G[k].cpu().numpy()
        assert np.abs(got - g[ref]).max() <= tol * np.abs(g[ref]).max(), (k, np.abs(got - g[ref]).max(), np.abs(g[ref]).max())
    # the reference evaluated in float64 (its knots c -/+ r are float64 there, fp32 in the kernel as in the fp32 reference)
    assert abs(L["interlevel"] - float(g["loss_interlevel_f64"])) <= 1e-5 * L["interlevel"]
    assert abs(L["distortion"] - float(g["loss_distortion_f64"])) <= 2e-6 * L["distortion"]
    for k in ("w0", "w1", "w2"):
        ref = g[f"g_{k}_f64"]
        assert np.abs(G[k].cpu().numpy() - ref).max() <= 5e-5 * np.abs(ref).max(), k


def _rand_hist(rng, R, S, peaky=False, lo=0.0, hi=1.0):
    s = np.sort(rng.random((R, S + 1)), -1) * (hi - lo) + lo
    s[:, 0], s[:, -1] = 0.0, 1.0
    w = rng.random((R, S)) ** (6 if peaky else 2)
    w = w / w.sum(-1, keepdims=True) * rng.random((R, 1))
    return s.astype(np.float32), w.astype(np.float32)


@gpu
@pytest.mark.parametrize("R,S0,S1,S2,C", [(4099, 64, 64, 32, 19), (65, 7, 130, 3, 1), (1, 64, 64, 32, 4), (70, 128, 128, 64, 2), (70, 300, 64, 32, 2)])
def test_zip_loss_tail_vs_oracle(R, S0, S1, S2, C):
    """waymo.gin level sizes at a non-multiple-of-64 ray count, odd level sizes (a proposal level wider than the blurred NeRF
    histogram has knots, a 3-interval NeRF level), and a single ray; masks with zeros, both depth masks, duplicated fence posts
    in the proposal levels and exactly-zero weights.  The last two cases: a workgroup's staged rows above 64 KB of LDS (the interlevel
    kernel's opt-in limit) and above 160 KB (the walk falls back to global memory)."""
    from snerf_amd import ops
    rng = np.random.default_rng(R + S1)
    s0, w0 = _rand_hist(rng, R, S0)
    s1, w1 = _rand_hist(rng, R, S1)
    s2, w2 = _rand_hist(rng, R, S2, peaky=True)
    if S0 > 4:
        s0[:, 3] = s0[:, 2]                       # zero-width proposal interval
    w1[:, ::5] = 0.0
    w2[:, 0] = 0.0
    rgb, tgt = rng.random((R, 3), dtype=np.float32), rng.random((R, 3), dtype=np.float32)
    lm = (rng.random(R) < 0.7).astype(np.float32)
    lm[0] = 1.0
    depth = (rng.random(R) * 60 + 1).astype(np.float32)
    td = (rng.random(R) * 60 + 1).astype(np.float32)
    dm = (lm * (rng.random(R) < 0.5)).astype(np.float32)
    cm = ((1 - lm) * (rng.random(R) < 0.5)).astype(np.float32)
    sem = rng.random((R, C), dtype=np.float32)
    sem = sem / sem.sum(-1, keepdims=True) * rng.random((R, 1), dtype=np.float32)
    lab = rng.integers(0, C, R).astype(np.int32)
    kw = dict(charb_padding=0.001, data_mult=1.0, depth_lambda=0.5, com_mult=0.2, sem_mult=0.04, pulse_width=(0.03, 0.003), interlevel_mult=0.01,
              distortion_mult=0.005)
    Lr, Gr = oc.zip_loss_tail(rgb, tgt, lm, depth, td, dm, cm, sem, lab, lm, [s0, s1, s2], [w0, w1, w2], *kw.values())
    c = lambda a: torch.from_numpy(a).cuda()
    out, G = ops.zip_loss_tail(c(rgb), c(tgt), c(lm), depth=c(depth), tdepth=c(td), dmask=c(dm), cmask=c(cm), sem=c(sem), labels=c(lab), smask=c(lm),
                               hist=[(c(s0), c(w0)), (c(s1), c(w1)), (c(s2), c(w2))], **kw)
    out = out.cpu().numpy().astype(np.float64)
    for i, k in enumerate(ops.ZIP_LOSS_NAMES):
        ref = float(Lr.get(k, 0.0))
        assert abs(out[4 + i] - ref) <= 1e-5 * abs(ref) + 1e-9, (k, out[4 + i], ref)
    for k in ("rgb", "depth", "semantic", "w0", "w1", "w2"):
        got, ref = G[k].cpu().numpy(), Gr[k]
        assert np.isfinite(got).all(), k
        assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max() + 1e-12, (k, np.abs(got - ref).max(), np.abs(ref).max())


@gpu
def test_zip_loss_tail_absent_terms_and_empty_masks():
    from snerf_amd import ops
    from snerf_amd._lib import SnerfHipError
    rng = np.random.default_rng(0)
    R = 130
    c = lambda a: torch.from_numpy(a).cuda()
    rgb, tgt = c(rng.random((R, 3), dtype=np.float32)), c(rng.random((R, 3), dtype=np.float32))
    out, G = ops.zip_loss_tail(rgb, tgt, mse=True)                                   # the 'mse' data term alone
    ref = float(((rgb - tgt) ** 2).mean())
    assert abs(float(out[4]) - ref) < 1e-6 * ref and abs(float(out[5]) - ref) < 1e-6 * ref and float(out[6:].abs().sum()) == 0
    assert torch.allclose(G["rgb"], 2 * (rgb - tgt) / (3 * R), rtol=1e-5, atol=1e-9)
    assert all(G[k] is None for k in ("depth", "semantic", "w0", "w1", "w2"))
    # empty masks: the terms are 0 (the reference's mean over an empty selection is NaN) and so are their gradients
    z = torch.zeros(R, device="cuda")
    depth, td = c((rng.random(R) + 1).astype(np.float32)), c((rng.random(R) + 1).astype(np.float32))
    sem, lab = c(rng.random((R, 5), dtype=np.float32)), c(rng.integers(0, 5, R).astype(np.int32))
    out, G = ops.zip_loss_tail(rgb, tgt, depth=depth, tdepth=td, dmask=z, cmask=z, sem=sem, labels=lab, smask=z)
    assert float(out[6]) == 0 and float(out[7]) == 0 and float(out[8]) == 0
    assert float(G["depth"].abs().max()) == 0 and float(G["semantic"].abs().max()) == 0
    # only one proposal level, distortion off
    s0, w0 = _rand_hist(rng, R, 16)
    s2, w2 = _rand_hist(rng, R, 8)
    out, G = ops.zip_loss_tail(rgb, tgt, hist=[(c(s0), c(w0)), (None, None), (c(s2), c(w2))], distortion_mult=0.0)
    Lr, Gr = oc.zip_loss_tail(rgb.cpu().numpy(), tgt.cpu().numpy(), None, None, None, None, None, None, None, None, [s0, s2], [w0, w2],
                              pulse_width=(0.03,), distortion_mult=0.0)
    assert abs(float(out[9]) - Lr["interlevel"]) <= 1e-5 * Lr["interlevel"] and float(out[10]) == 0
    assert G["w1"] is None and G["w2"] is None
    assert np.abs(G["w0"].cpu().numpy() - Gr["w0"]).max() <= 1e-5 * np.abs(Gr["w0"]).max()
    with pytest.raises(SnerfHipError):
        ops.zip_loss_tail(rgb, tgt, hist=[(c(s0), c(w0)), (None, None), (c(s2), c(w2))], pulse_width=(0.0, 0.003))


@gpu
@pytest.mark.parametrize("C", [1, 4])
def test_hash_decay_vs_oracle(C):
    """train_utils.py:184-203 through the C-ABI: value and accumulated gradient on a 3-level table with unequal level sizes."""
    from snerf_amd import ops
    rng = np.random.default_rng(C)
    off = np.array([0, 4920, 40864, 40864 + 2 ** 16], np.int32)
    tab = (rng.standard_normal((int(off[-1]), C)) * 0.1).astype(np.float32)
    g0 = rng.standard_normal(tab.shape).astype(np.float32) * 1e-3
    ref_l, ref_g = oc.hash_decay_loss(tab, off, 0.1)
    t, g = torch.from_numpy(tab).cuda(), torch.from_numpy(g0.copy()).cuda()
    loss = torch.zeros(1, device="cuda")
    ops.hash_decay(t, g, torch.from_numpy(off).cuda(), 3, C, 0.1, loss)
    assert abs(float(loss) - ref_l) <= 1e-5 * ref_l
    np.testing.assert_allclose(g.cpu().numpy() - g0, ref_g, rtol=1e-4, atol=1e-9)
    ops.hash_decay(t, g, torch.from_numpy(off).cuda(), 3, C, 0.0, loss)             # mult 0: nothing happens
    assert abs(float(loss) - ref_l) <= 1e-5 * ref_l


def test_trainer_ray_gradients_for_pose_refinement():
    """MipTrainer.step(ray_grads=True): the fused step also hands back d loss / d rays, equal to what autograd through
    MipNerfModel.forward gives for the same loss (host logic on the CPU emulation)."""
    from cpu_ops_emulation import emulate_ops
    from oracle import common
    from snerf_amd import mipnerf
    from snerf_amd.trainer import MipTrainer
    with emulate_ops():
        torch.manual_seed(0)
        m = mipnerf.MipNerfModel(n_samples=8, N_fine=9, no_warp_sample=0, ray_shape="cone", fn=1, radius=3., transform_idx=0, real=True, rgb_layer=3,
                                 hidden_layer=64, density_noise=0., max_deg_point=16, proposal_hidden_layer=64, proposal_loss=True, compute="f32",
                                 device="cpu")
        n = 12
        rc = common.synthetic_rays(n, seed=2)
        g = torch.Generator().manual_seed(3)
        tgt, td, conf = torch.rand(n, 3, generator=g), torch.rand(n, generator=g) * 20 + 2, torch.rand(n, generator=g)
        leaves = {k: rc[k].clone().requires_grad_(True) for k in ("origins", "directions", "viewdirs")}
        ret = m(mipnerf.Rays(**{**rc, **leaves}), False, False, 0.)
        tr = MipTrainer(m, lr=0.0, depth_lambda=0.2, coarse_depth_mult=0.2, proposal_loss=True, proposal_lambda=0.05)
        outs = (ret[0][1], ret[0][2], ret[0][3], ret[0][4], ret[1][0], ret[1][1], ret[1][2], ret[1][4], ret[1][5])
        # the same loss through autograd: dL/d outputs from the fused tail, applied to the differentiable outputs
        loss, gouts = tr.loss_and_grads(tuple(o.detach() for o in outs), tgt, td, conf)
        g_dist0, _, g_w0, g_rgb1, g_dist1, _, _ = gouts
        torch.autograd.backward([ret[0][1], ret[0][4], ret[1][0], ret[1][1]], [g_dist0, g_w0, g_rgb1, g_dist1])
        tr.step(mipnerf.Rays(**rc), tgt, td, conf, randomized=False, ray_grads=True)
        assert tr.last_ray_grads is not None
        for k, got in zip(("origins", "directions", "viewdirs"), tr.last_ray_grads):
            ref = leaves[k].grad
            assert float(ref.abs().max()) > 0 and float((got - ref).abs().max()) <= 1e-4 * float(ref.abs().max()), k
        tr.step(mipnerf.Rays(**rc), tgt, td, conf, randomized=False)
        assert tr.last_ray_grads is None


def test_apply_pose_transform_is_sample_rays_pose_branch_and_is_differentiable():
    """sample_utils.py:421-435: directions / viewdirs rotated by pose[:3,:3] (as a row-vector product), origins shifted by pose[:3,3];
    autograd reaches the pose (what `pose_refine = True` trains)."""
    from snerf_amd import sample_utils
    g = torch.Generator().manual_seed(0)
    n = 17
    mk = lambda *s: torch.randn(*s, generator=g)
    rays = sample_utils.Rays(mk(n, 3), mk(n, 3), mk(n, 3), mk(n, 1), torch.ones(n, 1), torch.ones(n, 1), torch.ones(n, 1) * 9, torch.zeros(n, 1))
    pose = torch.eye(4) + 0.05 * mk(4, 4)
    pose.requires_grad_(True)
    out = sample_utils.apply_pose_transform(rays, pose)
    assert torch.allclose(out.directions, (rays.directions[:, None, :] * pose[:3, :3]).sum(axis=-1))
    assert torch.allclose(out.viewdirs, rays.viewdirs @ pose[:3, :3].t()) and torch.allclose(out.origins, rays.origins + pose[:3, 3])
    assert out.radii is rays.radii and out.near is rays.near
    (out.origins.sum() + (out.directions ** 2).sum() + out.viewdirs[:, 0].sum()).backward()
    assert pose.grad is not None and float(pose.grad[:3, :].abs().min()) > 0 and float(pose.grad[3].abs().max()) == 0
