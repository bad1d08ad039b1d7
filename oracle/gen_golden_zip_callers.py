#!/usr/bin/env python3
"""Golden vectors for the zipnerf (path C) callers either side of the render path (rows 8f-1, 8f-2): runs the REFERENCE's
camera_utils.pixels_to_rays and its loss functions (train_utils.compute_data_loss / anti_interlevel_loss / distortion_loss,
stepfun.blur_stepfun, math.sorted_interp_quad) on seeded inputs, plus the depth / semantic terms that live inline in the
training loop (s-nerfpp/zipnerf/train.py:252-298, evaluated here with the same torch expressions), and records inputs + outputs
as tests/golden/g15_zip_rays.npz and g16_zip_losses.npz.  Build-container only (needs /root/reference).

Third-party packages the image lacks and the loss code never calls (absl, rawpy, imageio, pycolmap) get empty import stubs."""
import os
import sys
import types

os.environ["TORCHDYNAMO_DISABLE"] = "1"
sys.dont_write_bytecode = True
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
if REPO not in sys.path:
    sys.path.insert(0, REPO)
from oracle import common as _oracle_common  # noqa: E402  (save_golden: writes the fixture, or compares under --check)
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)

from oracle import gen_golden_zip as gz  # noqa: E402


class _Stub(types.ModuleType):
    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return type(k, (), {}) if k[0].isupper() else (lambda *a, **kw: None)


def import_reference():
    gz.import_reference()
    absl, fl = types.ModuleType("absl"), types.ModuleType("absl.flags")
    absl.flags = fl
    fl.DEFINE_multi_string = lambda *a, **k: None
    fl.FLAGS = types.SimpleNamespace()
    sys.modules["absl"], sys.modules["absl.flags"] = absl, fl
    sys.modules["gin"].add_config_file_search_path = lambda *a, **k: None
    for name in ("rawpy", "imageio", "pycolmap"):
        try:
            __import__(name)
        except Exception:
            sys.modules[name] = _Stub(name)
    from internal import camera_utils, train_utils, stepfun, math as rmath
    return camera_utils, train_utils, stepfun, rmath


def histogram(g, R, S, lo=0.0, hi=1.0, peaky=False):
    """sorted fence posts in [lo, hi] and non-negative weights summing to < 1 (what compute_alpha_weights emits)"""
    s = torch.sort(torch.rand(R, S + 1, generator=g), dim=-1).values * (hi - lo) + lo
    s[:, 0], s[:, -1] = lo, hi
    w = torch.rand(R, S, generator=g) ** (6 if peaky else 2)
    w = w / w.sum(-1, keepdim=True) * torch.rand(R, 1, generator=g)
    return s, w


def main():
    cu, tu, stepfun, rmath = import_reference()
    g = torch.Generator().manual_seed(11)
    rng = np.random.default_rng(3)
    # ---- G15: rays of two Waymo-like cameras (different intrinsics / poses), 300 random pixels
    ncam, n = 2, 300
    Ks = np.array([[[2050., 0., 960.], [0., 2050., 640.], [0., 0., 1.]], [[2031.5, 0., 948.2], [0., 2044.1, 651.7], [0., 0., 1.]]])
    pixtocams = np.linalg.inv(Ks).astype(np.float32)
    c2w = []
    for th in (0.3, -1.1):
        c2w.append([[np.cos(th), 0.04, np.sin(th), 0.15], [-0.04, 1.0, 0.02, -0.05], [-np.sin(th), -0.02, np.cos(th), 0.075]])
    c2w = np.asarray(c2w, np.float32)
    cam_idx = rng.integers(0, ncam, n)
    px = rng.integers(0, 1920, n).astype(np.int32)
    py = rng.integers(0, 1280, n).astype(np.int32)
    o, d, vd, rad, ip, bx, by = cu.pixels_to_rays(px, py, pixtocams[cam_idx], c2w[cam_idx])
    f32 = lambda a: np.asarray(a, np.float32)
    _oracle_common.save_golden(os.path.join(OUT, "g15_zip_rays.npz"), pixtocams=pixtocams, camtoworlds=c2w, cam_idx=cam_idx.astype(np.int32), pix_x=px, pix_y=py,
                        origins=f32(o), directions=f32(d), viewdirs=f32(vd), radii=f32(rad), imageplane=f32(ip), base_x=f32(bx), base_y=f32(by))
    # ---- G16: loss tail on seeded renderer outputs (waymo.gin level sizes 64 / 64 / 32, shrunk ray count)
    R, C = 80, 19
    cfg = types.SimpleNamespace(data_loss_type="charb", charb_padding=0.001, disable_multiscale_loss=False, compute_disp_metrics=False,
                                compute_normal_metrics=False, data_coarse_loss_mult=0.0, data_loss_mult=1.0, anti_interlevel_loss_mult=0.01,
                                pulse_width=[0.03, 0.003], distortion_loss_mult=0.005)
    s0, w0 = histogram(g, R, 64)
    s1, w1 = histogram(g, R, 64)
    s2, w2 = histogram(g, R, 32, peaky=True)
    # the NeRF level's fence posts crowd where the proposal put its mass: make some rays sharply peaked and narrow
    s2[: R // 2] = 0.4 + 0.2 * s2[: R // 2]
    s2[: R // 2, 0], s2[: R // 2, -1] = 0.0, 1.0
    rgb = torch.rand(R, 3, generator=g)
    tgt = torch.rand(R, 3, generator=g)
    mask_rgb = torch.rand(R, generator=g) < 0.8
    depth = torch.rand(R, generator=g) * 60 + 1
    tdepth = torch.rand(R, generator=g) * 60 + 1
    tdepth[torch.rand(R, generator=g) < 0.4] = 0
    sem = torch.softmax(torch.randn(R, C, generator=g) * 2, -1) * torch.rand(R, 1, generator=g)
    labels = torch.randint(0, C, (R,), generator=g)
    leaves = [t.clone().requires_grad_(True) for t in (rgb, depth, sem, w0, w1, w2)]
    rgb_, depth_, sem_, w0_, w1_, w2_ = leaves
    hist = [dict(sdist=s0, weights=w0_), dict(sdist=s1, weights=w1_), dict(sdist=s2, weights=w2_)]
    batch = dict(rgb=tgt, mask_rgb=mask_rgb)
    data, stats = tu.compute_data_loss(batch, [dict(rgb=rgb_)], cfg)
    # train.py:252-255,277-278 (depth) and :294-298 (semantic), same expressions
    depth_mask = torch.logical_and(tdepth > 0, mask_rgb)
    dep_dist = 1 / (depth_[depth_mask] + 1e-5) - 1 / (1e-5 + tdepth[depth_mask])
    dep_lam = 0.5
    l_depth = dep_lam * torch.abs(dep_dist).mean()
    l_sem = torch.nn.NLLLoss()(torch.log(sem_[mask_rgb] + 1e-6), labels[mask_rgb].long()) * 0.04
    l_inter = tu.anti_interlevel_loss(hist, cfg)
    l_dist = tu.distortion_loss(hist, cfg)
    total = data + l_depth + l_sem + l_inter + l_dist
    grads = torch.autograd.grad(total, leaves)
    # stage vectors: blurred histogram and resampled CDF of the first ray block for both pulse widths
    wn = w2 / (s2[..., 1:] - s2[..., :-1])
    stage = {}
    for i, r in enumerate(cfg.pulse_width):
        c_, w_ = stepfun.blur_stepfun(s2, wn, r)
        area = 0.5 * (w_[..., 1:] + w_[..., :-1]) * (c_[..., 1:] - c_[..., :-1])
        cdf = torch.cat([torch.zeros_like(area[..., :1]), torch.cumsum(area, -1)], -1)
        stage[f"blur_x{i}"], stage[f"blur_y{i}"] = c_.numpy(), w_.numpy()
        stage[f"cdf_interp{i}"] = rmath.sorted_interp_quad([s0, s1][i], c_, w_, cdf).numpy()
    # the same two regularisers evaluated by the reference in float64 (its fp32 cumsum of the +/- steps of the blurred histogram carries
    # ~1e-4 relative noise): pins the restatement's algorithm to ~1e-12, the fp32 values above show the reference's own rounding
    dd = lambda t: t.double()
    leaves64 = [dd(t).clone().requires_grad_(True) for t in (w0, w1, w2)]
    hist64 = [dict(sdist=dd(s0), weights=leaves64[0]), dict(sdist=dd(s1), weights=leaves64[1]), dict(sdist=dd(s2), weights=leaves64[2])]
    li64, ld64 = tu.anti_interlevel_loss(hist64, cfg), tu.distortion_loss(hist64, cfg)
    g64 = torch.autograd.grad(li64 + ld64, leaves64)
    wn64 = dd(w2) / (dd(s2)[..., 1:] - dd(s2)[..., :-1])
    for i, r in enumerate(cfg.pulse_width):
        c_, w_ = stepfun.blur_stepfun(dd(s2), wn64, r)
        area = 0.5 * (w_[..., 1:] + w_[..., :-1]) * (c_[..., 1:] - c_[..., :-1])
        cdf = torch.cat([torch.zeros_like(area[..., :1]), torch.cumsum(area, -1)], -1)
        stage[f"blur_y{i}_f64"] = w_.numpy()
        stage[f"cdf_interp{i}_f64"] = rmath.sorted_interp_quad(dd([s0, s1][i]), c_, w_, cdf).numpy()
    stage.update(loss_interlevel_f64=li64.detach().numpy(), loss_distortion_f64=ld64.detach().numpy(), g_w0_f64=g64[0].numpy(),
                 g_w1_f64=g64[1].numpy(), g_w2_f64=g64[2].numpy())
    _oracle_common.save_golden(
        os.path.join(OUT, "g16_zip_losses.npz"), s0=s0.numpy(), w0=w0.numpy(), s1=s1.numpy(), w1=w1.numpy(), s2=s2.numpy(), w2=w2.numpy(),
        rgb=rgb.numpy(), target_rgb=tgt.numpy(), mask_rgb=mask_rgb.numpy(), depth=depth.numpy(), target_depth=tdepth.numpy(),
        semantic=sem.numpy(), labels=labels.numpy().astype(np.int32), depth_lambda=np.float64(dep_lam), sem_mult=np.float64(0.04),
        pulse_width=np.asarray(cfg.pulse_width, np.float64), anti_interlevel_mult=np.float64(0.01), distortion_mult=np.float64(0.005),
        charb_padding=np.float64(0.001), loss_data=data.detach().numpy(), mse=np.float32(stats["mses"][0]), loss_depth=l_depth.detach().numpy(),
        loss_sem=l_sem.detach().numpy(), loss_interlevel=l_inter.detach().numpy(), loss_distortion=l_dist.detach().numpy(),
        distortion_per_ray=stepfun.lossfun_distortion(s2, w2).numpy(),
        g_rgb=grads[0].numpy(), g_depth=grads[1].numpy(), g_sem=grads[2].numpy(), g_w0=grads[3].numpy(), g_w1=grads[4].numpy(), g_w2=grads[5].numpy(),
        **stage)
    print("wrote g15_zip_rays.npz, g16_zip_losses.npz")


if __name__ == "__main__":
    main()
