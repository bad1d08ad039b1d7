#!/usr/bin/env python3
"""Early ray termination + sample compaction (csrc/ert.hip) on a scene with REAL opacity: the path-A model is fitted
for a few hundred steps to an analytic street-like scene -- four opaque boxes at 12-60 m, a checkered ground plane, sky beyond -- with
RGB + depth supervision (MipTrainer: the reference's RgbLoss + disparity DepthLoss), so that the proposal histograms are peaked at the
surfaces; then the 1600 x 900 frame is rendered without and with ert=(eps_t, eps_w).

    python tools/ert_scene.py [--steps 300] [--eps 1e-4 1e-4]

`fit_and_render()` is what bench.py's `ert_scene` leg calls."""
import argparse
import json
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

H, W, FOCAL, NEAR, FAR, TH = 900, 1600, 1266.0, 1.8, 110.0, 0.3
BOXES = [  # centre and half-size in the CAMERA frame (x right, y up, looking along -z), colour
    ((-4.0, 0.2, -12.0), (1.5, 1.8, 1.5), (0.85, 0.20, 0.20)),
    ((3.0, -0.3, -25.0), (2.5, 1.3, 2.0), (0.20, 0.70, 0.30)),
    ((-8.0, 1.0, -45.0), (4.0, 2.6, 3.0), (0.20, 0.30, 0.85)),
    ((10.0, 2.0, -60.0), (5.0, 3.6, 4.0), (0.85, 0.80, 0.20)),
]
GROUND_Y = -1.6


def pose():
    return np.array([[math.cos(TH), 0.0, math.sin(TH), 0.0], [0.0, 1.0, 0.0, 0.0], [-math.sin(TH), 0.0, math.cos(TH), 0.0]], dtype=np.float32)


def analytic_scene(origins, directions):
    """-> (rgb [n,3], t_hit [n]; 0 = sky).  t is in units of the un-normalised ray direction, the model's distance unit."""
    dev = origins.device
    R = torch.from_numpy(pose()[:, :3]).to(dev)
    o, d = origins @ R, directions @ R                                    # camera frame (R^T x as a row vector: x @ R)
    n = o.shape[0]
    best = torch.full((n,), float("inf"), device=dev)
    rgb = torch.zeros(n, 3, device=dev)
    sky = torch.stack([0.55 + 0.25 * d[:, 1].clamp(-1, 1), 0.70 + 0.15 * d[:, 1].clamp(-1, 1), torch.full((n,), 0.95, device=dev)], -1)
    inv = 1.0 / torch.where(d.abs() < 1e-9, torch.full_like(d, 1e-9), d)
    for c, h, col in BOXES:
        c, h, col = (torch.tensor(v, device=dev) for v in (c, h, col))
        t0, t1 = (c - h - o) * inv, (c + h - o) * inv
        tn, tf = torch.minimum(t0, t1), torch.maximum(t0, t1)
        tnear, axis = tn.max(-1)
        tfar = tf.min(-1).values
        hit = (tnear < tfar) & (tnear > NEAR) & (tnear < best)
        shade = torch.tensor([0.75, 1.0, 0.55], device=dev)[axis]         # face-dependent shading
        rgb = torch.where(hit[:, None], col[None] * shade[:, None], rgb)
        best = torch.where(hit, tnear, best)
    tg = (GROUND_Y - o[:, 1]) / torch.where(d[:, 1] < -1e-6, d[:, 1], torch.full_like(d[:, 1], -1e-6))
    hit = (d[:, 1] < -1e-6) & (tg > NEAR) & (tg < 100.0) & (tg < best)
    pt = o + tg[:, None] * d
    chk = ((torch.floor(pt[:, 0] / 2.0) + torch.floor(pt[:, 2] / 2.0)) % 2 == 0).float()
    gcol = (0.35 + 0.25 * chk)[:, None].expand(-1, 3)
    rgb = torch.where(hit[:, None], gcol, rgb)
    best = torch.where(hit, tg, best)
    miss = torch.isinf(best)
    return torch.where(miss[:, None], sky, rgb), torch.where(miss, torch.zeros_like(best), best)


def rays_of(coords, first, n, dev):
    from snerf_amd import ops
    from snerf_amd.mipnerf import Rays
    o, d, v, r, nr, fr = ops.pinhole_rays(coords, first, n, W, H, pose(), W * 0.5, H * 0.5, FOCAL, FOCAL, False, NEAR, FAR, dev)
    ones = torch.ones_like(r)
    return Rays(o, d, v, r, ones, nr, fr, ones * 0)


def precision_on_fitted_weights(model, build, rows=40, row0=430):
    """bf16 / split-bf16 against exact fp32 on TRAINED weights (the init-time bounds say nothing about a fitted
    model): `rows` image rows through the scene's boxes and ground, rendered by `model`'s weights in compute = "f32", "bf16x3", "f16f8"
    (fp16 tiles + e4m3 correction tiles: the two-pass-equivalent form) and "bf16".  `build(compute)` -> a fresh MipNerfModel of the same shape.  -> dict of PSNR / max errors vs the f32 render."""
    dev = model.arena.flat.device
    n = rows * W
    rays = rays_of(None, row0 * W, n, dev)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    outs = {}
    with torch.no_grad():
        for mode in ("f32", "bf16x3", "f16f8", "fp16", "bf16"):
            m = model if mode == "bf16" else build(mode)
            if m is not model:
                m.load_state_dict(sd)
            parts = [m(type(rays)(*[r[a:a + 16384] for r in rays]), False, False, 0.) for a in range(0, n, 16384)]
            outs[mode] = tuple(torch.cat([p[1][k] for p in parts], 0) for k in (0, 1, 2))      # rgb, distance, acc
            del m
    torch.cuda.empty_cache()
    ref = outs["f32"]
    res = {"rays": n, "what": f"rows {row0}..{row0 + rows - 1} of the 1600 x 900 frame of the FITTED model, each mode vs compute='f32' on the same weights"}
    for mode in ("bf16x3", "f16f8", "fp16", "bf16"):
        rgb, dist, acc = outs[mode]
        mse = float(((rgb.double() - ref[0].double()) ** 2).mean())
        res[mode] = {"psnr_db": float("inf") if mse == 0 else round(-10.0 * math.log10(mse), 2),
                     "max_abs_err_rgb": float((rgb - ref[0]).abs().max()),
                     "max_rel_err_depth": float(((dist - ref[1]).abs() / ref[1].abs().clamp(min=1e-6)).max()),
                     "p999_rel_err_depth": float(torch.quantile(((dist - ref[1]).abs() / ref[1].abs().clamp(min=1e-6)).float(), 0.999)),
                     "max_abs_err_acc": float((acc - ref[2]).abs().max())}
    return res


def fit(model, steps, rays_per_step=4096, lr=1e-3, seed=0, log=None):
    """`steps` train steps of `model` on the analytic scene (RGB + depth supervised) -> seconds"""
    from snerf_amd.trainer import MipTrainer
    dev = model.arena.flat.device
    tr = MipTrainer(model, lr=lr, depth_lambda=0.5, coarse_depth_mult=1.0)
    g = torch.Generator(device="cpu").manual_seed(seed)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for it in range(steps):
        pix = torch.randint(0, H * W, (rays_per_step,), generator=g)
        coords = torch.stack([pix // W, pix % W], -1).int().to(dev)
        rays = rays_of(coords, 0, rays_per_step, dev)
        rgb, t_hit = analytic_scene(rays.origins, rays.directions)
        tr.lr = lr * (0.1 ** (it / max(steps, 1)))
        loss, _ = tr.step(rays, rgb, t_hit, torch.ones_like(t_hit))
        if log is not None and (it % 50 == 0 or it == steps - 1):
            log(f"fit step {it}: loss {float(loss):.5f}")
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def fit_and_precision(model, steps, build):
    """bench.py's fitted_weights_precision leg: fit, then every compute mode against exact fp32 on the fitted weights"""
    t_fit = fit(model, steps)
    res = precision_on_fitted_weights(model, build)
    res["fit_steps"], res["fit_s"] = steps, round(t_fit, 2)
    return res


def fit_and_render(model, steps=300, eps=(1e-4, 1e-4), rays_per_step=4096, chunk=32768, lr=1e-3, seed=0, log=None, group=0, rows=None, row0=0,
                   build=None):
    """Fit `model` (a MipNerfModel on its device) to the analytic scene, then render the frame plain and with ERT.  -> dict
    `group` > 0: front-to-back termination on the fine network's own densities in groups of that many samples (eps_t is then an exact
    bound); 0: the selection from the proposal histogram (eps_t, eps_w).  `rows` (+ `row0`): render only that window of image rows (a
    short leg of bench.py); `build`: also run precision_on_fitted_weights with it."""
    from snerf_amd.mipnerf import Rays, render_image
    dev = model.arena.flat.device
    t_fit = fit(model, steps, rays_per_step, lr, seed, log)
    prec = precision_on_fitted_weights(model, build) if build is not None else None
    Hw = H if rows is None else rows                                   # (a window is rendered and scored like a frame of `rows` rows)
    fr = rays_of(None, row0 * W, Hw * W, dev)
    grid = Rays(*[r.reshape(Hw, W, -1) for r in fr])
    stats = {"kept": 0, "tot": 0}

    def render(ert):
        def fn(r):
            ret = model(r, False, False, 0., ert=ert)
            if ert is not None:
                stats["kept"] += model.last_ert_rows[0]; stats["tot"] += model.last_ert_rows[1]
            return ret
        with torch.no_grad():
            fn(Rays(*[r[:chunk] for r in fr]))                             # warm-up chunk
            stats["kept"] = stats["tot"] = 0
            torch.cuda.synchronize(); t1 = time.perf_counter()
            out = render_image(fn, grid, 0, chunk=chunk, world=1)
            torch.cuda.synchronize()
        return out, time.perf_counter() - t1
    (rgb_f, dist_f, acc_f, _), t_full = render(None)
    (rgb_e, dist_e, acc_e, _), t_ert = render((float(eps[0]), float(eps[1])) + ((int(group),) if group else ()))
    tgt_rgb, tgt_t = analytic_scene(fr.origins, fr.directions)
    mse = lambda a, b: float(((a - b) ** 2).mean())
    psnr = lambda m: float("inf") if m == 0 else -10.0 * math.log10(m)
    hitm = (tgt_t > 0).reshape(Hw, W)
    window = None if rows is None else f"rows {row0}..{row0 + rows - 1} of {H}"
    return {"window": window, "precision_on_fitted_weights": prec, "scene": "4 opaque boxes at 12-60 m + checkered ground plane + sky (tools/ert_scene.py), RGB + depth supervised",
            "fit_steps": steps, "fit_s": round(t_fit, 2), "fit_psnr_db": round(psnr(mse(rgb_f.reshape(-1, 3), tgt_rgb)), 2),
            "fit_depth_median_rel_err": round(float(((dist_f - tgt_t.reshape(Hw, W)).abs() / tgt_t.reshape(Hw, W).clamp(min=1))[hitm].median()), 4),
            "mode": f"front to back on the fine network's densities, groups of {group} samples (exact bound eps_t)" if group else "selection from the proposal histogram",
            "eps_t": eps[0], "eps_w": eps[1], "ms_per_frame_full": round(t_full * 1e3, 1), "ms_per_frame_ert": round(t_ert * 1e3, 1),
            "speedup": round(t_full / t_ert, 3), "fine_samples_evaluated": round(stats["kept"] / max(stats["tot"], 1), 4),
            "max_abs_err_rgb": float((rgb_e - rgb_f).abs().max()), "max_rel_err_depth": float(((dist_e - dist_f).abs() / dist_f.abs().clamp(min=1e-6)).max()),
            "max_abs_err_acc": float((acc_e - acc_f).abs().max()), "psnr_vs_full_db": round(psnr(mse(rgb_e, rgb_f)), 2),
            "note": "inference extension (csrc/ert.hip), not the reference's algorithm; errors = ERT render vs the un-skipped render of the same fitted model; "
                    "front-to-back mode: the skipped weights sum to <= eps_t, which bounds the acc and rgb errors"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--eps", type=float, nargs=2, default=[1e-4, 1e-4])
    ap.add_argument("--sweep", action="store_true", help="also (1e-3, 1e-3) and (1e-2, 1e-3) on the same fitted model")
    args = ap.parse_args()
    import bench
    dev = torch.device("cuda", 0)
    model = bench.build_model("bf16", dev)
    res = fit_and_render(model, args.steps, tuple(args.eps), log=lambda s: print(s, file=sys.stderr))
    print(json.dumps(res))
    if args.sweep:
        for e in ((1e-3, 1e-3), (1e-2, 1e-3)):
            print(json.dumps(fit_and_render(model, 0, e)))
        for e, g in ((1e-4, 32), (1e-4, 16), (1e-5, 32), (1e-3, 32), (1e-4, 64)):
            print(json.dumps(fit_and_render(model, 0, (e, 0.0), group=g)))


if __name__ == "__main__":
    main()
