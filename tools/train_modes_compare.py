#!/usr/bin/env python3
"""Do the compute modes TRAIN alike?  The path-A model is fitted to the analytic street scene (tools/ert_scene.py: RGB + depth supervised, the
reference's losses through MipTrainer) from the same initial weights, with the same pixel batches, in every compute mode; 40 held-out image rows are
then rendered by each fitted model IN ITS OWN MODE and scored against the analytic scene (PSNR, median relative depth error on the hit rays).
Beside it: how far each mode's parameters have drifted from the exact-fp32 run's (relative L2 of the whole arena) -- Adam turns the sign of a
near-zero gradient into a full step, so trajectories separate even in fp32-class arithmetic; what matters is where they arrive.

    python tools/train_modes_compare.py [--steps 300] [--modes f32,f16f8,bf16x3_fwd,fp16,bf16]"""
import argparse
import json
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ert_scene  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--modes", default="f32,f16f8,bf16x3_fwd,fp16,bf16")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    sd0 = None
    rows, row0 = 40, 430
    n = rows * ert_scene.W
    rays = ert_scene.rays_of(None, row0 * ert_scene.W, n, dev)
    tgt_rgb, tgt_t = ert_scene.analytic_scene(rays.origins, rays.directions)
    hit = tgt_t > 0
    res, flats = {}, {}
    for mode in args.modes.split(","):
        m = bench.build_model(mode, dev, seed=1)
        if sd0 is None:
            sd0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
        m.load_state_dict(sd0)
        t_fit = ert_scene.fit(m, args.steps, seed=0)
        with torch.no_grad():
            parts = [m(type(rays)(*[r[a:a + 16384] for r in rays]), False, False, 0.) for a in range(0, n, 16384)]
        rgb = torch.cat([p[1][0] for p in parts], 0)
        dist = torch.cat([p[1][1] for p in parts], 0).reshape(-1)
        mse = float(((rgb - tgt_rgb) ** 2).mean())
        flats[mode] = m.arena.flat.detach().clone()
        res[mode] = {"fit_s": round(t_fit, 1), "ms_per_step": round(t_fit / args.steps * 1e3, 1), "psnr_vs_scene_db": round(-10 * math.log10(mse), 2),
                     "depth_median_rel_err": round(float(((dist - tgt_t).abs() / tgt_t.clamp(min=1))[hit].median()), 4),
                     "depth_p90_rel_err": round(float(torch.quantile(((dist - tgt_t).abs() / tgt_t.clamp(min=1))[hit].float(), 0.9)), 4)}
        del m
        torch.cuda.empty_cache()
    ref = flats.get("f32")
    if ref is not None:
        for mode, f in flats.items():
            res[mode]["param_rel_l2_vs_f32_run"] = round(float((f - ref).norm() / ref.norm()), 4)
    print(json.dumps({"steps": args.steps, "rays_per_step": 4096, "eval_rays": n, "modes": res}))
    print(f"\n{'mode':12s} {'ms/step':>8s} {'PSNR dB':>8s} {'depth med':>10s} {'depth p90':>10s} {'params vs f32 run':>18s}")
    for mode, v in res.items():
        print(f"{mode:12s} {v['ms_per_step']:8.1f} {v['psnr_vs_scene_db']:8.2f} {v['depth_median_rel_err']:10.4f} {v['depth_p90_rel_err']:10.4f} {v.get('param_rel_l2_vs_f32_run', float('nan')):18.4f}")


if __name__ == "__main__":
    main()
