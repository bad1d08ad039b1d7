#!/usr/bin/env python3
"""Early ray termination + sample compaction on a 1600x900 frame (inference extension, csrc/ert.hip).
A random-init network is nearly transparent, so nothing terminates (bench.py --ert: 95-99 % of the samples survive); to show the
mechanism this tool makes the SAME network opaque by shifting the bias of both density heads (a fog-like medium: transmittance
drops below 1e-3 within the first ~10 proposal intervals), which is the regime of trained street scenes where rays end at
surfaces.  Reports frame time, surviving samples and PSNR against the full evaluation of the same scene."""
import argparse, json, math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shift", type=float, default=6.0, help="added to proposal.density_layer.bias and mlp.density_layer.bias")
    ap.add_argument("--chunk", type=int, default=32768)
    ap.add_argument("--rows", type=int, default=900)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    model = bench.build_model("bf16", dev)
    with torch.no_grad():
        p = dict(model.named_parameters())
        p["proposal.density_layer.bias"] += args.shift
        p["mlp.density_layer.bias"] += args.shift
    model.arena.bump()
    W = 1600
    npix = args.rows * W

    def frame(ert):
        kept = tot = 0
        outs = []
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with torch.no_grad():
            for i in range(0, npix, args.chunk):
                fr = bench.frame_rays(i, min(args.chunk, npix - i), dev)
                ret = model(fr, False, False, 0., ert=ert)
                if ert is not None:
                    kept += model.last_ert_rows[0]; tot += model.last_ert_rows[1]
                outs.append(ret[1][0])
        torch.cuda.synchronize()
        return time.perf_counter() - t0, torch.cat(outs, 0), (kept / tot if tot else 1.0)

    frame(None)
    t_full, img, _ = frame(None)
    res = {"scene": f"bench network with density-head biases + {args.shift} (opaque medium)", "rays": npix, "full_ms": round(t_full * 1e3, 1), "ert": []}
    for eps in ((1e-4, 1e-5), (1e-3, 1e-4), (1e-2, 1e-3)):
        frame(eps)
        t, im, frac = frame(eps)
        mse = float(((im - img) ** 2).mean())
        res["ert"].append({"eps_t": eps[0], "eps_w": eps[1], "ms": round(t * 1e3, 1), "speedup": round(t_full / t, 2), "fine_samples_evaluated": round(frac, 4),
                           "psnr_vs_full_db": float("inf") if mse == 0 else round(-10 * math.log10(mse), 2)})
    print(json.dumps(res))


if __name__ == "__main__":
    main()
