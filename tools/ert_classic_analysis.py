#!/usr/bin/env python3
"""Would early ray termination pay in path B's fine pass?  (VERDICT r4 item 5 / north_star's "early ray termination and sample
compaction".)  The classic renderer's fine network evaluates the sorted union of the 64 uniform coarse positions and the 128 importance
samples (render.py:380-389) -- the uniform positions behind the first surface are the one place on any of the three paths where
samples sit behind an opaque hit by construction.

The two classic NeRFs (8 x 256) are fitted to the analytic street scene of tools/ert_scene.py through the drop-in render_rays +
torch.optim.Adam, then a window of the 1600 x 900 frame is rendered and, per fine sample, two skip rules are evaluated on the
un-skipped render's own densities:
  exact    a sample is skippable when the FINE transmittance in front of it is <= eps (front-to-back termination: the skipped weights
           sum to <= eps -- the bound snerf_ert_f2b_step guarantees on path A);
  coarse   a sample is skippable when the COARSE pass's transmittance at its depth is <= eps (free: no extra evaluation, but the
           bound holds only as far as the two networks agree) -- with the acc error it would have caused.
-> the kept fraction of the 192 fine evaluations and the best-case frame speed-up 256 / (64 + 192 kept) (no selection / compaction cost).

    python tools/ert_classic_analysis.py [--steps 400] [--eps 1e-4]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch

import ert_scene as sc


def rows_of(first, n, dev):
    r = sc.rays_of(None, first, n, dev)
    return torch.cat([r.origins, r.directions, r.near, r.far, r.viewdirs], -1), r


def fit_and_measure(steps=400, eps_list=(1e-4, 1e-3, 1e-2), rows_n=96, lindisp=False, groups=(48, (96, 16)), ert_eps=1e-4, dev=None, chunk=524288, row0=400, stages=False):
    """-> dict: the analysis above + the MEASURED window renders with render_rays(ert=(ert_eps, G)) for G in `groups` against the plain
    render of the same fitted networks (time, evaluated fraction, errors).  bench.py's `path_b_ert` leg calls this with fewer steps."""
    import types
    args = types.SimpleNamespace(steps=steps, eps=list(eps_list), rows=rows_n, lindisp=lindisp)
    from snerf_amd import classic
    dev = torch.device("cuda", 0) if dev is None else dev
    torch.manual_seed(0)
    mk = lambda: classic.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=4, skips=[4], use_viewdirs=True, compute="bf16", device=dev)
    coarse, fine = mk(), mk()
    e, _ = classic.get_embedder(10, 0)
    ed, _ = classic.get_embedder(4, 0)
    q = classic.make_network_query_fn(e, ed, netchunk=1 << 30)
    opt = torch.optim.Adam(list(coarse.parameters()) + list(fine.parameters()), lr=5e-4)
    g = torch.Generator(device="cpu").manual_seed(0)
    n = 8192
    t0 = time.perf_counter()
    for it in range(args.steps):
        pix = torch.randint(0, sc.H * sc.W, (n,), generator=g)
        r = sc.rays_of(torch.stack([pix // sc.W, pix % sc.W], -1).int().to(dev), 0, n, dev)
        rows = torch.cat([r.origins, r.directions, r.near, r.far, r.viewdirs], -1)
        rgb, t_hit = sc.analytic_scene(r.origins, r.directions)
        for gp in opt.param_groups:
            gp["lr"] = 5e-4 * (0.1 ** (it / args.steps))
        opt.zero_grad(set_to_none=False)
        # (the 64 coarse positions over [1.8, 110]; the street scene's surfaces are at 12 .. 100 m)
        out = classic.render_rays(rows, coarse, q, 64, lindisp=args.lindisp, perturb=1.0, N_importance=128, network_fine=fine, raw_noise_std=0.0)
        hit = t_hit > 0
        loss = ((out["rgb_map"] - rgb) ** 2).mean() + ((out["rgb0"] - rgb) ** 2).mean()
        loss = loss + 0.05 * ((1 / out["depth_map"].clamp(min=1.0) - 1 / t_hit.clamp(min=1.0)).abs() * hit).mean()
        loss.backward()
        opt.step()
        coarse.arena.bump(); fine.arena.bump()
    torch.cuda.synchronize()
    t_fit = time.perf_counter() - t0
    first, nw = row0 * sc.W, args.rows * sc.W
    rows, r = rows_of(first, nw, dev)
    tgt, t_hit = sc.analytic_scene(r.origins, r.directions)
    outs = []
    with torch.no_grad():
        for a in range(0, nw, 32768):
            o = classic.render_rays(rows[a:a + 32768], coarse, q, 64, lindisp=args.lindisp, perturb=0.0, N_importance=128, network_fine=fine, retraw=True, return_inds=True)
            outs.append({k: o[k] for k in ("rgb_map", "acc_map", "raw", "z_vals_fine", "weights", "z_vals_map")})
    cat = lambda k: torch.cat([o[k] for o in outs], 0)
    rgb, raw, z, w0, z0 = cat("rgb_map"), cat("raw"), cat("z_vals_fine"), cat("weights"), cat("z_vals_map")
    mse = float(((rgb - tgt) ** 2).mean())
    dn = r.directions.norm(dim=-1, keepdim=True)
    dists = torch.cat([z[:, 1:] - z[:, :-1], torch.full_like(z[:, :1], 1e10)], -1) * dn
    alpha = 1 - torch.exp(-torch.relu(raw[..., 3]) * dists)
    T = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1 - alpha + 1e-10], -1), -1)[:, :-1]       # transmittance in front of sample i
    wf = alpha * T
    # the coarse pass's transmittance at the depth of every fine sample
    acc0 = torch.cumsum(w0, -1)                                     # sum of coarse weights up to and including sample j = 1 - T after it
    idx = (torch.searchsorted(z0.contiguous(), z.contiguous(), right=True) - 1).clamp(min=0)
    Tc = 1 - torch.gather(torch.cat([torch.zeros_like(acc0[:, :1]), acc0], -1), -1, idx)   # coarse T in front of coarse sample idx (conservative)
    res = {"lindisp": bool(args.lindisp), "fit_steps": args.steps, "fit_s": round(t_fit, 1), "fit_psnr_db": round(-10 * torch.log10(torch.tensor(mse)).item(), 2), "window_rays": nw,
           "opaque_rays_fraction": round(float((cat("acc_map") > 0.99).float().mean()), 4), "rules": {}}
    for eps in args.eps:
        skip_exact = T <= eps
        skip_coarse = Tc <= eps
        lost = (wf * skip_coarse).sum(-1)
        for name, sk in (("exact", skip_exact), ("coarse", skip_coarse)):
            kept = 1 - float(sk.float().mean())
            res["rules"][f"{name}_eps_{eps:g}"] = {"kept_fraction_of_fine_evaluations": round(kept, 4),
                                                  "best_case_frame_speedup": round(256 / (64 + 192 * kept), 3)}
        res["rules"][f"coarse_eps_{eps:g}"]["max_abs_err_acc_if_skipped"] = float(lost.max())
        res["rules"][f"coarse_eps_{eps:g}"]["p999_abs_err_acc_if_skipped"] = float(torch.quantile(lost, 0.999))
    # ---- measured: the window rendered plain and with the front-to-back termination
    evals = [0, 0]

    def render(ert):
        parts = []
        evals[0] = evals[1] = 0
        with torch.no_grad():
            torch.cuda.synchronize(); t1 = time.perf_counter()
            for a in range(0, nw, chunk):
                o = classic.render_rays(rows[a:a + chunk], coarse, q, 64, lindisp=args.lindisp, perturb=0.0, N_importance=128, network_fine=fine, ert=ert)
                parts.append((o["rgb_map"], o["acc_map"], o["depth_map"]))
                if "ert_evals" in o:
                    evals[0] += int(o["ert_evals"][0, 0]); evals[1] += int(o["ert_evals"][0, 1])
            torch.cuda.synchronize()
        return [torch.cat(x, 0) for x in zip(*parts)], time.perf_counter() - t1
    render(None)
    (rgb_p, acc_p, dep_p), t_plain = render(None)
    res["window_ms_plain"] = round(t_plain * 1e3, 2)
    res["render_chunk"] = chunk
    res["measured"] = {}
    for G in groups:
        render((ert_eps, G))
        (rgb_e, acc_e, dep_e), t_e = render((ert_eps, G))
        res["measured"][f"eps_{ert_eps:g}_G{G if isinstance(G, int) else '_'.join(map(str, G))}"] = {
            "window_ms": round(t_e * 1e3, 2), "speedup": round(t_plain / t_e, 3),
            "fine_evaluations_kept": round(evals[0] / max(evals[1], 1), 4),
            "max_abs_err_rgb": float((rgb_e - rgb_p).abs().max()), "max_abs_err_acc": float((acc_e - acc_p).abs().max()),
            "max_abs_err_depth": float((dep_e - dep_p).abs().max())}
    if stages:
        # where the ERT window's time goes (synchronised timers around the three stages of a group; the syncs cost a little themselves)
        from snerf_amd import ops
        acc = {"points": 0.0, "network": 0.0, "step": 0.0}
        def timed(name, fn):
            def w(*a, **k):
                torch.cuda.synchronize(); t = time.perf_counter()
                r = fn(*a, **k)
                torch.cuda.synchronize(); acc[name] += time.perf_counter() - t
                return r
            return w
        keep = (ops.classic_ert_points, ops.classic_ert_step)
        ops.classic_ert_points, ops.classic_ert_step = timed("points", keep[0]), timed("step", keep[1])
        q0 = q
        q = timed("network", q0)                      # (coarse pass + every fine group)
        _, t_all = render((ert_eps, groups[-1]))
        ert_acc = dict(acc)
        for k in acc:
            acc[k] = 0.0
        _, t_pl = render(None)
        res["stages_ms_plain"] = {"network": round(acc["network"] * 1e3, 2), "window": round(t_pl * 1e3, 2)}
        acc.update(ert_acc)
        ops.classic_ert_points, ops.classic_ert_step = keep
        q = q0
        res["stages_ms"] = {k: round(v * 1e3, 2) for k, v in acc.items()}
        res["stages_ms"]["window"] = round(t_all * 1e3, 2)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--eps", type=float, nargs="+", default=[1e-4, 1e-3, 1e-2])
    ap.add_argument("--rows", type=int, default=96)
    ap.add_argument("--chunk", type=int, default=524288)
    ap.add_argument("--row0", type=int, default=400)
    ap.add_argument("--lindisp", action="store_true", help="coarse positions uniform in disparity (render_rays(lindisp=True)); default: uniform in depth, the reference's default")
    ap.add_argument("--groups", default="48;96,16", help="group schedules to measure, ';'-separated (a schedule: ','-separated sizes, the last repeats)")
    ap.add_argument("--stages", action="store_true", help="also time the stages of the front-to-back pass (points / network / step)")
    args = ap.parse_args()
    torch.cuda.set_device(0)
    print(json.dumps(fit_and_measure(args.steps, args.eps, args.rows, args.lindisp, chunk=args.chunk, row0=args.row0, stages=args.stages,
                                     groups=tuple((lambda v: v[0] if len(v) == 1 else v)(tuple(int(x) for x in g.split(','))) for g in args.groups.split(';')))))


if __name__ == "__main__":
    main()
