#!/usr/bin/env python3
"""Per-level cost of the fused featurisation forward / backward (cumulative over the first k levels) -- where do the
table-gradient atomics hurt?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from snerf_amd import ops, zipnerf


def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    m = zipnerf.Model(config=None, raydist_fn='power_transformation', opaque_background=True, compute="bf16", table_dtype="f16", init_std=0.1)
    R, S = 16384, 32
    g = torch.Generator().manual_seed(1)
    d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g) * torch.tensor([0.3, 0.2, 1.0]), dim=-1)
    up = torch.tensor([0.0, 1.0, 0.0]).expand(R, 3)
    bx = torch.nn.functional.normalize(torch.cross(d, up, dim=-1), dim=-1); by = torch.nn.functional.normalize(torch.cross(d, bx, dim=-1), dim=-1)
    o = torch.randn(R, 3, generator=g) * 0.05
    radii = torch.full((R,), 2.0 / 2050 / 12 ** 0.5)
    near, far = torch.full((R,), 0.1), torch.full((R,), 10.0)
    sd = torch.cat([torch.zeros(R, 1), torch.ones(R, 1)], -1).cuda(); w = torch.ones(R, 1).cuda()
    u = torch.linspace(1 / 64, 1 - 1 / 64 - 1e-7, S).cuda()
    _, tdist = ops.zip_resample(sd, w, u, S, near.cuda(), far.cuda(), 0.0, False, 1.0)
    args = [tdist, o.cuda(), d.cuda(), radii.cuda(), bx.cuda(), by.cuda(), None]
    for lvl in (2, 0):
        e = m.encs[lvl]
        tab = m._table(lvl)
        Fb = torch.zeros(R * S, 64, dtype=torch.bfloat16, device="cuda")
        dF = (torch.randn(R * S, 64, generator=g) * 0.01).bfloat16().cuda()
        gt = torch.zeros(e.rows, e.C, device="cuda")
        prev_f = prev_b = 0.0
        for k in range(1, e.L + 1):
            tf = timeit(lambda: ops.zip_encode_fwd(*args, tab, m.dev_offsets[lvl], m.dev_sizes[lvl], Fb, k, e.C, 7, 3, e.Sl, e.H, 0.35))
            tb = timeit(lambda: ops.zip_encode_bwd(*args, m.dev_offsets[lvl], m.dev_sizes[lvl], dF, gt, k, e.C, 7, 3, e.Sl, e.H, 0.35, min(k, e.lds_levels), e.lds_cells, sum(-(-int(x) // e.lds_cells) for x in (e.offsets[1:min(k, e.lds_levels) + 1] - e.offsets[:min(k, e.lds_levels)]))))
            print(f"enc {lvl} (C={e.C}) level {k - 1} res {int(e.res[k - 1])}: fwd +{tf - prev_f:7.3f} ms   bwd +{tb - prev_b:8.3f} ms", flush=True)
            prev_f, prev_b = tf, tb


if __name__ == "__main__":
    main()
