#!/usr/bin/env python3
"""Per-launch table of the NT/TN GEMMs of one path-A train step at the bench shape (HIP events around every launch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench


def main():
    from snerf_amd import ops
    from snerf_amd.trainer import MipTrainer
    dev = torch.device("cuda", 0)
    model = bench.build_model("bf16", dev)
    tr = MipTrainer(model, lr=5e-4)
    n = 4096
    rays = bench.synth_rays(n, 1000, dev)
    g = torch.Generator().manual_seed(2000)
    tgt = torch.rand(n, 3, generator=g).to(dev)
    depth = torch.where(torch.rand(n, generator=g) < 0.5, torch.rand(n, generator=g) * 78 + 2, torch.zeros(n)).to(dev)
    conf = torch.rand(n, generator=g).to(dev)
    for _ in range(3):
        tr.step(rays, tgt, depth, conf)
    rec = []
    of, ow = ops.linear_fwd, ops.linear_wgrad

    def lf(A, W, bias, Y, K, n_store, act, dt, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); of(A, W, bias, Y, K, n_store, act, dt, **kw); e1.record()
        rec.append(("NT act=%d%s" % (act, " cs" if kw.get("colsum") is not None else ""), A.shape[0], W.shape[0], K, e0, e1))

    def lw(dZ, X, dW, nv, kv, dt, variant=0):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ow(dZ, X, dW, nv, kv, dt, variant=variant); e1.record()
        rec.append(("TN", dZ.shape[0], dZ.shape[1], X.shape[1], e0, e1))
    ops.linear_fwd, ops.linear_wgrad = lf, lw
    tr.step(rays, tgt, depth, conf)
    torch.cuda.synchronize()
    ops.linear_fwd, ops.linear_wgrad = of, ow
    tot = 0.0
    print(f"{'kind':14s} {'M':>8s} {'N':>6s} {'K':>6s} {'ms':>8s} {'TFLOP/s':>9s}")
    for kind, M, N, K, e0, e1 in rec:
        ms = e0.elapsed_time(e1)
        tot += ms
        print(f"{kind:14s} {M:8d} {N:6d} {K:6d} {ms:8.3f} {2.0 * M * N * K / ms / 1e9:9.1f}")
    print(f"total {tot:.2f} ms over {len(rec)} launches")


if __name__ == "__main__":
    main()
