#!/usr/bin/env python3
"""Per-launch durations of every GEMM of one path-A train step (HIP events around snerf_linear_fwd / snerf_linear_wgrad), by shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from snerf_amd import ops
from snerf_amd.trainer import MipTrainer

dev = torch.device("cuda")
model = bench.build_model(sys.argv[1] if len(sys.argv) > 1 else "bf16", dev)      # bf16 (default) | bf16x3 | f32
R = int(os.environ.get("RAYS", "4096"))                                            # RAYS=512: the per-rank batch of an 8-GPU strong-scaling run
rays = bench.synth_rays(R, 0, dev)
tgt = torch.rand(R, 3, device=dev); depth = torch.rand(R, device=dev) * 20 + 2; conf = torch.ones(R, device=dev)
tr = MipTrainer(model, lr=5e-4, proposal_loss=True)
for _ in range(3):
    tr.step(rays, tgt, depth, conf)
rec = []
of, ow = ops.linear_fwd, ops.linear_wgrad
def tf(A, W, bias, Y, K, n_store, act, dt, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); of(A, W, bias, Y, K, n_store, act, dt, **kw); e1.record()
    rec.append(("NT act=%d%s" % (act, " +colsum" if kw.get("colsum") is not None else ""), A.shape[0], W.shape[0], K, e0, e1))
def tw(dZ, X, dW, n_valid, k_valid, dt, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ow(dZ, X, dW, n_valid, k_valid, dt, **kw); e1.record()
    rec.append(("TN", dZ.shape[0], n_valid, k_valid, e0, e1))
ops.linear_fwd, ops.linear_wgrad = tf, tw
fused = []                                           # the fused launches of the step (not GEMM launches): timed beside them
def wrap(name):
    orig = getattr(ops, name)
    def timed(*a, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = orig(*a, **kw); e1.record()
        fused.append((name, e0, e1))
        return r
    setattr(ops, name, timed)
    return orig
saved = {n: wrap(n) for n in ("fcolour_fwd", "fcolour_bwd", "fchain_bwd", "fmlp_proposal_train_fwd", "mip_encode", "mip_composite_fwd", "mip_composite_bwd", "mip_resample", "adam_step")}
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
tr.step(rays, tgt, depth, conf)
ev[1].record()
torch.cuda.synchronize()
ops.linear_fwd, ops.linear_wgrad = of, ow
for n, f in saved.items():
    setattr(ops, n, f)
print(f"instrumented step {ev[0].elapsed_time(ev[1]):.2f} ms")
for n, e0, e1 in fused:
    print(f"fused/other {n:28s} {e0.elapsed_time(e1) * 1e3:8.1f} us")
rows = [(k, M, N, K, e0.elapsed_time(e1)) for k, M, N, K, e0, e1 in rec]
tot = sum(r[4] for r in rows)
print(f"{len(rows)} GEMM launches, {tot:.2f} ms")
for k, M, N, K, ms in sorted(rows, key=lambda r: -r[4]):
    print(f"{k:18s} M={M:7d} N={N:5d} K={K:5d}  {ms * 1e3:8.1f} us  {2.0 * M * N * K / ms / 1e9:7.1f} TFLOP/s")
