#!/usr/bin/env python3
"""Path C (S-NeRF++ / zipnerf background) throughput on one MI355X: train step (forward + backward through the fused
featurisation, MLPs, compositing; table gradients by fp32 atomics) and forward-only rays/s, plus the achieved gather
bandwidth of the dominant kernel (zip_encode_kernel) against the HBM roofline.
Algorithmic gather bytes / ray: 7 * [64*6*8*1*T0 + 64*8*8*1*T1 + 32*10*8*4*T2] bytes."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--rays", type=int, default=16384, help="rays per GPU per step (weak scaling)")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--compute", default="fp16", help="fp16 (BASELINE config 4's fp16 MLP; static loss scale), bf16 or f32")
    ap.add_argument("--table", default="ref", choices=["ref", "f16", "f32"],
                    help="gather tables: ref = the reference's autocast policy (C = 4 NeRF table fp16, C = 1 proposal tables fp32, gridencoder/grid.py:41-44); "
                         "f16 = all three halved (narrower than the reference on the proposal levels); f32 = none")
    ap.add_argument("--log2T", type=int, default=21)
    ap.add_argument("--table-grad", default="auto", choices=["auto", "table", "f32", "f16", "bf16"])
    ap.add_argument("--table-grad-mode", default="binned", choices=["binned", "atomic"],
                    help="binned (default): records binned by destination + per-bin LDS fixed-point accumulation (no L2 atomics, bit-reproducible); "
                         "atomic: the reference's scatter with global atomics")
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--semantic", action="store_true", help="19-class semantic head on (Config.use_semantic): rendered and trained")
    ap.add_argument("--train-only", action="store_true", help="time the training loop and exit (profiling the train step)")
    ap.add_argument("--frame-only", action="store_true", help="skip the training / forward timing loops (profiling the frame)")
    ap.add_argument("--frame-chunk", type=int, default=65536, help="render_chunk_size of the measured 1920x1280 frame")
    ap.add_argument("--same-device", action="store_true", help="functional test of the N > 1 flow on a 1-GPU box")
    args = ap.parse_args()
    import torch.distributed as dist
    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    if args.same_device:
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend)
    from snerf_amd import ops, zipnerf
    from snerf_amd.trainer import ZipTrainer
    torch.manual_seed(0)
    m = zipnerf.Model(config=None, raydist_fn='power_transformation', opaque_background=True, compute=args.compute, table_dtype=args.table,
                      grid_log2_hashmap_size=args.log2T, init_std=0.1, table_grad_dtype=args.table_grad, table_grad_mode=args.table_grad_mode, device=torch.device("cuda", local),
                      use_semantic=args.semantic)
    tr = ZipTrainer(m, lr=1e-2)
    tr.broadcast_parameters(0)
    R = args.rays
    g = torch.Generator().manual_seed(1 + rank)
    # Waymo-like pinhole rays (1920x1280, focal 2050), scene rescaled so that near = 0.1, far = 10 (configs/waymo.gin)
    dev = torch.device("cuda", local)
    pix = torch.randint(0, 1920 * 1280, (R,), generator=g)
    K = torch.tensor([[2050.0, 0.0, 960.0], [0.0, 2050.0, 640.0], [0.0, 0.0, 1.0]])
    c2w = torch.eye(4)[:3].clone()
    c2w[:, 3] = torch.tensor([0.02, -0.01, 0.03])
    rays = ops.zip_pixels_to_rays((pix % 1920).int().to(dev), (pix // 1920).int().to(dev), None, torch.linalg.inv(K)[None].to(dev), c2w[None].to(dev))
    batch = dict(rays, near=torch.full((R, 1), 0.1, device=dev), far=torch.full((R, 1), 10.0, device=dev))
    batch["origins"] = batch["origins"] + (torch.randn(R, 3, generator=g) * 0.05).to(dev)
    tgt = torch.rand(R, 3, generator=g).to(dev)
    a = m.arena
    # LiDAR-like depth targets on half of the rays; the proposal levels are supervised by the fused tail's
    # anti-interlevel term, the NeRF level also by the distortion term (reference defaults)
    tdepth = (torch.rand(R, generator=g) * 4 + 0.2).to(dev)
    targets = dict(depth=tdepth, depth_mask=(torch.rand(R, generator=g) < 0.5).float().to(dev))
    if args.semantic:
        targets.update(semantic=torch.randint(0, 19, (R,), generator=g).int().to(dev))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def train_step(t):
        tr.step(batch, tgt, train_frac=0.5, rand=True, targets=targets)

    def fwd_only():
        m.scattered_rays = True                          # the bench batch is random pixels, not image rows
        with torch.no_grad():
            m(None, batch, 1.0, False)
        m.scattered_rays = False

    if args.frame_only:
        args.steps = 1
    for t in range(0 if args.frame_only else 3):
        train_step(t)
    barrier(); t0 = time.perf_counter()
    for t in range(args.steps):
        train_step(t)
    barrier(); dt_train = (time.perf_counter() - t0) / args.steps
    if world > 1:
        te = torch.tensor([dt_train], device=dev, dtype=torch.float64)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        dt_train = te.item()
    if args.train_only:
        if rank == 0:
            print(json.dumps({"path": "C train step only", "rays_per_gpu": R, "train_ms": round(dt_train * 1e3, 3), "table_grad_mode": m.table_grad_mode}))
        return
    fwd_only(); barrier(); t0 = time.perf_counter()
    for _ in range(args.steps):
        fwd_only()
    barrier(); dt_fwd = (time.perf_counter() - t0) / args.steps
    # BASELINE config 5: one whole 1920 x 1280 frame (2 457 600 rays) through render_image (ray generation on device, contiguous ray
    # blocks per rank, compute_extras like random_render_waymo_seq.py:197, one all-gather per buffer)
    import types
    W_, H_ = 1920, 1280
    pidx = torch.arange(W_ * H_, device=dev)
    fr = ops.zip_pixels_to_rays((pidx % W_).int(), (pidx // W_).int(), None, torch.linalg.inv(K)[None].to(dev), c2w[None].to(dev))
    fr.update(near=torch.full((W_ * H_, 1), 0.1, device=dev), far=torch.full((W_ * H_, 1), 10.0, device=dev))
    frame = {k: v.reshape(H_, W_, -1) for k, v in fr.items()}
    cfg = types.SimpleNamespace(render_chunk_size=args.frame_chunk, vis_num_rays=16)
    m.config = cfg
    rfn = lambda rand, b: m(rand, b, train_frac=1.0, compute_extras=True)
    zipnerf.render_image(rfn, None, frame, False, cfg)
    barrier(); t0 = time.perf_counter()
    img = zipnerf.render_image(rfn, None, frame, False, cfg)
    barrier(); dt_frame = time.perf_counter() - t0
    assert img["rgb"].shape == (H_, W_, 3) and bool(torch.isfinite(img["rgb"]).all())
    # dominant kernel: the fused featurisation (forward), timed with events around its three launches
    rec = []
    orig, orig_p = ops.zip_encode_fwd, ops.zip_encode_prop_fwd

    def timed(*x, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); orig(*x, **k); e1.record(); rec.append((e0, e1))

    def timed_p(*x, **k):                                # inference: the proposal levels' featurisation carries their MLP
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = orig_p(*x, **k); e1.record(); rec.append((e0, e1))
        return r
    ops.zip_encode_fwd, ops.zip_encode_prop_fwd = timed, timed_p
    fwd_only(); torch.cuda.synchronize()
    ops.zip_encode_fwd, ops.zip_encode_prop_fwd = orig, orig_p
    enc_ms = [e0.elapsed_time(e1) for e0, e1 in rec]
    # the fused loss tail (data + depth + anti-interlevel + distortion, values and gradients)
    tail = []
    orig_tail = ops.zip_loss_tail

    def timed_tail(*x, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = orig_tail(*x, **k); e1.record(); tail.append((e0, e1))
        return r
    ops.zip_loss_tail = timed_tail
    train_step(0); train_step(1); torch.cuda.synchronize()
    ops.zip_loss_tail = orig_tail
    tail_ms = min(e0.elapsed_time(e1) for e0, e1 in tail)
    losses = dict(zip(ops.ZIP_LOSS_NAMES + ("hash_decay",), [round(v, 6) for v in tr.last_losses.cpu().tolist()]))
    tbs = {"ref": (4, 4, 2), "f16": (2, 2, 2), "f32": (4, 4, 4)}[args.table]          # bytes per table entry channel: prop 0, prop 1, NeRF
    bytes_lvl = [R * 7 * 64 * 6 * 8 * 1 * tbs[0], R * 7 * 64 * 8 * 8 * 1 * tbs[1], R * 7 * 32 * 10 * 8 * 4 * tbs[2]]
    out = {"path": "C (zipnerf Model, waymo.gin shape: 64+64+32 intervals x 7 multisamples, grids L=6/8/10)", "n_gpus": world, "rays_per_gpu": R,
           "compute": args.compute,
           "compute_note": "BASELINE config 4 names an fp16 MLP (torch autocast): compute=fp16 runs the fp16 MFMA with fp32 accumulation and a static loss "
                           "scale folded into Adam (loss_scale below); bf16 is the same kernels at the same rate without a loss scale; tables: see `table` "
                           "(ref = fp16 where the reference halves them, i.e. only the C = 4 NeRF table)",
           "loss_scale": tr.loss_scale,
           "table_grad": args.table_grad, "table_grad_mode": m.table_grad_mode,
           "table": args.table, "train_ms": round(dt_train * 1e3, 3), "train_rays_per_s": round(world * R / dt_train, 1), "fwd_ms": round(dt_fwd * 1e3, 3),
           "fwd_rays_per_s": round(R / dt_fwd, 1), "frame_1920x1280_s": round(dt_frame, 3), "frame_outputs": sorted(k for k in img if not k.startswith("ray_")), "encode_fwd_ms_per_level": [round(x, 3) for x in enc_ms], "encode_note": "inference: proposal levels = featurisation + MLP fused",
           "encode_fwd_gather_GBps_per_level": [round(b / (ms * 1e-3) / 1e9, 1) for b, ms in zip(bytes_lvl, enc_ms)],
           "roofline": {"bound": "hbm", "kernel": "zip_encode_kernel (nerf level)", "achieved": round(bytes_lvl[2] / (enc_ms[2] * 1e-3) / 1e9, 1),
                        "peak": 8000.0, "unit": "GB/s", "frac": round(bytes_lvl[2] / (enc_ms[2] * 1e-3) / 1e9 / 8000.0, 4),
                        "note": "algorithmic (useful) gather bytes; table is Infinity-Cache resident when it fits 256 MB"},
           "loss_tail_ms": round(tail_ms, 4), "losses_last_step": losses, "params": int(a.numel)}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
