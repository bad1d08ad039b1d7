#!/usr/bin/env python3
"""Benchmark of the S-NeRF background hot path on MI355X (contract: see DESIGN.md "Measurement").

    python bench.py --gpus N --steps K --warmup W        (N > 1 without a launcher: re-executes itself under torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one training step of the live S-NeRF renderer (path A, MipNerfModel) on one batch of synthetic
nuScenes-like rays: sample -> IPE encode -> proposal MLP -> composite -> resample -> IPE encode -> NeRF MLP
(8 x 1024) -> composite, RGB + depth loss, backward through every kernel, gradient all-reduce (N > 1) and fused Adam.
Workload = BASELINE.json configs[1]: 64 proposal + 128 fine network evaluations per ray (192 spp), hidden 1024,
bf16 MFMA with fp32 accumulation, 4096 rays per GPU per step (weak scaling: per-GPU work is fixed).
Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import math
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # this host's driver only supports dmabuf IPC (RCCL across processes needs it); before torch loads the runtime

import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

S0, P1, HIDDEN, RGB_LAYERS = 64, 129, 1024, 3      # 64 proposal intervals + 128 fine intervals = 192 evals / ray
MAC_PROP, MAC_NERF = 221440, 8753920               # MACs per sample
PEAK_BF16_TFLOPS = 2500.0                          # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)


def synth_rays(n, seed, device, H=900, W=1600, focal=1266.0, near=1.8, far=110.0):
    """nuScenes-like pinhole rays (workload M2)."""
    rng = np.random.default_rng(seed)
    pix = rng.choice(H * W, size=n, replace=False) if n <= H * W else rng.integers(0, H * W, size=n)
    return rays_from_pixels(pix, rng.normal(0.0, 0.1, size=(n, 3)), device, H, W, focal, near, far)


def rays_from_pixels(pix, origins, device, H=900, W=1600, focal=1266.0, near=1.8, far=110.0):
    from snerf_amd.mipnerf import Rays
    j, i = (pix // W).astype(np.float64), (pix % W).astype(np.float64)
    th = 0.3
    R = np.array([[math.cos(th), 0.0, math.sin(th)], [0.0, 1.0, 0.0], [-math.sin(th), 0.0, math.cos(th)]])
    dirs = lambda ii, jj: np.stack([(ii - W * 0.5 + 0.5) / focal, -(jj - H * 0.5 + 0.5) / focal, -np.ones_like(ii)], -1) @ R.T
    d = dirs(i, j)
    dx = np.where((j + 1 <= H - 1)[:, None], dirs(i, np.minimum(j + 1, H - 1)) - d, d - dirs(i, np.maximum(j - 1, 0)))
    radii = np.sqrt((dx ** 2).sum(-1, keepdims=True)) * 2.0 / math.sqrt(12.0)
    ones = np.ones((len(pix), 1))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)
    return Rays(t(origins), t(d), t(d / np.linalg.norm(d, axis=-1, keepdims=True)), t(radii), t(ones), t(ones * near), t(ones * far), t(ones * 0))


def relaunch_under_torchrun(n):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks ourselves (one process per GPU over RCCL)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def frame_rays(first, n, device, H=900, W=1600, focal=1266.0, near=1.8, far=110.0):
    """Rays of n consecutive pixels of the frame, generated on the device (snerf_pinhole_rays: the reference's
    get_rays_single_img, sample_utils.py:286-345) -- same camera as rays_from_pixels."""
    from snerf_amd import ops
    from snerf_amd.mipnerf import Rays
    th = 0.3
    pose = np.array([[math.cos(th), 0.0, math.sin(th), 0.0], [0.0, 1.0, 0.0, 0.0], [-math.sin(th), 0.0, math.cos(th), 0.0]], dtype=np.float32)
    o, d, v, r, nr, fr = ops.pinhole_rays(None, first, n, W, H, pose, W * 0.5, H * 0.5, focal, focal, False, near, far, device)
    ones = torch.ones_like(r)
    return Rays(o, d, v, r, ones, nr, fr, ones * 0)


def build_model(compute, device, seed=0):
    from snerf_amd.mipnerf import MipNerfModel
    torch.manual_seed(seed)
    return MipNerfModel(n_samples=S0, N_fine=P1, no_warp_sample=0, ray_shape="cone", fn=1, radius=3., transform_idx=0, real=True,
                        rgb_layer=RGB_LAYERS, hidden_layer=HIDDEN, density_noise=0., max_deg_point=16, proposal_hidden_layer=256,
                        proposal_loss=True, compute=compute, device=device)


def roofline_traffic(compute, variant):
    """HBM bytes of ONE launch of the dominant kernel at its largest layer shape, from the committed PMC measurement
    (profiles/roofline_traffic.json: FETCH_SIZE x2 + WRITE_SIZE of separate rocprofv3 --pmc passes) -- reported only while the kernel's
    source still hashes to what was measured (tools/kernel_hash.py), so an edit of the tile walk cannot leave a stale number in the line."""
    if compute != "bf16" or variant != 8:
        return None, "no PMC measurement for this kernel selection"
    try:
        sys.path.insert(0, os.path.join(REPO, "tools"))
        from kernel_hash import kernel_hash
        d = json.load(open(os.path.join(REPO, "profiles", "roofline_traffic.json")))
        if d["kernel_sha256"] != kernel_hash(d["kernel"]):
            return None, f"STALE: {d['kernel']} changed since {d['source']} was measured; rerun tools/pmc_gemm_traffic.sh and tools/kernel_hash.py --update"
        return d["fetch_bytes"] + d["write_bytes"], f"{d['source']} (per launch at {d['launch']}; algorithmic {d['algorithmic_bytes']:.3g} B; kernel sha256 {d['kernel_sha256'][:12]})"
    except (OSError, KeyError, ValueError) as e:
        return None, f"unavailable: {e}"


def measure_gemm_kernel(trainer, rays, tgt, depth, conf):
    """One instrumented step: HIP events around every launch of the dominant kernel (the NT MFMA GEMM used by the forward and
    data-gradient layers that are not fused), on the stream the kernels run on (torch's current stream).
    -> (launches, kernel ms, padded FLOPs, algorithmic FLOPs).  Algorithmic = 2 x rows x the number of REAL weights the launch multiplies
    with: the packed operand W [N_pad, K_pad] holds exact zeros wherever the tile layout pads (extra rows of narrow heads, the K padding of
    the 96 / 1120 / 1051-wide inputs, encoding columns whose data gradient is not needed), so nnz(W[:, :K]) is the launch's unpadded
    N x K whatever the layer -- counted after the step from the operands the launches actually received."""
    from snerf_amd import ops
    rec = []
    orig = ops.linear_fwd

    def timed(A, W, bias, Y, K, n_store, act, dt, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        orig(A, W, bias, Y, K, n_store, act, dt, **kw)
        e1.record()
        rec.append((e0, e1, A.shape[0], K, W, 3 if dt == ops.BF16X3 else 1))
    ops.linear_fwd = timed
    try:
        trainer.step(rays, tgt, depth, conf)
        torch.cuda.synchronize()
    finally:
        ops.linear_fwd = orig
    ms = sum(e0.elapsed_time(e1) for e0, e1, *_ in rec)
    padded = sum(2.0 * M * K * W.shape[0] * sp for _, _, M, K, W, sp in rec)
    nnz = {}
    for _, _, M, K, W, sp in rec:                       # (an optimiser step does not create or remove zeros: counting afterwards is exact)
        key = (W.data_ptr(), K)
        if key not in nnz:
            nnz[key] = int((W[:, :K] != 0).sum())
    alg = sum(2.0 * M * nnz[(W.data_ptr(), K)] * sp for _, _, M, K, W, sp in rec)
    return len(rec), ms, padded, alg


def cpu_baseline(n_rays, model_sd, rays, seed=0):
    """Reference algorithm on the host CPU cores (oracle/ = CPU restatement pinned to the imported reference):
    forward + backward of the same network shape on a bounded sample; also returns the oracle's rgb for a parity read-out."""
    from oracle import mip as om
    sd = {k: v.detach().float().cpu() for k, v in model_sd.items()}
    rc = {k: getattr(rays, k)[:n_rays].detach().float().cpu() for k in rays._fields}
    g = torch.Generator().manual_seed(seed)
    tgt = torch.rand(n_rays, 3, generator=g)
    pr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    t0 = time.perf_counter()
    ret = om.mipnerf_forward(pr, rc, S0, P1)
    dt_fwd = time.perf_counter() - t0
    loss = ((ret[1][0] - tgt) ** 2).mean() + 0.2 * (1.0 / ret[1][1]).mean() + 0.04 * (1.0 / ret[0][1]).mean()
    loss.backward()
    dt_s = time.perf_counter() - t0
    return n_rays / dt_s, dt_s, ret[1][0].detach(), ret[1][1].detach(), n_rays / dt_fwd


def eager_baseline(model_sd, rays, n_rays, steps, bf16):
    """The same train step as plain PyTorch-ROCm eager ops on the SAME GPU (the oracle's torch restatement of the reference
    model moved to cuda; its numpy resampler replaced by the equivalent torch ops), autograd backward + torch.optim.Adam.
    A reported baseline for BASELINE.json's ">= 2x over the PyTorch-ROCm eager path" target, never part of the product."""
    from oracle import mip as om
    dev = rays.origins.device

    from oracle.eager import mip_resample_torch as resample_torch
    saved, om.warp_resample_s = om.warp_resample_s, resample_torch
    prev = torch.get_default_device() if hasattr(torch, "get_default_device") else None
    torch.set_default_device(dev)
    try:
        pr = {k: v.detach().float().clone().requires_grad_(True) for k, v in model_sd.items()}
        rc = {k: getattr(rays, k)[:n_rays].detach().float() for k in rays._fields}
        tgt = torch.rand(n_rays, 3, device=dev)
        opt = torch.optim.Adam(list(pr.values()), lr=5e-4)

        def step():
            opt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf16):
                ret = om.mipnerf_forward(pr, rc, S0, P1)
            loss = ((ret[1][0].float() - tgt) ** 2).mean() + 0.2 * (1.0 / ret[1][1]).mean() + 0.04 * (1.0 / ret[0][1]).mean()
            loss.backward()
            opt.step()

        step(); step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dt_s = (time.perf_counter() - t0) / steps
        # forward only (what eval.py's render_image runs per chunk, models.py:328-360): the eager side of north_star's frame target
        def fwd():
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf16):
                om.mipnerf_forward(pr, rc, S0, P1)
        fwd()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fwd()
        torch.cuda.synchronize()
        dt_f = (time.perf_counter() - t0) / steps
    finally:
        om.warp_resample_s = saved
        torch.set_default_device(prev if prev is not None else "cpu")
    return n_rays / dt_s, dt_s * 1e3, n_rays / dt_f


def dropin_autograd_leg(model_sd, rays, tgt, depth, conf, device, steps):
    """What an UNMODIFIED s-nerf/train.py executes per step with the drop-in module (train.py:112-115, 150-221): `model(rays)` through
    the autograd Function, the losses as torch expressions (RgbLoss loss_factory.py:5-11, the confidence-weighted disparity DepthLoss on
    both levels :26-37 / confidence.py:209-224), `loss.backward()`, `torch.optim.Adam.step()` -- no fused loss tail, no fused Adam, no
    flat-arena shortcuts.  The product path only; timed like the headline."""
    m = build_model("bf16", device)
    m.load_state_dict({k: v.clone() for k, v in model_sd.items()})
    opt = torch.optim.Adam(m.parameters(), lr=5e-4)
    mask = depth > 0

    def step():
        opt.zero_grad(set_to_none=True)
        ret = m(rays, True, False, 0.)
        loss = ((ret[1][0] - tgt) ** 2).mean()
        for lvl, mult in ((1, 1.0), (0, 0.2)):
            d = (1.0 / ret[lvl][1] - 1.0 / depth.clamp(min=1e-6)).abs() * conf
            loss = loss + 0.2 * mult * torch.where(mask, d, torch.zeros_like(d)).sum() / mask.sum().clamp(min=1)
        loss.backward()
        opt.step()
        return loss
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return dt, float(loss)


def _timeit(fn, steps, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def _timed_calls(module, names):
    """Wrap `module.<name>` so that every call is bracketed by events on torch's current stream (the stream the kernels are launched
    on); -> (records {name: [(e0, e1)]}, restore())."""
    rec, saved = {n: [] for n in names}, {n: getattr(module, n) for n in names}

    def wrap(n):
        def f(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); r = saved[n](*a, **k); e1.record(); rec[n].append((e0, e1))
            return r
        return f
    for n in names:
        setattr(module, n, wrap(n))

    def restore():
        for n in names:
            setattr(module, n, saved[n])
    return rec, restore


def counter_bytes(key):
    """PMC-measured HBM bytes per launch of a path-B / path-C kernel from the committed measurement (profiles/roofline_traffic_paths.json,
    written by tools/pmc_paths.sh + tools/pmc_summary.py on the GPU box; scale factors per kernel class from the gather calibration probe)."""
    try:
        d = json.load(open(os.path.join(REPO, "profiles", "roofline_traffic_paths.json")))
        e = d[key]
        return e["bytes"], f"{e['source']} ({e['how']})"
    except (OSError, KeyError, ValueError) as e:
        return None, f"unavailable: {e}"


def path_c_leg(device, n_rays=65536, steps=10, compute="fp16", also=("bf16",), scale_factor=0.01):
    """BASELINE configs 4-5 (S-NeRF++ / zipnerf background, waymo.gin shape: 64 + 64 + 32 intervals x 7 multisamples, hash grids
    L = 6 / 8 / 10, T = 2^21): ZipTrainer train step at 65 536 rays (configs.py:29), forward only, and the whole 1920 x 1280 frame through
    zipnerf.render_image (compute_extras like random_render_waymo_seq.py:197).  Roofline of the dominant kernel (hash-grid gather of the
    NeRF level inside the fused featurisation): useful bytes / its event-timed launch.
    `compute`: the networks' arithmetic -- "fp16" is what BASELINE configs[3] names ("fp16 MLP": zipnerf/train.py:215 autocast); the train
    step of every mode in `also` is timed beside it on the same batch."""
    import types
    from snerf_amd import ops, zipnerf
    from snerf_amd.trainer import ZipTrainer
    torch.manual_seed(0)
    m = zipnerf.Model(config=None, raydist_fn='power_transformation', opaque_background=True, compute=compute, table_dtype="ref",
                      init_std=0.1, device=device)
    tr = ZipTrainer(m, lr=1e-2)
    R = n_rays
    g = torch.Generator().manual_seed(1)
    pix = torch.randint(0, 1920 * 1280, (R,), generator=g)
    K = torch.tensor([[2050.0, 0.0, 960.0], [0.0, 2050.0, 640.0], [0.0, 0.0, 1.0]])
    c2w = torch.eye(4)[:3].clone()
    c2w[:, 3] = torch.tensor([0.02, -0.01, 0.03])
    Kinv, c2wd = torch.linalg.inv(K)[None].to(device), c2w[None].to(device)
    rays = ops.zip_pixels_to_rays((pix % 1920).int().to(device), (pix // 1920).int().to(device), None, Kinv, c2wd)
    # the WAYMO loader's ray range after its PCA rescale (internal/datasets.py:837-841: near = 2 scale_factor, far = 10000 scale_factor;
    # scale_factor = 1 / (largest camera distance): 0.01 for a 100 m sequence), not waymo.gin's 0.1 / 10, which that loader overrides
    t_near, t_far = 2.0 * scale_factor, 10000.0 * scale_factor
    batch = dict(rays, near=torch.full((R, 1), t_near, device=device), far=torch.full((R, 1), t_far, device=device))
    batch["origins"] = batch["origins"] + (torch.randn(R, 3, generator=g) * 0.05).to(device)
    tgt = torch.rand(R, 3, generator=g).to(device)
    # (LiDAR returns at 2 .. 80 m in scene units)
    targets = dict(depth=(torch.rand(R, generator=g) * 0.78 + 0.02).to(device), depth_mask=(torch.rand(R, generator=g) < 0.5).float().to(device))
    train = lambda: tr.step(batch, tgt, train_frac=0.5, rand=True, targets=targets)

    def fwd():
        m.scattered_rays = True
        with torch.no_grad():
            m(None, batch, 1.0, False)
        m.scattered_rays = False
    dt_train = _timeit(train, steps, warm=3)
    # the dominant kernels of the train step, event-timed inside one more step: featurisation (gather + record count) and the table gradient
    rec, restore = _timed_calls(ops, ["zip_encode_fwd_count", "zip_encode_bwd_binned"])
    try:
        train(); torch.cuda.synchronize()
    finally:
        restore()
    enc_train = [e0.elapsed_time(e1) for e0, e1 in rec["zip_encode_fwd_count"]]
    tgrad = [e0.elapsed_time(e1) for e0, e1 in rec["zip_encode_bwd_binned"]]          # backward order: NeRF level, proposal 1, proposal 0
    dt_fwd = _timeit(fwd, steps, warm=1)
    rec, restore = _timed_calls(ops, ["zip_encode_fwd", "zip_encode_prop_fwd"])
    try:
        fwd(); torch.cuda.synchronize()
    finally:
        restore()
    enc_inf = [e0.elapsed_time(e1) for e0, e1 in rec["zip_encode_prop_fwd"]] + [e0.elapsed_time(e1) for e0, e1 in rec["zip_encode_fwd"]]
    W_, H_ = 1920, 1280
    pidx = torch.arange(W_ * H_, device=device)
    fr = ops.zip_pixels_to_rays((pidx % W_).int(), (pidx // W_).int(), None, Kinv, c2wd)
    fr.update(near=torch.full((W_ * H_, 1), t_near, device=device), far=torch.full((W_ * H_, 1), t_far, device=device))
    frame = {k: v.reshape(H_, W_, -1) for k, v in fr.items()}
    cfg = types.SimpleNamespace(render_chunk_size=65536, vis_num_rays=16)
    m.config = cfg
    rfn = lambda rand, b: m(rand, b, train_frac=1.0, compute_extras=True)
    zipnerf.render_image(rfn, None, frame, False, cfg)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    img = zipnerf.render_image(rfn, None, frame, False, cfg)
    torch.cuda.synchronize(); dt_frame = time.perf_counter() - t0
    ok = img["rgb"].shape == (H_, W_, 3) and bool(torch.isfinite(img["rgb"]).all())
    # BASELINE configs[4] (M5): the S-NeRF++ background frame as random_render_waymo_seq.py:197-220 produces it -- the model with its
    # 19-class semantic head (internal/models.py:586-703), compute_extras, and the frame's files (rgb PNG, 16-bit depth PNG, argmax
    # semantic PNG + palette) written through FrameWriter (device-side quantisation, PNG encoding on a background thread)
    import shutil
    import tempfile
    from snerf_amd.frame_writer import FrameWriter
    torch.manual_seed(0)
    msem = zipnerf.Model(config=None, raydist_fn='power_transformation', opaque_background=True, compute=compute, table_dtype="ref",
                         init_std=0.1, device=device, use_semantic=True)
    msem.config = cfg
    rsem = lambda rand, b: msem(rand, b, train_frac=1.0, compute_extras=True)
    tmpd = tempfile.mkdtemp(prefix="snerf_frames_")
    try:
        with FrameWriter(tmpd, scale_factor=1.0) as fw:
            fw.write(0, zipnerf.render_image(rsem, None, frame, False, cfg))
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for k in (1, 2):
                img_s = zipnerf.render_image(rsem, None, frame, False, cfg)
                fw.write(k, img_s)
            torch.cuda.synchronize(); dt_sem_render = (time.perf_counter() - t0) / 2
        dt_sem_total = (time.perf_counter() - t0) / 2               # (the context exit waits for the writer thread)
        files = sorted(os.path.relpath(os.path.join(d, f), tmpd) for d, _, fs in os.walk(tmpd) for f in fs)
        sem_ok = img_s["semantic"].shape == (H_, W_, 19) and bool(torch.isfinite(img_s["semantic"]).all()) and len(files) >= 9
    finally:
        shutil.rmtree(tmpd, ignore_errors=True)
    sem_frame = {"what": "Model(use_semantic=True) frame 1920 x 1280 with the 19-class head + compute_extras, written through FrameWriter (rgb / depth16 / semantic / paint PNGs)",
                 "render_ms": round(dt_sem_render * 1e3, 1), "render_and_files_ms": round(dt_sem_total * 1e3, 1), "frames": 2, "ok": sem_ok, "files_per_frame": len(files) // 3}
    del msem, img_s
    torch.cuda.empty_cache()
    useful = [R * 7 * 64 * 6 * 8 * 4, R * 7 * 64 * 8 * 8 * 4, R * 7 * 32 * 10 * 8 * 4 * 2]       # prop 0 / prop 1 (fp32, C = 1), NeRF (fp16, C = 4)
    cb, cb_src = counter_bytes("zip_encode_fwd_all_nerf_train")
    rl = {"bound": "hbm", "kernel": "zip_encode_fwd_all_kernel<half, %s, 4, COUNT> (NeRF-level hash-grid gather of the train step)" % {"fp16": "_Float16", "bf16": "bf16", "f32": "float"}[compute],
          "achieved": round(useful[2] / (enc_train[2] * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
          "frac": round(useful[2] / (enc_train[2] * 1e-3) / 1e9 / 8000.0, 4), "useful_bytes_per_launch": useful[2],
          "traffic": cb, "traffic_source": cb_src, "launch_ms": round(enc_train[2], 3)}
    out = {"workload": "BASELINE configs[3] / [4]: zipnerf Model (waymo.gin: 64 + 64 + 32 intervals x 7 multisamples, grids L = 6 / 8 / 10, T = 2^21, "
                       "NeRF table fp16), rays in [2, 10000] x scale_factor 0.01 (datasets.py:837-841), ZipTrainer step with depth targets; %s MLPs%s" % (compute, " (static loss scale %g folded into Adam)" % tr.loss_scale if tr.loss_scale != 1 else ""),
           "dtype": compute,
           "rays_per_step": R, "steps": steps, "train_ms_per_step": round(dt_train * 1e3, 3), "train_rays_per_s": round(R / dt_train, 1),
           "fwd_ms": round(dt_fwd * 1e3, 3), "fwd_rays_per_s": round(R / dt_fwd, 1),
           "frame_1920x1280_ms": round(dt_frame * 1e3, 1), "frame_ok": ok, "frame_semantic": sem_frame,
           "encode_train_ms_per_level": [round(x, 3) for x in enc_train], "encode_inference_ms_per_level": [round(x, 3) for x in enc_inf],
           "encode_train_useful_GBps_per_level": [round(b / (ms * 1e-3) / 1e9, 1) for b, ms in zip(useful, enc_train)],
           "table_gradient_ms": {"nerf": round(tgrad[0], 3), "prop1": round(tgrad[1], 3), "prop0": round(tgrad[2], 3)},
           "train_useful_gather_bytes_per_step": 2 * sum(useful),
           "train_step_useful_TBps": round(2 * sum(useful) / dt_train / 1e12, 3),
           "roofline": rl, "losses_last_step": [round(v, 6) for v in tr.last_losses.cpu().tolist()]}
    sd_c = {k: v.detach().float().cpu().clone() for k, v in m.state_dict().items() if v.is_floating_point()}     # (the encoders' integer buffers stay behind)
    del tr, m, frame, fr, img
    torch.cuda.empty_cache()
    for other in also:                                   # the same train step in the other reduced-precision mode(s)
        torch.manual_seed(0)
        m2 = zipnerf.Model(config=None, raydist_fn='power_transformation', opaque_background=True, compute=other, table_dtype="ref", init_std=0.1, device=device)
        tr2 = ZipTrainer(m2, lr=1e-2)
        dt2 = _timeit(lambda: tr2.step(batch, tgt, train_frac=0.5, rand=True, targets=targets), steps, warm=3)
        out["train_ms_per_step_" + other] = round(dt2 * 1e3, 3)
        del tr2, m2
        torch.cuda.empty_cache()
    out.update(path_c_baselines(sd_c, batch, tgt, device, R / dt_train))
    out["eager_baseline"]["frame_speedup_vs_fp32"] = round((W_ * H_ / dt_frame) / out["eager_baseline"]["forward_only_fp32"], 2)
    return out


def _zip_oracle_step(oz, p, specs, b, tgt, grid_fn):
    """forward + backward of the reference zipnerf Model's train step as plain torch ops (oracle/zip.py = internal/models.py:98-349):
    Charbonnier data loss on the final level, autograd backward into every parameter incl. the three hash tables"""
    saved, oz.grid_features = oz.grid_features, grid_fn
    try:
        jit = [torch.rand(b["origins"].shape[0], 1, device=tgt.device) for _ in range(3)]
        ren, _ = oz.model_forward(p, specs, b, jitters=jit)
        loss = torch.sqrt((ren[-1]["rgb"] - tgt) ** 2 + 0.001 ** 2).mean()
        loss.backward()
    finally:
        oz.grid_features = saved
    return loss


def path_c_baselines(sd, batch, tgt, device, build_rays_per_s, n_cpu=512, n_eager=16384):
    """The two baselines SURVEY section 8(d) asks for beside path C's number, on the same synthetic batch and weights:
    cpu_baseline = the oracle's torch restatement (hash-grid gather as torch ops, oracle/eager.py) forward + backward on the box's host
    cores, bounded sample; eager_baseline = the same Python on the GPU as the reference would run it on ROCm -- torch eager ops +
    autograd + torch.optim.Adam, with the hash-grid encoder as its native extension's kernels (the per-(point, level) gather and the
    atomic scatter of gridencoder.cu, here csrc/grid.hip with the fast path off), fp32 (no autocast: the bf16 / fp16 numbers are the build's)."""
    from oracle import zip as oz, eager
    from snerf_amd import ops
    from snerf_amd.gridencoder import grid_encode
    specs = [oz.GridSpec(6, 1, 512), oz.GridSpec(8, 1, 2048), oz.GridSpec(10, 4, 8192)]
    keys = ("origins", "directions", "viewdirs", "radii", "base_x", "base_y", "near", "far")
    out = {}
    # ---- host CPU
    bc = {k: batch[k][:n_cpu].detach().float().cpu() for k in keys}
    bc["radii"] = bc["radii"].reshape(n_cpu, 1)
    pc = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    t0 = time.perf_counter()
    _zip_oracle_step(oz, pc, specs, bc, tgt[:n_cpu].cpu(), eager.grid_features_torch)
    dt = time.perf_counter() - t0
    out["cpu_baseline"] = {"value": round(n_cpu / dt, 2), "unit": "rays/s", "cores": torch.get_num_threads(), "kind": "port",
                           "sample": f"{n_cpu} rays forward + backward of the zipnerf Model (oracle/zip.py + oracle/eager.py: torch-CPU restatement of the reference), {dt:.1f} s"}
    del pc
    # ---- PyTorch-ROCm eager on this GPU, reference-form grid kernels
    prev = torch.get_default_device() if hasattr(torch, "get_default_device") else None
    torch.set_default_device(device)
    prev_fast = ops.grid_set_fast_path(False)
    try:
        pg = {k: v.to(device).requires_grad_(True) for k, v in sd.items()}
        bg = {k: batch[k][:n_eager].detach().float() for k in keys}
        bg["radii"] = bg["radii"].reshape(n_eager, 1)
        tg = tgt[:n_eager]
        offs = [torch.from_numpy(sp.offsets).to(device) for sp in specs]
        scale = {id(sp): float(2.0 ** sp.S) for sp in specs}
        off_of = {id(sp): o for sp, o in zip(specs, offs)}

        def grid_fn(spec, emb, means):
            y = grid_encode(((means.reshape(-1, 3) + 1) / 2).contiguous(), emb, off_of[id(spec)], scale[id(spec)], spec.H, False, 0, False, 0)
            return y.reshape(list(means.shape[:-1]) + [spec.L, spec.C])
        opt = torch.optim.Adam(list(pg.values()), lr=1e-2, eps=1e-15)

        def step():
            opt.zero_grad(set_to_none=True)
            _zip_oracle_step(oz, pg, specs, bg, tg, grid_fn)
            opt.step()
        step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        dte = (time.perf_counter() - t0) / 2
        def fwd_only():
            saved_g, oz.grid_features = oz.grid_features, grid_fn
            try:
                with torch.no_grad():
                    oz.model_forward(pg, specs, bg, jitters=None)
            finally:
                oz.grid_features = saved_g
        fwd_only()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(2):
            fwd_only()
        torch.cuda.synchronize()
        dtf = (time.perf_counter() - t0) / 2
        out["eager_baseline"] = {"unit": "rays/s", "fp32": round(n_eager / dte, 1), "fp32_ms_per_step": round(dte * 1e3, 2), "rays_per_step": n_eager, "steps": 2,
                                 "forward_only_fp32": round(n_eager / dtf, 1),
                                 "kind": "the reference's Model as PyTorch-ROCm eager ops (oracle/zip.py on the GPU) + autograd + torch.optim.Adam, its hash-grid "
                                         "extension as the per-(point, level) gather / atomic-scatter kernels (csrc/grid.hip, fast path off), same GPU",
                                 "speedup_of_the_build": round(build_rays_per_s / (n_eager / dte), 2)}
        del pg, opt
    finally:
        ops.grid_set_fast_path(prev_fast)
        torch.set_default_device(prev if prev is not None else "cpu")
        torch.cuda.empty_cache()
    return out


def path_b_baselines(sd_c, sd_f, rays, tgt, device, build_rays_per_s, n_cpu=1024, n_eager=8192):
    """cpu_baseline (oracle/classic.py = render.py:281-409 + run_nerf_helpers.py on the host cores, forward + backward, bounded sample) and
    eager_baseline (the same torch ops on this GPU with torch.searchsorted as the reference calls it, autograd + torch.optim.Adam, fp32)
    for the classic render_rays path."""
    from oracle import classic as oc, eager
    out = {}

    def step(pc, pf, r, t, opt=None):
        if opt is not None:
            opt.zero_grad(set_to_none=True)
        n = r.shape[0]
        ret = oc.render_rays(r, pc, pf, 64, 128, t_rand=torch.rand(n, 64, device=r.device), u=torch.rand(n, 128, device=r.device))
        loss = ((ret["rgb_map"] - t) ** 2).mean() + ((ret["rgb0"] - t) ** 2).mean()
        loss.backward()
        if opt is not None:
            opt.step()
    pc = {k: v.clone().requires_grad_(True) for k, v in sd_c.items()}
    pf = {k: v.clone().requires_grad_(True) for k, v in sd_f.items()}
    t0 = time.perf_counter()
    step(pc, pf, rays[:n_cpu].cpu(), tgt[:n_cpu].cpu())
    dt = time.perf_counter() - t0
    out["cpu_baseline"] = {"value": round(n_cpu / dt, 2), "unit": "rays/s", "cores": torch.get_num_threads(), "kind": "port",
                           "sample": f"{n_cpu} rays forward + backward of render_rays (64 + 128, two NeRF 8 x 256), oracle/classic.py, {dt:.1f} s"}
    prev = torch.get_default_device() if hasattr(torch, "get_default_device") else None
    saved, oc.sample_pdf = oc.sample_pdf, eager.sample_pdf_torch
    torch.set_default_device(device)
    try:
        pc = {k: v.to(device).requires_grad_(True) for k, v in sd_c.items()}
        pf = {k: v.to(device).requires_grad_(True) for k, v in sd_f.items()}
        opt = torch.optim.Adam(list(pc.values()) + list(pf.values()), lr=5e-4)
        r, t = rays[:n_eager], tgt[:n_eager]
        step(pc, pf, r, t, opt)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            step(pc, pf, r, t, opt)
        torch.cuda.synchronize()
        dte = (time.perf_counter() - t0) / 3

        def fwd_only():
            with torch.no_grad():
                oc.render_rays(r, pc, pf, 64, 128, t_rand=None, u=None)
        fwd_only()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            fwd_only()
        torch.cuda.synchronize()
        dtf = (time.perf_counter() - t0) / 3
        out["eager_baseline"] = {"unit": "rays/s", "fp32": round(n_eager / dte, 1), "fp32_ms_per_step": round(dte * 1e3, 2), "rays_per_step": n_eager, "steps": 3,
                                 "forward_only_fp32": round(n_eager / dtf, 1),
                                 "kind": "plain PyTorch-ROCm eager ops (torch restatement of render_rays / NeRF), autograd + torch.optim.Adam, same GPU",
                                 "speedup_of_the_build": round(build_rays_per_s / (n_eager / dte), 2)}
    finally:
        oc.sample_pdf = saved
        torch.set_default_device(prev if prev is not None else "cpu")
        torch.cuda.empty_cache()
    return out


def grid_encoder_leg(device, R=65536, S=32, n=7, steps=3):
    """The reference's only native FFI as a stand-alone operator (gridencoder/grid.py:24-89 -> bindings.cpp:5-9): GridEncoder of the NeRF
    level (L = 10, C = 4, H = 16 -> 8192, T = 2^21: internal/models.py:381-386) on B = 65 536 x 32 x 7 points under fp16 autocast (half
    table, half gradients: grid.py:41-44), forward + backward through the module's autograd Function.  Points in the order the
    reference's callers pass them (the 7 helix multisamples of an interval one after the other, contracted: models.py:488-494).
    Timed: the corner-cached gather + binned table gradient (default) and, beside it, the one-thread-per-(point, level) gather + atomic
    scatter of kernel_grid / kernel_grid_backward's restatement (SNERF_GRID_FAST=0), same inputs."""
    from snerf_amd import ops
    from snerf_amd.gridencoder import GridEncoder
    L, C = 10, 4
    x = grid_points(device, R, S, n)
    B = x.shape[0]
    g = torch.Generator(device=device).manual_seed(4)
    enc = GridEncoder(input_dim=3, num_levels=L, level_dim=C, base_resolution=16, desired_resolution=8192, log2_hashmap_size=21, device=device)
    with torch.no_grad():
        enc.embeddings.uniform_(-0.1, 0.1)
    w = (torch.randn(B, L * C, generator=g, device=device) * 1e-3).half()
    return _grid_encoder_measure(enc, x, w, R, S, n, L, C, steps)


def grid_points(device, R=65536, S=32, n=7):
    """R x S x n contracted multisample points in [-1, 1]^3, in the order zipnerf passes them to its encoders (models.py:488-494)"""
    g = torch.Generator(device=device).manual_seed(3)
    o = torch.randn(R, 1, 1, 3, generator=g, device=device) * 0.05
    d = torch.nn.functional.normalize(torch.randn(R, 1, 1, 3, generator=g, device=device), dim=-1)
    edges = torch.exp(torch.linspace(math.log(0.1), math.log(30.0), S + 1, device=device))
    t0, t1 = edges[:-1].view(1, S, 1, 1), edges[1:].view(1, S, 1, 1)
    j = (torch.arange(n, device=device, dtype=torch.float32) + 0.5).view(1, 1, n, 1) / n
    t = t0 + (t1 - t0) * j
    ang = 2 * math.pi * 3 * j
    e1 = torch.nn.functional.normalize(torch.cross(d, torch.tensor([0.0, 0.0, 1.0], device=device).expand_as(d), dim=-1), dim=-1)
    e2 = torch.cross(d, e1, dim=-1)
    x = o + d * t + 1.5e-3 * t * (torch.cos(ang) * e1 + torch.sin(ang) * e2)
    mag = x.norm(dim=-1, keepdim=True).clamp(min=1e-6)
    x = torch.where(mag <= 1, x, (2 - 1 / mag) * x / mag) / 2                       # contraction into the ball of radius 2, halved: [-1, 1]^3
    return x.reshape(-1, 3).contiguous()


def _grid_encoder_measure(enc, x, w, R, S_, n, L, C, steps):
    from snerf_amd import ops
    import numpy as np
    B = x.shape[0]

    def fwd_bwd():
        enc.embeddings.grad = None
        with torch.autocast("cuda", dtype=torch.float16):
            y = enc(x, bound=1)
        y.backward(w)

    def fwd_only():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            enc(x, bound=1)
    res = {}
    prev_fast = ops.GRID_FAST
    try:
        for name, fast, k in (("fast", True, steps), ("reference_form", False, 1)):
            ops.grid_set_fast_path(fast)
            f_ms = _timeit(fwd_only, k, warm=1) * 1e3
            fb_ms = _timeit(fwd_bwd, k, warm=1) * 1e3
            res[name] = (f_ms, fb_ms - f_ms, enc.embeddings.grad.clone())
    finally:
        ops.grid_set_fast_path(prev_fast)
    gf, gs = res["fast"][2], res["reference_form"][2]
    # an fp32-ACCUMULATED gradient of the same half gradients as the yardstick for both (fp32 atomics into an fp32 table gradient: order
    # noise ~1e-7): which of the two 16-bit forms is off, and by how much, at the full size
    x01 = ((x + 1) / 2).contiguous()
    S, H = float(np.log2(enc.per_level_scale)), enc.base_resolution
    truth, _ = ops.grid_encode_bwd(w.float(), x01, torch.zeros_like(enc.embeddings.data), enc.offsets, L, S, H, 0, False, 0)
    tn = float(truth.norm())
    err = lambda g: {"rel_l2": float((g.float() - truth).norm() / tn), "max_abs_over_max": float((g.float() - truth).abs().max() / truth.abs().max())}
    e_fast, e_atomic = err(gf), err(gs)
    plan = ops.grid_encode_bwd_binned_plan(B, C, L, enc.offsets.cpu().numpy(), True, torch.float16)
    useful = B * L * 8 * C * 2                                                # 8 corner rows of C halves per (point, level)
    cb, cb_src = counter_bytes("grid_encoder_fwd")
    out = {"workload": f"GridEncoder (L = {L}, C = {C}, T = 2^21, 16 -> 8192; half table under autocast) forward + backward on B = {R} x {S_} x {n} = {B} "
                       "ray-ordered contracted multisample points; autograd through snerf_amd.gridencoder (the drop-in of gridencoder/grid.py)",
           "points": B, "fwd_ms": round(res["fast"][0], 3), "bwd_ms": round(res["fast"][1], 3),
           "reference_form_fwd_ms": round(res["reference_form"][0], 3), "reference_form_bwd_ms": round(res["reference_form"][1], 3),
           "bwd_speedup_vs_atomic_scatter": round(res["reference_form"][1] / res["fast"][1], 2),
           "fwd_speedup_vs_per_point_gather": round(res["reference_form"][0] / res["fast"][0], 2),
           "bwd_workspace_bytes": plan["bytes_used"], "bwd_chunks_per_level": plan["chunks"], "bwd_levels_per_transposed_group": plan["levels_per_transposed_group"],
           "bwd_launches": plan["launches"],
           "table_gradient_error_vs_fp32_accumulated": {"fast_binned_half_records": e_fast, "atomic_half2_scatter": e_atomic},
           "roofline": {"bound": "hbm", "kernel": "g3_fwd_kernel<_Float16, 4> (corner-cached hash-grid gather)", "achieved": round(useful / (res["fast"][0] * 1e-3) / 1e9, 1),
                        "peak": 8000.0, "unit": "GB/s", "frac": round(useful / (res["fast"][0] * 1e-3) / 1e9 / 8000.0, 4), "useful_bytes_per_launch": useful,
                        "traffic": cb, "traffic_source": cb_src},
           "bwd_useful_GBps": round(useful / (res["fast"][1] * 1e-3) / 1e9, 1)}
    del truth, x01
    del enc, x, w, res, gf, gs
    torch.cuda.empty_cache()
    return out


def path_b_leg(device, n_rays=32768, steps=5, compute="bf16"):
    """The classic render_rays path (path B, behind the signatures north_star names): 64 coarse + 192 fine evaluations per ray through two
    NeRF 8 x 256 networks, autograd through the drop-in operators + torch.optim.Adam (the route of render.py:281-409 callers)."""
    from snerf_amd import classic
    torch.manual_seed(0)
    mk = lambda: classic.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=4, skips=[4], use_viewdirs=True, compute=compute, device=device)
    coarse, fine = mk(), mk()
    embed_fn, _ = classic.get_embedder(10, 0)
    embeddirs_fn, _ = classic.get_embedder(4, 0)
    q = classic.make_network_query_fn(embed_fn, embeddirs_fn, netchunk=1 << 30)
    N = n_rays
    g = torch.Generator().manual_seed(1)
    d = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1)
    o = torch.randn(N, 3, generator=g) * 0.1 + torch.tensor([0.0, 0.0, 4.0])
    rays = torch.cat([o, -d, torch.full((N, 1), 2.0), torch.full((N, 1), 6.0), -d], -1).to(device)
    tgt = torch.rand(N, 3, generator=g).to(device)
    opt = torch.optim.Adam(list(coarse.parameters()) + list(fine.parameters()), lr=5e-4)
    fwd = lambda: classic.render_rays(rays, coarse, q, 64, perturb=1.0, N_importance=128, network_fine=fine, white_bkgd=False, raw_noise_std=0.0)

    def train():
        opt.zero_grad(set_to_none=False)
        r = fwd()
        loss = ((r["rgb_map"] - tgt) ** 2).mean() + ((r["rgb0"] - tgt) ** 2).mean()
        loss.backward()
        opt.step()
        coarse.arena.bump(); fine.arena.bump()
    dt_train = _timeit(train, steps)
    with torch.no_grad():
        dt_fwd = _timeit(fwd, steps)
    # the 1600 x 900 frame through the drop-in render() (render.py:22-91: rays from c2w, batchify_rays in chunks, both networks)
    Hh, Ww, focal = 900, 1600, 1266.0
    c2w = torch.tensor([[1.0, 0.0, 0.0, 0.0], [0.0, 1.0, 0.0, 0.0], [0.0, 0.0, 1.0, 4.0]], device=device)
    rk = dict(network_fn=coarse, network_query_fn=q, N_samples=64, perturb=0.0, N_importance=128, network_fine=fine, white_bkgd=False, raw_noise_std=0.0)
    with torch.no_grad():
        classic.render(Hh, Ww, focal, chunk=1 << 17, c2w=c2w, ndc=False, near=2.0, far=6.0, use_viewdirs=True, **rk)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fr = classic.render(Hh, Ww, focal, chunk=1 << 17, c2w=c2w, ndc=False, near=2.0, far=6.0, use_viewdirs=True, **rk)
        torch.cuda.synchronize(); dt_frame = time.perf_counter() - t0
    frame_ok = fr[0].shape == (Hh, Ww, 3) and bool(torch.isfinite(fr[0]).all())
    del fr
    flops = 2.0 * 593408 * (64 + 192)                                  # 303.8 MFLOP / ray forward
    a_tr, a_fw = 3 * N * flops / dt_train / 1e12, N * flops / dt_fwd / 1e12
    cb, cb_src = counter_bytes("fmlp_kernel_train_fwd")
    out = {"workload": "path B: classic render_rays (render.py:281-409), 64 coarse + 128 importance samples (fine net on 192), NeRF 8 x 256 x 2, "
                       "autograd + torch.optim.Adam; bf16 MFMA, fp32 accumulate",
           "rays_per_step": N, "steps": steps, "train_ms_per_step": round(dt_train * 1e3, 3), "train_rays_per_s": round(N / dt_train, 1),
           "fwd_ms": round(dt_fwd * 1e3, 3), "fwd_rays_per_s": round(N / dt_fwd, 1), "frame_1600x900_ms": round(dt_frame * 1e3, 1), "frame_ok": frame_ok,
           "frame_what": "classic.render(H, W, focal, c2w=...) (render.py:22-91): on-device rays, 11 chunks of 131 072 rays, coarse + fine network; measured",
           "frame_tflops": round(Hh * Ww * flops / dt_frame / 1e12, 1),
           "roofline": {"bound": "mfma", "kernel": "fmlp_kernel (fused register-resident 8 x 256 MLP) + fchain_bwd + gemm_tn (whole step's MLP work)",
                        "achieved": round(a_tr, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(a_tr / PEAK_BF16_TFLOPS, 4),
                        "forward_only_achieved": round(a_fw, 1), "forward_only_frac": round(a_fw / PEAK_BF16_TFLOPS, 4),
                        "algorithmic_flops_per_ray_forward": flops, "traffic": cb, "traffic_source": cb_src}}
    sd_c = {k: v.detach().float().cpu().clone() for k, v in coarse.state_dict().items()}
    sd_f = {k: v.detach().float().cpu().clone() for k, v in fine.state_dict().items()}
    del coarse, fine, opt
    torch.cuda.empty_cache()
    # the modes whose render_rays outputs stay inside the 1e-4 contract (per-layer launches: the fused kernels are bf16): same step, 3 repetitions
    inside = {}
    for mode in ("f16f8", "f32"):
        compute_keep, compute = compute, mode
        coarse, fine = mk(), mk()
        compute = compute_keep
        opt = torch.optim.Adam(list(coarse.parameters()) + list(fine.parameters()), lr=5e-4)
        t_tr = _timeit(train, 3, warm=1)
        with torch.no_grad():
            t_fw = _timeit(fwd, 3, warm=1)
        inside[mode] = {"train_ms_per_step": round(t_tr * 1e3, 2), "fwd_ms": round(t_fw * 1e3, 2)}
        del coarse, fine, opt
        torch.cuda.empty_cache()
    out["modes_inside_1e-4"] = inside
    out.update(path_b_baselines(sd_c, sd_f, rays, tgt, device, N / dt_train))
    out["eager_baseline"]["frame_speedup_vs_fp32"] = round((Hh * Ww / dt_frame) / out["eager_baseline"]["forward_only_fp32"], 2)
    return out


def small_step_leg(device, n_rays=512, steps=20):
    """The per-GPU step of the reference's 4096-ray batch split over 8 GPUs (8 x 512 rays): launched kernel by kernel and
    as a hipGraph replay (MipTrainer.capture).  The strong-scaling ceiling of an 8-GPU node is headline_ms / (8 x this)."""
    from snerf_amd.trainer import MipTrainer
    m = build_model("bf16", device)
    tr = MipTrainer(m, lr=5e-4)
    rays = synth_rays(n_rays, 77, device)
    g = torch.Generator(device="cpu").manual_seed(78)
    tgt = torch.rand(n_rays, 3, generator=g).to(device)
    depth = torch.where(torch.rand(n_rays, generator=g) < 0.5, torch.rand(n_rays, generator=g) * 78 + 2, torch.zeros(n_rays)).to(device)
    conf = torch.rand(n_rays, generator=g).to(device)
    dt_e = _timeit(lambda: tr.step(rays, tgt, depth, conf), steps, warm=5)
    tr.capture(rays, tgt, depth, conf, warmup=2)
    dt_g = _timeit(tr.replay, steps, warm=3)
    del tr, m
    torch.cuda.empty_cache()
    return {"rays": n_rays, "steps": steps, "ms": round(min(dt_e, dt_g) * 1e3, 3), "ms_eager_launches": round(dt_e * 1e3, 3), "ms_hipgraph": round(dt_g * 1e3, 3),
            "what": "path-A train step at 512 rays (1/8 of the 4096-ray batch): draws, forward, loss tail, backward, fused Adam; bf16"}


def shipped_shape_leg(device, n_rays=4096, steps=5):
    """SURVEY 8(d) M2's second shape: the shipped config's 128 proposal + 127 fine intervals (configs/nuScenes_depth_6cams:36; mip.py:308-316)
    instead of BASELINE's 64 + 128 -- same train step, same kernels."""
    global S0, P1
    from snerf_amd.trainer import MipTrainer
    keep = (S0, P1)
    S0, P1 = 128, 128
    try:
        m = build_model("bf16", device)
        tr = MipTrainer(m, lr=5e-4)
        rays = synth_rays(n_rays, 1000, device)
        g = torch.Generator(device="cpu").manual_seed(2000)
        tgt = torch.rand(n_rays, 3, generator=g).to(device)
        depth = torch.where(torch.rand(n_rays, generator=g) < 0.5, torch.rand(n_rays, generator=g) * 78 + 2, torch.zeros(n_rays)).to(device)
        conf = torch.rand(n_rays, generator=g).to(device)
        dt = _timeit(lambda: tr.step(rays, tgt, depth, conf), steps, warm=3)
        fwd = 2.0 * (S0 * MAC_PROP + (P1 - 1) * MAC_NERF)
        del tr, m
    finally:
        S0, P1 = keep
    torch.cuda.empty_cache()
    return {"shape": "128 proposal + 127 fine intervals (255 network evaluations per ray), hidden 1024", "rays_per_step": n_rays, "steps": steps,
            "ms_per_step": round(dt * 1e3, 3), "rays_per_s": round(n_rays / dt, 1), "whole_step_tflops": round(3 * fwd * n_rays / dt / 1e12, 1)}


PROSE_KEYS = ("what", "note", "frame_what", "frame_note", "includes", "traffic_source", "mode", "scene", "window", "kernel", "parallelism", "collective", "step",
              # numbers only the verbose line carries (the compact one must stay under the driver's 8 KB tail)
              "losses_last_step", "leg_wall_s", "p999_rel_err_depth", "max_abs_err_acc")
LAST_LEGS = ("path_b_ert", "path_b", "grid_encoder", "path_c")          # the driver keeps the line's 8 KB tail: the other configs' numbers go last


def compact_line(out, limit=7900):
    """The one JSON line without its prose (the strings say what each leg ran: profiles/README.md holds them, keyed by leg; --verbose-json prints them
    inline), floats at 6 significant digits, the legs of BASELINE configs 4-5 / path B last.  Contract fields (metric .. config.workload, roofline,
    cpu_baseline{value, unit, cores, kind, sample}) stay."""
    def strip(v, top=False):
        if isinstance(v, dict):
            o = {}
            for k, x in v.items():
                if k in PROSE_KEYS or (k == "kind" and isinstance(x, str) and len(x) > 16) or (k == "workload" and not top):
                    continue
                if k == "sample" and isinstance(x, str):
                    x = x[:72]
                o[k] = strip(x)
            return o
        if isinstance(v, list):
            return [strip(x) for x in v]
        if isinstance(v, float) and v == v and abs(v) != float("inf"):
            return float(f"{v:.6g}")
        return v
    c = {k: (strip(v) if k != "config" else {kk: (vv if kk == "workload" else strip(vv)) for kk, vv in v.items() if kk not in PROSE_KEYS}) for k, v in out.items()}
    c["roofline"] = dict(c["roofline"], kernel=out["roofline"]["kernel"].split(" ")[0])
    ordered = {k: v for k, v in c.items() if k not in LAST_LEGS}
    for k in LAST_LEGS:
        if k in c:
            ordered[k] = c[k]
    n = len(json.dumps(ordered))
    ordered_final = dict(ordered)
    if n > limit:                                          # (never cut a number silently: say so in the line)
        ordered_final = {"line_bytes_over_limit": n, **ordered}
    return ordered_final


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rays", type=int, default=4096, help="rays per GPU per step (weak scaling) / global rays per step (--scaling strong)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --rays per GPU whatever N (default); strong: the reference's --rays-ray batch split across the N GPUs")
    ap.add_argument("--compute", default="bf16", choices=["bf16", "f32", "bf16x3", "bf16x3_fwd", "f16f8", "fp16"])
    ap.add_argument("--variant", type=int, default=8, help="NT GEMM variant: 8 = persistent 8-phase 256x256 (default), 4 = 8-phase, 1 = 256x256 block-issue, 0 = 128x128")
    ap.add_argument("--no-frame", action="store_true", help="skip the 1600x900 frame render")
    ap.add_argument("--no-cpu", action="store_true", help="skip the host-CPU baseline")
    ap.add_argument("--no-eager", action="store_true", help="skip the PyTorch-ROCm eager baseline (3 steps each of fp32 and bf16 autocast on this GPU)")
    ap.add_argument("--no-f32", action="store_true", help="skip the fp32-parity-mode leg (3 train steps with exact-fp32 MFMA)")
    ap.add_argument("--eager", action="store_true", help="(kept for compatibility: the eager baseline is on by default)")
    ap.add_argument("--no-ert-scene", action="store_true", help="skip the path-B early-termination leg and the fitted-weights precision read-out (fit steps on an analytic scene)")
    ap.add_argument("--ert-scene", action="store_true", help="also run the path-A early-termination leg (retired from the default line in round 6: 0.96x, "
                                                             "the fine samples of path A sit ON the surface; profiles/README.md)")
    ap.add_argument("--verbose-json", action="store_true", help="print the line with every leg's prose (what / note / kind / traffic_source strings); the default "
                                                               "line is compact (< 8 KB: the driver keeps an 8 KB tail) -- the prose is in profiles/README.md, keyed by leg")
    ap.add_argument("--no-paths", action="store_true", help="skip the path-C (zipnerf, 65 536 rays) and path-B (classic render_rays, 32 768 rays) legs")
    ap.add_argument("--no-dropin", action="store_true", help="skip the drop-in-autograd and pose-refinement legs (5 steps each)")
    ap.add_argument("--cpu-rays", type=int, default=512, help="rays of the bounded host-CPU baseline sample (about 15 s on the GPU box)")
    ap.add_argument("--shape", default="baseline", choices=["baseline", "shipped"],
                    help="baseline = BASELINE.json's 64 proposal + 128 fine evals/ray (192 spp); shipped = the reference config's "
                         "128 proposal + 127 fine intervals (configs/nuScenes_depth_6cams)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only for functional tests)")
    ap.add_argument("--same-device", action="store_true", help="functional test of the N > 1 control flow on a 1-GPU box: every rank uses cuda:0")
    ap.add_argument("--ert", type=float, nargs=2, default=None, metavar=("EPS_T", "EPS_W"),
                    help="also render the frame with early ray termination + sample compaction (inference extension, not the reference's "
                         "algorithm): skip fine samples whose proposal-predicted transmittance <= EPS_T or weight <= EPS_W")
    ap.add_argument("--frame-chunk", type=int, default=32768)
    ap.add_argument("--graph", action="store_true",
                    help="time the hipGraph-captured step (MipTrainer.capture / replay: forward + loss tail + backward as one graph launch; "
                         "N > 1: the all-reduce and Adam follow outside the graph) -- the small-batch / strong-scaling step")
    args = ap.parse_args()
    global S0, P1
    if args.shape == "shipped":
        S0, P1 = 128, 128

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(relaunch_under_torchrun(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.same_device:
        local = 0
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend)
    else:
        torch.cuda.set_device(0)
    device = torch.device("cuda", local if world > 1 else 0)

    from snerf_amd import ops
    from snerf_amd.mipnerf import Rays, render_image
    from snerf_amd.trainer import MipTrainer
    model = build_model(args.compute, device)
    model.nerf.variant = model.prop.variant = args.variant
    trainer = MipTrainer(model, lr=5e-4)
    trainer.broadcast_parameters(0)

    n = args.rays if args.scaling == "weak" else max(args.rays // world, 1)
    rays = synth_rays(n, 1000 + rank, device)
    g = torch.Generator(device="cpu").manual_seed(2000 + rank)
    tgt = torch.rand(n, 3, generator=g).to(device)
    depth = torch.where(torch.rand(n, generator=g) < 0.5, torch.rand(n, generator=g) * 78 + 2, torch.zeros(n)).to(device)
    conf = torch.rand(n, generator=g).to(device)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.graph:
        trainer.capture(rays, tgt, depth, conf, warmup=max(args.warmup, 1))
        step_fn = trainer.replay
    else:
        step_fn = lambda: trainer.step(rays, tgt, depth, conf)
    for _ in range(args.warmup):
        step_fn()
    barrier()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(args.steps):
        loss, _ = step_fn()
        ev[i + 1].record()                                   # per-step device timestamps (no sync inside the timed region)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        te = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = te.item()
    final_loss = float(loss)
    per_step = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps))
    ms_median = per_step[len(per_step) // 2] if len(per_step) % 2 else 0.5 * (per_step[len(per_step) // 2 - 1] + per_step[len(per_step) // 2])

    # ---- roofline of the dominant kernel (NT MFMA GEMM: all forward + data-gradient layers); the instrumented step contains
    # the gradient all-reduce, so EVERY rank runs it
    launches, gemm_ms, padded_flops, alg_nt = measure_gemm_kernel(trainer, rays, tgt, depth, conf)
    barrier()
    comm = None
    if world > 1:                                            # the step's one exchange, timed alone: all-reduce of the flat gradient arena
        buf = torch.zeros_like(model.arena.grad)
        for _ in range(2):
            dist.all_reduce(buf)
        barrier()
        tc = time.perf_counter()
        for _ in range(5):
            dist.all_reduce(buf)
        barrier()
        comm = {"collective": "all-reduce(sum) of the flat fp32 gradient arena in per-layer buckets, each started as soon as its layer's weight gradient is final (heads, the eight 4 MB trunk layers, the proposal network), overlapped with the rest of the backward",
                "backend": "rccl" if args.backend == "nccl" else args.backend, "ranks": world, "bytes": buf.numel() * 4,
                "allreduce_ms_alone": round((time.perf_counter() - tc) / 5 * 1e3, 3)}
        del buf
    # ---- N > 1: the STRONG-scaling step beside the weak-scaling headline (north_star / SURVEY 8e: the reference's 4096-ray batch split
    # across the ranks, 512 rays per GPU at N = 8) -- a second trainer on the same model, launched per kernel and as a hipGraph
    strong = None
    if world > 1 and args.scaling == "weak":
        try:
            ns = max(args.rays // world, 1)
            rs = synth_rays(ns, 3000 + rank, device)
            gs = torch.Generator(device="cpu").manual_seed(4000 + rank)
            tg_s = torch.rand(ns, 3, generator=gs).to(device)
            dp_s = torch.where(torch.rand(ns, generator=gs) < 0.5, torch.rand(ns, generator=gs) * 78 + 2, torch.zeros(ns)).to(device)
            cf_s = torch.rand(ns, generator=gs).to(device)
            strong = {"global_rays_per_step": ns * world, "rays_per_gpu_per_step": ns, "steps": args.steps}
            for mode in ("launches", "hipgraph"):
                ts = MipTrainer(model, lr=5e-4)
                if mode == "hipgraph":
                    ts.capture(rs, tg_s, dp_s, cf_s, warmup=2)
                    fn = ts.replay
                else:
                    fn = lambda: ts.step(rs, tg_s, dp_s, cf_s)
                for _ in range(3):
                    fn()
                barrier()
                t_s = time.perf_counter()
                for _ in range(args.steps):
                    fn()
                barrier()
                el = time.perf_counter() - t_s
                te = torch.tensor([el], device=device, dtype=torch.float64)
                dist.all_reduce(te, op=dist.ReduceOp.MAX)
                strong["ms_per_step_" + mode] = round(te.item() / args.steps * 1e3, 3)
                strong["rays_per_s_" + mode] = round(ns * world * args.steps / te.item(), 1)
                del ts
            strong["note"] = "scaling: strong -- total work fixed at --rays per step; compare rays_per_s with the N = 1 headline (same 4096-ray batch)"
        except Exception as e:          # (the headline above is already measured: a failure of this extra must not cost the line; every rank runs the same code)
            strong = {"error": f"{type(e).__name__}: {e}"[:300]}
    fwd = 2.0 * (S0 * MAC_PROP + (P1 - 1) * MAC_NERF)                        # 2.269 GFLOP / ray
    out = None
    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        rays_per_s = world * n * args.steps / elapsed
        achieved = alg_nt / (gemm_ms * 1e-3) / 1e12
        peak = 157.3 if args.compute == "f32" else PEAK_BF16_TFLOPS
        traffic, traffic_src = roofline_traffic(args.compute, args.variant)
        roofline = {"bound": "mfma", "kernel": {8: "gemm_nt8p_kernel (bf16, 256x256 persistent 8-phase)", 4: "gemm_nt8_kernel (bf16, 256x256 8-phase)", 1: "gemm_nt_kernel<bf16,256,256,2,4>"}.get(args.variant, "gemm_nt_kernel<bf16,128,128,2,2>") if args.compute == "bf16" else ("gemm_nt_kernel<f32,128,128,2,2>" if args.compute == "f32" else "gemm_nt8p_kernel<.., SPLIT> (three bf16 MFMA passes per product; executed work)"),
                    "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                    "traffic": traffic, "traffic_source": traffic_src,
                    "launches_per_step": launches, "avg_launch_ms": round(gemm_ms / launches, 4), "kernel_ms_per_step": round(gemm_ms, 3),
                    "algorithmic_flops_per_step": alg_nt, "padded_flops_per_step": padded_flops,
                    "whole_step_tflops": round(3 * fwd * n / (ms_step * 1e-3) / 1e12, 1)}
        out = {"metric": "rays/sec (train step)", "value": round(rays_per_s, 1), "unit": "rays/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
               "dtype": args.compute, "data": "synthetic",
               "config": {"workload": f"S-NeRF path A (MipNerfModel) train step, nuScenes-like 1600x900 rays, {S0} proposal + {P1 - 1} fine evals/ray "
                                      f"({S0 + P1 - 1} spp), hidden 1024, rgb_layer 3, cone + contraction + IPE-96",
                          "rays_per_gpu_per_step": n, "global_rays_per_step": n * world, "parallelism": f"dp{world} (ray-sharded, flat-arena RCCL all-reduce overlapped with the backward)",
                          "train_flops_per_ray": 3 * fwd},
               "roofline": roofline, "final_loss": final_loss, "ms_per_step_median": round(ms_median, 3)}
        if args.graph:
            out["config"]["step"] = "hipGraph-captured (MipTrainer.capture / replay)"
        if comm is not None:
            out["comm"] = comm
        if strong is not None:
            out["strong_scaling"] = strong

    # ---- full-frame inference (forward only): 1600 x 900 rays through the drop-in render_image (models.py:328-360; eval.py:146),
    # one contiguous block of the frame per rank + one all-gather per output buffer
    if not args.no_frame:
        H, W = 900, 1600
        chunk = args.frame_chunk
        render_fn = lambda r: model(r, False, False, 0.)
        with torch.no_grad():
            render_fn(frame_rays(0, min(chunk, H * W), device))                      # warm-up chunk (packing, allocator)
            barrier()
            t0 = time.perf_counter()
            fr = frame_rays(0, H * W, device)                                         # ray generation on the device (80 MB), inside the timed region
            grid = Rays(*[r.reshape(H, W, -1) for r in fr])
            rgb_f, dist_f, acc_f, _ = render_image(render_fn, grid, rank, chunk=chunk, world=world)
            barrier()
            t_frame = time.perf_counter() - t0
        if world > 1:
            te = torch.tensor([t_frame], device=device, dtype=torch.float64)
            dist.all_reduce(te, op=dist.ReduceOp.MAX)
            t_frame = te.item()
        if args.ert is not None:
            with torch.no_grad():
                barrier()
                t1 = time.perf_counter()
                kept = tot = 0

                def render_ert(r):
                    nonlocal kept, tot
                    ret = model(r, False, False, 0., ert=tuple(args.ert))
                    kept += model.last_ert_rows[0]; tot += model.last_ert_rows[1]
                    return ret
                rgb_e, _, _, _ = render_image(render_ert, grid, rank, chunk=chunk, world=world)
                barrier()
                t_ert = time.perf_counter() - t1
                mse = float(((rgb_e - rgb_f) ** 2).mean())
            if rank == 0:
                out["frame_ert"] = {"eps_t": args.ert[0], "eps_w": args.ert[1], "ms_per_frame": round(t_ert * 1e3, 1),
                                    "fine_samples_evaluated": round(kept / max(tot, 1), 4),
                                    "psnr_vs_full_db": (float("inf") if mse == 0 else round(-10.0 * math.log10(mse), 2)),
                                    "note": "inference extension (csrc/ert.hip), not the reference's algorithm; sample count of the rank-0 block"}
        if rank == 0:
            out["ms_per_frame"] = round(t_frame * 1e3, 1)
            out["frame"] = {"resolution": "1600x900", "rays": H * W, "spp": S0 + P1 - 1, "chunk": chunk, "rays_per_s": round(H * W / t_frame, 1),
                            "tflops": round(fwd * H * W / t_frame / 1e12, 1),
                            "includes": "on-device ray generation, snerf_amd.mipnerf.render_image (chunk loop), all-gather of rgb / distance / acc"}
        del rgb_f, dist_f, acc_f, grid, fr

    # ---- the other two renderers of the reference on the same GPU (BASELINE configs 4-5 = path C, the classic render_rays = path B):
    # compact legs so that every config has a driver-timed number; the headline above stays path A (configs[1])
    if rank == 0 and world == 1 and not args.no_paths and args.compute == "bf16":
        torch.cuda.empty_cache()
        t0 = time.perf_counter()
        out["path_c"] = path_c_leg(device)
        out["path_c"]["leg_wall_s"] = round(time.perf_counter() - t0, 1)
        t0 = time.perf_counter()
        out["path_b"] = path_b_leg(device)
        out["path_b"]["leg_wall_s"] = round(time.perf_counter() - t0, 1)
        if not args.no_ert_scene:
            # north_star's "early ray termination and sample compaction" where it pays: path B's fine pass re-evaluates the 64
            # uniform coarse positions (render.py:380-389), a good part of which lies behind the first surface.  Two classic NeRFs fitted for
            # 500 steps to the analytic street scene, the 1600 x 900 frame plain and with render_rays(ert=(1e-4, (96, 16))) (exact bound on acc / rgb)
            sys.path.insert(0, os.path.join(REPO, "tools"))
            import ert_classic_analysis
            t0 = time.perf_counter()
            r = ert_classic_analysis.fit_and_measure(steps=500, eps_list=(1e-4,), rows_n=900, groups=((96, 16),), ert_eps=1e-4, dev=device, row0=0)
            m48 = r["measured"]["eps_0.0001_G96_16"]
            out["path_b_ert"] = {"what": "classic render_rays frame 1600 x 900 (64 + 128, two fitted NeRF 8 x 256), plain vs ert=(1e-4, (96, 16)): fine pass front to back -- the "
                                         "first 96 sorted samples in one piece, then groups of 16 --, rays leave at fine transmittance <= 1e-4, rows compacted (per-group scatter + transmittance + compaction: snerf_classic_ert_step, evaluation by fmlp_kernel)",
                                 "fit_steps": r["fit_steps"], "fit_psnr_db": r["fit_psnr_db"], "frame_ms_plain": r["window_ms_plain"], "frame_ms_ert": m48["window_ms"],
                                 "speedup": m48["speedup"], "fine_evaluations_kept": m48["fine_evaluations_kept"],
                                 "best_case_speedup_at_sample_granularity": r["rules"]["exact_eps_0.0001"]["best_case_frame_speedup"],
                                 "max_abs_err_rgb": m48["max_abs_err_rgb"], "max_abs_err_acc": m48["max_abs_err_acc"], "max_abs_err_depth": m48["max_abs_err_depth"],
                                 "leg_wall_s": round(time.perf_counter() - t0, 1)}
            torch.cuda.empty_cache()
        t0 = time.perf_counter()
        out["grid_encoder"] = grid_encoder_leg(device)
        out["grid_encoder"]["leg_wall_s"] = round(time.perf_counter() - t0, 1)

    # ---- early ray termination on a scene with real opacity (north_star: "early ray termination and sample compaction"):
    # a fresh model fitted for 200 steps to an analytic street scene (tools/ert_scene.py), its 1600 x 900 frame rendered plain and with
    # the front-to-back termination inside north_star's tolerance (eps_t = 1e-4: an exact bound on acc / rgb).  A labelled extra: the
    # headline frame above stays un-skipped.
    if rank == 0 and world == 1 and not args.no_ert_scene and args.compute == "bf16":
        sys.path.insert(0, os.path.join(REPO, "tools"))
        import ert_scene
        torch.cuda.empty_cache()
        t0 = time.perf_counter()
        if args.ert_scene and not args.no_frame:
            out["ert_scene"] = ert_scene.fit_and_render(build_model("bf16", device, seed=1), steps=120, eps=(1e-4, 0.0), group=16, rows=96, row0=400,
                                                        build=lambda mode: build_model(mode, device, seed=1))
            out["fitted_weights_precision"] = out["ert_scene"].pop("precision_on_fitted_weights")
            out["ert_scene"]["leg_wall_s"] = round(time.perf_counter() - t0, 1)
        else:       # the precision of every compute mode on FITTED weights (120 fit steps), without the two ERT frames
            out["fitted_weights_precision"] = ert_scene.fit_and_precision(build_model("bf16", device, seed=1), steps=120, build=lambda mode: build_model(mode, device, seed=1))
            out["fitted_weights_precision"]["leg_wall_s"] = round(time.perf_counter() - t0, 1)
        torch.cuda.empty_cache()

    # ---- the two routes a user of the reference takes besides MipTrainer.step: the unmodified train.py loop (autograd + torch losses +
    # torch.optim.Adam) and the shipped config's pose_refine = True (configs/nuScenes_depth_6cams:31: the step also back-propagates to
    # the rays, MipTrainer.step(..., ray_grads=True))
    if rank == 0 and world == 1 and not args.no_dropin and args.compute == "bf16":
        sd0 = {k: v.clone() for k, v in model.state_dict().items()}
        dt_d, loss_d = dropin_autograd_leg(sd0, rays, tgt, depth, conf, device, 5)
        out["dropin_autograd"] = {"rays_per_s": round(n / dt_d, 1), "ms_per_step": round(dt_d * 1e3, 3), "steps": 5, "final_loss": loss_d,
                                  "vs_headline": round((n / dt_d) / out["value"], 4),
                                  "what": "model(rays) -> torch losses -> loss.backward() -> torch.optim.Adam.step(): the loop of s-nerf/train.py:112-221 unmodified"}
        for _ in range(2):
            trainer.step(rays, tgt, depth, conf, ray_grads=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            trainer.step(rays, tgt, depth, conf, ray_grads=True)
        torch.cuda.synchronize()
        dt_p = (time.perf_counter() - t0) / 5
        out["pose_refine"] = {"rays_per_s": round(n / dt_p, 1), "ms_per_step": round(dt_p * 1e3, 3), "steps": 5, "vs_headline": round((n / dt_p) / out["value"], 4),
                              "what": "MipTrainer.step(..., ray_grads=True): the train step + d loss / d (origins, directions, viewdirs) for the pose optimiser"}
        torch.cuda.empty_cache()

    if rank == 0 and world == 1 and not args.no_dropin and args.compute == "bf16" and args.shape == "baseline":
        out["small_step"] = small_step_leg(device)
        out["small_step"]["strong_scaling_ceiling_at_8_gpus"] = round(out["ms_per_step"] / out["small_step"]["ms"], 2)
        out["shipped_shape"] = shipped_shape_leg(device)

    # ---- the fp32-parity mode's speed (north_star's 1e-4-relative contract holds in compute="f32": exact-fp32 MFMA 32x32x2, 157.3 TF peak)
    if rank == 0 and world == 1 and not args.no_f32 and args.compute == "bf16":
        m32 = build_model("f32", device)
        m32.load_state_dict({k: v.clone() for k, v in model.state_dict().items()})
        t32 = MipTrainer(m32, lr=5e-4)
        t32.step(rays, tgt, depth, conf)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            t32.step(rays, tgt, depth, conf)
        torch.cuda.synchronize()
        dt32 = (time.perf_counter() - t0) / 3
        l32, g32, _, alg32 = measure_gemm_kernel(t32, rays, tgt, depth, conf)
        a32 = alg32 / (g32 * 1e-3) / 1e12
        out["f32_mode"] = {"rays_per_s": round(n / dt32, 1), "ms_per_step": round(dt32 * 1e3, 2), "steps": 3,
                           "roofline": {"bound": "mfma", "kernel": "gemm_nt_kernel<f32,128,128,2,2> (v_mfma_f32_32x32x2_f32: exact fp32 FMA chain)",
                                        "achieved": round(a32, 2), "peak": 157.3, "unit": "TFLOP/s", "frac": round(a32 / 157.3, 4), "launches_per_step": l32},
                           "note": "the mode in which rgb / depth match the reference within 1e-4 relative (parity read-out below)"}
        del t32, m32
        torch.cuda.empty_cache()
        # the same contract at bf16 MFMA rates: split-bf16 operands (hi = bf16(x), lo = bf16(x - hi); hi.hi + lo.hi + hi.lo, fp32 accumulate)
        mx3 = build_model("bf16x3", device)
        mx3.load_state_dict({k: v.clone() for k, v in model.state_dict().items()})
        tx3 = MipTrainer(mx3, lr=5e-4)
        for _ in range(2):
            tx3.step(rays, tgt, depth, conf)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            tx3.step(rays, tgt, depth, conf)
        torch.cuda.synchronize()
        dtx3 = (time.perf_counter() - t0) / 5
        lx3, gx3, _, algx3 = measure_gemm_kernel(tx3, rays, tgt, depth, conf)
        ax3 = algx3 / (gx3 * 1e-3) / 1e12                                           # three bf16 MFMA passes per algorithmic product (counted in algx3)
        out["split_bf16_mode"] = {"rays_per_s": round(n / dtx3, 1), "ms_per_step": round(dtx3 * 1e3, 2), "steps": 5,
                                  "roofline": {"bound": "mfma", "kernel": "gemm_nt8p_kernel<.., SPLIT> (three v_mfma_f32_32x32x16_bf16 passes per product)",
                                               "achieved": round(ax3, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s (executed bf16 MFMA work)",
                                               "frac": round(ax3 / PEAK_BF16_TFLOPS, 4), "launches_per_step": lx3},
                                  "note": "compute='bf16x3': 16-bit mantissas through every GEMM; held to the same 1e-4 rgb / depth bounds as the exact-fp32 mode (parity read-out below)"}
        if not args.no_frame:                         # the 1600 x 900 frame in the mode whose renders are inside the 1e-4 contract
            with torch.no_grad():
                fn3 = lambda r: mx3(r, False, False, 0.)
                fn3(frame_rays(0, min(args.frame_chunk, 900 * 1600), device))
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                fr3 = frame_rays(0, 900 * 1600, device)
                render_image(fn3, Rays(*[r.reshape(900, 1600, -1) for r in fr3]), rank, chunk=args.frame_chunk, world=world)
                torch.cuda.synchronize()
                t3 = time.perf_counter() - t0
                del fr3
            out["split_bf16_mode"]["ms_per_frame"] = round(t3 * 1e3, 1)
            out["split_bf16_mode"]["frame_rays_per_s"] = round(900 * 1600 / t3, 1)
        del tx3, mx3
        torch.cuda.empty_cache()
        # the same three-pass FORWARD (bit-identical renders and losses) with a ONE-pass bf16 backward: compute="bf16x3_fwd"
        mxf = build_model("bf16x3_fwd", device)
        mxf.load_state_dict({k: v.clone() for k, v in model.state_dict().items()})
        txf = MipTrainer(mxf, lr=5e-4)
        for _ in range(2):
            txf.step(rays, tgt, depth, conf)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            txf.step(rays, tgt, depth, conf)
        torch.cuda.synchronize()
        dtxf = (time.perf_counter() - t0) / 5
        out["split_fwd_mode"] = {"rays_per_s": round(n / dtxf, 1), "ms_per_step": round(dtxf * 1e3, 2), "steps": 5,
                                 "speedup_vs_split_bf16_mode": round(dtx3 / dtxf, 3),
                                 "note": "compute='bf16x3_fwd': the split-bf16 forward (renders / losses bit-identical to compute='bf16x3', inside the 1e-4 contract) with "
                                         "single-pass bf16 data and weight gradients (hi halves of the saved activations); gradient error between the two pure modes "
                                         "(tests/test_paths.py::test_mipnerf_split_forward_plain_backward)"}
        del txf, mxf
        torch.cuda.empty_cache()
        # one fp16 pass (compute="fp16" on the mip path: 11 significant bits per operand instead of bf16's 8, per-layer launches, scaled fp16 backward)
        mh = build_model("fp16", device)
        mh.load_state_dict({k: v.clone() for k, v in model.state_dict().items()})
        th = MipTrainer(mh, lr=5e-4)
        for _ in range(2):
            th.step(rays, tgt, depth, conf)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            th.step(rays, tgt, depth, conf)
        torch.cuda.synchronize()
        dth = (time.perf_counter() - t0) / 5
        out["fp16_mode"] = {"rays_per_s": round(n / dth, 1), "ms_per_step": round(dth * 1e3, 2), "steps": 5,
                            "note": "compute='fp16': the bf16 step's GEMMs on the f16 MFMA (same rate); the 256- / 128-wide networks per layer (the fused register-resident "
                                    "kernels are bf16); precision on fitted weights in fitted_weights_precision.fp16"}
        del th, mh
        torch.cuda.empty_cache()
        # the same contract in TWO pass-equivalents: fp16 tiles + e4m3 correction tiles on the block-scaled MFMA (compute="f16f8"), scaled fp16 backward
        m8 = build_model("f16f8", device)
        m8.load_state_dict({k: v.clone() for k, v in model.state_dict().items()})
        t8 = MipTrainer(m8, lr=5e-4)
        for _ in range(2):
            t8.step(rays, tgt, depth, conf)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            t8.step(rays, tgt, depth, conf)
        torch.cuda.synchronize()
        dt8 = (time.perf_counter() - t0) / 5
        out["f16f8_mode"] = {"rays_per_s": round(n / dt8, 1), "ms_per_step": round(dt8 * 1e3, 2), "steps": 5, "speedup_vs_split_bf16_mode": round(dtx3 / dt8, 3),
                             "note": "compute='f16f8': x = fp16(x) + r; hi.hi on the fp16 MFMA + the correction r.w + x.(w - fp16(w)) as OCP e4m3 on v_mfma_scale_f32_32x32x64_f8f6f4 "
                                     "(twice the 16-bit rate): the forward in two pass-equivalents instead of three; backward = plain fp16 on power-of-two scaled gradients"}
        if not args.no_frame:
            with torch.no_grad():
                fn8 = lambda r: m8(r, False, False, 0.)
                fn8(frame_rays(0, min(args.frame_chunk, 900 * 1600), device))
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                fr8 = frame_rays(0, 900 * 1600, device)
                render_image(fn8, Rays(*[r.reshape(900, 1600, -1) for r in fr8]), rank, chunk=args.frame_chunk, world=world)
                torch.cuda.synchronize()
                t8f = time.perf_counter() - t0
                del fr8
            out["f16f8_mode"]["ms_per_frame"] = round(t8f * 1e3, 1)
            out["f16f8_mode"]["frame_rays_per_s"] = round(900 * 1600 / t8f, 1)
        del t8, m8
        torch.cuda.empty_cache()

    # ---- parity read-out + host-CPU baseline (rank 0, N = 1 only)
    if rank == 0 and world == 1 and not args.no_cpu:
        ncpu = min(args.cpu_rays, n)
        sd = {k: v.clone() for k, v in model.state_dict().items()}
        cpu_rps, cpu_s, rgb_ref, dist_ref, cpu_fwd_rps = cpu_baseline(ncpu, sd, rays)
        sub = Rays(*[r[:ncpu] for r in rays])
        with torch.no_grad():
            rb = model(sub, False, False, 0.)
            m32 = build_model("f32", device)
            m32.load_state_dict(sd)
            r32 = m32(sub, False, False, 0.)
            mx3 = build_model("bf16x3", device)
            mx3.load_state_dict(sd)
            rx3 = mx3(sub, False, False, 0.)
            del mx3
            m8 = build_model("f16f8", device)
            m8.load_state_dict(sd)
            r8 = m8(sub, False, False, 0.)
            del m8
        mse = lambda a, b: float(((a.double() - b.double()) ** 2).mean())
        psnr = lambda a, b: (float("inf") if mse(a, b) == 0 else -10.0 * math.log10(mse(a, b)))
        out["cpu_baseline"] = {"value": round(cpu_rps, 2), "unit": "rays/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": f"{ncpu} rays forward+backward, same network/shape, oracle (torch-CPU restatement of the reference), {cpu_s:.1f} s",
                               "forward_only_value": round(cpu_fwd_rps, 2)}
        out["parity"] = {"rays": ncpu,
                         "f32_kernels_vs_cpu_oracle_max_rel_err_rgb": float(((r32[1][0].cpu() - rgb_ref).abs() / (rgb_ref.abs() + 1e-3)).max()),
                         "f32_kernels_vs_cpu_oracle_max_rel_err_depth": float(((r32[1][1].cpu() - dist_ref).abs() / dist_ref.abs()).max()),
                         "split_bf16_kernels_vs_cpu_oracle_max_rel_err_rgb": float(((rx3[1][0].cpu() - rgb_ref).abs() / (rgb_ref.abs() + 1e-3)).max()),
                         "split_bf16_kernels_vs_cpu_oracle_max_rel_err_depth": float(((rx3[1][1].cpu() - dist_ref).abs() / dist_ref.abs()).max()),
                         "f16f8_kernels_vs_cpu_oracle_max_rel_err_rgb": float(((r8[1][0].cpu() - rgb_ref).abs() / (rgb_ref.abs() + 1e-3)).max()),
                         "f16f8_kernels_vs_cpu_oracle_max_rel_err_depth": float(((r8[1][1].cpu() - dist_ref).abs() / dist_ref.abs()).max()),
                         "psnr_f16f8_kernels_vs_cpu_oracle_db": psnr(r8[1][0].cpu(), rgb_ref),
                         "psnr_split_bf16_kernels_vs_cpu_oracle_db": psnr(rx3[1][0].cpu(), rgb_ref),
                         "psnr_f32_kernels_vs_cpu_oracle_db": psnr(r32[1][0].cpu(), rgb_ref),
                         "psnr_bf16_kernels_vs_cpu_oracle_db": psnr(rb[1][0].cpu(), rgb_ref)}
        del m32
    if rank == 0 and world == 1 and not args.no_eager:
        sd = {k: v.clone() for k, v in model.state_dict().items()}
        e32, ms32, f32_fwd = eager_baseline(sd, rays, n, 3, False)
        e16, ms16, f16_fwd = eager_baseline(sd, rays, n, 3, True)
        out["eager_baseline"] = {"unit": "rays/s", "fp32": round(e32, 1), "fp32_ms_per_step": round(ms32, 2), "bf16_autocast": round(e16, 1),
                                 "bf16_autocast_ms_per_step": round(ms16, 2), "rays_per_step": n, "steps": 3,
                                 "kind": "plain PyTorch-ROCm eager ops (torch restatement of the reference model), autograd + torch.optim.Adam, same GPU",
                                 "speedup_vs_fp32": round(out["value"] / e32, 2), "speedup_vs_bf16_autocast": round(out["value"] / e16, 2),
                                 "forward_only_fp32": round(f32_fwd, 1), "forward_only_bf16_autocast": round(f16_fwd, 1)}
        if "frame" in out:      # north_star: ">= 2x single-GPU rays/sec over the PyTorch-ROCm eager path on a 1600x900 frame at 192 samples/ray"
            out["eager_baseline"]["frame_speedup_vs_fp32"] = round(out["frame"]["rays_per_s"] / f32_fwd, 2)
            out["eager_baseline"]["frame_speedup_vs_bf16_autocast"] = round(out["frame"]["rays_per_s"] / f16_fwd, 2)
            if "frame_rays_per_s" in out.get("split_bf16_mode", {}):      # north_star's frame target in the mode that also holds its 1e-4
                out["eager_baseline"]["split_bf16_frame_speedup_vs_fp32"] = round(out["split_bf16_mode"]["frame_rays_per_s"] / f32_fwd, 2)
            if "frame_rays_per_s" in out.get("f16f8_mode", {}):
                out["eager_baseline"]["f16f8_frame_speedup_vs_fp32"] = round(out["f16f8_mode"]["frame_rays_per_s"] / f32_fwd, 2)
                out["eager_baseline"]["f16f8_frame_speedup_vs_bf16_autocast"] = round(out["f16f8_mode"]["frame_rays_per_s"] / f16_fwd, 2)
            out["eager_baseline"]["frame_note"] = ("the build's measured 1600 x 900 frame rate (render_image, ray generation and gathers included) over the eager forward's "
                                                   "rate on a 4096-ray chunk (eval.py's chunk size; a frame is 352 such chunks)")
    if rank == 0:
        print(json.dumps(out if args.verbose_json else compact_line(out)), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
