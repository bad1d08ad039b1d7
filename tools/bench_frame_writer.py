#!/usr/bin/env python3
"""Frame-writer throughput on the GPU box (row 8f-3): a 1920x1280 frame (rgb + depth + 19-class semantic) through
ops.frame_quantize + the native PNG writer, against the reference's route (full float buffers to the host, numpy quantisation,
PIL save) on the same synthetic frame."""
import io, json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from PIL import Image


def main():
    from snerf_amd import frame_writer as fw, ops
    H, W, C = 1280, 1920, 19
    dev = torch.device("cuda", 0)
    yy, xx = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing="ij")
    g = torch.Generator(device=dev).manual_seed(0)
    # a smooth "render" with sensor-like noise: compresses like a photograph, not like white noise
    rgb = torch.stack([(torch.sin(xx / 90.0) + 1) / 2, (yy / H) ** 1.5, ((xx + yy) % 400) / 400.0], -1) * 0.9 + torch.rand(H, W, 3, device=dev, generator=g) * 0.04
    depth = 3.0 + 60.0 * (yy / H) + torch.rand(H, W, device=dev, generator=g) * 0.05
    sem = torch.rand(H, W, C, device=dev, generator=g) * 0.1
    sem[..., 3] += (yy > H // 2).float(); sem[..., 11] += (xx % 500 < 120).float() * 0.7
    cmap = np.array([[(i * 883011) // 65536 % 256, (i * 883011) // 256 % 256, (i * 883011) % 256] for i in range(C)], np.uint8)
    res = {"frame": f"{W}x{H}, rgb + depth + {C}-class semantic", "host_cores": os.cpu_count()}
    for threads in (1, 8, 16):
        with tempfile.TemporaryDirectory() as td:
            w = fw.FrameWriter(td, scale_factor=0.5, color_map=cmap, threads=threads)
            w.write(0, dict(rgb=rgb, depth=depth, semantic=sem)); w.close()          # warm
            w = fw.FrameWriter(td, scale_factor=0.5, color_map=cmap, threads=threads)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            n = 4
            for i in range(n):
                w.write(i, dict(rgb=rgb, depth=depth, semantic=sem))
            t_enq = time.perf_counter() - t0
            w.close(); dt = (time.perf_counter() - t0) / n
            res[f"native_threads{threads}_ms_per_frame"] = round(dt * 1e3, 1)
            res[f"native_threads{threads}_enqueue_ms_per_frame"] = round(t_enq / n * 1e3, 2)
            res["png_bytes"] = {k: os.path.getsize(os.path.join(td, k, "00000.png")) for k in ("rgb", "depth", "semantic", "paint")}
    # the reference's route
    with tempfile.TemporaryDirectory() as td:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = {k: v.detach().cpu().numpy() for k, v in dict(rgb=rgb, depth=depth, semantic=sem).items()}
        t_copy = time.perf_counter() - t0
        # the reference's expressions (random_render_waymo_seq.py:214-227, internal/utils.py:111-116)
        labels = np.argmax(r["semantic"], axis=-1)
        q = dict(rgb=(np.clip(np.nan_to_num(r["rgb"]), 0., 1.) * 255.).astype(np.uint8), depth=(r["depth"] * 256 / 0.5).astype(np.uint16),
                 semantic=labels.astype(np.uint8), paint=cmap[labels].astype(np.uint8))
        for k, a in q.items():
            Image.fromarray(a).save(os.path.join(td, k + ".png"))
        res["reference_route_ms_per_frame"] = round((time.perf_counter() - t0) * 1e3, 1)
        res["reference_route_d2h_ms"] = round(t_copy * 1e3, 1)
        for k, a in q.items():
            assert np.array_equal(np.array(Image.open(os.path.join(td, k + ".png"))), a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ops.frame_quantize(rgb, depth, sem, torch.from_numpy(cmap).to(dev), 0.5)
    e0.record(); ops.frame_quantize(rgb, depth, sem, torch.from_numpy(cmap).to(dev), 0.5); e1.record(); torch.cuda.synchronize()
    px = H * W
    res["quantize_kernel_ms"] = round(e0.elapsed_time(e1), 4)
    res["quantize_kernel_GBps"] = round(px * ((3 + 1 + C) * 4 + 9) / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
