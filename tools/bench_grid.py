#!/usr/bin/env python3
"""Stand-alone GridEncoder timings (bench.py's grid_encoder leg, plus probes): `--sweep` times the binned table gradient over workspace
sizes (chunks per level / levels per transposed group follow from the size) and in both gradient layouts, and the forward in its two
forms; `--once` runs one forward + backward (for rocprofv3); `--truth` adds the error of the fast and the atomic gradients against an
fp32-accumulated gradient at full size."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sweep", action="store_true")
    ap.add_argument("--once", action="store_true")
    args = ap.parse_args()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    if args.once:
        bench.grid_encoder_leg(dev, steps=1)
        return
    print(json.dumps(bench.grid_encoder_leg(dev)))
    if args.sweep:
        from snerf_amd import ops
        from snerf_amd.gridencoder import GridEncoder
        for C, L, des in ((4, 10, 8192), (1, 8, 2048)):                  # ray-ordered points of the leg's geometry, both encoders of the reference
            x = (bench.grid_points(dev) + 1) / 2
            B = x.shape[0]
            enc = GridEncoder(input_dim=3, num_levels=L, level_dim=C, base_resolution=16, desired_resolution=des, log2_hashmap_size=21, device=dev)
            S, H = float(np.log2(enc.per_level_scale)), 16
            oh = enc.offsets.cpu().numpy()
            g = torch.Generator(device=dev).manual_seed(4)
            for dt in (torch.float16, torch.float32):
                if C == 1 and dt == torch.float16:
                    continue                                              # (odd C: the reference does not halve the table, grid.py:41-44)
                w = (torch.randn(B, L * C, generator=g, device=dev) * 1e-3).to(dt)
                w_lm = w.reshape(B, L, C).permute(1, 0, 2).contiguous()
                half = dt == torch.float16
                for ws in (256 << 20, 512 << 20, None, 2 << 30, 4 << 30, 24 << 30):
                    for lm in (False, True):
                        try:
                            plan = ops.grid_encode_bwd_binned_plan(B, C, L, oh, half, dt, level_major=lm, ws_bytes=ws)
                        except Exception as e:
                            print(f"C={C} {dt} ws={ws}: {e}")
                            continue
                        f = lambda: ops.grid_encode_bwd_binned(w_lm if lm else w, x, enc.offsets, C, L, S, H, out_dtype=dt, level_major=lm, offsets_host=oh, ws_bytes=ws)
                        ms = bench._timeit(f, 3, warm=1) * 1e3
                        print(f"C={C} L={L} grad {str(dt)[6:]} {'[L,B,C]' if lm else '[B,L*C]'} ws {plan['ws_bytes'] / 2**20:8.0f} MiB (used {plan['bytes_used'] / 2**20:7.0f}): "
                              f"{plan['chunks']:3d} chunks x {plan['chunk_points']:9d} points, {plan['levels_per_transposed_group']:2d} levels per transposed group, "
                              f"{plan['launches']:4d} launches: bwd {ms:7.3f} ms")
                del w, w_lm
            for ref in (True, False):
                def f():
                    with torch.no_grad():
                        ops.grid_encode_fwd(x, enc.embeddings.data.half() if C == 4 else enc.embeddings.data, enc.offsets, L, S, H, 0, False, 0, reference_form=ref)
                print(f"C={C} L={L} forward, {'reference form' if ref else 'fast'}: {bench._timeit(f, 3, warm=1) * 1e3:.3f} ms")
            del enc, x
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
