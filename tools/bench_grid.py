#!/usr/bin/env python3
"""Stand-alone GridEncoder timings (bench.py's grid_encoder leg, plus probes): `--sweep` times the forward with 0 (reference form) /
1 / 2 / 4 / 8 points per thread on ray-ordered and on uniformly random points; `--once` runs one forward + backward (for rocprofv3)."""
import argparse
import json
import os
import sys

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
            x = bench.grid_points(dev)
            enc = GridEncoder(input_dim=3, num_levels=L, level_dim=C, base_resolution=16, desired_resolution=des, log2_hashmap_size=21, device=dev)
            for G, name in ((32, "level-major"), (16, "point-major")):
                ops.grid_set_fast_path(G)

                def f():
                    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
                        enc(x, bound=1)
                print(f"C={C} L={L} ray-ordered points, {name} mapping: fwd {bench._timeit(f, 3, warm=1) * 1e3:.3f} ms")
        ops.grid_set_fast_path(1)
        from snerf_amd.gridencoder import GridEncoder
        g = torch.Generator(device=dev).manual_seed(3)
        B = 65536 * 32 * 7
        for C in (4, 1):
            enc = GridEncoder(input_dim=3, num_levels=10 if C == 4 else 8, level_dim=C, base_resolution=16, desired_resolution=8192 if C == 4 else 2048,
                              log2_hashmap_size=21, device=dev)
            x = torch.rand(B, 3, device=dev, generator=g) * 2 - 1
            for G in (0, 32, 16, 2, 8):
                ops.grid_set_fast_path(G)

                def f():
                    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
                        enc(x, bound=1)
                print(f"C={C} uniformly random points, points per thread {G} (0 = reference form, 32 = level-major, 16 = point-major mapping): fwd {bench._timeit(f, 3, warm=1) * 1e3:.3f} ms")
            ops.grid_set_fast_path(1)


if __name__ == "__main__":
    main()
