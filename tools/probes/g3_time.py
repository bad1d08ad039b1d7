#!/usr/bin/env python3
"""time snerf_grid_encode_bwd_binned at the grid_encoder leg's size with the library SNERF_HIP_LIB names (ablation builds)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from snerf_amd import ops
from snerf_amd.gridencoder import GridEncoder
dev = torch.device("cuda", 0)
L, C = 10, 4
x = (bench.grid_points(dev) + 1) / 2
B = x.shape[0]
enc = GridEncoder(input_dim=3, num_levels=L, level_dim=C, base_resolution=16, desired_resolution=8192, log2_hashmap_size=21, device=dev)
S, H = float(np.log2(enc.per_level_scale)), 16
oh = enc.offsets.cpu().numpy()
g = torch.Generator(device=dev).manual_seed(4)
w = (torch.randn(B, L * C, generator=g, device=dev) * 1e-3).half()
for ws in (None, 4 << 30):
    f = lambda: ops.grid_encode_bwd_binned(w, x, enc.offsets, C, L, S, H, out_dtype=torch.float16, offsets_host=oh, ws_bytes=ws)
    print(os.environ.get("SNERF_HIP_LIB", "shipped"), "ws", ws, f"bwd {bench._timeit(f, 3, warm=1) * 1e3:.3f} ms")
