#!/usr/bin/env python3
"""Golden vectors for the fp16 compute mode of path C (BASELINE configs[3] "fp16 MLP": s-nerfpp/zipnerf/train.py:215
`with accelerator.autocast():` around Model.forward, internal/models.py:772 for rendering): the reference Model of g11 (same grids, same
rays) run under torch.autocast('cpu', dtype=torch.float16) -- every nn.Linear computes on fp16 operands and returns fp16, the rest of the
renderer follows autocast's per-op policy -- and in fp32 on the same parameters, forward plus the gradients of the dense parameters of
a loss on the final rgb / depth (the loss is multiplied by 4096 before backward() and the gradients divided back, like
accelerate's GradScaler does) -> tests/golden/g26_zip_fp16.npz.  The linear layers carry full-rank random weights (g11's formula
weights are rank 2 and amplify any rounding to O(1)).  Build-container only."""
import os
import sys
import types

sys.dont_write_bytecode = True
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
if REPO not in sys.path:
    sys.path.insert(0, REPO)
from oracle import common as _oracle_common  # noqa: E402  (save_golden: writes the fixture, or compares under --check)
sys.path.insert(0, REPO)
from oracle import zip as oz  # noqa: E402
from oracle import gen_golden_zip as gz  # noqa: E402

SCALE = 4096.0


def main():
    coord, rmath, models, render, stepfun = gz.import_reference()
    specs = gz.small_specs()
    cfg = types.SimpleNamespace(use_semantic=False, vis_num_rays=8, zero_glo=True)
    torch.manual_seed(0)
    model = models.Model(config=cfg, raydist_fn='power_transformation', opaque_background=True)
    model.nerf_mlp = models.NerfMLP(disable_density_normals=True, deg_view=1, grid_log2_hashmap_size=14, use_semantic=False)
    model.prop_mlp_0 = models.PropMLP(disable_density_normals=True, disable_rgb=True, grid_level_dim=1, grid_disired_resolution=512, grid_log2_hashmap_size=14)
    model.prop_mlp_1 = models.PropMLP(disable_density_normals=True, disable_rgb=True, grid_level_dim=1, grid_disired_resolution=2048, grid_log2_hashmap_size=14)
    shapes = oz.param_shapes(specs)
    p = gz.formula_params(shapes)
    g = torch.Generator().manual_seed(41)
    d = {}
    for k, s in shapes:
        if k.endswith(".weight"):
            p[k] = torch.randn(*s, generator=g) * (1.2 / s[1] ** 0.5)                  # full rank, activations stay O(1)
            d["w." + k] = p[k]
        elif k.endswith(".bias"):
            p[k] = torch.randn(*s, generator=g) * 0.1
            d["w." + k] = p[k]
    model.load_state_dict(p, strict=False)
    batch = gz.make_batch(20, 11)
    target = torch.rand(20, 3, generator=g)
    d.update({"b_" + k: v for k, v in batch.items()})
    d["target"] = target
    for tag, half in (("f32", False), ("f16", True)):
        model.zero_grad()
        with torch.autocast("cpu", dtype=torch.float16, enabled=half):
            rend, hist = model(None, dict(batch), train_frac=1.0, compute_extras=False)
        rgb, depth = rend[-1]["rgb"].float(), rend[-1]["depth"].float()
        loss = ((rgb - target) ** 2).mean() + 0.01 * depth.mean()
        (loss * SCALE).backward()
        d[tag + "_rgb"], d[tag + "_depth"], d[tag + "_loss"] = rgb, depth, loss.detach()
        d[tag + "_dtypes"] = np.array([str(rend[-1]["rgb"].dtype), str(hist[-1]["weights"].dtype)])
        for lvl in range(3):
            d[f"{tag}_sdist{lvl}"] = hist[lvl]["sdist"].float()
            d[f"{tag}_weights{lvl}"] = hist[lvl]["weights"].float()
        for k, v in model.named_parameters():
            if v.grad is not None and not k.endswith("embeddings"):
                d[f"{tag}_grad.{k}"] = v.grad.float() / SCALE
    arr = {k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()}
    _oracle_common.save_golden(os.path.join(REPO, "tests", "golden", "g26_zip_fp16.npz"), **arr)
    print("wrote g26_zip_fp16.npz", sum(a.nbytes for a in arr.values() if a.dtype.kind != "U"), "bytes")
    print("reference fp16 vs fp32: max |d rgb|", float(np.abs(arr["f16_rgb"] - arr["f32_rgb"]).max()), " max rel d depth",
          float((np.abs(arr["f16_depth"] - arr["f32_depth"]) / np.abs(arr["f32_depth"])).max()), arr["f16_dtypes"])
    for k in arr:
        if k.startswith("f32_grad."):
            a, b = arr[k], arr["f16" + k[3:]]
            print(f"  {k[9:]:45s} |g| {np.linalg.norm(a):.3e}  fp16 vs fp32 rel L2 {np.linalg.norm(a - b) / (np.linalg.norm(a) + 1e-30):.3e}")


if __name__ == "__main__":
    main()
