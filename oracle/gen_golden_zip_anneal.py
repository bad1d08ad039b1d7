#!/usr/bin/env python3
"""Golden vectors for the near-bound annealing of the zipnerf Model (s-nerfpp/zipnerf/internal/models.py:47-48 near_anneal_rate /
near_anneal_init, :147-158 the initial interval [init_s_near, 1], :170-171 dilation scaled by its length, :182-186 / :207-213 the
resampling domain): the reference Model of g11 (same grids, formula weights and rays) with near_anneal_rate = 0.5 at train_frac =
0.2 (init_s_near = 0.6) and at train_frac = 0.01 (clipped to near_anneal_init = 0.95) -> tests/golden/g24_zip_near_anneal.npz.
Build-container only (needs /root/reference)."""
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


def main():
    coord, rmath, models, render, stepfun = gz.import_reference()
    specs = gz.small_specs()
    cfg = types.SimpleNamespace(use_semantic=False, vis_num_rays=8, zero_glo=True)
    torch.manual_seed(0)
    model = models.Model(config=cfg, raydist_fn='power_transformation', opaque_background=True)
    model.nerf_mlp = models.NerfMLP(disable_density_normals=True, deg_view=1, grid_log2_hashmap_size=14, use_semantic=False)
    model.prop_mlp_0 = models.PropMLP(disable_density_normals=True, disable_rgb=True, grid_level_dim=1, grid_disired_resolution=512, grid_log2_hashmap_size=14)
    model.prop_mlp_1 = models.PropMLP(disable_density_normals=True, disable_rgb=True, grid_level_dim=1, grid_disired_resolution=2048, grid_log2_hashmap_size=14)
    model.load_state_dict(gz.formula_params(oz.param_shapes(specs)), strict=False)
    model.near_anneal_rate, model.near_anneal_init = 0.5, 0.95
    batch = gz.make_batch(20, 11)
    d = {"b_" + k: v for k, v in batch.items()}
    d.update(near_anneal_rate=np.float32(0.5), near_anneal_init=np.float32(0.95))
    for tag, tf in (("a", 0.2), ("b", 0.01)):
        with torch.no_grad():
            rend, hist = model(None, dict(batch), train_frac=tf, compute_extras=False)
        d[tag + "_train_frac"] = np.float32(tf)
        for lvl in range(3):
            d[f"{tag}_sdist{lvl}"] = hist[lvl]["sdist"]; d[f"{tag}_weights{lvl}"] = hist[lvl]["weights"]
        d[tag + "_rgb"], d[tag + "_depth"] = rend[-1]["rgb"], rend[-1]["depth"]
    arr = {k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()}
    _oracle_common.save_golden(os.path.join(REPO, "tests", "golden", "g24_zip_near_anneal.npz"), **arr)
    print("wrote g24_zip_near_anneal.npz", sum(a.nbytes for a in arr.values()), "bytes; first posts:", float(d["a_sdist0"][0, 0]), float(d["b_sdist0"][0, 0]))


if __name__ == "__main__":
    main()
