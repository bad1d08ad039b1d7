#!/usr/bin/env python3
"""Golden vectors for the GLO vectors of the zipnerf Model (s-nerfpp/zipnerf/internal/models.py:44-45,75-77 `glo_vecs =
Embedding(num_glo_embeddings, num_glo_features)`, :131-139 per-ray lookup by batch['cam_idx'] or zeros with `zero_glo`, :454-459 +
:620-630 NerfMLP.lin_glo_0 / lin_glo_1 -> (scale, shift) -> bottleneck * exp(scale) + shift; configs/360_glo4.gin): the reference Model
of g11 (same grids and rays) with num_glo_features = 4, forward with the embedding and with zero_glo, and the gradients of every
dense parameter (+ the embedding rows) of a loss on the final rgb / depth -> tests/golden/g25_zip_glo.npz.  Build-container only."""
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

F_GLO, N_EMB = 4, 12
KEEP = ("nerf_mlp.density_layer.2.weight", "nerf_mlp.density_layer.2.bias", "nerf_mlp.density_layer.0.bias", "nerf_mlp.lin_second_stage_0.bias",
        "nerf_mlp.rgb_layer.weight", "prop_mlp_1.density_layer.0.bias")        # (the other dense gradients are covered by g11's tests)


def main():
    coord, rmath, models, render, stepfun = gz.import_reference()
    specs = gz.small_specs()
    cfg = types.SimpleNamespace(use_semantic=False, vis_num_rays=8, zero_glo=False)
    torch.manual_seed(0)
    model = models.Model(config=cfg, raydist_fn='power_transformation', opaque_background=True, num_glo_features=F_GLO, num_glo_embeddings=N_EMB)
    model.nerf_mlp = models.NerfMLP(disable_density_normals=True, deg_view=1, grid_log2_hashmap_size=14, use_semantic=False,
                                    num_glo_features=F_GLO, num_glo_embeddings=N_EMB)
    model.prop_mlp_0 = models.PropMLP(disable_density_normals=True, disable_rgb=True, grid_level_dim=1, grid_disired_resolution=512, grid_log2_hashmap_size=14)
    model.prop_mlp_1 = models.PropMLP(disable_density_normals=True, disable_rgb=True, grid_level_dim=1, grid_disired_resolution=2048, grid_log2_hashmap_size=14)
    shapes = oz.param_shapes(specs, num_glo_features=F_GLO, num_glo_embeddings=N_EMB, zero_glo=False)
    sd_ref = {k: v for k, v in model.state_dict().items() if k.endswith(("weight", "bias", "embeddings"))}
    assert [k for k, _ in shapes] == list(sd_ref.keys()), (list(sd_ref.keys()), [k for k, _ in shapes])
    assert all(tuple(sd_ref[k].shape) == tuple(s) for k, s in shapes)
    p = gz.formula_params(shapes)
    g = torch.Generator().manual_seed(31)
    p["glo_vecs.weight"] = torch.randn(N_EMB, F_GLO, generator=g) * 0.5
    p["nerf_mlp.lin_glo_0.weight"] = torch.randn(128, F_GLO, generator=g) * 0.4            # (the formula weights are rank 2: full-rank here)
    p["nerf_mlp.lin_glo_1.weight"] = torch.randn(512, 128, generator=g) * 0.05
    model.load_state_dict(p, strict=False)
    batch = gz.make_batch(20, 11)
    batch["cam_idx"] = torch.randint(0, N_EMB, (20, 1), generator=g)
    target = torch.rand(20, 3, generator=g)
    d = {"b_" + k: v for k, v in batch.items()}
    d.update(glo_vecs=p["glo_vecs.weight"], lin_glo_0_weight=p["nerf_mlp.lin_glo_0.weight"], lin_glo_1_weight=p["nerf_mlp.lin_glo_1.weight"], target=target,
             num_glo_features=np.int32(F_GLO), num_glo_embeddings=np.int32(N_EMB))
    for tag, zero in (("emb", False), ("zero", True)):
        model.zero_grad()
        rend, hist = model(None, dict(batch), train_frac=1.0, compute_extras=False, zero_glo=zero)
        loss = ((rend[-1]["rgb"] - target) ** 2).mean() + 0.01 * rend[-1]["depth"].mean()
        loss.backward()
        d[tag + "_rgb"], d[tag + "_depth"], d[tag + "_loss"] = rend[-1]["rgb"], rend[-1]["depth"], loss.detach()
        for lvl in range(3):
            d[f"{tag}_sdist{lvl}"] = hist[lvl]["sdist"]
        for k, v in model.named_parameters():
            if v.grad is not None and ("glo" in k or k in KEEP):
                d[f"{tag}_grad.{k}"] = v.grad.clone()
    arr = {k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()}
    _oracle_common.save_golden(os.path.join(REPO, "tests", "golden", "g25_zip_glo.npz"), **arr)
    print("wrote g25_zip_glo.npz", sum(a.nbytes for a in arr.values()), "bytes;", [k for k in arr if "glo" in k and "grad" in k])


if __name__ == "__main__":
    main()
