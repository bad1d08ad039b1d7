#!/usr/bin/env python3
"""Golden vectors for the ray gradients of the live mip path (the reference's pose refinement, s-nerf/utils/sample_utils.py:410-435):
the reference's MipNerfModel on seeded rays that require grad (near 0.5 / far 30: samples on both sides of the contraction radius),
loss on both levels, and d loss / d (origins, directions, viewdirs) from its autograd -> tests/golden/g19_mipnerf_raygrad.npz.
Build-container only (needs /root/reference)."""
import os
import sys
from collections import namedtuple

sys.dont_write_bytecode = True
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
if REPO not in sys.path:
    sys.path.insert(0, REPO)
from oracle import common as _oracle_common  # noqa: E402  (save_golden: writes the fixture, or compares under --check)
sys.path.insert(0, REPO)
from oracle import common  # noqa: E402
from oracle.gen_golden import _import_reference  # noqa: E402


def main():
    _, _, models, _, _ = _import_reference()
    torch.manual_seed(0)
    model = models.MipNerfModel(n_samples=16, N_fine=17, no_warp_sample=0, ray_shape="cone", fn=1, radius=3., transform_idx=0, real=True,
                                rgb_layer=3, hidden_layer=64, density_noise=0., max_deg_point=16, proposal_hidden_layer=64,
                                proposal_loss=True)
    sd = common.fill_state_dict_(model.state_dict())
    model.load_state_dict(sd)
    n = 24
    rays = common.synthetic_rays(n, seed=6)
    rays["near"], rays["far"] = torch.full_like(rays["near"], 0.5), torch.full_like(rays["far"], 30.0)
    leaves = {k: rays[k].clone().requires_grad_(True) for k in ("origins", "directions", "viewdirs")}
    Rays = namedtuple("Rays", ("origins", "directions", "viewdirs", "radii", "lossmult", "near", "far", "app"))
    g = torch.Generator().manual_seed(10)
    target, tdepth = torch.rand(n, 3, generator=g), torch.rand(n, generator=g) * 20 + 2
    ret = model(Rays(**{**rays, **leaves}), False, False, 0.)
    loss = (((ret[1][0] - target) ** 2).mean() + 0.2 * ((1 / ret[1][1] - 1 / tdepth).abs()).mean() + 0.04 * ((1 / ret[0][1] - 1 / tdepth).abs()).mean()
            + 0.01 * (ret[0][4] ** 2).sum() + 0.01 * ret[1][2].mean())
    loss.backward()
    out = dict(**{f"rays_{k}": v for k, v in rays.items()}, target=target, target_depth=tdepth, loss=loss.detach(), l1_rgb=ret[1][0], l1_distance=ret[1][1],
               l0_distance=ret[0][1], **{"grad_" + k: v.grad for k, v in leaves.items()})
    arr = {k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in out.items()}
    _oracle_common.save_golden(os.path.join(REPO, "tests", "golden", "g19_mipnerf_raygrad.npz"), **arr)
    print("wrote g19_mipnerf_raygrad.npz", {k: float(np.abs(arr["grad_" + k]).max()) for k in leaves})


if __name__ == "__main__":
    main()
