#!/usr/bin/env python3
"""Golden vectors for the OPTIONAL semantic head of the live mip path (s-nerf/model/models.py:256-261,279-282 MLP.semantic_layer,
mip.py:175-176 semantic = sum w raw_semantic): the reference's MipNerfModel with semantic=True on seeded rays and formula weights,
outputs and parameter gradients of a loss that includes the semantic rendering -> tests/golden/g14_mipnerf_semantic.npz.
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
from oracle import common  # noqa: E402
from oracle.gen_golden import _import_reference  # noqa: E402


def main():
    _, _, models, _, _ = _import_reference()
    torch.manual_seed(0)
    C = 7
    model = models.MipNerfModel(n_samples=16, N_fine=17, no_warp_sample=0, ray_shape="cone", fn=1, radius=3., transform_idx=0, real=True,
                                rgb_layer=3, hidden_layer=64, density_noise=0., max_deg_point=16, proposal_hidden_layer=64,
                                proposal_loss=True, semantic=True, semantic_class_num=C)
    sd = common.fill_state_dict_(model.state_dict())
    model.load_state_dict(sd)
    rays = common.synthetic_rays(24, seed=4)
    from collections import namedtuple
    Rays = namedtuple("Rays", ("origins", "directions", "viewdirs", "radii", "lossmult", "near", "far", "app"))
    g = torch.Generator().manual_seed(9)
    target, semw = torch.rand(24, 3, generator=g), torch.randn(24, C, generator=g)
    ret = model(Rays(**rays), False, False, 0.)
    loss = ((ret[1][0] - target) ** 2).mean() + 0.01 * ret[0][1].mean() + 0.2 * (ret[1][3] * semw).sum() / 24
    loss.backward()
    out = dict(**{f"rays_{k}": v for k, v in rays.items()}, target=target, semw=semw, l1_rgb=ret[1][0], l1_distance=ret[1][1], l1_semantic=ret[1][3],
               l0_distance=ret[0][1], loss=loss.detach(), param_names=np.array(list(sd.keys())))
    for k, v in model.named_parameters():
        out["grad." + k] = v.grad
    arr = {k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in out.items()}
    _oracle_common.save_golden(os.path.join(REPO, "tests", "golden", "g14_mipnerf_semantic.npz"), **arr)
    print("wrote g14_mipnerf_semantic.npz", sum(a.nbytes for a in arr.values()), "bytes; semantic params:", [k for k in sd if "semantic" in k])


if __name__ == "__main__":
    main()
