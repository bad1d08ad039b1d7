#!/usr/bin/env python3
"""Golden vectors for the OPTIONAL appearance embedding of the live mip path (s-nerf/model/models.py:57,63-64 `emb = Embedding(N_vocab,
48)`, :153-159 condition = cat([view encoding, emb(rays.app)]); utils/arg_parser.py:222 --encode_appearance): the reference's
MipNerfModel with encode_appearance=True on seeded rays (per-ray image indices in rays.app) and formula weights, outputs and ALL
parameter gradients (incl. emb.weight) -> tests/golden/g22_mipnerf_appearance.npz.  Build-container only (needs /root/reference)."""
import os
import sys

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
    V, n = 6, 24
    model = models.MipNerfModel(n_samples=16, N_fine=17, no_warp_sample=0, ray_shape="cone", fn=1, radius=3., transform_idx=0, real=True,
                                rgb_layer=3, hidden_layer=64, density_noise=0., max_deg_point=16, proposal_hidden_layer=64,
                                proposal_loss=True, encode_appearance=True, N_vocab=V)
    sd = common.fill_state_dict_(model.state_dict())
    model.load_state_dict(sd)
    rays = common.synthetic_rays(n, seed=11)
    g = torch.Generator().manual_seed(12)
    rays["app"] = torch.randint(0, V, (n, 1), generator=g).float()           # image index of every ray (sample_utils.py:206)
    from collections import namedtuple
    Rays = namedtuple("Rays", ("origins", "directions", "viewdirs", "radii", "lossmult", "near", "far", "app"))
    target = torch.rand(n, 3, generator=g)
    ret = model(Rays(**rays), False, False, 0.)
    loss = ((ret[1][0] - target) ** 2).mean() + 0.01 * ret[0][1].mean() + 0.05 * (1.0 / ret[1][1]).mean()
    loss.backward()
    out = dict(**{f"rays_{k}": v for k, v in rays.items()}, target=target, l1_rgb=ret[1][0], l1_distance=ret[1][1], l1_acc=ret[1][2],
               l0_distance=ret[0][1], loss=loss.detach(), param_names=np.array(list(sd.keys())))
    for k, v in model.named_parameters():
        out["grad." + k] = v.grad
    arr = {k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in out.items()}
    _oracle_common.save_golden(os.path.join(REPO, "tests", "golden", "g22_mipnerf_appearance.npz"), **arr)
    print("wrote g22_mipnerf_appearance.npz", sum(a.nbytes for a in arr.values()), "bytes; params:", [k for k in sd if "emb" in k or "cond_layers.0" in k])


if __name__ == "__main__":
    main()
