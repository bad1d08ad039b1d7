#!/usr/bin/env python3
"""Golden vectors for the view-centred warp of the live mip path (`fn = 0`: s-nerf/model/mip.py:367-378 warp_fn -> fn1 + Jacobi_f,
:323-340; `viewc` = mean camera centre, train.py:36 / eval.py:50): the reference's sample2enc with fn_idx = 0 on seeded rays, and its
MipNerfModel(fn=0) end to end (outputs + every parameter gradient) -> tests/golden/g23_warp0.npz.  Build-container only."""
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
    _, mip, models, _, _ = _import_reference()
    torch.manual_seed(0)
    n, S = 24, 16
    rays = common.synthetic_rays(n, seed=21)
    g = torch.Generator().manual_seed(22)
    rays["far"] = rays["far"] * (0.5 + torch.rand(n, 1, generator=g))            # per-ray far: fn1 divides by it, Jacobi_f by its maximum
    viewc = torch.tensor([0.3, -0.2, 0.1])
    sv = torch.sort(torch.rand(n, S + 1, generator=g), -1)[0]
    sv[:, 0] = 0; sv[:, -1] = 1
    out = {}
    for shape in ("cone", "cylinder"):
        fm, fc = mip.sample2enc(sv, rays["origins"], rays["directions"], rays["radii"], shape, rays["near"], rays["far"], S, 0, viewc=viewc,
                                radius=3., transform_idx=0)
        out[f"{shape}_f_means"], out[f"{shape}_f_covs"] = fm, fc
    model = models.MipNerfModel(n_samples=16, N_fine=17, no_warp_sample=0, ray_shape="cone", fn=0, radius=3., transform_idx=0, real=True,
                                rgb_layer=3, hidden_layer=64, density_noise=0., max_deg_point=16, proposal_hidden_layer=64,
                                proposal_loss=True)
    sd = common.fill_state_dict_(model.state_dict())
    model.load_state_dict(sd)
    from collections import namedtuple
    Rays = namedtuple("Rays", tuple(rays.keys()))
    target = torch.rand(n, 3, generator=g)
    ret = model(Rays(**rays), False, False, viewc)
    loss = ((ret[1][0] - target) ** 2).mean() + 0.01 * ret[0][1].mean() + 0.05 * (1.0 / ret[1][1]).mean()
    loss.backward()
    out.update(**{f"rays_{k}": v for k, v in rays.items()}, s_vals=sv, viewc=viewc, target=target, l1_rgb=ret[1][0], l1_distance=ret[1][1],
               l1_acc=ret[1][2], l0_distance=ret[0][1], l1_s_vals=ret[1][4], loss=loss.detach(), param_names=np.array(list(sd.keys())))
    for k, v in model.named_parameters():
        out["grad." + k] = v.grad
    arr = {k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in out.items()}
    _oracle_common.save_golden(os.path.join(REPO, "tests", "golden", "g23_warp0.npz"), **arr)
    print("wrote g23_warp0.npz", sum(a.nbytes for a in arr.values()), "bytes")


if __name__ == "__main__":
    main()
