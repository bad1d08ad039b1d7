#!/usr/bin/env python3
"""Golden vectors for SURVEY.md row A16: the reference's own ``render_image`` (s-nerf/model/models.py:328-360, called by
eval.py:146) on a small [H,W] ray grid with RAGGED chunks -> tests/golden/g21_render_image.npz.  The reference's function
needs a semantic model (it concatenates the fourth output) and proposal_loss off (it unpacks exactly four columns); on this CPU-only
container ``torch.cuda.device_count()`` is 0, which the reference divides by, so the count is pinned to 1 for the call (an
environment stub, nothing of the reference is changed).  Build-container only (needs /root/reference)."""
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
    C, H, W, chunk = 7, 5, 7, 8
    model = models.MipNerfModel(n_samples=16, N_fine=17, no_warp_sample=0, ray_shape="cone", fn=1, radius=3., transform_idx=0, real=True,
                                rgb_layer=3, hidden_layer=64, density_noise=0., max_deg_point=16, proposal_hidden_layer=64,
                                proposal_loss=False, semantic=True, semantic_class_num=C)
    sd = common.fill_state_dict_(model.state_dict())
    model.load_state_dict(sd)
    flat = common.synthetic_rays(H * W, seed=21)
    Rays = namedtuple("Rays", ("origins", "directions", "viewdirs", "radii", "lossmult", "near", "far", "app"))
    rays = Rays(**{k: v.reshape(H, W, -1) for k, v in flat.items()})
    saved = torch.cuda.device_count
    torch.cuda.device_count = lambda: 1
    try:
        with torch.no_grad():
            rgb, dist, acc, sem = models.render_image(lambda r: model(r, False, False, 0.), rays, 0, chunk=chunk)
    finally:
        torch.cuda.device_count = saved
    out = dict(**{f"rays_{k}": v for k, v in flat.items()}, H=np.int64(H), W=np.int64(W), chunk=np.int64(chunk), rgb=rgb, distance=dist, acc=acc,
               semantic=sem, param_names=np.array(list(sd.keys())))
    arr = {k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in out.items()}
    _oracle_common.save_golden(os.path.join(REPO, "tests", "golden", "g21_render_image.npz"), **arr)
    print("wrote g21_render_image.npz", sum(a.nbytes for a in arr.values()), "bytes", tuple(rgb.shape), tuple(dist.shape), tuple(sem.shape))


if __name__ == "__main__":
    main()
