#!/usr/bin/env python3
"""Golden vectors for the callers either side of the path (rows 8f-1, 8f-2): runs the REFERENCE's ray generation
(s-nerf/utils/sample_utils.py) and loss modules (s-nerf/model/loss_factory.py, confidence.calc_depth_loss) on seeded inputs
and records inputs + outputs as tests/golden/g12_rays.npz, g13_losses.npz.  Build-container only (needs /root/reference)."""
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
OUT = os.path.join(REPO, "tests", "golden")
REF = "/root/reference/s-nerf"


def main():
    if not os.path.isdir(REF):
        raise SystemExit("reference tree not present")
    for name in ("turtle", "cv2", "imageio", "lpips", "kornia", "pyquaternion", "matplotlib", "matplotlib.pyplot", "nuscenes", "open3d", "skimage", "tqdm", "torchvision", "torchvision.models", "torchvision.transforms", "scipy.spatial.transform"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                class _Stub(types.ModuleType):          # plotting / dataset packages the loss + ray code never calls
                    __path__ = []
                    def __getattr__(self, k):
                        if k.startswith("__"):
                            raise AttributeError(k)
                        return type(k, (), {})
                sys.modules[name] = _Stub(name)
    sys.path.insert(0, REF)
    import utils.sample_utils as su
    import model.loss_factory as lf
    args = types.SimpleNamespace(smooth_loss=False, no_ndc=True, N_rgb=300, encode_appearance=False, coarse_depth_mult=0.2,
                                 disparity_depth=True, proposal_lambda=0.05)
    # ---- G12: rays of a 37 x 53 nuScenes-like camera: selected pixels (training) and the whole frame (evaluation)
    H, W = 37, 53
    g = torch.Generator().manual_seed(5)
    th = 0.3
    pose = torch.tensor([[np.cos(th), 0.05, np.sin(th), 1.5], [-0.05, 1.0, 0.02, -0.25], [-np.sin(th), -0.02, np.cos(th), 0.75]], dtype=torch.float32)
    K = torch.tensor([[61.3, 0.0, 26.2], [0.0, 60.9, 18.4], [0.0, 0.0, 1.0]], dtype=torch.float32)
    image = torch.rand(H, W, 3, generator=g)
    depth = torch.rand(H, W, generator=g) * 50 + 2
    depth[torch.rand(H, W, generator=g) < 0.4] = 0
    np.random.seed(7)
    rays, trgb, tdep, sel, _ = su.sample_single_img(args, image, depth, pose, K, near=2.0, far=100.0, near_far=False, batch_n=300)
    full = su.get_rays_single_img(args, image, depth, pose, K, near=2.0, far=100.0, factor=1)
    out = dict(pose=pose.numpy(), intrinsic=K.numpy(), H=np.int64(H), W=np.int64(W), sel_coords=sel.numpy(), target_rgb=trgb.numpy(), target_depth=tdep.numpy(),
               image=image.numpy(), depth=depth.numpy())
    for k in rays._fields:
        out["sel_" + k] = getattr(rays, k).numpy()
        out["full_" + k] = getattr(full, k).reshape(H * W, -1).numpy()
    _oracle_common.save_golden(os.path.join(OUT, "g12_rays.npz"), **out)
    # ---- G13: loss tail on seeded renderer outputs (shapes of the shipped config: 128 + 127 intervals, shrunk ray count)
    N, Sc, Pf = 96, 128, 128
    def fence(n, P):
        s = torch.sort(torch.rand(n, P, generator=g), dim=-1).values
        s[:, 0] = 0.0; s[:, -1] = 1.0
        return s
    s_c, s_f = fence(N, Sc + 1), fence(N, Pf)
    s_f[:7] = s_c[:7, :Pf]                                     # ties between the two grids
    w_c = torch.rand(N, Sc, generator=g) ** 4; w_c = (w_c / w_c.sum(-1, keepdim=True) * torch.rand(N, 1, generator=g)).requires_grad_(True)
    w_f = torch.rand(N, Pf - 1, generator=g) ** 6; w_f = w_f / w_f.sum(-1, keepdim=True) * torch.rand(N, 1, generator=g)
    w_f[3] = 0.0
    pl = lf.ProposalLoss(args)(s_f, w_f, s_c, w_c)
    g_wc, = torch.autograd.grad(pl, w_c)
    rgb = torch.rand(N, 3, generator=g, requires_grad=True); tgt = torch.rand(N, 3, generator=g)
    rl = lf.RgbLoss(args)(rgb, tgt)
    g_rgb, = torch.autograd.grad(rl, rgb)
    d1 = (torch.rand(N, generator=g) * 60 + 2).requires_grad_(True); d0 = (torch.rand(N, generator=g) * 60 + 2).requires_grad_(True)
    td = torch.rand(N, generator=g) * 78 + 2; td[torch.rand(N, generator=g) < 0.5] = 0
    conf = torch.rand(N, generator=g)
    # confidence.calc_depth_loss (:209-224) with args.depth_conf: the confidence gathered at the valid rays
    mask = td != 0
    dl = lf.DepthLoss(args)(d1.unsqueeze(-1)[mask.unsqueeze(-1)], d0.unsqueeze(-1)[mask.unsqueeze(-1)], td.unsqueeze(-1)[mask.unsqueeze(-1)])
    dl = (dl * conf[mask]).mean()
    g_d1, g_d0 = torch.autograd.grad(dl, [d1, d0])
    n = lambda t: t.detach().numpy()
    _oracle_common.save_golden(os.path.join(OUT, "g13_losses.npz"), s_c=n(s_c), s_f=n(s_f), w_c=n(w_c), w_f=n(w_f), proposal_loss=n(pl), g_wc=n(g_wc),
                        rgb=n(rgb), tgt=n(tgt), rgb_loss=n(rl), g_rgb=n(g_rgb), d1=n(d1), d0=n(d0), td=n(td), conf=n(conf), depth_loss=n(dl),
                        g_d1=n(g_d1), g_d0=n(g_d0), proposal_lambda=np.float32(0.05), coarse_depth_mult=np.float32(0.2))
    print("wrote g12_rays.npz, g13_losses.npz")


if __name__ == "__main__":
    main()
