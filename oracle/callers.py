"""TEST INFRASTRUCTURE -- CPU restatement of the two steps either side of the render path:
ray generation (the step before) and the per-ray loss tail (the step after).  Only tests/, __graft_entry__.smoke()
and bench.py's baseline legs may import this package.

Pinned against the imported reference by oracle/gen_golden_callers.py -> tests/golden/g12_rays.npz, g13_losses.npz.
Citations are relative to /root/reference/.
"""
import numpy as np
import torch


# ---------------------------------------------------------------- rays ----
def pinhole_rays(coords, pose, intrinsic, H, near, far, training=False, W=None):
    """Rays of the pixels ``coords`` [N,2] = (row j, col i) of a pinhole camera: s-nerf/utils/sample_utils.py:286-345
    (get_rays_single_img, whole frame) and :92-211 (sample_single_img, selected pixels), no-NDC branch.

    camera_dirs = [(i-cx+.5)/f, -(j-cy+.5)/f, -1] with f = (fx+fy)/2 (:295-299); directions = sum_k cam[k]*pose[c,k]
    (:301-302); origins = pose[:3,3]; viewdirs = directions/||directions|| (:305); radii = ||dir(j,i) - dir(j+1,i)|| * 2/sqrt(12),
    the last row taking dx[-2:-1] = the value of row H-3 (:308-314).  near/far are the caller's already scaled bounds (:331-332).
    ``training=True`` is sample_single_img: the ray directions come from get_rays_by_coord
    (s-nerf/model/run_nerf_helpers.py:300-312: integer pixel coordinates WITHOUT the half-pixel offset, focal =
    mean([fx, fy])) while the radii still come from the half-pixel grid above (sample_utils.py:105-124,201).
    Returns the 8 Rays fields (sample_utils.py:11-13)."""
    coords = torch.as_tensor(coords)
    pose = torch.as_tensor(pose, dtype=torch.float32)
    K = torch.as_tensor(intrinsic, dtype=torch.float32)
    cx, cy, f = K[0, 2], K[1, 2], (K[0, 0] + K[1, 1]) / 2

    def dirs(jj, ii):
        cam = torch.stack([(ii - cx + 0.5) / f, -(jj - cy + 0.5) / f, -torch.ones_like(ii)], -1)
        return (cam[..., None, :] * pose[:3, :3]).sum(-1)

    j, i = coords[:, 0].float(), coords[:, 1].float()
    if training:
        focal = torch.stack([K[0, 0], K[1, 1]]).mean()
        cam = torch.stack([(i - cx) / focal, -(j - cy) / focal, -torch.ones_like(i)], -1)
        d = torch.sum(cam[..., None, :] * pose[:3, :3], -1)
    else:
        d = dirs(j, i)
    # the reference evaluates dx on the whole [H, W] grid and then indexes it; do the same so that torch's reduction order
    # (which depends on the tensor shape) is the reference's.  A lane-per-pixel evaluation ((x^2 + y^2) + z^2) differs from
    # it by <= 1 ulp in a few percent of the pixels; radii feed no index computation.
    if W is None:
        W = int(coords[:, 1].max()) + 1
    gi, gj = torch.meshgrid(torch.linspace(0, W - 1, W), torch.linspace(0, H - 1, H), indexing="ij")
    gi, gj = gi.t(), gj.t()
    grid = dirs(gj, gi)                                # [H, W, 3]
    dxg = torch.sqrt(torch.sum((grid[:-1, :, :] - grid[1:, :, :]) ** 2, -1))
    dxg = torch.cat([dxg, dxg[-2:-1, :]], 0)           # as the reference: the last image row repeats dx row H-3
    radii = (dxg[..., None] * 2 / np.sqrt(12))[coords[:, 0].long(), coords[:, 1].long()].reshape(-1, 1)
    o = torch.broadcast_to(pose[None, :3, -1], d.shape).contiguous()
    v = d / torch.linalg.norm(d, axis=-1, keepdims=True)
    ones = torch.ones_like(o[..., :1])
    return dict(origins=o, directions=d, viewdirs=v, radii=radii, lossmult=ones, near=ones * near, far=ones * far, app=ones * 0.0)


# -------------------------------------------------------------- losses ----
def rgb_loss(pred, tgt):
    """RgbLoss, s-nerf/model/loss_factory.py:5-11."""
    return torch.mean((pred - tgt) ** 2)


def depth_loss(pred, pred_c, tgt, confidence=None, coarse_depth_mult=0.2, disparity=True):
    """calc_depth_loss + DepthLoss: s-nerf/model/confidence.py:209-224, loss_factory.py:26-37.  Rays with tgt == 0 are
    masked out; |1/p - 1/t| + c*|1/pc - 1/t| (disparity) times the per-ray confidence, mean over the valid rays.
    (train.py:208 multiplies by depth_lambda.)"""
    m = tgt != 0
    fn = (lambda x, y: torch.abs(1 / x - 1 / y)) if disparity else (lambda x, y: torch.abs(x - y))
    l = fn(pred[m], tgt[m]) + coarse_depth_mult * fn(pred_c[m], tgt[m])
    if confidence is not None:
        l = l * confidence[m]
    return l.mean()


def proposal_loss(s_vals_f, weights_f, s_vals_c, weights_c, weight=0.05):
    """ProposalLoss, s-nerf/model/loss_factory.py:59-74 (fine level detached; note the reference clamps the right index
    with the FINE interval count and the left index at 0 rather than treating "before the first fence post" as 0)."""
    s_vals_f, weights_f = s_vals_f.detach(), weights_f.detach()
    inds = torch.searchsorted(s_vals_c.contiguous(), s_vals_f.contiguous(), right=True)
    W_c = torch.cumsum(weights_c, dim=1)
    left = torch.gather(W_c, 1, torch.clamp(inds[:, :-1] - 1, min=0).long())
    right = torch.gather(W_c, 1, torch.clamp(inds[:, 1:] - 1, max=weights_f.shape[1] - 1))
    bound = right - left
    l = torch.clamp(weights_f - bound, min=0) ** 2 / (weights_f + 1e-8)
    return l.sum(dim=1).mean() * weight
