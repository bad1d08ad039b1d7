"""Ray generation on the device: host-side mirror of s-nerf/utils/sample_utils.py (the step immediately before the render
path).  Same function names and argument meaning as the reference; the per-pixel arithmetic runs in
`snerf_pinhole_rays` (no [H, W, 3] direction grid is materialised to pick 4096 pixels).  `args` needs `no_ndc=True` (the
S-NeRF street-scene configuration); NDC rays raise NotImplementedError."""
import collections

import numpy as np
import torch

from . import ops

Rays = collections.namedtuple('Rays', ('origins', 'directions', 'viewdirs', 'radii', 'lossmult', 'near', 'far', 'app'))


def _intr(intrinsic):
    K = np.asarray(intrinsic.detach().cpu() if torch.is_tensor(intrinsic) else intrinsic, dtype=np.float32)
    return float(K[0, 2]), float(K[1, 2]), float(K[0, 0]), float(K[1, 1])


def _pose(pose):
    return np.asarray(pose.detach().cpu() if torch.is_tensor(pose) else pose, dtype=np.float32)


def _check(args):
    if not getattr(args, "no_ndc", True):
        raise NotImplementedError("NDC rays (no_ndc=False) are outside the accelerated path")


def get_rays_single_img(args, image, depth_gt, pose, intrinsic, near=0., far=1., factor=4, device=None):
    """sample_utils.py:286-345: rays of the whole (H//factor x W//factor) frame -> Rays with [H, W, .] fields."""
    _check(args)
    H, W = image.shape[0] // factor, image.shape[1] // factor
    device = device if device is not None else (image.device if torch.is_tensor(image) and image.is_cuda else torch.device("cuda"))
    cx, cy, fx, fy = (v / factor for v in _intr(intrinsic))
    # float32 arithmetic of `intrinsic / factor` (sample_utils.py:290)
    cx, cy, fx, fy = (float(np.float32(v)) for v in (cx, cy, fx, fy))
    o, d, v, r, nr, fr = ops.pinhole_rays(None, 0, H * W, W, H, _pose(pose), cx, cy, fx, fy, False, near * 0.9, far * 1.1, device)
    ones = torch.ones(H, W, 1, dtype=torch.float32, device=device)
    sh = lambda t: t.view(H, W, -1)
    return Rays(sh(o), sh(d), sh(v), sh(r), ones, sh(nr), sh(fr), ones * 0.)


def rays_of_pixels(coords, pose, intrinsic, H, W, near, far, training=False, device="cuda"):
    """Rays of selected pixels; coords [N,2] = (row, col) (any integer tensor / array)."""
    c = torch.as_tensor(coords).to(device=device, dtype=torch.int32).contiguous()
    cx, cy, fx, fy = _intr(intrinsic)
    o, d, v, r, nr, fr = ops.pinhole_rays(c, 0, c.shape[0], W, H, _pose(pose), cx, cy, fx, fy, training, near, far, device)
    ones = torch.ones_like(r)
    return Rays(o, d, v, r, ones, nr, fr, ones * 0.)


def sample_patches(H, W, patch_sz, n_patch):
    """sample_utils.py:68-89 (sample_patches_pt): `n_patch` centres drawn with np.random.randint among the pixels strictly more than `patch_sz` from
    every border (row-major order of that window), each giving the (2 (patch_sz // 2))^2 pixels [c - patch_sz // 2, c + patch_sz // 2) around it,
    rows outer.  -> int64 [n_patch * (2 (patch_sz // 2))^2, 2] of (row, col)."""
    hr, wc = H - 2 * patch_sz - 1, W - 2 * patch_sz - 1                  # rows / columns r with patch_sz < r < H - patch_sz
    if hr <= 0 or wc <= 0:
        raise ValueError("image smaller than the smooth-loss patch margin (the reference's np.random.randint(0) raises here too)")
    idx = np.random.randint(hr * wc, size=n_patch)
    cr, cc = patch_sz + 1 + idx // wc, patch_sz + 1 + idx % wc
    h = patch_sz // 2
    dr, dc = np.meshgrid(np.arange(-h, h), np.arange(-h, h), indexing="ij")
    rows = (cr[:, None, None] + dr[None]).reshape(-1)
    cols = (cc[:, None, None] + dc[None]).reshape(-1)
    return np.stack([rows, cols], -1).astype(np.int64)


def smooth_loss(image, skymask, sel_coords_smooth, pred_distance_smooth, n_patch, patch_sz, weight, use_skymask=True):
    """loss_factory.py:38-57 (SmoothLoss) on loss.py:14-35 (edge_aware_loss_v2): the edge-aware smoothness of the patches' disparities, in torch
    ops (autograd carries it to `pred_distance_smooth` = the renderer's distances of the batch's patch rays, train.py:154-177).  A caller-side
    loss: off in the shipped config (configs/nuScenes_depth_6cams:61), no kernel."""
    img = torch.as_tensor(image)
    c = torch.as_tensor(sel_coords_smooth).long().to(img.device)
    dev = pred_distance_smooth.device
    rgb = img[c[:, 0], c[:, 1]].to(dev).view(n_patch, patch_sz, patch_sz, -1)
    disp = (1 / torch.clamp(pred_distance_smooth, min=1e-5)).view(n_patch, patch_sz, patch_sz, -1)
    disp = disp / (disp.mean(1, True).mean(2, True) + 1e-7)
    gx = torch.abs(disp[:, :, :-1, :] - disp[:, :, 1:, :]) * torch.exp(-torch.mean(torch.abs(rgb[:, :, :-1, :] - rgb[:, :, 1:, :]), 3, keepdim=True))
    gy = torch.abs(disp[:, :-1, :, :] - disp[:, 1:, :, :]) * torch.exp(-torch.mean(torch.abs(rgb[:, :-1, :, :] - rgb[:, 1:, :, :]), 3, keepdim=True))
    if use_skymask:
        sky = torch.as_tensor(skymask)[c[:, 0].to(torch.as_tensor(skymask).device), c[:, 1].to(torch.as_tensor(skymask).device)].to(dev).view(n_patch, patch_sz, patch_sz, -1)
        gx = gx + sky[:, :, :-1, :] * gx
        gy = gy + sky[:, :-1, :, :] * gy
    return (gx.mean() + gy.mean()) * weight


def sample_single_img(args, image, depth_gt, pose, intrinsic, near=0., far=1., near_far=False, batch_n=None, app=0.):
    """sample_utils.py:92-211: a random pixel batch of one image -> (Rays, target_rgb, target_depth, sel_coords, sel_inds).
    The pixel choice uses numpy's global RNG exactly like the reference (np.random.choice without replacement)."""
    _check(args)
    H, W = image.shape[:2]
    n = batch_n if batch_n is not None else args.N_rgb
    patches = None
    if getattr(args, "smooth_loss", False):
        # --smooth_loss (sample_utils.py:102-103, 136-138): N_patch random patches appended BEHIND the random pixels; their centres are drawn first
        patches = sample_patches(H, W, int(args.patch_sz), int(args.N_patch))
    sel = np.random.choice(H * W, size=[n], replace=False)
    coords = np.stack([sel // W, sel % W], -1)
    if patches is not None:
        coords = np.concatenate([coords, patches], 0)
    if not near_far:
        near, far = near * 0.9, far * 1.1
    else:
        nz = depth_gt[depth_gt != 0]
        near, far = float(nz.min()) * 0.9, float(nz.max()) * 1.1
    dev = image.device if torch.is_tensor(image) and image.is_cuda else torch.device("cuda")
    rays = rays_of_pixels(coords, pose, intrinsic, H, W, near, far, training=True, device=dev)
    rays = rays._replace(app=rays.lossmult * float(app))
    ct = torch.as_tensor(coords, device=image.device if torch.is_tensor(image) else "cpu").long()
    target_rgb = image[ct[:, 0], ct[:, 1]]
    target_dep = depth_gt[ct[:, 0], ct[:, 1]]
    return rays, target_rgb, target_dep, ct, sel


def apply_pose_transform(rays, pose):
    """The pose-refinement half of `sample_rays` (sample_utils.py:421-435): rotate directions / viewdirs by pose[:3, :3] and shift the
    origins by pose[:3, 3], in torch, so that autograd links the rays to a learnable pose (`pose = pose_param_net(img_i,
    transform_only=True)`).  The rays this module generates are detached from any pose tensor (the per-pixel arithmetic runs in a
    kernel); with `pose.requires_grad` the returned origins / directions / viewdirs require grad, `MipNerfModel` / zipnerf `Model`
    deliver d loss / d rays for them, and `loss.backward()` reaches the pose parameters as in the reference."""
    R, t = pose[:3, :3], pose[:3, 3]
    rot = lambda v: (v[..., None, :] * R).sum(-1)
    return rays._replace(origins=rays.origins + t, directions=rot(rays.directions), viewdirs=rot(rays.viewdirs))
