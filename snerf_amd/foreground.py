"""Image-space foreground composite of S-NeRF++ stage 1 on the device.

Reference counterpart: s-nerfpp/stage1_code/generate_images.py:53-200 (the per-frame loop over the instances in occlusion order) with
utils_render.py:826-1005 handle_occlusion_paste, :306-324 get_bound_im, :327-361 fuse_bound_and_im / fuse_bound and
ip_utils.py:10-19 set_diff.  There every step round-trips PIL images and numpy index lists on the host; here the frame (background
rgb, depth, semantic) stays resident in HBM as uint8 / float tensors and every step is one launch.

Not here (they stay with the caller): mesh placement and the ray-traced mesh depth (`raytracing.RayTracer`, nvdiffrast, trimesh --
`fg_depth` is an input), the bounding-box export, and `handle_lighting` (cv2's 8-bit HSV round trip)."""
import torch

from . import _lib
from .ops import _p, _stream

SEMANTIC_ID = {"vehicle": 13, "person": 11, "object": 0, "bicycle": 18, "motorcycle": 17}     # utils_render.py:934-936


def _u8(t, shape=None):
    assert t.is_cuda and t.dtype == torch.uint8 and t.is_contiguous() and (shape is None or tuple(t.shape) == tuple(shape)), \
        "expected a contiguous uint8 CUDA tensor" + ("" if shape is None else f" of shape {tuple(shape)}")
    return t


def handle_occlusion_paste(bg_im, vehicle_im, mask_im, depth_mat, semantic_mat, fg_depth, category="vehicle"):
    """In place on bg_im [H,W,3] u8, mask_im [H,W,3] u8, depth_mat [H,W] f32, semantic_mat [H,W] u8; `fg_depth` [H,W] f32 = mesh depth
    per pixel (None for category "person").  Returns occlusion_per as a device scalar (no host sync)."""
    H, W = depth_mat.shape
    _u8(bg_im, (H, W, 3)); _u8(vehicle_im, (H, W, 3)); _u8(mask_im, (H, W, 3)); _u8(semantic_mat, (H, W))
    assert depth_mat.dtype == torch.float32 and depth_mat.is_contiguous() and depth_mat.is_cuda
    person = category == "person"
    if not person:
        assert fg_depth is not None and fg_depth.dtype == torch.float32 and fg_depth.is_contiguous() and tuple(fg_depth.shape) == (H, W)
    cnt = torch.empty(2, dtype=torch.int32, device=bg_im.device)
    _lib.call("snerf_fg_paste", _p(bg_im), _p(vehicle_im), _p(mask_im), _p(depth_mat), _p(semantic_mat), _p(None if person else fg_depth), H * W,
              SEMANTIC_ID[category], int(person), _p(cnt), _stream())
    c = cnt.double()
    return 1 - c[1] / (c[0] + 1)


def bound_radius(mask_im, category="vehicle"):
    """generate_images.py:127-134 (one small host sync: the band width is a kernel-shape parameter)."""
    if category in ("motorcycle", "bicycle"):
        cols = (mask_im > 0).any(dim=2).any(dim=0)
        return 3 if bool(cols.any()) else 1
    cols = torch.nonzero((mask_im > 0).any(dim=2).any(dim=0))
    if cols.numel() == 0:
        return 1
    return int((float(cols.max() - cols.min()) / 80) ** .82)


def get_bound_im(mask_im, r, return_mask_diff=False):
    """-> bound_im [H,W,3] u8 (and, with return_mask_diff, set_diff(mask_im, bound_im) from the same launch)."""
    H, W, _ = mask_im.shape
    _u8(mask_im, (H, W, 3))
    bound = torch.empty_like(mask_im)
    diff = torch.empty_like(mask_im) if return_mask_diff else None
    _lib.call("snerf_fg_bound", _p(mask_im), H, W, max(1, int(r)), _p(bound), _p(diff), _stream())
    return (bound, diff) if return_mask_diff else bound


def accumulate(total_mask, total_bound, bound_im, mask_im):
    """In place: total_bound <- fuse_bound(total_mask, total_bound, bound_im, mask_im); total_mask <- mask_im | total_mask.
    Zero-initialised totals reproduce the reference's first-instance special case."""
    for t in (total_mask, total_bound, bound_im, mask_im):
        _u8(t, total_mask.shape)
    _lib.call("snerf_fg_accumulate", _p(total_mask), _p(total_bound), _p(bound_im), _p(mask_im), total_mask.numel(), _stream())


def fuse_bound_and_im(fuse_im, bound_im):
    """In place: fuse_im[bound_im > 0] = 0."""
    _u8(fuse_im); _u8(bound_im, fuse_im.shape)
    _lib.call("snerf_fg_blank", _p(fuse_im), _p(bound_im), fuse_im.numel(), _stream())
    return fuse_im


def composite_frame(bg_im, depth_mat, semantic_mat, instances):
    """The instance loop of generate_images.py:80-166 for one frame.  `instances`: iterable (already in the reference's occlusion
    order) of dict(image [H,W,3] u8, mask [H,W,3] u8, fg_depth [H,W] f32 or None, category, r (optional band width)).
    In place on bg_im / depth_mat / semantic_mat; returns dict(fuse, mask, bound, occluded_mask, depth, semantic, occlusion [per
    instance], instance_masks, instance_bounds) = the images the reference saves per frame (:153-154,186-196)."""
    total_mask, total_bound, total_occluded = torch.zeros_like(bg_im), torch.zeros_like(bg_im), torch.zeros_like(bg_im)
    occ, masks, bounds = [], [], []
    for inst in instances:
        cat = inst.get("category", "vehicle")
        r = inst.get("r")
        if r is None:
            r = bound_radius(inst["mask"], cat)
        # the depth test works on a copy of the mask (utils_render.py:280: np.array(mask_im)); the band and the frame's mask come from
        # the instance's full mask, the depth-tested one only feeds the `occluded_mask` image of vehicles (generate_images.py:162-166)
        tested = inst["mask"].clone()
        occ.append(handle_occlusion_paste(bg_im, inst["image"], tested, depth_mat, semantic_mat, inst.get("fg_depth"), cat))
        bound, mask_d = get_bound_im(inst["mask"], r, return_mask_diff=True)
        accumulate(total_mask, total_bound, bound, mask_d)
        if cat == "vehicle":
            total_occluded = torch.where((tested > 0) | (total_occluded > 0), 255, 0).to(torch.uint8)
        masks.append(mask_d); bounds.append(bound)
    fuse_bound_and_im(bg_im, total_bound)
    return dict(fuse=bg_im, mask=total_mask, bound=total_bound, occluded_mask=total_occluded, depth=depth_mat, semantic=semantic_mat, occlusion=occ,
                instance_masks=masks, instance_bounds=bounds)
