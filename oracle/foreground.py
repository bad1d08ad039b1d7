"""TEST INFRASTRUCTURE -- CPU restatement of the image-space foreground composite of S-NeRF++ stage 1: the
depth-tested paste of a rendered foreground instance into the background frame with its occlusion bookkeeping, and the boundary /
mask algebra around it.  Only tests/, __graft_entry__.smoke() and bench baselines may import this package.

Pinning (oracle/gen_golden_foreground.py -> tests/golden/g18_foreground.npz):
  * set_diff, fuse_bound, fuse_bound_and_im and the mask union: vectors from the reference's own functions.
  * occlusion_paste: vectors from the reference's handle_occlusion_paste, run with the un-vendored CUDA ray tracer replaced by an
    injected per-pixel mesh depth (the quantity under test is what the reference does WITH the mesh depth, not the ray tracing).
  * get_bound_im: PARITY UNPINNED -- it calls cv2.dilate / cv2.erode (opencv_python 4.8.0.76, s-nerfpp/requirements.txt:21-22), which
    is not installed here; `rect_morph` restates OpenCV's published definition (dst(x,y) = max / min over the kernel window
    src(x + x' - anchor.x, y + y' - anchor.y), anchor = size // 2, pixels outside the image never win).
  * handle_lighting (utils_render.py:1008-1051) is NOT restated: it goes through cv2's 8-bit fixed-point RGB<->HSV tables and stays
    with the caller.
Citations are relative to /root/reference/s-nerfpp/stage1_code/."""
import numpy as np

SEMANTIC_ID = {"vehicle": 13, "person": 11, "object": 0, "bicycle": 18, "motorcycle": 17}     # utils_render.py:934-936
OVERWRITABLE = (0, 1, 8)                                                                         # road / sidewalk / ... :972-974


def occlusion_paste(bg_im, fg_im, mask_im, depth_mat, semantic_mat, fg_depth, category="vehicle"):
    """utils_render.py:826-1005 (handle_occlusion_paste) given `fg_depth` [H,W] = the mesh depth along each pixel's ray (what
    get_depth_from_mesh_batch returns for the masked pixels :913-947; -1 everywhere for category "person" :940-941).
    A masked pixel (mask[..., 0] > 0) takes the foreground colour, depth and class id when the mesh is in front of the background
    depth or the background there is one of the overwritable classes; otherwise the mask is cleared.  Returns copies
    (bg_im, depth_mat, semantic_mat, mask_im, occlusion_per) with occlusion_per = 1 - #pasted / (#masked + 1) (:1002-1003)."""
    bg, fg, mask = np.array(bg_im), np.asarray(fg_im), np.array(mask_im)
    depth, sem = np.array(depth_mat), np.array(semantic_mat)
    ii, jj = np.where(mask[..., 0] > 0)
    bgd = depth[ii, jj]
    fgd = -1 * np.ones_like(bgd) if category == "person" else np.asarray(fg_depth)[ii, jj]
    s = sem[ii, jj]
    valid = (fgd < bgd) | (s == 0) | (s == 1) | (s == 8)
    vi, vj, xi, xj = ii[valid], jj[valid], ii[~valid], jj[~valid]
    bg[vi, vj, ...] = fg[vi, vj, ...]
    depth[vi, vj] = fgd[valid]
    sem[vi, vj] = SEMANTIC_ID[category]
    mask[xi, xj] = 0
    return bg, depth, sem, mask, 1 - valid.sum() / (ii.size + 1)


def rect_morph(img, r, dilate):
    """cv2.dilate / cv2.erode with cv2.getStructuringElement(cv2.MORPH_RECT, (r, r)), default anchor (r // 2, r // 2) and default
    border (a constant that never wins the max / min)."""
    H, W = img.shape
    a = r // 2
    pad = np.full((H + r, W + r), 0 if dilate else 255, img.dtype)
    pad[a:a + H, a:a + W] = img
    out = None
    for dy in range(r):
        for dx in range(r):
            w = pad[dy:dy + H, dx:dx + W]
            out = w.copy() if out is None else (np.maximum(out, w) if dilate else np.minimum(out, w))
    return out


def get_bound_im(mask_im, r):
    """utils_render.py:306-324: boundary band = dilate(mask[..., 0]) XOR erode(mask[..., 0]) with a max(1, r)-square, as a 3-channel
    0 / 255 image."""
    m = np.asarray(mask_im)[..., 0]
    k = max(1, int(r))
    b = np.logical_xor(rect_morph(m, k, True), rect_morph(m, k, False))
    return b[..., None].repeat(3, -1).astype(np.uint8) * 255


def bound_radius(mask_im, category="vehicle"):
    """generate_images.py:127-134: r = int((mask width / 80) ** .82) (1 for an empty mask, 3 for two-wheelers)."""
    _, jj, _ = np.where(np.asarray(mask_im) > 0)
    if jj.size == 0:
        return 1
    return 3 if category in ("motorcycle", "bicycle") else int(((jj.max() - jj.min()) / 80) ** .82)


def set_diff(a, b):
    """ip_utils.py:10-19: A minus B on the > 0 supports, as 0 / 255."""
    A, B = np.asarray(a) > 0, np.asarray(b) > 0
    return np.logical_and(A, np.logical_not(np.logical_and(A, B))).astype(np.uint8) * 255


def fuse_bound(total_mask, total_bound, bound, mask):
    """utils_render.py:338-361: the new instance's band outside the already pasted masks, plus the old bands outside the new mask."""
    tm, tb, b, m = (np.asarray(x) > 0 for x in (total_mask, total_bound, bound, mask))
    b = np.logical_and(b, np.logical_not(np.logical_and(b, tm)))
    tb = np.logical_and(tb, np.logical_not(np.logical_and(tb, m)))
    return np.logical_or(b, tb).astype(np.uint8) * 255


def mask_union(total_mask, mask):
    """generate_images.py:161: (mask | total_mask) * 255."""
    return (np.asarray(mask).astype(bool) | np.asarray(total_mask).astype(bool)).astype(np.uint8) * 255


def fuse_bound_and_im(im, bound):
    """utils_render.py:327-335: blank the pixels (per channel) under the boundary band, for the inpainting stage to fill."""
    out = np.array(im)
    out[np.asarray(bound) > 0] = 0
    return out
