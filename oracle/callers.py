"""TEST INFRASTRUCTURE -- CPU restatement of the two steps either side of the render path:
ray generation (the step before) and the per-ray loss tail (the step after).  Only tests/, __graft_entry__.smoke()
and bench.py's baseline legs may import this package.

Pinned against the imported reference by oracle/gen_golden_callers.py -> tests/golden/g12_rays.npz, g13_losses.npz (mip path),
oracle/gen_golden_zip_callers.py -> g15_zip_rays.npz, g16_zip_losses.npz (zipnerf path; the two regularisers also against the reference
evaluated in float64) and oracle/gen_golden_frame.py -> g17_frame_writer.npz (pixels decoded from the files the reference's save path
wrote).  PARITY UNPINNED: `hash_decay_loss` only -- it rests on torch_scatter.segment_coo (pinned 2.1.1, not installed), whose
documented segment mean is restated.
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


# ---------------------------------------------------------------- zipnerf (path C) callers ----
def zip_pixels_to_rays(pix_x, pix_y, cam_idx, pixtocams, camtoworlds):
    """s-nerfpp/zipnerf/internal/camera_utils.py:453-563 (pixels_to_rays), perspective camera, no distortion, no NDC:
    dirs_cam = pixtocam @ [x+.5, y+.5, 1] for the pixel and its +x / +y neighbours (:491-507), OpenCV->OpenGL flip
    diag(1,-1,-1) (:527), rotate by camtoworld[:3,:3] (:534), origins = camtoworld[:3,3], viewdirs normalised (:539-540),
    radii = 0.5 (|dx-d| + |dy-d|) * 2/sqrt(12) (:544-563), base_x/base_y = normalised neighbour offsets (:548-549).
    numpy promotes to float64 there (pix ints + .5); outputs are cast to fp32 by the dataset.  -> dict of fp32 arrays."""
    x = np.asarray(pix_x, np.float64)
    y = np.asarray(pix_y, np.float64)
    p2c = np.asarray(pixtocams, np.float64)[cam_idx]
    c2w = np.asarray(camtoworlds, np.float64)[cam_idx]
    def cast(xx, yy):
        cam = np.einsum("nij,nj->ni", p2c, np.stack([xx + .5, yy + .5, np.ones_like(xx)], -1))
        cam = cam * np.array([1., -1., -1.])
        return cam, np.einsum("nij,nj->ni", c2w[:, :3, :3], cam)
    cam0, d = cast(x, y)
    _, dx = cast(x + 1, y)
    _, dy = cast(x, y + 1)
    px, py = dx - d, dy - d
    nx, ny = np.linalg.norm(px, axis=-1), np.linalg.norm(py, axis=-1)
    f = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    return dict(origins=f(c2w[:, :3, 3]), directions=f(d), viewdirs=f(d / np.linalg.norm(d, axis=-1, keepdims=True)),
                radii=f((0.5 * (nx + ny))[:, None] * 2 / np.sqrt(12)), imageplane=f(cam0[:, :2]), base_x=f(px / nx[:, None]), base_y=f(py / ny[:, None]))


def zip_data_loss(rgb, target, lossmult=None, charb_padding=0.001, loss_type="charb"):
    """train_utils.py:62-90 (compute_data_loss, one level): lossmult [R] broadcast over the 3 channels, denom = lossmult.sum();
    -> (data loss, mse) with data = sqrt(resid^2 + pad^2) ('charb', :76) or resid^2 ('mse', :73)."""
    rgb, target = np.asarray(rgb, np.float64), np.asarray(target, np.float64)
    lm = np.ones(rgb.shape[0]) if lossmult is None else np.asarray(lossmult, np.float64)
    lm = np.broadcast_to(lm[:, None], rgb.shape)
    r2 = (rgb - target) ** 2
    d = np.sqrt(r2 + charb_padding ** 2) if loss_type == "charb" else r2
    return (lm * d).sum() / lm.sum(), (lm * r2).sum() / lm.sum()


def zip_depth_loss(depth, target, mask):
    """s-nerfpp/zipnerf/train.py:252-277: mean over the masked rays of |1/(depth+1e-5) - 1/(1e-5+target)| (caller multiplies by
    depth_lambda, and by 0.2 for the 'd_complete' mask).  An empty mask gives 0 (the reference takes the mean of an empty tensor)."""
    depth, target, m = np.asarray(depth, np.float64), np.asarray(target, np.float64), np.asarray(mask, np.float64)
    if m.sum() == 0:
        return 0.0
    return float((m * np.abs(1 / (depth + 1e-5) - 1 / (1e-5 + target))).sum() / m.sum())


def zip_semantic_nll(semantic, labels, mask):
    """train.py:294-298: NLLLoss(log(semantic[mask] + 1e-6), labels[mask]) = -mean log(semantic[r, label_r] + 1e-6)."""
    sem, m = np.asarray(semantic, np.float64), np.asarray(mask, np.float64)
    if m.sum() == 0:
        return 0.0
    pick = sem[np.arange(sem.shape[0]), np.asarray(labels, np.int64)]
    return float(-(m * np.log(pick + 1e-6)).sum() / m.sum())


def lossfun_distortion(t, w):
    """stepfun.py:297-307, per ray: sum_ij w_i w_j |u_i - u_j| + sum_i w_i^2 (t_{i+1} - t_i) / 3, u = interval midpoints."""
    t, w = np.asarray(t, np.float64), np.asarray(w, np.float64)
    ut = (t[..., 1:] + t[..., :-1]) / 2
    dut = np.abs(ut[..., :, None] - ut[..., None, :])
    inter = np.sum(w * np.sum(w[..., None, :] * dut, -1), -1)
    intra = np.sum(w ** 2 * (t[..., 1:] - t[..., :-1]), -1) / 3
    return inter + intra


def blur_stepfun(x, y, r):
    """stepfun.py:425-433: convolve the step function (x, y) with a box of half-width r -> piecewise-linear (xr, yr).
    The knots x -/+ r are formed in x's own dtype (the reference's fp32 tensors round them to fp32; r becomes an fp32 scalar there),
    everything after that in float64."""
    x = np.asarray(x)
    if x.dtype != np.float32:
        x = x.astype(np.float64)
    rr = x.dtype.type(r)
    y = np.asarray(y, np.float64)
    cat = np.concatenate([x - rr, x + rr], -1).astype(np.float64)
    r = float(rr)
    idx = np.argsort(cat, -1, kind="stable")
    xr = np.take_along_axis(cat, idx, -1)
    z = np.zeros_like(y[..., :1])
    y1 = (np.concatenate([y, z], -1) - np.concatenate([z, y], -1)) / (2 * r)
    y2 = np.take_along_axis(np.concatenate([y1, -y1], -1), idx[..., :-1], -1)
    yr = np.maximum(np.cumsum((xr[..., 1:] - xr[..., :-1]) * np.cumsum(y2, -1), -1), 0)
    return xr, np.concatenate([np.zeros_like(yr[..., :1]), yr], -1)


def sorted_interp_quad(x, xp, fpdf, fcdf):
    """math.py:133-156: piecewise-quadratic CDF lookup; idx = #(xp <= x) from the joint sort (:141-147)."""
    n = xp.shape[-1]
    idx = np.stack([np.searchsorted(xp[i], x[i], side="right") for i in range(x.shape[0])])
    i0, i1 = np.maximum(idx - 1, 0), np.minimum(idx, n - 1)
    g = lambda a, i: np.take_along_axis(a, i, -1)
    f0, f1, c0, x0, x1 = g(fpdf, i0), g(fpdf, i1), g(fcdf, i0), g(xp, i0), g(xp, i1)
    with np.errstate(divide="ignore", invalid="ignore"):
        off = np.clip(np.nan_to_num((x - x0) / (x1 - x0), nan=0.0), 0, 1)
    return c0 + (x - x0) * (f0 + f1 * off + f0 * (1 - off)) / 2


def anti_interlevel_loss(sdists, weights, pulse_width=(0.03, 0.003), mult=0.01):
    """train_utils.py:132-164: blur the (detached) final-level histogram with each proposal level's pulse width, integrate it
    to a piecewise-quadratic CDF, resample on the proposal's intervals, penalise proposal weights below it:
    mean(clamp(w_s - wp, 0)^2 / (wp + 1e-5)) summed over the proposal levels, times `mult`.
    sdists / weights: per level, last = the NeRF level.  -> (loss, [per-level terms])."""
    c, w = np.asarray(sdists[-1], np.float64), np.asarray(weights[-1], np.float64)
    wn = w / (c[..., 1:] - c[..., :-1])
    terms = []
    for i in range(len(sdists) - 1):
        cp, wp = np.asarray(sdists[i], np.float64), np.asarray(weights[i], np.float64)
        c_, w_ = blur_stepfun(sdists[-1], wn, pulse_width[i])
        area = 0.5 * (w_[..., 1:] + w_[..., :-1]) * (c_[..., 1:] - c_[..., :-1])
        cdf = np.concatenate([np.zeros_like(area[..., :1]), np.cumsum(area, -1)], -1)
        ws = np.diff(sorted_interp_quad(cp, c_, w_, cdf), axis=-1)
        terms.append(float((np.maximum(ws - wp, 0) ** 2 / (wp + 1e-5)).mean()))
    return mult * sum(terms), terms


def hash_decay_loss(table, offsets, mult=0.1):
    """train_utils.py:184-203 for one encoder: torch_scatter.segment_coo(param ** 2, idx, out=zeros(L, C), reduce='mean') -- the mean of
    param^2 over the rows of each level (idx = level of each row, gridencoder/grid.py:128-141) -- then .mean() over [L, C]; times `mult`.
    torch_scatter (pinned 2.1.1, s-nerfpp/requirements.txt:39) is not installed: its documented reduction is restated here, so this term
    is PARITY-UNPINNED by the reference.  -> (loss, d loss / d table)."""
    t = np.asarray(table, np.float64)
    L = len(offsets) - 1
    per = np.stack([(t[offsets[l]:offsets[l + 1]] ** 2).mean(0) for l in range(L)])
    grad = np.zeros_like(t)
    for l in range(L):
        grad[offsets[l]:offsets[l + 1]] = 2 * mult * t[offsets[l]:offsets[l + 1]] / ((offsets[l + 1] - offsets[l]) * L * t.shape[1])
    return mult * per.mean(), grad


def zip_loss_tail(rgb, target, lossmult, depth, target_depth, depth_mask, com_mask, semantic, labels, sem_mask, sdists, weights,
                  charb_padding=0.001, data_mult=1.0, depth_lambda=0.5, com_mult=0.2, sem_mult=0.04, pulse_width=(0.03, 0.003),
                  interlevel_mult=0.01, distortion_mult=0.005):
    """The whole per-ray loss tail of the zipnerf training step (s-nerfpp/zipnerf/train.py:250-311) with its analytic gradients
    w.r.t. the renderer outputs -- what snerf_zip_loss_tail computes in one launch.  Absent terms: pass None.
    -> (losses dict, grads dict); grads of the anti-interlevel term reach only the proposal weights (the NeRF histogram is detached,
    train_utils.py:136-137), the distortion term only the NeRF weights (sdist is detached by stop_level_grad, models.py:216-217)."""
    f = lambda a: None if a is None else np.asarray(a, np.float64)
    rgb, target, depth, target_depth, semantic = f(rgb), f(target), f(depth), f(target_depth), f(semantic)
    R = rgb.shape[0]
    L, G = {}, {}
    lm = np.ones(R) if lossmult is None else f(lossmult)
    diff = rgb - target
    root = np.sqrt(diff ** 2 + charb_padding ** 2)
    den = 3 * lm.sum()
    L["data"] = data_mult * (lm[:, None] * root).sum() / den
    L["mse"] = (lm[:, None] * diff ** 2).sum() / den
    G["rgb"] = data_mult * lm[:, None] * diff / root / den
    gd = np.zeros(R)
    for name, m, k in (("depth", depth_mask, depth_lambda), ("d_complete", com_mask, depth_lambda * com_mult)):
        if m is None or depth is None:
            continue
        m = f(m)
        e = 1 / (depth + 1e-5) - 1 / (1e-5 + target_depth)
        n = m.sum()
        L[name] = k * (m * np.abs(e)).sum() / n if n > 0 else 0.0
        if n > 0:
            gd += k * m * np.sign(e) * (-1 / (depth + 1e-5) ** 2) / n
    if depth is not None:
        G["depth"] = gd
    if semantic is not None:
        m = f(sem_mask)
        lab = np.asarray(labels, np.int64)
        pick = semantic[np.arange(R), lab]
        n = m.sum()
        L["sem"] = -sem_mult * (m * np.log(pick + 1e-6)).sum() / n if n > 0 else 0.0
        gs = np.zeros_like(semantic)
        if n > 0:
            gs[np.arange(R), lab] = -sem_mult * m / (n * (pick + 1e-6))
        G["semantic"] = gs
    c, w = f(sdists[-1]), f(weights[-1])
    if distortion_mult > 0:
        L["distortion"] = distortion_mult * lossfun_distortion(c, w).mean()
        ut = (c[..., 1:] + c[..., :-1]) / 2
        inner = np.sum(w[..., None, :] * np.abs(ut[..., :, None] - ut[..., None, :]), -1)
        G["w%d" % (len(sdists) - 1)] = distortion_mult / R * (2 * inner + 2 * w * (c[..., 1:] - c[..., :-1]) / 3)
    if interlevel_mult > 0:
        wn = w / (c[..., 1:] - c[..., :-1])
        tot = 0.0
        for i in range(len(sdists) - 1):
            cp, wp = f(sdists[i]), f(weights[i])
            c_, w_ = blur_stepfun(sdists[-1], wn, pulse_width[i])
            area = 0.5 * (w_[..., 1:] + w_[..., :-1]) * (c_[..., 1:] - c_[..., :-1])
            cdf = np.concatenate([np.zeros_like(area[..., :1]), np.cumsum(area, -1)], -1)
            ws = np.diff(sorted_interp_quad(cp, c_, w_, cdf), axis=-1)
            over = np.maximum(ws - wp, 0)
            tot += (over ** 2 / (wp + 1e-5)).mean()
            G["w%d" % i] = interlevel_mult / wp.size * (-2 * over / (wp + 1e-5) - over ** 2 / (wp + 1e-5) ** 2)
        L["interlevel"] = interlevel_mult * tot
    L["total"] = sum(v for k, v in L.items() if k != "mse")
    return L, G


# ---------------------------------------------------------------- frame writer (row 8f-3) ----
def frame_quantize(rgb=None, depth=None, semantic=None, color_map=None, scale_factor=1.0):
    """The quantisation the reference applies before PIL writes the frame (s-nerfpp/zipnerf/random_render_waymo_seq.py:214-227):
    rgb: internal/utils.py:111-116 `(np.clip(np.nan_to_num(img), 0., 1.) * 255.).astype(np.uint8)`; depth :218-219
    `(dep * 256 / scale_factor).astype(np.uint16)`; labels :222-223 `np.argmax(semantic, -1).astype(np.uint8)`; paint :225
    `color_map[labels].astype(np.uint8)`.  The same numpy expressions, so the (platform-defined) float -> integer casts are numpy's."""
    out = {}
    with np.errstate(invalid="ignore", over="ignore"):
        if rgb is not None:
            out["rgb"] = (np.clip(np.nan_to_num(np.asarray(rgb, np.float32)), 0., 1.) * 255.).astype(np.uint8)
        if depth is not None:
            out["depth"] = (np.asarray(depth, np.float32) * 256 / scale_factor).astype(np.uint16)
        if semantic is not None:
            labels = np.argmax(np.asarray(semantic, np.float32), axis=-1)
            out["semantic"] = labels.astype(np.uint8)
            if color_map is not None:
                out["paint"] = np.asarray(color_map)[labels].astype(np.uint8)
    return out
