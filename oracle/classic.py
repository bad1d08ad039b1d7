"""Oracle for path B -- the classic NeRF renderer behind ``render_rays`` /
``run_network`` (test infrastructure only, see oracle/__init__.py).

Every function restates one reference function on the CPU; citations are
relative to /root/reference/s-nerf/model/.

Weights are passed as plain ``dict[str, Tensor]`` with the reference module's
``state_dict`` key names (``pts_linears.{i}.weight`` ...), so a checkpoint of
the reference's ``NeRF`` module can be fed in unchanged.
"""
import numpy as np
import torch

from .common import seq_cumsum_f32, seq_sum_f32


# --------------------------------------------------------------------------
# B2  positional encoding -- run_nerf_helpers.py:22-70 (Embedder/get_embedder)
# --------------------------------------------------------------------------
def embed(x: torch.Tensor, num_freqs: int) -> torch.Tensor:
    """[x, sin(2^0 x), cos(2^0 x), sin(2^1 x), cos(2^1 x), ...] on the last axis.

    Interleaved (sin, cos) per frequency, log-sampled bands 2^0..2^(L-1),
    identity first (run_nerf_helpers.py:32-46, 59-66).  Output width
    3 * (1 + 2 L): 63 for L=10 (points), 27 for L=4 (view directions).
    """
    outs = [x]
    for k in range(num_freqs):
        f = float(2 ** k)
        outs.append(torch.sin(x * f))
        outs.append(torch.cos(x * f))
    return torch.cat(outs, -1)


# --------------------------------------------------------------------------
# B4  NeRF MLP -- run_nerf_helpers.py:74-126
# --------------------------------------------------------------------------
def nerf_mlp(p: dict, x: torch.Tensor, input_ch: int = 63, input_ch_views: int = 27,
             D: int = 8, skips=(4,), use_viewdirs: bool = True) -> torch.Tensor:
    """8 x W ReLU trunk with ``cat([input_pts, h])`` AFTER layer ``skips`` (input
    first), alpha head W->1, feature W->W (no activation), views layer
    [feature, views] -> W/2 ReLU, rgb W/2->3; returns cat([rgb, alpha])
    (run_nerf_helpers.py:103-126)."""
    F = torch.nn.functional
    pts, views = x[..., :input_ch], x[..., input_ch:input_ch + input_ch_views]
    h = pts
    for i in range(D):
        h = F.relu(F.linear(h, p[f"pts_linears.{i}.weight"], p[f"pts_linears.{i}.bias"]))
        if i in skips:
            h = torch.cat([pts, h], -1)
    if use_viewdirs:
        alpha = F.linear(h, p["alpha_linear.weight"], p["alpha_linear.bias"])
        feat = F.linear(h, p["feature_linear.weight"], p["feature_linear.bias"])
        h = torch.cat([feat, views], -1)
        h = F.relu(F.linear(h, p["views_linears.0.weight"], p["views_linears.0.bias"]))
        rgb = F.linear(h, p["rgb_linear.weight"], p["rgb_linear.bias"])
        return torch.cat([rgb, alpha], -1)
    return F.linear(h, p["output_linear.weight"], p["output_linear.bias"])


# --------------------------------------------------------------------------
# B3  run_network -- run_nerf_helpers.py:460-474
# --------------------------------------------------------------------------
def run_network(pts: torch.Tensor, viewdirs, p: dict, multires: int = 10, multires_views: int = 4,
                **mlp_kw) -> torch.Tensor:
    """Flatten [N,S,3] points, embed (63-d), broadcast + embed the per-ray view
    direction (27-d), concatenate, run the MLP, reshape to [N,S,4]."""
    flat = pts.reshape(-1, pts.shape[-1])
    e = embed(flat, multires)
    if viewdirs is not None:
        d = viewdirs[:, None].expand(pts.shape).reshape(-1, pts.shape[-1])
        e = torch.cat([e, embed(d, multires_views)], -1)
    out = nerf_mlp(p, e, input_ch=3 * (1 + 2 * multires),
                   input_ch_views=3 * (1 + 2 * multires_views) if viewdirs is not None else 0,
                   use_viewdirs=viewdirs is not None, **mlp_kw)
    return out.reshape(list(pts.shape[:-1]) + [out.shape[-1]])


# --------------------------------------------------------------------------
# B5  raw2outputs -- run_nerf_helpers.py:381-424
# --------------------------------------------------------------------------
def raw2outputs(raw: torch.Tensor, z_vals: torch.Tensor, rays_d: torch.Tensor,
                noise=None, white_bkgd: bool = False):
    """dists = [diff z, 1e10] * |d|; rgb = sigmoid; alpha = 1 - exp(-relu(sigma + noise) dists);
    w = alpha * exclusive-cumprod(1 - alpha + 1e-10); rgb_map, depth = sum w z,
    disp = 1 / max(1e-10, depth / sum w), acc = sum w; optional white background.
    Returns (rgb_map, disp_map, acc_map, weights, depth_map)."""
    dists = z_vals[..., 1:] - z_vals[..., :-1]
    dists = torch.cat([dists, torch.full_like(dists[..., :1], 1e10)], -1)
    dists = dists * torch.norm(rays_d[..., None, :], dim=-1)
    rgb = torch.sigmoid(raw[..., :3])
    sigma = raw[..., 3] if noise is None else raw[..., 3] + noise
    alpha = 1.0 - torch.exp(-torch.relu(sigma) * dists)
    trans = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1.0 - alpha + 1e-10], -1), -1)[:, :-1]
    weights = alpha * trans
    rgb_map = torch.sum(weights[..., None] * rgb, -2)
    depth_map = torch.sum(weights * z_vals, -1)
    acc_map = torch.sum(weights, -1)
    disp_map = 1.0 / torch.max(1e-10 * torch.ones_like(depth_map), depth_map / acc_map)
    if white_bkgd:
        rgb_map = rgb_map + (1.0 - acc_map[..., None])
    return rgb_map, disp_map, acc_map, weights, depth_map


# --------------------------------------------------------------------------
# B6  sample_pdf -- run_nerf_helpers.py:336-379
# --------------------------------------------------------------------------
def sample_pdf(bins: torch.Tensor, weights: torch.Tensor, u: torch.Tensor, sum_mode: str = "canonical"):
    """Inverse-CDF sampling with explicit uniforms ``u`` [N, Nf].

    pdf = (w + 1e-5) / sum; cdf = [0, cumsum(pdf)]; inds = searchsorted(cdf, u,
    right=True) = #(cdf <= u); below = max(0, inds-1), above = min(len-1, inds);
    t = (u - cdf[below]) / denom with denom < 1e-5 -> 1; lerp of bins.
    Sums use the canonical order of oracle/common.py.  Returns (samples fp32
    [N,Nf], inds int64 [N,Nf]) -- ``inds`` is the bit-exact target.
    (det=True in the reference is u = linspace(0, 1, Nf).)
    """
    w = weights.detach().cpu().numpy().astype(np.float32) + np.float32(1e-5)
    b = bins.detach().cpu().numpy().astype(np.float32)
    un = np.ascontiguousarray(u.detach().cpu().numpy().astype(np.float32))
    # sum_mode="torch": row sum exactly as the reference's torch.sum on this machine (pinning only)
    wsum = torch.sum(torch.from_numpy(w), -1, keepdim=True).numpy() if sum_mode == "torch" else seq_sum_f32(w)
    pdf = (w / wsum).astype(np.float32)
    cdf = np.concatenate([np.zeros_like(pdf[..., :1]), seq_cumsum_f32(pdf)], -1)
    n, nb = cdf.shape
    inds = np.empty(un.shape, dtype=np.int64)
    for r in range(n):
        inds[r] = np.searchsorted(cdf[r], un[r], side="right")
    below = np.maximum(0, inds - 1)
    above = np.minimum(nb - 1, inds)
    c0 = np.take_along_axis(cdf, below, -1)
    c1 = np.take_along_axis(cdf, above, -1)
    b0 = np.take_along_axis(b, below, -1)
    b1 = np.take_along_axis(b, above, -1)
    denom = (c1 - c0).astype(np.float32)
    denom = np.where(denom < np.float32(1e-5), np.float32(1.0), denom)
    t = ((un - c0) / denom).astype(np.float32)
    samples = (b0 + (t * (b1 - b0)).astype(np.float32)).astype(np.float32)
    return torch.from_numpy(samples), torch.from_numpy(inds)


# --------------------------------------------------------------------------
# B1  render_rays -- render.py:281-409
# --------------------------------------------------------------------------
def stratified_z(near, far, n_samples: int, lindisp: bool, t_rand=None):
    """render.py:330-352: linspace in depth (or disparity) and optional
    stratified jitter with explicit uniforms ``t_rand`` [N, S]."""
    t = torch.linspace(0.0, 1.0, steps=n_samples)
    if not lindisp:
        z = near * (1.0 - t) + far * t
    else:
        z = 1.0 / (1.0 / near * (1.0 - t) + 1.0 / far * t)
    z = z.expand([near.shape[0], n_samples])
    if t_rand is not None:
        mids = 0.5 * (z[..., 1:] + z[..., :-1])
        upper = torch.cat([mids, z[..., -1:]], -1)
        lower = torch.cat([z[..., :1], mids], -1)
        z = lower + (upper - lower) * t_rand
    return z


def render_rays(ray_batch: torch.Tensor, p_coarse: dict, p_fine, N_samples: int, N_importance: int = 0,
                lindisp: bool = False, t_rand=None, u=None, white_bkgd: bool = False,
                noise0=None, noise1=None, retraw: bool = False, sum_mode: str = "canonical"):
    """Classic coarse -> fine pipeline.  Random draws are explicit inputs:
    ``t_rand`` [N,S] (None = perturb 0), ``u`` [N,Nimp] (None = det linspace).
    Returns the reference's dict keys (render.py:394-401) plus ``inds``."""
    N = ray_batch.shape[0]
    rays_o, rays_d = ray_batch[:, 0:3], ray_batch[:, 3:6]
    viewdirs = ray_batch[:, -3:] if ray_batch.shape[-1] > 9 else None
    near, far = ray_batch[:, 6:7], ray_batch[:, 7:8]
    z_vals = stratified_z(near, far, N_samples, lindisp, t_rand)
    pts = rays_o[..., None, :] + rays_d[..., None, :] * z_vals[..., :, None]
    raw = run_network(pts, viewdirs, p_coarse)
    rgb_map, disp_map, acc_map, weights, depth_map = raw2outputs(raw, z_vals, rays_d, noise0, white_bkgd)
    ret = {}
    if N_importance > 0:
        rgb0, disp0, acc0 = rgb_map, disp_map, acc_map
        z_mid = 0.5 * (z_vals[..., 1:] + z_vals[..., :-1])
        if u is None:
            u = torch.linspace(0.0, 1.0, steps=N_importance).expand(N, N_importance)
        z_samples, inds = sample_pdf(z_mid, weights[..., 1:-1], u, sum_mode)
        z_all, _ = torch.sort(torch.cat([z_vals, z_samples], -1), -1)
        pts = rays_o[..., None, :] + rays_d[..., None, :] * z_all[..., :, None]
        raw = run_network(pts, viewdirs, p_fine if p_fine is not None else p_coarse)
        rgb_map, disp_map, acc_map, _, depth_map = raw2outputs(raw, z_all, rays_d, noise1, white_bkgd)
        ret.update(rgb0=rgb0, disp0=disp0, acc0=acc0,
                   z_std=torch.std(z_samples, dim=-1, unbiased=False), inds=inds, z_samples=z_samples,
                   z_vals_fine=z_all)
    ret.update(rgb_map=rgb_map, disp_map=disp_map, acc_map=acc_map, depth_map=depth_map,
               z_vals_map=z_vals, weights=weights)
    if retraw:
        ret["raw"] = raw
    return ret


def nerf_param_shapes(D=8, W=256, input_ch=63, input_ch_views=27, skips=(4,)):
    """Ordered (name, shape) list of the reference ``NeRF`` module with
    use_viewdirs=True (run_nerf_helpers.py:86-101)."""
    out = []
    for i in range(D):
        k = input_ch if i == 0 else (W + input_ch if (i - 1) in skips else W)
        out += [(f"pts_linears.{i}.weight", (W, k)), (f"pts_linears.{i}.bias", (W,))]
    out += [("views_linears.0.weight", (W // 2, input_ch_views + W)), ("views_linears.0.bias", (W // 2,)),
            ("feature_linear.weight", (W, W)), ("feature_linear.bias", (W,)),
            ("alpha_linear.weight", (1, W)), ("alpha_linear.bias", (1,)),
            ("rgb_linear.weight", (3, W // 2)), ("rgb_linear.bias", (3,))]
    return out


# --------------------------------------------------------------------------
# B4'  NeRF_RGB -- run_nerf_helpers.py:157-212
# --------------------------------------------------------------------------
def nerf_rgb_mlp(p: dict, p_alpha: dict, x: torch.Tensor, input_ch: int = 63, input_ch_views: int = 27,
                 D: int = 8, skips=(4,)) -> torch.Tensor:
    """The colour network of the two-stage variant: same trunk / feature / views / rgb layers as ``NeRF`` but NO alpha head;
    the density column comes from a frozen, separately trained ``alpha_model`` evaluated on the same embedded input under
    no_grad (run_nerf_helpers.py:188-212).  ``p`` holds the NeRF_RGB parameters without the ``alpha_model.`` entries."""
    F = torch.nn.functional
    pts, views = x[..., :input_ch], x[..., input_ch:input_ch + input_ch_views]
    h = pts
    for i in range(D):
        h = F.relu(F.linear(h, p[f"pts_linears.{i}.weight"], p[f"pts_linears.{i}.bias"]))
        if i in skips:
            h = torch.cat([pts, h], -1)
    with torch.no_grad():
        alpha = nerf_mlp(p_alpha, x, input_ch, input_ch_views, D, skips)[..., 3:4]
    feat = F.linear(h, p["feature_linear.weight"], p["feature_linear.bias"])
    h = F.relu(F.linear(torch.cat([feat, views], -1), p["views_linears.0.weight"], p["views_linears.0.bias"]))
    return torch.cat([F.linear(h, p["rgb_linear.weight"], p["rgb_linear.bias"]), alpha], -1)


# --------------------------------------------------------------------------
# B7  get_rays / ndc_rays / render -- run_nerf_helpers.py:247-258, 314-332; render.py:22-91
# --------------------------------------------------------------------------
def get_rays(H: int, W: int, focal: float, c2w: torch.Tensor, ori_points=None):
    """Pinhole rays of the whole frame, pixel centres at +0.5 (run_nerf_helpers.py:247-258): dirs = [((i + .5) - cx) / f,
    -((j + .5) - cy) / f, -1], rays_d = sum_k dirs[k] c2w[:, k] (left to right), rays_o = c2w[:, 3].  -> ([H,W,3], [H,W,3])"""
    cx, cy = (W * 0.5, H * 0.5) if not ori_points else ori_points
    j, i = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    c0 = ((i + 0.5) - cx) / focal
    c1 = -((j + 0.5) - cy) / focal
    c2w = c2w.float()
    rays_d = torch.stack([(c0 * c2w[r, 0] + c1 * c2w[r, 1]) + (-1.0) * c2w[r, 2] for r in range(3)], -1)
    return c2w[:3, -1].expand(rays_d.shape), rays_d


def ndc_rays(H: int, W: int, focal: float, near: float, rays_o: torch.Tensor, rays_d: torch.Tensor):
    """Shift the origins to the near plane and project into NDC (run_nerf_helpers.py:314-332)."""
    t = -(near + rays_o[..., 2]) / rays_d[..., 2]
    rays_o = rays_o + t[..., None] * rays_d
    cw, ch = -1.0 / (W / (2.0 * focal)), -1.0 / (H / (2.0 * focal))
    o0 = cw * rays_o[..., 0] / rays_o[..., 2]
    o1 = ch * rays_o[..., 1] / rays_o[..., 2]
    o2 = 1.0 + 2.0 * near / rays_o[..., 2]
    d0 = cw * (rays_d[..., 0] / rays_d[..., 2] - rays_o[..., 0] / rays_o[..., 2])
    d1 = ch * (rays_d[..., 1] / rays_d[..., 2] - rays_o[..., 1] / rays_o[..., 2])
    d2 = -2.0 * near / rays_o[..., 2]
    return torch.stack([o0, o1, o2], -1), torch.stack([d0, d1, d2], -1)


def ray_batch(H, W, focal, rays=None, c2w=None, ndc=True, near=0.0, far=1.0, use_viewdirs=False, c2w_staticcam=None,
              depths=None, ori_points=None):
    """The ray-batch assembly of render() (render.py:50-77): -> (rows [N, 8 (+1) (+3)], sh = shape of rays_d)."""
    if c2w is not None:
        rays_o, rays_d = get_rays(H, W, focal, c2w, ori_points)
    else:
        rays_o, rays_d = rays
    viewdirs = None
    if use_viewdirs:
        viewdirs = rays_d
        if c2w_staticcam is not None:
            rays_o, rays_d = get_rays(H, W, focal, c2w_staticcam)
        viewdirs = viewdirs / torch.norm(viewdirs, dim=-1, keepdim=True)
        viewdirs = viewdirs.reshape(-1, 3).float()
    sh = rays_d.shape
    if ndc:
        rays_o, rays_d = ndc_rays(H, W, focal, 1.0, rays_o, rays_d)
    rays_o, rays_d = rays_o.reshape(-1, 3).float(), rays_d.reshape(-1, 3).float()
    cols = [rays_o, rays_d, near * torch.ones_like(rays_d[..., :1]), far * torch.ones_like(rays_d[..., :1])]
    if depths is not None:
        cols.append(depths.reshape(-1, 1))
    if use_viewdirs:
        cols.append(viewdirs)
    return torch.cat(cols, -1), sh


def render(H, W, focal, p_coarse, p_fine=None, chunk=1024 * 32, rays=None, c2w=None, ndc=True, near=0.0, far=1.0,
           use_viewdirs=False, c2w_staticcam=None, depths=None, ori_points=None, **render_rays_kw):
    """render() (render.py:22-91) on top of this module's render_rays: chunk loop, reshape to the ray grid's shape,
    -> [rgb_map, disp_map, acc_map, depth_map, extras]."""
    rows, sh = ray_batch(H, W, focal, rays, c2w, ndc, near, far, use_viewdirs, c2w_staticcam, depths, ori_points)
    parts = {}
    for i in range(0, rows.shape[0], chunk):
        ret = render_rays(rows[i:i + chunk], p_coarse, p_fine, **render_rays_kw)
        for k, v in ret.items():
            parts.setdefault(k, []).append(v)
    all_ret = {k: torch.cat(v, 0) for k, v in parts.items()}
    all_ret = {k: v.reshape(list(sh[:-1]) + list(v.shape[1:])) for k, v in all_ret.items()}
    main = ["rgb_map", "disp_map", "acc_map", "depth_map"]
    return [all_ret[k] for k in main] + [{k: v for k, v in all_ret.items() if k not in main}]
