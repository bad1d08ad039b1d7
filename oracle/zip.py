"""Oracle for path C -- the S-NeRF++ / zipnerf background model (test infrastructure only, see oracle/__init__.py).

Citations are relative to /root/reference/s-nerfpp/zipnerf/internal/.  Restated: the branch `configs/waymo.gin` runs
(`Model.raydist_fn='power_transformation'`, `opaque_background=True`, `PropMLP(disable_density_normals, disable_rgb,
grid_level_dim=1)`, `NerfMLP(disable_density_normals, deg_view=1)`), class defaults otherwise (2 proposal levels of 64
intervals + 32 NeRF intervals, 7 multisamples on a 3-turn helix, no GLO, no semantic head).  The hash-grid features come
from oracle/grid.py.  Pinned by golden vectors G10/G11 captured from the imported reference with this oracle's grid
encoder injected as its `gridencoder` module (oracle/gen_golden_zip.py).
"""
import math

import numpy as np
import torch

from . import grid as og

F = torch.nn.functional
EPS32 = float(torch.finfo(torch.float32).eps)


# ------------------------------------------------------------------ C1 ----
def power_transformation(x, lam):
    """coord.py:103-108"""
    lam_1 = abs(lam - 1)
    return lam_1 / lam * ((x / lam_1 + 1) ** lam - 1)


def inv_power_transformation(x, lam):
    """coord.py:111-118 (note the + eps inside the power)"""
    lam_1 = abs(lam - 1)
    return ((x * lam / lam_1 + 1 + EPS32) ** (1 / lam) - 1) * lam_1


def s_to_t(s, t_near, t_far, lam=-1.5):
    """construct_ray_warps('power_transformation', near, far, lam)[1], coord.py:121-162:
    s_to_t(s) = inv_power(s * s_far + (1 - s) * s_near) / 2 with s_x = power(2 t_x)."""
    s_near, s_far = power_transformation(t_near * 2, lam), power_transformation(t_far * 2, lam)
    return inv_power_transformation(s * s_far + (1 - s) * s_near, lam) / 2


# ------------------------------------------------------------------ C2 ----
def max_dilate_weights(t, w, dilation, domain=(0.0, 1.0)):
    """stepfun.py:75-105 with renormalize=True: pdf = w / dt; fence posts = sort(cat(t, t0 - d, t1 + d)) clipped to the
    domain; dilated pdf = max over the intervals that cover each new post; back to weights; renormalise.  The reference
    evaluates an [R, 3S+2, S] mask; this is the same max expressed per new post."""
    dt = (t[..., 1:] - t[..., :-1]).clamp_min(EPS32)
    p = w / dt
    t0, t1 = t[..., :-1] - dilation, t[..., 1:] + dilation
    t_dil, _ = torch.sort(torch.cat([t, t0, t1], dim=-1), dim=-1)
    t_dil = torch.clip(t_dil, *domain)
    cover = (t0[..., None, :] <= t_dil[..., None]) & (t1[..., None, :] > t_dil[..., None])
    p_dil = torch.max(torch.where(cover, p[..., None, :], torch.zeros_like(p[..., None, :])), dim=-1).values[..., :-1]
    w_dil = p_dil * (t_dil[..., 1:] - t_dil[..., :-1])
    w_dil = w_dil / torch.sum(w_dil, dim=-1, keepdim=True).clamp_min(EPS32)
    return t_dil, w_dil


# ------------------------------------------------------------------ C3 ----
def det_centers_u(num_samples):
    """stepfun.py:199-205 (rand=None, deterministic_center=True): linspace(pad, 1 - pad - eps, n), pad = 1/(2n)"""
    pad = 1 / (2 * num_samples)
    return torch.linspace(pad, 1. - pad - EPS32, num_samples)


def rand_u(num_samples, jitter01):
    """stepfun.py:209-216: linspace(0, 1 - u_max, n) + rand * max_jitter; `jitter01` = the torch.rand draw, [R,1] (single
    jitter) or [R,n]."""
    u_max = EPS32 + (1 - EPS32) / num_samples
    max_jitter = (1 - u_max) / (num_samples - 1) - EPS32
    return torch.linspace(0, 1 - u_max, num_samples) + jitter01 * max_jitter


def sorted_interp(x, xp, fp):
    """math.py:88-107 via the interval index #(x >= xp) - 1 (mask-free; identical for sorted xp)."""
    idx = (torch.searchsorted(xp.contiguous(), x.contiguous(), right=True) - 1).clamp(0, xp.shape[-1] - 1)
    idx1 = (idx + 1).clamp(max=xp.shape[-1] - 1)
    fp0, fp1 = torch.gather(fp, -1, idx), torch.gather(fp, -1, idx1)
    xp0, xp1 = torch.gather(xp, -1, idx), torch.gather(xp, -1, idx1)
    off = torch.clip(torch.nan_to_num((x - xp0) / (xp1 - xp0), 0), 0, 1)
    return fp0 + off * (fp1 - fp0), idx


def sample_intervals(t, w_logits, u, domain=(0.0, 1.0)):
    """stepfun.py:251-294 + 175-218 + 154-161 + 108-128: softmax -> CDF [0, min(1, cumsum(w[:-1])), 1] -> centres by
    inverse-CDF interpolation at `u` -> fence posts = midpoints, end posts reflected and clamped to the domain.
    Returns (sdist [R, n+1], idx [R, n])."""
    w = torch.softmax(w_logits, dim=-1)
    cw = torch.cumsum(w[..., :-1], dim=-1).clamp_max(1)
    one = w.new_ones(w.shape[:-1] + (1,))
    cw0 = torch.cat([0 * one, cw, one], dim=-1)
    if u.dim() == 1:
        u = torch.broadcast_to(u, t.shape[:-1] + u.shape)
    centers, idx = sorted_interp(u, cw0, t)
    mid = (centers[..., 1:] + centers[..., :-1]) / 2
    first = (2 * centers[..., :1] - mid[..., :1]).clamp_min(domain[0])
    last = (2 * centers[..., -1:] - mid[..., -1:]).clamp_max(domain[1])
    return torch.cat([first, mid, last], dim=-1), idx


def resample_logits(sdist, weights, anneal=1.0, resample_padding=0.0):
    """models.py:196-203: anneal * log(w + padding), -inf for zero-width intervals."""
    return torch.where(sdist[..., 1:] > sdist[..., :-1], anneal * torch.log(weights + resample_padding),
                       torch.full_like(sdist[..., :-1], -float("inf")))


# ------------------------------------------------------------------ C4 ----
def cast_rays(tdist, origins, directions, radii, base_x, base_y, deg_jitter=None, n=7, m=3, std_scale=0.35):
    """render.py:129-168: n multisamples per interval on an m-turn helix; local (r t cos/2, r t sin/2, t) rotated by
    [base_x, base_y, directions]; std = std_scale * r * t.  deg_jitter (= the torch.rand_like(deg) draw) or None."""
    t0, t1 = tdist[..., :-1], tdist[..., 1:]
    j = torch.arange(n)
    t = t0[..., None] + (t1[..., None] - t0[..., None]) * (j + 0.5) / n
    deg = torch.broadcast_to(2 * torch.pi * m * j / n, t.shape)
    if deg_jitter is not None:
        deg = deg + deg_jitter * torch.pi * 2
    means = torch.stack([radii[..., None] * t * torch.cos(deg) / 2, radii[..., None] * t * torch.sin(deg) / 2, t], dim=-1)
    stds = std_scale * radii[..., None] * t
    basis = torch.stack([base_x, base_y, directions], dim=-1)
    means = torch.matmul(means, basis[..., None, :, :].transpose(-1, -2))
    return means + origins[..., None, None, :], stds


# ------------------------------------------------------------------ C5 ----
def contract_mean_std(x, std):
    """coord.py:51-63: z = |x|^2 <= 1 ? x : (2|x| - 1)/|x|^2 x; std *= det(J)^(1/3), det = (1/|x|^2) (2/|x| - 1/|x|^2)^2."""
    x_mag_sq = torch.sum(x ** 2, dim=-1, keepdim=True).clamp_min(EPS32)
    x_mag = torch.sqrt(x_mag_sq)
    mask = x_mag_sq <= 1
    z = torch.where(mask, x, ((2 * torch.sqrt(x_mag_sq) - 1) / x_mag_sq) * x)
    det = ((1 / x_mag_sq) * ((2 / x_mag - 1 / x_mag_sq) ** 2))[..., 0]
    std = torch.where(mask[..., 0], std, (det ** (1 / x.shape[-1])) * std)
    return z, std


# --------------------------------------------------------------- C6, C7 ----
class GridSpec:
    """level layout of one encoder (grid.py:96-149)"""

    def __init__(self, num_levels, level_dim, desired_resolution, base_resolution=16, log2_hashmap_size=21):
        self.L, self.C, self.H = num_levels, level_dim, base_resolution
        self.offsets, self.res, self.scale = og.level_layout(3, num_levels, level_dim, 2.0, base_resolution, log2_hashmap_size,
                                                             desired_resolution, False)
        self.S = float(np.log2(self.scale))
        self.rows = int(self.offsets[-1])


def grid_features(spec: GridSpec, emb, means):
    """GridEncoder.forward(means, bound=1) (grid.py:158-177) -> [..., L, C] via oracle/grid.py"""
    x = ((means.reshape(-1, 3) + 1) / 2).numpy().astype(np.float32)
    out = og.grid_encode_forward(x, emb.detach().numpy(), spec.offsets, spec.S, spec.H, 0, False, 0)   # [L,B,C]
    return torch.from_numpy(out).permute(1, 0, 2).reshape(list(means.shape[:-1]) + [spec.L, spec.C])


def predict_density(p, prefix, spec, means, stds):
    """models.py:481-519: contract, /2, grid features, erf down-weighting w_l = erf(1/sqrt(8 std^2 res_l^2)), mean over the
    multisamples, density_layer (Linear-ReLU-Linear).  Returns (raw_density [...], x [..., out])."""
    pre = means.shape[:-1]
    m, s = contract_mean_std(means.reshape(-1, 3), stds.reshape(-1))
    m, s = m.reshape(*pre, 3) / 2, s.reshape(*pre) / 2
    feats = grid_features(spec, p[prefix + "encoder.embeddings"], m)
    grid_sizes = torch.from_numpy(spec.res.astype(np.int32)).to(s.device)
    w = torch.erf(1 / torch.sqrt(8 * s[..., None] ** 2 * grid_sizes ** 2))
    feats = (feats * w[..., None]).mean(dim=-3).flatten(-2, -1)
    x = F.linear(F.relu(F.linear(feats, p[prefix + "density_layer.0.weight"], p[prefix + "density_layer.0.bias"])),
                 p[prefix + "density_layer.2.weight"], p[prefix + "density_layer.2.bias"])
    return x[..., 0], x


# ------------------------------------------------------------------ C8 ----
def pos_enc(x, min_deg, max_deg, append_identity=True):
    """coord.py:199-210"""
    scales = 2.0 ** torch.arange(min_deg, max_deg)
    xb = (x[..., None, :] * scales[:, None]).reshape(list(x.shape[:-1]) + [-1])
    four = torch.sin(torch.cat([xb, xb + 0.5 * torch.pi], dim=-1))
    return torch.cat([x, four], dim=-1) if append_identity else four


def mlp_forward(p, prefix, spec, means, stds, viewdirs, disable_rgb, deg_view=1, density_bias=-1.0, rgb_padding=0.001,
                net_depth_viewdirs=2, skip_layer_dir=0, use_semantic=False, class_num=19, glo_vec=None, net_depth_glo=2):
    """MLP.forward (models.py:521-714) on the waymo.gin branch.  Returns dict(density, rgb, semantic); with `use_semantic`
    semantic = softmax(x[..., 1:1+class_num]) of the density network's output (models.py:594-597), else None.  `glo_vec` [R, F]
    (num_glo_features > 0, models.py:620-630): the per-ray GLO vector goes through lin_glo_0 (ReLU) .. lin_glo_{depth-1} to a
    (scale, shift) pair that modulates the bottleneck, bottleneck * exp(scale) + shift, before the view-dependent layers."""
    raw_density, x = predict_density(p, prefix, spec, means, stds)
    density = F.softplus(raw_density + density_bias)
    if disable_rgb:
        return dict(density=density, rgb=torch.zeros(density.shape + (3,)), semantic=None)
    sem = torch.softmax(x[..., 1:1 + class_num], -1) if use_semantic else None
    if glo_vec is not None:
        g = glo_vec
        for i in range(net_depth_glo):
            g = F.linear(g, p[prefix + f"lin_glo_{i}.weight"], p[prefix + f"lin_glo_{i}.bias"])
            if i != net_depth_glo - 1:
                g = F.relu(g)
        scale, shift = torch.broadcast_to(g[..., None, :], x.shape[:-1] + g.shape[-1:]).chunk(2, dim=-1)
        x = x * torch.exp(scale) + shift
    dir_enc = pos_enc(viewdirs, 0, deg_view, True)
    dir_enc = torch.broadcast_to(dir_enc[..., None, :], x.shape[:-1] + (dir_enc.shape[-1],))
    h = torch.cat([x, dir_enc], dim=-1)
    inputs = h
    for i in range(net_depth_viewdirs):
        h = F.relu(F.linear(h, p[prefix + f"lin_second_stage_{i}.weight"], p[prefix + f"lin_second_stage_{i}.bias"]))
        if i == skip_layer_dir:
            h = torch.cat([h, inputs], dim=-1)
    rgb = torch.sigmoid(F.linear(h, p[prefix + "rgb_layer.weight"], p[prefix + "rgb_layer.bias"]))
    return dict(density=density, rgb=rgb * (1 + 2 * rgb_padding) - rgb_padding, semantic=sem)


# ------------------------------------------------------------------ C9 ----
def compute_alpha_weights(density, tdist, dirs, opaque_background=True):
    """render.py:170-189"""
    delta = (tdist[..., 1:] - tdist[..., :-1]) * torch.norm(dirs[..., None, :], dim=-1)
    dd = density * delta
    if opaque_background:
        dd = torch.cat([dd[..., :-1], torch.full_like(dd[..., -1:], float("inf"))], dim=-1)
    alpha = 1 - torch.exp(-dd)
    trans = torch.exp(-torch.cat([torch.zeros_like(dd[..., :1]), torch.cumsum(dd[..., :-1], dim=-1)], dim=-1))
    return alpha * trans


def weighted_percentile(t, w, ps):
    """stepfun.py:329-339: percentiles `ps` (in %) of the step function (t, w), w summing to 1: interpolate t at ps/100 in the
    integrated weights [0, min(1, cumsum(w[:-1])), 1] (:106-126)."""
    cw = torch.cumsum(w[..., :-1], dim=-1).clamp_max(1)
    cw0 = torch.cat([torch.zeros_like(cw[..., :1]), cw, torch.ones_like(cw[..., :1])], dim=-1)
    x = (torch.tensor(ps, dtype=t.dtype) / 100).expand(t.shape[0], len(ps))
    return sorted_interp(x, cw0, t)[0]


def volumetric_rendering(rgbs, weights, tdist, bg_rgbs=1.0, semantic=None, compute_extras=False, t_far=None):
    """render.py:192-269: rgb = sum w c + max(0, 1 - acc) bg; depth = clip(exp(sum w log t_mid / acc));
    semantic = sum detach(w) semantic (:237-241: "no influences to density").  compute_extras (:243-267): acc, distance_mean (the same
    log-space expectation) and the 5 / 50 / 95 % distance percentiles of the histogram extended by a far-plane fence post that
    carries the background weight."""
    acc = weights.sum(dim=-1)
    bg_w = (1 - acc[..., None]).clamp_min(0.)
    rgb = (weights[..., None] * rgbs).sum(dim=-2) + bg_w * bg_rgbs
    t_mids = 0.5 * (tdist[..., :-1] + tdist[..., 1:])
    expect = (weights * torch.log(t_mids)).sum(dim=-1) / acc.clamp_min(EPS32)
    depth = torch.clip(torch.nan_to_num(torch.exp(expect), float("inf")), tdist[..., 0], tdist[..., -1])
    out = dict(rgb=rgb, depth=depth, acc=acc)
    if semantic is not None:
        out["semantic"] = (weights.detach()[..., None] * semantic).sum(dim=-2)
    if compute_extras:
        out["distance_mean"] = depth
        pct = weighted_percentile(torch.cat([tdist, t_far], dim=-1), torch.cat([weights, bg_w], dim=-1), [5, 50, 95])
        out["distance_percentile_5"], out["distance_median"], out["distance_percentile_95"] = pct[..., 0], pct[..., 1], pct[..., 2]
    return out


# ------------------------------------------------------------------ C11 ---
def model_forward(p, specs, batch, num_prop_samples=(64, 64), num_nerf_samples=32, train_frac=1.0, anneal_slope=10.0,
                  dilation_multiplier=0.5, dilation_bias=0.0025, power_lambda=-1.5, std_scale=0.35, jitters=None,
                  deg_jitters=None, bg=1.0, use_semantic=False, compute_extras=False, vis_num_rays=16, sdist_override=None,
                  near_anneal_rate=None, near_anneal_init=0.95, num_glo_features=0, zero_glo=True):
    """Model.forward (models.py:98-349) with rand=None (jitters None) or explicit draws; `use_semantic`: the final level also
    renders the 19-class semantic distribution (models.py:297-305).  `specs` = [prop0, prop1, nerf]
    GridSpec; parameter names follow the reference's state_dict (`prop_mlp_0.encoder.embeddings`, `nerf_mlp.rgb_layer.weight`...).
    `compute_extras`: acc / distance_* per level and the first `vis_num_rays` rays' ray_sdist / ray_weights / ray_rgbs, the proposal
    levels' ray_rgbs replaced by the final level's average colour (models.py:316-346).
    Returns (renderings, ray_history) with the reference's keys (rgb, depth / sdist, weights, tdist)."""
    # GLO (models.py:131-139): the embedding row of every ray's camera, or zeros (`zero_glo`, the default of Model.forward)
    glo_vec = None
    if num_glo_features > 0:
        glo_vec = (torch.zeros(batch["origins"].shape[:-1] + (num_glo_features,)) if zero_glo
                   else p["glo_vecs.weight"][batch["cam_idx"][..., 0].long()])
    near, far = batch["near"], batch["far"]
    # near-bound annealing (models.py:147-158): the first interval starts at clip(1 - train_frac / rate, 0, init) instead of 0
    s_near = 0.0 if near_anneal_rate is None else float(np.clip(1 - train_frac / near_anneal_rate, 0, near_anneal_init))
    sdist = torch.cat([torch.full_like(near, s_near), torch.ones_like(far)], dim=-1)
    weights = torch.ones_like(near)
    prod = 1
    renderings, history = [], []
    n_levels = len(num_prop_samples) + 1
    for lvl in range(n_levels):
        is_prop = lvl < n_levels - 1
        ns = num_prop_samples[lvl] if is_prop else num_nerf_samples
        dilation = dilation_bias + dilation_multiplier * (1.0 - s_near) / prod
        prod *= ns
        if lvl > 0:
            sdist, weights = max_dilate_weights(sdist, weights, dilation, (s_near, 1.0))
            sdist, weights = sdist[..., 1:-1], weights[..., 1:-1]
        anneal = (anneal_slope * train_frac) / ((anneal_slope - 1) * train_frac + 1) if anneal_slope > 0 else 1.0
        logits = resample_logits(sdist, weights, anneal, 0.0)
        u = det_centers_u(ns) if jitters is None else rand_u(ns, jitters[lvl])
        sdist, _ = sample_intervals(sdist, logits, u, (s_near, 1.0))
        sdist = sdist.detach()                      # stop_level_grad, models.py:215-218
        if sdist_override is not None:
            # teacher forcing for conditioning-aware gradient checks (tests only): every level evaluated at GIVEN fence posts -- those
            # another evaluation resampled -- so that both sides see identical sample positions (the posts carry no gradient)
            sdist = sdist_override[lvl].detach()
        tdist = s_to_t(sdist, near, far, power_lambda)
        means, stds = cast_rays(tdist, batch["origins"], batch["directions"], batch["radii"], batch["base_x"], batch["base_y"],
                                None if deg_jitters is None else deg_jitters[lvl], std_scale=std_scale)
        prefix = f"prop_mlp_{lvl}." if is_prop else "nerf_mlp."
        res = mlp_forward(p, prefix, specs[lvl], means, stds, batch["viewdirs"], disable_rgb=is_prop, use_semantic=use_semantic and not is_prop,
                          glo_vec=None if is_prop else glo_vec)
        weights = compute_alpha_weights(res["density"], tdist, batch["directions"], True)
        renderings.append(volumetric_rendering(res["rgb"], weights, tdist, bg, semantic=res["semantic"], compute_extras=compute_extras, t_far=far))
        if compute_extras:
            n = vis_num_rays
            renderings[-1].update(ray_sdist=sdist[:n], ray_weights=weights[:n], ray_rgbs=res["rgb"][:n])
        history.append(dict(sdist=sdist.clone(), weights=weights.clone(), tdist=tdist.clone(), density=res["density"], rgb=res["rgb"]))
    if compute_extras:
        final = torch.sum(renderings[-1]["ray_rgbs"] * renderings[-1]["ray_weights"][..., None], dim=-2)
        for r in renderings[:-1]:
            r["ray_rgbs"] = torch.broadcast_to(final[:, None, :], r["ray_sdist"].shape[:1] + (r["ray_weights"].shape[1], 3))
    return renderings, history


def param_shapes(specs, bottleneck=256, width_view=256, deg_view=1, num_glo_features=0, num_glo_embeddings=1000, net_width_glo=128, zero_glo=True):
    """Ordered (name, shape) of the reference Model's learnable parameters on the waymo.gin branch
    (models.py:393-479; nerf_mlp registered first, then prop_mlp_0, prop_mlp_1; with num_glo_features > 0 the NeRF MLP has lin_glo_0 /
    lin_glo_1 in front of its second stage (:454-459) and, unless config.zero_glo, the model ends with glo_vecs (:75-77))."""
    dim_dir = 3 + 6 * deg_view
    nerf = specs[2]
    glo = [] if num_glo_features <= 0 else [
        ("nerf_mlp.lin_glo_0.weight", (net_width_glo, num_glo_features)), ("nerf_mlp.lin_glo_0.bias", (net_width_glo,)),
        ("nerf_mlp.lin_glo_1.weight", (2 * bottleneck, net_width_glo)), ("nerf_mlp.lin_glo_1.bias", (2 * bottleneck,))]
    out = [("nerf_mlp.encoder.embeddings", (nerf.rows, nerf.C)),
           ("nerf_mlp.density_layer.0.weight", (64, nerf.L * nerf.C)), ("nerf_mlp.density_layer.0.bias", (64,)),
           ("nerf_mlp.density_layer.2.weight", (bottleneck, 64)), ("nerf_mlp.density_layer.2.bias", (bottleneck,))] + glo + [
           ("nerf_mlp.lin_second_stage_0.weight", (width_view, bottleneck + dim_dir)), ("nerf_mlp.lin_second_stage_0.bias", (width_view,)),
           ("nerf_mlp.lin_second_stage_1.weight", (width_view, width_view + bottleneck + dim_dir)), ("nerf_mlp.lin_second_stage_1.bias", (width_view,)),
           ("nerf_mlp.rgb_layer.weight", (3, width_view)), ("nerf_mlp.rgb_layer.bias", (3,))]
    for i in range(2):
        s = specs[i]
        out += [(f"prop_mlp_{i}.encoder.embeddings", (s.rows, s.C)),
                (f"prop_mlp_{i}.density_layer.0.weight", (64, s.L * s.C)), (f"prop_mlp_{i}.density_layer.0.bias", (64,)),
                (f"prop_mlp_{i}.density_layer.2.weight", (1, 64)), (f"prop_mlp_{i}.density_layer.2.bias", (1,))]
    if num_glo_features > 0 and not zero_glo:
        out.append(("glo_vecs.weight", (num_glo_embeddings, num_glo_features)))
    return out
