"""Oracle for path A -- the live S-NeRF ``MipNerfModel`` forward (test
infrastructure only, see oracle/__init__.py).

Citations are relative to /root/reference/s-nerf/model/.  Only the branch the
shipped nuScenes config runs is restated: warp sampling (``no_warp_sample=0``),
contraction ``fn=1`` with the hard-coded radius 3 (mip.py:386), log spacing
(``transform_idx=0``; 1 = disparity and 2 = linear are also provided), cone or
cylinder rays, full-covariance IPE -- of which only the diagonal reaches the
encoding (SURVEY.md section 7 "Full-covariance IPE").

Random draws are explicit inputs (``s_rand``, ``u_jitter``) so that the HIP
path and the oracle consume identical numbers.
"""
import math

import numpy as np
import torch

from .common import seq_cumsum_f32, seq_sum_f32

F = torch.nn.functional
EPS32 = float(torch.finfo(torch.float32).eps)
RADIUS = 3.0  # sample2enc ignores its radius argument: mip.py:386


# ------------------------------------------------------------------ A2 ----
def transform(s, near, far, idx: int = 0):
    """s in [0,1] -> metric t.  idx 0: near*exp(s*log(far/near)) (mip.py:8),
    1: disparity (mip.py:7), else linear (mip.py:9)."""
    if idx == 0:
        return near * torch.exp(s * torch.log(far / near))
    if idx == 1:
        return 1.0 / ((1 - s) / near + s / far)
    return near * (1 - s) + far * s


# ------------------------------------------------------------------ A1 ----
def warp_sample_s(n_rays: int, num_samples: int, s_rand=None):
    """mip.py:268-291: s = linspace(0,1,S+1); stratified jitter between
    midpoints with explicit uniforms ``s_rand`` [N,S+1] (None = not randomized)."""
    s = torch.linspace(0.0, 1.0, num_samples + 1)
    if s_rand is not None:
        mids = 0.5 * (s[..., 1:] + s[..., :-1])
        upper = torch.cat([mids, s[..., -1:]], -1)
        lower = torch.cat([s[..., :1], mids], -1)
        return lower + (upper - lower) * s_rand
    return torch.broadcast_to(s, [n_rays, num_samples + 1])


# ------------------------------------------------------------------ A3 ----
def cast_rays(t_vals, origins, directions, radii, ray_shape: str = "cone"):
    """mip.py:80-91 with diag=True: conical frustum (stable form, mip.py:56-70)
    or cylinder (mip.py:73-77) -> Gaussian mean [N,S,3] and diagonal cov [N,S,3]
    (lift_gaussian, mip.py:31-53)."""
    t0, t1 = t_vals[..., :-1], t_vals[..., 1:]
    d = directions
    if ray_shape == "cone":
        mu = (t0 + t1) / 2
        hw = (t1 - t0) / 2
        t_mean = mu + (2 * mu * hw ** 2) / (3 * mu ** 2 + hw ** 2)
        t_var = (hw ** 2) / 3 - (4 / 15) * ((hw ** 4 * (12 * mu ** 2 - hw ** 2)) / (3 * mu ** 2 + hw ** 2) ** 2)
        r_var = radii ** 2 * ((mu ** 2) / 4 + (5 / 12) * hw ** 2 - 4 / 15 * (hw ** 4) / (3 * mu ** 2 + hw ** 2))
    elif ray_shape == "cylinder":
        t_mean = (t0 + t1) / 2
        r_var = radii ** 2 / 4
        t_var = (t1 - t0) ** 2 / 12
        r_var = torch.broadcast_to(r_var, t_mean.shape)
    else:
        raise ValueError(ray_shape)
    mean = d[..., None, :] * t_mean[..., None]
    d_mag_sq = torch.clamp_min(torch.sum(d ** 2, dim=-1, keepdim=True), 1e-10)
    d_outer_diag = d ** 2
    null_outer_diag = 1 - d_outer_diag / d_mag_sq
    cov = t_var[..., None] * d_outer_diag[..., None, :] + r_var[..., None] * null_outer_diag[..., None, :]
    return mean + origins[..., None, :], cov


# --------------------------------------------------------------- A4, A5 ----
def contract(x):
    """warp_fn.fn2, mip.py:371-374: l = |x| + 1e-8; l > 3 ? (2 - 3/l) x / l : x / 3."""
    l = (torch.linalg.norm(x, dim=-1) + 1e-8)[..., None]
    return (2.0 - RADIUS / l) * x / l * (l > RADIUS) + x / RADIUS * (l <= RADIUS)


def contract_jacobian(x):
    """Jacobi_g, mip.py:343-364: J = (l >= 3) ? (2/l - 3/l^2) I + (6/l^4 - 2/l^3) x x^T : I/3
    with l = |x| + 1e-5 (note: different eps and >= vs > than ``contract``)."""
    nrm = torch.linalg.norm(x, dim=-1)
    ln = 1.0 / (nrm + 1e-5)
    eye = torch.eye(3)
    L = x[..., :, None] * x[..., None, :]
    P1 = (-RADIUS * (ln ** 2) + 2.0 * ln)[..., None, None] * eye
    P2 = (2.0 * RADIUS * ln ** 4 - 2.0 * ln ** 3)[..., None, None] * L
    J1 = P1 + P2
    l = (nrm + 1e-5)[..., None, None]
    return torch.where(l >= RADIUS, J1, (1.0 / RADIUS) * eye.expand(J1.shape))


def warp_view(x, viewc, far):
    """warp_fn.fn1 (``fn == 0``), mip.py:368-369: (x - viewc) / sqrt(|x - viewc| * far), far [N,1] per ray."""
    dx = x - viewc
    return dx / torch.sqrt((torch.linalg.norm(dx, dim=-1) * far)[..., None])


def warp_view_jacobian(x, far):
    """Jacobi_f, mip.py:323-340: (l I - x x^T) / l^(3/2) / sqrt(max far), l = |x| + 1e-5 -- evaluated at the UNSHIFTED means (the
    reference does not subtract viewc here) and scaled by the batch-wide maximum of far."""
    ln = torch.linalg.norm(x, dim=-1) + 1e-5
    L = x[..., :, None] * x[..., None, :]
    J = ln[..., None, None] * torch.eye(3)
    J = (J - L) / (ln ** (3 / 2))[..., None, None]
    return J / torch.sqrt(far.max())


# ------------------------------------------------------------------ A6 ----
def sample2enc(s_vals, origins, directions, radii, near, far, ray_shape="cone", transform_idx=0,
               full_cov: bool = False, fn_idx: int = 1, viewc=0.0):
    """mip.py:381-395: s -> t -> Gaussians -> warped mean + warped covariance
    J diag(c) J^T (``fn_idx`` 1: the contraction, 0: the view-centred warp of mip.py:368-369 + Jacobi_f).  Returns
    (f_means [N,S,3], cov) where cov is the diagonal [N,S,3] (default) or the full matrix [N,S,3,3] (golden comparison)."""
    t = transform(s_vals, near, far, transform_idx)
    means, covs = cast_rays(t, origins, directions, radii, ray_shape)
    if fn_idx == 0:
        f_means = warp_view(means, viewc, far)
        J = warp_view_jacobian(means, far)
    else:
        f_means = contract(means)
        J = contract_jacobian(means)
    if full_cov:
        return f_means, torch.einsum("...ai,...i,...bi->...ab", J, covs, J)
    return f_means, torch.sum(J * J * covs[..., None, :], dim=-1)


# ------------------------------------------------------------------ A7 ----
def safe_sin(x):
    """math_ops.py:6-12: sin(|x| < 100 pi ? x : x mod 100 pi) (python-sign mod)."""
    t = 100 * math.pi
    return torch.sin(torch.where(torch.abs(x) < t, x, x % t))


def integrated_pos_enc(means, cov_diag, min_deg: int = 0, max_deg: int = 16):
    """mip.py:94-118 + expected_sin (mip.py:24-28), first output only:
    y[deg*3+d] = 2^deg x_d, var = 4^deg cov_dd; enc = exp(-var/2) * safe_sin([y, y + pi/2])."""
    scales = torch.tensor([2.0 ** i for i in range(min_deg, max_deg)])
    shape = list(means.shape[:-1]) + [-1]
    y = torch.reshape(means[..., None, :] * scales[:, None], shape)
    y_var = torch.reshape((cov_diag[..., None, :] * scales[:, None]) * scales[:, None], shape)
    x = torch.cat([y, y + 0.5 * math.pi], dim=-1)
    x_var = torch.cat([y_var] * 2, dim=-1)
    return torch.exp(-0.5 * x_var) * safe_sin(x)


# ------------------------------------------------------------------ A10 ---
def pos_enc(x, min_deg: int = 0, max_deg: int = 4, append_identity: bool = True):
    """mip.py:12-21: [x, sin(2^i x) (deg-major), sin(2^i x + pi/2)]."""
    scales = torch.tensor([2.0 ** i for i in range(min_deg, max_deg)])
    xb = torch.reshape(x[..., None, :] * scales[:, None], list(x.shape[:-1]) + [-1])
    four = torch.sin(torch.cat([xb, xb + 0.5 * math.pi], dim=-1))
    return torch.cat([x, four], dim=-1) if append_identity else four


# ------------------------------------------------------------------ A8 ----
def proposal_mlp(p: dict, enc, prefix: str = "proposal."):
    """models.py:299-325: 4 x (Linear + ReLU) then density Linear -> [N,S,1]."""
    n, s = enc.shape[:2]
    x = enc.reshape(-1, enc.shape[-1])
    i = 0
    while f"{prefix}layers.{i}.layers.0.weight" in p:
        x = F.relu(F.linear(x, p[f"{prefix}layers.{i}.layers.0.weight"], p[f"{prefix}layers.{i}.layers.0.bias"]))
        i += 1
    return F.linear(x, p[f"{prefix}density_layer.weight"], p[f"{prefix}density_layer.bias"]).reshape(n, s, -1)


# ------------------------------------------------------------------ A9 ----
def nerf_mlp(p: dict, enc, cond, prefix: str = "mlp.", skip_layer: int = 4):
    """models.py:217-296: 8 DenseBlocks; after layer i (i % 4 == 0, i > 0) the
    input is re-concatenated as cat([x, inputs]) (trunk first); density head
    from the trunk; bottleneck (Linear+ReLU); cat([bottleneck, cond tiled]);
    cond layers (Linear+ReLU each); rgb head; optional semantic head.
    Returns (raw_rgb [N,S,3], raw_density [N,S,1], raw_semantic or None)."""
    n, s = enc.shape[:2]
    x = enc.reshape(-1, enc.shape[-1])
    inputs = x
    i = 0
    while f"{prefix}layers.{i}.layers.0.weight" in p:
        x = F.relu(F.linear(x, p[f"{prefix}layers.{i}.layers.0.weight"], p[f"{prefix}layers.{i}.layers.0.bias"]))
        if i % skip_layer == 0 and i > 0:
            x = torch.cat([x, inputs], dim=-1)
        i += 1
    raw_density = F.linear(x, p[f"{prefix}density_layer.weight"], p[f"{prefix}density_layer.bias"]).reshape(n, s, -1)
    raw_sem = None
    if f"{prefix}semantic_layer.0.layers.0.weight" in p:
        h = F.relu(F.linear(x, p[f"{prefix}semantic_layer.0.layers.0.weight"], p[f"{prefix}semantic_layer.0.layers.0.bias"]))
        raw_sem = F.linear(h, p[f"{prefix}semantic_layer.1.weight"], p[f"{prefix}semantic_layer.1.bias"]).reshape(n, s, -1)
    if cond is not None:
        b = F.relu(F.linear(x, p[f"{prefix}bottleneck_layer.layers.0.weight"], p[f"{prefix}bottleneck_layer.layers.0.bias"]))
        c = cond[:, None, :].expand(n, s, cond.shape[-1]).reshape(-1, cond.shape[-1])
        x = torch.cat([b, c], dim=-1)
        j = 0
        while f"{prefix}cond_layers.{j}.layers.0.weight" in p:
            x = F.relu(F.linear(x, p[f"{prefix}cond_layers.{j}.layers.0.weight"], p[f"{prefix}cond_layers.{j}.layers.0.bias"]))
            j += 1
    raw_rgb = F.linear(x, p[f"{prefix}rgb_layer.weight"], p[f"{prefix}rgb_layer.bias"]).reshape(n, s, -1)
    return raw_rgb, raw_density, raw_sem


# ------------------------------------------------------------------ A11 ---
def activate(raw_rgb, raw_density, rgb_padding: float = 0.001, density_bias: float = -1.0):
    """models.py:166-175."""
    rgb = None if raw_rgb is None else torch.sigmoid(raw_rgb) * (1 + 2 * rgb_padding) - rgb_padding
    return rgb, F.softplus(raw_density + density_bias)


# ------------------------------------------------------------------ A12 ---
def volumetric_rendering(rgb, density, s_vals, dirs, near, far, white_bkgd: bool = False,
                         raw_semantic=None, transform_idx: int = 0):
    """real_volumetric_rendering, mip.py:151-189.  Returns (comp_rgb or None,
    distance, acc, weights, semantic or None); distance is the UN-normalised
    sum w t_mid, nan->inf then clipped to [t_0, t_S]."""
    t = transform(s_vals, near, far, transform_idx)
    t_mids = 0.5 * (t[..., :-1] + t[..., 1:])
    t_dists = t[..., 1:] - t[..., :-1]
    delta = t_dists * torch.linalg.norm(dirs[..., None, :], dim=-1)
    dd = density[..., 0] * delta
    alpha = 1 - torch.exp(-dd)
    trans = torch.exp(-torch.cat([torch.zeros_like(dd[..., :1]), torch.cumsum(dd[..., :-1], dim=-1)], dim=-1))
    w = alpha * trans
    comp = None if rgb is None else (w[..., None] * rgb).sum(dim=-2)
    sem = None if raw_semantic is None else (w[..., None] * raw_semantic).sum(dim=-2)
    acc = w.sum(dim=-1)
    dist = (w * t_mids).sum(dim=-1)
    dist = torch.clip(torch.nan_to_num(dist, float("inf")), t[:, 0], t[:, -1])
    if white_bkgd and comp is not None:
        comp = comp + (1.0 - acc[..., None])
    return comp, dist, acc, w, sem


# --------------------------------------------------------------- A13, A14 --
def blur_weights(weights: np.ndarray, resample_padding: float = 0.01) -> np.ndarray:
    """mip.py:296-306: pad edges, max of neighbours, mean of adjacent maxes, + padding."""
    w = weights.astype(np.float32)
    wp = np.concatenate([w[..., :1], w, w[..., -1:]], -1)
    wmax = np.maximum(wp[..., :-1], wp[..., 1:])
    blur = (np.float32(0.5) * (wmax[..., :-1] + wmax[..., 1:])).astype(np.float32)
    return (blur + np.float32(resample_padding)).astype(np.float32)


def det_u(num_samples: int) -> torch.Tensor:
    """u for randomized=False: linspace(0, 1 - eps32, num) (math_ops.py:57)."""
    return torch.linspace(0.0, 1.0 - EPS32, num_samples)


def rand_u(num_samples: int, jitter: torch.Tensor) -> torch.Tensor:
    """u for randomized=True from explicit jitter ~ U[0, 1/num - eps32)
    (math_ops.py:50-54): min(arange * (1/num) + jitter, 1 - eps32)."""
    s = 1 / num_samples
    u = torch.arange(num_samples) * s + jitter
    return torch.minimum(u, torch.ones_like(u) - EPS32)


def sorted_piecewise_constant_pdf(bins, weights, u, sum_mode: str = "canonical"):
    """math_ops.py:19-76 with explicit ``u`` [N,num] (or [num], broadcast).

    ``sum_mode="torch"`` takes the row sum with ``torch.sum`` exactly as the
    reference does (SIMD-width dependent order; used only to show bit-equality
    with the golden vectors on the machine that captured them);
    ``"canonical"`` (default, what the HIP kernel implements) uses the fixed
    order of oracle/common.py.  The two differ by at most 1 ulp in the sum.

    The reference finds the interval with a [N, nb+1, num] mask and max/min; for
    sorted inputs that equals idx = #(u >= cdf) - 1 followed by a gather of
    (idx, idx+1).  Sums use the canonical order (oracle/common.py).
    Returns (samples fp32 [N,num], idx int32 [N,num]); ``idx`` is the
    bit-exact target for the HIP kernel.
    """
    b = bins.detach().cpu().numpy().astype(np.float32)
    w = weights.detach().cpu().numpy().astype(np.float32) if torch.is_tensor(weights) else weights.astype(np.float32)
    un = u.detach().cpu().numpy().astype(np.float32)
    if un.ndim == 1:
        un = np.broadcast_to(un, (b.shape[0], un.shape[0]))
    eps = np.float32(1e-5)
    if sum_mode == "torch":
        wsum = torch.sum(torch.from_numpy(w), dim=-1, keepdim=True).numpy()
    else:
        wsum = seq_sum_f32(w)
    padding = np.maximum(np.float32(0), eps - wsum).astype(np.float32)
    w = (w + padding / np.float32(w.shape[-1])).astype(np.float32)
    wsum = (wsum + padding).astype(np.float32)
    pdf = (w / wsum).astype(np.float32)
    cdf = np.minimum(np.float32(1), seq_cumsum_f32(np.ascontiguousarray(pdf[..., :-1])))
    cdf = np.concatenate([np.zeros_like(cdf[..., :1]), cdf, np.ones_like(cdf[..., :1])], -1)
    n = b.shape[0]
    idx = np.empty(un.shape, dtype=np.int32)
    for r in range(n):
        idx[r] = np.searchsorted(cdf[r], un[r], side="right") - 1
    idx1 = idx + 1  # u < 1 == cdf[-1] always, so idx1 <= nb
    b0, b1 = np.take_along_axis(b, idx, -1), np.take_along_axis(b, idx1, -1)
    c0, c1 = np.take_along_axis(cdf, idx, -1), np.take_along_axis(cdf, idx1, -1)
    with np.errstate(divide="ignore", invalid="ignore"):
        t = ((un - c0) / (c1 - c0)).astype(np.float32)
    t = np.clip(np.nan_to_num(t, nan=0.0, posinf=np.finfo(np.float32).max, neginf=np.finfo(np.float32).min), 0, 1)
    samples = (b0 + (t.astype(np.float32) * (b1 - b0)).astype(np.float32)).astype(np.float32)
    return torch.from_numpy(samples), torch.from_numpy(idx)


def warp_resample_s(s_vals, weights, u, resample_padding: float = 0.01):
    """warp_resample_along_rays up to the new fence posts (mip.py:294-316)."""
    w = blur_weights(weights.detach().cpu().numpy(), resample_padding)
    return sorted_piecewise_constant_pdf(s_vals, torch.from_numpy(w), u)


# ------------------------------------------------------------------ A15 ---
def mipnerf_forward(p: dict, rays: dict, n_samples: int, n_fine: int, s_rand=None, u=None,
                    white_bg: bool = False, ray_shape: str = "cone", transform_idx: int = 0,
                    max_deg_point: int = 16, deg_view: int = 4, resample_padding: float = 0.01,
                    density_noise0=None, density_noise1=None, return_aux: bool = False, s1_override=None, fn_idx: int = 1, viewc=0.0,
                    disable_integration: bool = False):
    """MipNerfModel.forward, models.py:72-187 (warp branch, use_viewdirs; the appearance
    embedding of encode_appearance when `p` holds "emb.weight").  ``u`` [N,n_fine] or [n_fine]; None = det_u(n_fine).
    Returns [[None, distance, acc, s_vals, weights], [rgb, distance, acc, semantic, s_vals, weights]]
    (the proposal_loss=True layout, models.py:180-185) and, if ``return_aux``,
    a dict with the resample indices and raw network outputs."""
    o, d, r = rays["origins"], rays["directions"], rays["radii"]
    near, far = rays["near"], rays["far"]
    n = o.shape[0]
    aux = {}
    # level 0: proposal
    s0 = warp_sample_s(n, n_samples, s_rand)
    m0, c0 = sample2enc(s0, o, d, r, near, far, ray_shape, transform_idx, fn_idx=fn_idx, viewc=viewc)
    if disable_integration:            # models.py:132-133: samples = (samples[0], torch.zeros_like(samples[1]))
        c0 = torch.zeros_like(c0)
    enc0 = integrated_pos_enc(m0, c0, 0, max_deg_point)
    raw_d0 = proposal_mlp(p, enc0)
    if density_noise0 is not None:
        raw_d0 = raw_d0 + density_noise0
    _, dens0 = activate(None, raw_d0)
    _, dist0, acc0, w0, _ = volumetric_rendering(None, dens0, s0, d, near, far, white_bg, None, transform_idx)
    # level 1: resample + NeRF MLP
    if u is None:
        u = det_u(n_fine)
    s1, idx = warp_resample_s(s0, w0, u, resample_padding)
    if s1_override is not None:
        # teacher forcing for conditioning-aware gradient checks (tests only): level 1 evaluated at GIVEN fence posts -- those a
        # reduced-precision run resampled -- so that both sides see identical sample positions (the posts carry no gradient)
        s1 = s1_override
    m1, c1 = sample2enc(s1, o, d, r, near, far, ray_shape, transform_idx, fn_idx=fn_idx, viewc=viewc)
    if disable_integration:
        c1 = torch.zeros_like(c1)
    enc1 = integrated_pos_enc(m1, c1, 0, max_deg_point)
    cond = pos_enc(rays["viewdirs"], 0, deg_view, True)
    if "emb.weight" in p:
        # encode_appearance (models.py:63-64,153-159): the per-image embedding row rays.app selects, appended to the view condition
        cond = torch.cat([cond, p["emb.weight"][rays["app"].long().reshape(-1)]], dim=1)
    raw_rgb, raw_d1, raw_sem = nerf_mlp(p, enc1, cond)
    if density_noise1 is not None:
        raw_d1 = raw_d1 + density_noise1
    rgb, dens1 = activate(raw_rgb, raw_d1)
    comp, dist1, acc1, w1, sem = volumetric_rendering(rgb, dens1, s1, d, near, far, white_bg, raw_sem, transform_idx)
    ret = [[None, dist0, acc0, s0, w0], [comp, dist1, acc1, sem, s1, w1]]
    if return_aux:
        aux.update(idx=idx, raw_density0=raw_d0, raw_rgb=raw_rgb, raw_density1=raw_d1, enc0=enc0, enc1=enc1, cond=cond)
        return ret, aux
    return ret


def mipnerf_param_shapes(hidden: int = 1024, rgb_layers: int = 3, prop_hidden: int = 256, feature_dim: int = 96,
                         cond_dim: int = 27, n_layers: int = 8, n_prop_layers: int = 4, cond_units: int = 128,
                         skip_layer: int = 4, semantic_class_num: int = 0, n_vocab: int = 0):
    """Ordered (name, shape) list of ``MipNerfModel.state_dict()`` for the
    shipped config (models.py:55-68, 232-255, 307-315); with ``semantic_class_num`` > 0 the
    semantic head Sequential(DenseBlock(hidden, hidden // 2), Linear(hidden // 2, C)) (models.py:258-260) is registered after
    the rgb head; with ``n_vocab`` > 0 (encode_appearance, models.py:57,63-64) cond_dim must be 27 + 48 and the embedding table
    ``emb.weight`` [n_vocab, 48] sits between the NeRF MLP and the proposal MLP."""
    out = []
    for i in range(n_layers):
        k = feature_dim if i == 0 else (hidden + feature_dim if ((i - 1) % skip_layer == 0 and (i - 1) > 0) else hidden)
        out += [(f"mlp.layers.{i}.layers.0.weight", (hidden, k)), (f"mlp.layers.{i}.layers.0.bias", (hidden,))]
    out += [("mlp.density_layer.weight", (1, hidden)), ("mlp.density_layer.bias", (1,)),
            ("mlp.bottleneck_layer.layers.0.weight", (hidden, hidden)), ("mlp.bottleneck_layer.layers.0.bias", (hidden,))]
    for j in range(rgb_layers):
        k = hidden + cond_dim if j == 0 else cond_units
        out += [(f"mlp.cond_layers.{j}.layers.0.weight", (cond_units, k)), (f"mlp.cond_layers.{j}.layers.0.bias", (cond_units,))]
    out += [("mlp.rgb_layer.weight", (3, cond_units)), ("mlp.rgb_layer.bias", (3,))]
    if semantic_class_num > 0:
        out += [("mlp.semantic_layer.0.layers.0.weight", (hidden // 2, hidden)), ("mlp.semantic_layer.0.layers.0.bias", (hidden // 2,)),
                ("mlp.semantic_layer.1.weight", (semantic_class_num, hidden // 2)), ("mlp.semantic_layer.1.bias", (semantic_class_num,))]
    if n_vocab > 0:
        out += [("emb.weight", (n_vocab, 48))]
    for i in range(n_prop_layers):
        k = feature_dim if i == 0 else prop_hidden
        out += [(f"proposal.layers.{i}.layers.0.weight", (prop_hidden, k)), (f"proposal.layers.{i}.layers.0.bias", (prop_hidden,))]
    out += [("proposal.density_layer.weight", (1, prop_hidden)), ("proposal.density_layer.bias", (1,))]
    return out


# --------------------------------------------------------------------------
# A16  render_image -- models.py:328-360 (eval.py:146)
# --------------------------------------------------------------------------
def render_image(p: dict, rays: dict, H: int, W: int, chunk: int, n_samples: int, n_fine: int, **kw):
    """Chunked full-frame inference: the [H,W,.] ray grid is flattened, rendered in chunks of `chunk` rays (the last one ragged)
    and the last level's (rgb, distance, acc) are concatenated and reshaped to the image (models.py:328-360; the reference's
    reflect-padding of a chunk to a multiple of the device count is the identity for one device).
    -> (rgb [H,W,3], distance [H,W], acc [H,W])"""
    n = H * W
    flat = {k: v.reshape(n, -1) for k, v in rays.items()}
    cols = [[], [], []]
    for i in range(0, n, chunk):
        out = mipnerf_forward(p, {k: v[i:i + chunk] for k, v in flat.items()}, n_samples, n_fine, **kw)[-1]
        for c, o in zip(cols, out[:3]):
            c.append(o)
    rgb, dist, acc = (torch.cat(c, 0) for c in cols)
    return rgb.reshape(H, W, -1), dist.reshape(H, W), acc.reshape(H, W)
