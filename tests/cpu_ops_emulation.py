"""TEST INFRASTRUCTURE: a CPU emulation of the `snerf_amd.ops` entry points (same signatures, in-place
semantics) built from the oracle, so that the HOST logic above the C-ABI -- weight packing, buffer/column
plumbing, the hand-written backward chains of snerf_amd.mlp, the autograd glue and the trainer -- can be
exercised by `-m "not gpu"` tests in a container without a GPU.  Never imported by the product."""
import contextlib

import numpy as np

import torch

from oracle import classic as oc
from oracle import mip as om


_BITS = {}


def _split_unpack(T, K):
    """split-bf16 activation [M, >= 2 K] (hi / lo interleaved per 64 logical columns) -> (hi [M, K], lo [M, K]) as fp32"""
    t = T[:, :2 * K].float().reshape(T.shape[0], K // 64, 2, 64)
    return t[:, :, 0].reshape(T.shape[0], K), t[:, :, 1].reshape(T.shape[0], K)


def _split_pack(y, Y, n):
    """fp32 [M, n] -> the interleaved split-bf16 layout in Y[:, :2 * roundup(n, 64)] (columns n .. are left alone where a group is partial)"""
    hi = y.to(torch.bfloat16)
    lo = (y - hi.float()).to(torch.bfloat16)
    for j in range((n + 63) // 64):
        w = min(64, n - 64 * j)
        Y[:, 128 * j:128 * j + w] = hi[:, 64 * j:64 * j + w]
        Y[:, 128 * j + 64:128 * j + 64 + w] = lo[:, 64 * j:64 * j + w]


_F8 = torch.float8_e4m3fn
_S8_ACT, _S8_W = (2.0 ** 13, 2.0 ** 2), (2.0 ** 20, 2.0 ** 9)       # (residual scale, value scale) of activations / weights; 13 + 9 = 2 + 20 = 22


def _e4m3(x, mul):
    return (x * mul).clamp(-448.0, 448.0).to(_F8)


def _s8_unpack(T, K, weight=False):
    """fp16 + fp8 split rows [M, >= 2 K] (fp16-typed; per 64 logical columns [fp16 x 64 | 128 e4m3 bytes]) -> (hi, e4m3 residual, e4m3 value) as fp32,
    the e4m3 parts still carrying their power-of-two scales"""
    M = T.shape[0]
    t = T[:, :2 * K].contiguous().view(M, K // 64, 128)
    hi = t[:, :, :64].float().reshape(M, K)
    by = t[:, :, 64:].contiguous().view(torch.uint8).reshape(M, K // 64, 128)
    first, second = by[:, :, :64].contiguous().view(_F8).float().reshape(M, K), by[:, :, 64:].contiguous().view(_F8).float().reshape(M, K)
    return (hi, second, first) if weight else (hi, first, second)


def _s8_pack(y, Y, n, weight=False):
    mr, mx = _S8_W if weight else _S8_ACT
    hi = y.to(torch.float16)
    r8, x8 = _e4m3(y - hi.float(), mr).view(torch.uint8), _e4m3(y, mx).view(torch.uint8)
    yb = Y.view(torch.uint8).view(Y.shape[0], -1)              # [M, 2 x physical columns] bytes
    for j in range((n + 63) // 64):
        w = min(64, n - 64 * j)
        Y[:, 128 * j:128 * j + w] = hi[:, 64 * j:64 * j + w]
        b0 = 256 * j + 128
        first, second = (x8, r8) if weight else (r8, x8)
        yb[:, b0:b0 + w] = first[:, 64 * j:64 * j + w]
        yb[:, b0 + 64:b0 + 64 + w] = second[:, 64 * j:64 * j + w]


def split8_cast(src, C, dst, Cpad, weight=False):
    y = torch.zeros(src.shape[0], Cpad)
    y[:, :C] = src[:, :C]
    _s8_pack(y, dst, Cpad, weight)


def split_cast(src, C, dst, Cpad):
    y = torch.zeros(src.shape[0], Cpad)
    y[:, :C] = src[:, :C]
    _split_pack(y, dst, Cpad)


def linear_fwd(A, W, bias, Y, K, n_store, act, dt, out_f32=False, aux=None, colsum=None, variant=0, deterministic=False, aux_split=False):
    assert W.shape[0] % 128 == 0 and K % (32 if dt == 0 else 64) == 0
    if aux_split and act == 2:                                   # plain bf16 / fp16 data gradient, mask = the hi half of a split activation
        assert dt in (1, 2)
        aux = _split_unpack(aux, 64 * ((n_store + 63) // 64))[0]
    if dt == 5:                                                 # fp16 + fp8: hi.hi on fp16 + e4m3 corrections r.w + x.(w - hi) (csrc/gemm.hip, GemmNT::split == 2)
        assert act in (0, 1, 3) and colsum is None
        ah, ar, ax = _s8_unpack(A, K)
        wh, wr, wx = _s8_unpack(W, K, weight=True)
        y = ah @ wh.t() + (ar @ wx.t() + ax @ wr.t()) * 2.0 ** -22
        if bias is not None:
            y = y + bias
        y = y[:, :n_store]
        if act in (1, 3):
            y = torch.relu(y)
        if act == 3:
            _BITS[aux.data_ptr()] = y.to(torch.float16).float() > 0
        if out_f32:
            Y[:, :n_store] = y
        else:
            _s8_pack(y, Y, n_store)
        return
    if dt == 4:                                                 # split-bf16: hi.hi + lo.hi + hi.lo, fp32 accumulation (csrc/gemm.hip)
        ah, al = _split_unpack(A, K)
        w = W[:, :3 * K].float().reshape(W.shape[0], K // 64, 3, 64)
        assert torch.equal(w[:, :, 0], w[:, :, 1]), "weights of the split mode are [hi | hi | lo] per 64 columns"
        wh, wl = w[:, :, 0].reshape(W.shape[0], K), w[:, :, 2].reshape(W.shape[0], K)
        y = ah @ wh.t() + al @ wh.t() + ah @ wl.t()
        if bias is not None:
            y = y + bias
        y = y[:, :n_store]
        if act in (1, 3):
            y = torch.relu(y)
        if act == 2:
            y = y * (_split_unpack(aux, 64 * ((n_store + 63) // 64))[0][:, :n_store] > 0)
        if act == 3:
            _BITS[aux.data_ptr()] = y.to(torch.bfloat16).float() > 0
        if act == 4:
            y = y * _BITS[aux.data_ptr()][:, :n_store]
        if out_f32:
            Y[:, :n_store] = y
        else:
            _split_pack(y, Y, n_store)
            if colsum is not None:
                hi = y.to(torch.bfloat16).float()
                colsum[:n_store] += (hi + (y - hi).to(torch.bfloat16).float()).sum(0)
        return
    y = A[:, :K].float() @ W[:, :K].float().t()
    if bias is not None:
        y = y + bias
    y = y[:, :n_store]
    if act == 1:
        y = torch.relu(y)
    elif act == 2:
        y = y * (aux[:, :n_store].float() > 0)
    elif act == 3:                                              # ReLU + bit mask (opaque buffer: keep the mask beside it)
        y = torch.relu(y)
        _BITS[aux.data_ptr()] = y.to(Y.dtype).float() > 0
    elif act == 4:
        y = y * _BITS[aux.data_ptr()][:, :n_store]
    Y[:, :n_store] = y.to(Y.dtype)
    if colsum is not None:
        colsum[:n_store] += y.sum(0)


def linear_wgrad(dZ, X, dW, n_valid, k_valid, dt, variant=0, deterministic=False, x_split_hi=False):
    if x_split_hi:                                              # plain bf16 / fp16 dZ x the hi half of a split activation
        assert dt in (1, 2) and X.shape[1] % 128 == 0
        dW[:n_valid, :k_valid] += (dZ.float().t() @ _split_unpack(X, X.shape[1] // 2)[0])[:n_valid, :k_valid]
        return
    if dt == 4:                                                 # the kernels multiply the physical matrices: all four hi / lo combinations
        zh, zl = _split_unpack(dZ, dZ.shape[1] // 2)
        xh, xl = _split_unpack(X, X.shape[1] // 2)
        dW[:n_valid, :k_valid] += ((zh + zl).t() @ (xh + xl))[:n_valid, :k_valid]
        return
    dW[:n_valid, :k_valid] += (dZ.float().t() @ X.float())[:n_valid, :k_valid]


def classic_embed(pts, viewdirs, S, L, Lv, dst1, dst2, w_pts, dstv, w_views, dt):
    e = oc.embed(pts, L)
    pad = torch.zeros(pts.shape[0], w_pts - e.shape[1])
    v = torch.cat([e, pad], -1)
    if dt in (4, 5):                                            # the split layouts: fp32 images, then the operand layout per destination
        cast_pad(v, w_pts, dst1, w_pts, dt)
        if dst2 is not None:
            cast_pad(v, w_pts, dst2, w_pts, dt)
        if viewdirs is not None:
            ev = oc.embed(viewdirs[:, None].expand(-1, S, -1).reshape(-1, 3), Lv)
            cast_pad(torch.cat([ev, torch.zeros(ev.shape[0], w_views - ev.shape[1])], -1), w_views, dstv, w_views, dt)
        return
    dst1[:, :w_pts] = v.to(dst1.dtype)
    if dst2 is not None:
        dst2[:, :w_pts] = v.to(dst2.dtype)
    if viewdirs is not None:
        ev = oc.embed(viewdirs[:, None].expand(-1, S, -1).reshape(-1, 3), Lv)
        dstv[:, :w_views] = torch.cat([ev, torch.zeros(ev.shape[0], w_views - ev.shape[1])], -1).to(dstv.dtype)


def mip_encode(s_vals, origins, directions, radii, near, far, cone, transform_idx, max_deg, dst1, dst2, width, dt,
               means_out=None, covs_out=None, sample_id=None, warp=None):
    assert sample_id is None, "compacted rows are an inference-only GPU mode"
    kw = {} if warp is None else dict(fn_idx=0, viewc=torch.tensor(warp[0]))
    assert warp is None or float(warp[1]) == float(far.max())
    fm, fc = om.sample2enc(s_vals, origins, directions, radii[:, None], near[:, None], far[:, None], "cone" if int(cone) & 1 else "cylinder", transform_idx, **kw)
    if int(cone) & 2:                      # --disable_integration
        fc = torch.zeros_like(fc)
    enc = om.integrated_pos_enc(fm, fc, 0, max_deg).reshape(-1, 6 * max_deg)
    v = torch.cat([enc, torch.zeros(enc.shape[0], width - enc.shape[1])], -1)
    if dt == 4:
        return split_cast(v, width, dst1, width)
    if dt == 5:
        return split8_cast(v, width, dst1, width)
    dst1[:, :width] = v.to(dst1.dtype)
    if dst2 is not None:
        dst2[:, :width] = v.to(dst2.dtype)


def mip_viewenc(viewdirs, S, deg, dst, width, dt, sample_id=None):
    assert sample_id is None
    e = om.pos_enc(viewdirs, 0, deg, True)[:, None].expand(-1, S, -1).reshape(-1, 3 + 6 * deg)
    if dt in (4, 5):
        return (split_cast if dt == 4 else split8_cast)(torch.cat([e, torch.zeros(e.shape[0], width - e.shape[1])], -1), width, dst, width)
    dst[:, :width] = torch.cat([e, torch.zeros(e.shape[0], width - e.shape[1])], -1).to(dst.dtype)


def app_embed(emb, app, S, dst, dt, sample_id=None, check=True):
    assert sample_id is None
    rows = emb[app.reshape(-1).long().clamp(0, emb.shape[0] - 1)]
    dst[:, :emb.shape[1]] = rows[:, None].expand(-1, S, -1).reshape(-1, emb.shape[1]).to(dst.dtype)


def app_embed_bwd(dV, app, S, g_emb, deterministic=False):
    idx = app.reshape(-1).long().clamp(0, g_emb.shape[0] - 1)
    g_emb.index_add_(0, idx, dV[:, :g_emb.shape[1]].reshape(idx.numel(), S, -1).sum(1))


def classic_sample_pdf(bins, weights, u, mid_mode, want_inds=False, want_std=False):
    n = bins.shape[0]
    uu = u if u.dim() == 2 else u.expand(n, u.shape[0])
    if mid_mode:
        s, i = oc.sample_pdf(0.5 * (bins[:, 1:] + bins[:, :-1]), weights[:, 1:-1], uu)
    else:
        s, i = oc.sample_pdf(bins, weights, uu)
    return s, (i.int() if want_inds else None), (torch.std(s, dim=-1, unbiased=False) if want_std else None)


def classic_points(rays, z_vals):
    return rays[:, None, 0:3] + rays[:, None, 3:6] * z_vals[..., None]


def classic_merge_sort(a, b):
    return torch.sort(torch.cat([a, b], -1), -1)[0]


def mip_resample(s_vals, weights, u, resample_padding=0.01, want_idx=False):
    s, i = om.warp_resample_s(s_vals, weights, u, resample_padding)
    return s, (i if want_idx else None)


def stratified(base, rnd, near, far, n, mode, lindisp=False):
    if mode == 1:
        return om.warp_sample_s(n, base.shape[0] - 1, rnd).contiguous()
    return oc.stratified_z(near[:, None], far[:, None], base.shape[0], lindisp, rnd).contiguous()


def jitter_u(jit, s):
    eps = float(torch.finfo(torch.float32).eps)
    jit.copy_(torch.minimum(torch.arange(jit.shape[1], device=jit.device) * s + jit, torch.ones_like(jit) - eps))   # math_ops.py:50-54
    return jit


def mip_composite_fwd(raw_rgb, raw_density, noise, s_vals, dirs, near, far, transform_idx, white, rgb_padding, density_bias, row_index=None):
    assert row_index is None
    n, P = s_vals.shape
    rd = raw_density.reshape(n, P - 1, 1)
    if noise is not None:
        rd = rd + noise.reshape(n, P - 1, 1)
    rgb, den = om.activate(None if raw_rgb is None else raw_rgb.reshape(n, P - 1, 3), rd, rgb_padding, density_bias)
    c, d, a, w, _ = om.volumetric_rendering(rgb, den, s_vals, dirs, near[:, None], far[:, None], white, None, transform_idx)
    return c, d, a, w


def mip_composite_bwd(raw_rgb, raw_density, noise, s_vals, dirs, near, far, transform_idx, white, rgb_padding, density_bias,
                      weights, distance, g_rgb, g_dist, g_acc, g_w, d_raw_rgb, d_raw_density, g_dirs=None):
    with torch.enable_grad():
        rr = None if raw_rgb is None else raw_rgb.detach().clone().requires_grad_(True)
        rd = raw_density.detach().clone().requires_grad_(True)
        dd = dirs.detach().clone().requires_grad_(True)
        c, d, a, w = mip_composite_fwd(rr, rd, noise, s_vals, dd, near, far, transform_idx, white, rgb_padding, density_bias)
        loss = 0
        for o, g in ((c, g_rgb), (d, g_dist), (a, g_acc), (w, g_w)):
            if g is not None and o is not None:
                loss = loss + (o * g).sum()
        loss.backward()
    d_raw_density.copy_(rd.grad)
    if rr is not None:
        d_raw_rgb.copy_(rr.grad)
    if g_dirs is not None:
        g_dirs.copy_(dd.grad if dd.grad is not None else torch.zeros_like(dd))


def mip_encode_bwd(s_vals, origins, directions, radii, near, far, cone, transform_idx, max_deg, dE, warp=None):
    kw = {} if warp is None else dict(fn_idx=0, viewc=torch.tensor(warp[0]))
    with torch.enable_grad():
        o, d = origins.detach().clone().requires_grad_(True), directions.detach().clone().requires_grad_(True)
        fm, fc = om.sample2enc(s_vals, o, d, radii[:, None], near[:, None], far[:, None], "cone" if int(cone) & 1 else "cylinder", transform_idx, **kw)
        if int(cone) & 2:
            fc = torch.zeros_like(fc)
        enc = om.integrated_pos_enc(fm, fc, 0, max_deg).reshape(-1, 6 * max_deg)
        (enc * dE[:, :6 * max_deg]).sum().backward()
    return o.grad, d.grad


def mip_viewenc_bwd(viewdirs, S, deg, dV):
    with torch.enable_grad():
        v = viewdirs.detach().clone().requires_grad_(True)
        e = om.pos_enc(v, 0, deg, True)[:, None].expand(-1, S, -1).reshape(-1, 3 + 6 * deg)
        (e * dV[:, :3 + 6 * deg]).sum().backward()
    return v.grad


def classic_composite_fwd(raw, noise, z_vals, rays_d, white):
    n, S = z_vals.shape
    return oc.raw2outputs(raw.reshape(n, S, -1), z_vals, rays_d, noise, white)


def classic_composite_bwd(raw, noise, z_vals, rays_d, white, weights, acc, depth, g_rgb, g_disp, g_acc, g_depth, g_w, d_raw):
    with torch.enable_grad():
        r = raw.detach().clone().requires_grad_(True)
        outs = classic_composite_fwd(r, noise, z_vals, rays_d, white)
        loss = 0
        for o, g in zip(outs, (g_rgb, g_disp, g_acc, g_w, g_depth)):
            if g is not None:
                loss = loss + (o * g).sum()
        loss.backward()
    d_raw.copy_(r.grad)


def adam_step(p, g, m, v, lr, b1, b2, eps, step, grad_scale=1.0, zero_grad=True, nonfinite="zero", grad_max_val=0.0, clip_coef=None,
              step_dev=None, lr_dev=None, dropped=None):
    if dropped is not None:
        dropped += int((~torch.isfinite(g)).sum())
    if step_dev is not None:
        step_dev += 1
        step = int(step_dev)
    if lr_dev is not None:
        lr = float(lr_dev)
    gg = g * grad_scale * (1.0 if clip_coef is None else float(clip_coef[0]))
    if grad_max_val > 0:
        gg = torch.clamp(gg, -grad_max_val, grad_max_val)
    if nonfinite == "zero":
        gg = torch.where(torch.isfinite(gg), gg, torch.zeros_like(gg))
    elif nonfinite == "nan_to_num":
        gg = torch.nan_to_num(gg)
    m.mul_(b1).add_(gg, alpha=1 - b1)
    v.mul_(b2).addcmul_(gg, gg, value=1 - b2)
    p.sub_((lr / (1 - b1 ** step)) * m / (v.sqrt() / (1 - b2 ** step) ** 0.5 + eps))
    if zero_grad:
        g.zero_()


def grad_clip_coef(g, grad_scale, max_norm):
    norm = abs(grad_scale) * float(g.double().pow(2).sum().sqrt())
    return torch.tensor([min(max_norm / (norm + 1e-6), 1.0), norm], dtype=torch.float32)


def nonfinite_flag(g, flag):
    if not bool(torch.isfinite(g).all()):
        flag |= 1


def colsum_f32(x, C, out, deterministic=False):
    out[:C] += x[:, :C].sum(0)


def cast_pad(src, C, dst, Cpad, dt):
    if dt == 4:
        return split_cast(src, C, dst, Cpad)
    if dt == 5:
        return split8_cast(src, C, dst, Cpad)
    dst[:, :Cpad] = 0
    dst[:, :C] = src[:, :C].to(dst.dtype)


# ---------------------------------------------------------------- zipnerf path ----
def zip_resample(sdist, weights, u, n, near, far, dilation, dilate, anneal, resample_padding=0.0, lam=-1.5, dom=(0.0, 1.0)):
    from oracle import zip as oz
    t, w = sdist, weights
    if dilate:
        t, w = oz.max_dilate_weights(t, w, dilation, dom)
        t, w = t[..., 1:-1], w[..., 1:-1]
    logits = oz.resample_logits(t, w, anneal, resample_padding)
    sd, _ = oz.sample_intervals(t, logits, u, dom)
    return sd.contiguous(), oz.s_to_t(sd, near[:, None], far[:, None], lam).contiguous()


def _zip_points(tdist, origins, directions, radii, base_x, base_y, deg_jitter, n, m, std_scale):
    from oracle import zip as oz
    means, stds = oz.cast_rays(tdist, origins, directions, radii[:, None], base_x, base_y, deg_jitter, n, m, std_scale)
    pre = means.shape[:-1]
    z, s = oz.contract_mean_std(means.reshape(-1, 3), stds.reshape(-1))
    return (z.reshape(*pre, 3) / 2 + 1) / 2, s.reshape(*pre) / 2


def zip_encode_fwd(tdist, origins, directions, radii, base_x, base_y, deg_jitter, table, offsets, grid_sizes, feat, L, C, n, m, Sl, H, std_scale,
                   levels_per_thread=0):
    from oracle import grid as og
    x01, s = _zip_points(tdist, origins, directions, radii, base_x, base_y, deg_jitter, n, m, std_scale)
    out = og.grid_encode_forward(x01.reshape(-1, 3).numpy().astype("float32"), table.float().numpy(), offsets.numpy(), Sl, H, 0, False, 0)
    f = torch.from_numpy(out).permute(1, 0, 2).reshape(list(x01.shape[:-1]) + [L, C])
    w = torch.erf(1 / torch.sqrt(8 * s[..., None] ** 2 * grid_sizes.float() ** 2))
    f = (f * w[..., None]).mean(dim=-3).flatten(-2, -1)
    feat[:, :L * C] = f.reshape(-1, L * C).to(feat.dtype)


def zip_encode_fwd_count(tdist, origins, directions, radii, base_x, base_y, deg_jitter, table, offsets, grid_sizes, feat, L, C, n, m, Sl, H, std_scale,
                         ksplit, level_rows):
    assert len(ksplit) == L and all(k >= 1 for k in ksplit) and len(level_rows) == L and n <= 8 and C in (1, 4)
    zip_encode_fwd(tdist, origins, directions, radii, base_x, base_y, deg_jitter, table, offsets, grid_sizes, feat, L, C, n, m, Sl, H, std_scale, 1)
    return ("precounted", tuple(ksplit))                # (what the emulated backward checks it was handed back)


def _zpm_rb(t, rnd):
    """rnd: rounding mode (0 / False none, 1 / True bf16, 2 fp16)"""
    return t.half().float() if int(rnd) == 2 else (t.bfloat16().float() if rnd else t)


def _zpm_hidden(F, L, w1, b1, rnd):
    f = F[:, :L].float()
    return _zpm_rb(torch.relu(f @ _zpm_rb(w1.reshape(-1, L), rnd).T + b1), rnd), f


def zip_prop_mlp_fwd(F, L, w1, b1, w2, b2, round_bf16):
    H, _ = _zpm_hidden(F, L, w1, b1, round_bf16)
    return (H @ _zpm_rb(w2.reshape(-1), round_bf16) + b2).reshape(-1, 1)


def zip_prop_mlp_bwd(F, d_raw, L, w1, b1, w2, b2, round_bf16, g_w1, g_b1, g_w2, g_b2):
    rnd = round_bf16
    H, f = _zpm_hidden(F, L, w1, b1, rnd)
    g = _zpm_rb(d_raw.reshape(-1, 1).float(), rnd)
    dh = torch.where(H > 0, _zpm_rb(g * _zpm_rb(w2.reshape(1, -1), rnd), rnd), torch.zeros_like(H))
    dF = torch.zeros(F.shape, dtype=F.dtype)
    dF[:, :L] = (dh @ _zpm_rb(w1.reshape(-1, L), rnd)).to(F.dtype)
    g_w1 += (dh.T @ f).reshape(g_w1.shape)
    g_b1 += dh.sum(0)
    g_w2 += (g * H).sum(0).reshape(g_w2.shape)
    g_b2 += d_raw.sum()
    return dF


def zip_encode_bwd(tdist, origins, directions, radii, base_x, base_y, deg_jitter, offsets, grid_sizes, grad_feat, grad_table, L, C, n, m, Sl, H,
                   std_scale, lds_levels=0, lds_cells=0, lds_slabs=0, grad_table_bf16=None):
    from oracle import grid as og
    x01, s = _zip_points(tdist, origins, directions, radii, base_x, base_y, deg_jitter, n, m, std_scale)
    w = torch.erf(1 / torch.sqrt(8 * s[..., None] ** 2 * grid_sizes.float() ** 2))             # [R,S,n,L]
    g = grad_feat[:, :L * C].float().reshape(list(x01.shape[:-2]) + [1, L, C]) * w[..., None] / n  # [R,S,n,L,C]
    G = g.reshape(-1, L, C).permute(1, 0, 2).contiguous().numpy()
    gE, _ = og.grid_encode_backward(G, x01.reshape(-1, 3).numpy().astype("float32"), offsets.numpy(), grad_table.shape[0], Sl, H, 0, False, 0)
    grad_table += torch.from_numpy(gE)


def zip_encode_bwd_binned(tdist, origins, directions, radii, base_x, base_y, deg_jitter, offsets, grid_sizes, grad_feat, grad_table, L, C, n, m, Sl, H,
                          std_scale, ksplit, g64_rows, level_rows, precounted=None, half_records=False):
    assert len(ksplit) == L and all(k >= 1 for k in ksplit) and len(level_rows) == L
    assert precounted is None or precounted == ("precounted", tuple(ksplit)), "the forward's counts belong to another bin plan"
    zip_encode_bwd(tdist, origins, directions, radii, base_x, base_y, deg_jitter, offsets, grid_sizes, grad_feat, grad_table, L, C, n, m, Sl, H, std_scale)


def colsum_wide_f32(x, C, out, deterministic=False):
    out[:C] += x[:, :C].sum(0)


def zip_glo_modulate(X, SS, S, out):
    B = SS.shape[1] // 2
    ss = SS.repeat_interleave(S, 0)
    out[:, :B] = (X[:, :B].float() * torch.exp(ss[:, :B]) + ss[:, B:]).to(out.dtype)


def zip_glo_modulate_bwd(dXm, X, SS, d_head, S, dX):
    B = SS.shape[1] // 2
    R = SS.shape[0]
    e = torch.exp(SS[:, :B]).repeat_interleave(S, 0)
    g = dXm[:, :B].float()
    dx = g * e
    if d_head is not None:
        dx[:, :d_head.shape[1]] += d_head
    dX[:, :B] = dx.to(dX.dtype)
    dSS = torch.cat([((g * X[:, :B].float()) * e).view(R, S, B).sum(1), g.view(R, S, B).sum(1)], 1)
    return dSS, dX[:, :B].float().view(R, S, B).sum(1)


def zip_composite_fwd(raw_rgb, raw_density, tdist, dirs, opaque, bg, rgb_padding, density_bias):
    from oracle import zip as oz
    R, P = tdist.shape
    dens = torch.nn.functional.softplus(raw_density.reshape(R, P - 1) + density_bias)
    w = oz.compute_alpha_weights(dens, tdist, dirs, opaque)
    rgbs = torch.zeros(R, P - 1, 3) if raw_rgb is None else torch.sigmoid(raw_rgb.reshape(R, P - 1, 3)) * (1 + 2 * rgb_padding) - rgb_padding
    r = oz.volumetric_rendering(rgbs, w, tdist, bg)
    return r["rgb"], r["depth"], r["acc"], w


class GridFn(torch.autograd.Function):
    """Differentiable wrapper of the oracle's hash-grid encoder (oracle/grid.py): features [B, L, C] of positions x01 [B,3] in [0,1];
    backward = the reference's kernel_grid_backward (table) and kernel_input_backward via dy_dx (positions)."""

    @staticmethod
    def forward(ctx, x01, table, offsets, Sl, H):
        from oracle import grid as og
        x = x01.detach().numpy().astype("float32")
        out, dy_dx = og.grid_encode_forward(x, table.detach().float().numpy(), offsets, Sl, H, 0, False, 0, want_dy_dx=True)
        ctx.save = (x, offsets, Sl, H, dy_dx, table.shape)
        return torch.from_numpy(out).permute(1, 0, 2).contiguous()

    @staticmethod
    def backward(ctx, g):
        from oracle import grid as og
        x, offsets, Sl, H, dy_dx, tshape = ctx.save
        gE, g_in = og.grid_encode_backward(g.permute(1, 0, 2).contiguous().numpy(), x, offsets, tshape[0], Sl, H, 0, False, 0, dy_dx=dy_dx)
        return torch.from_numpy(g_in), torch.from_numpy(gE), None, None, None


def zip_encode_ray_bwd(tdist, origins, directions, radii, base_x, base_y, deg_jitter, table, offsets, grid_sizes, grad_feat, L, C, n, m, Sl, H,
                       std_scale, g_o, g_d, g_bx, g_by):
    with torch.enable_grad():
        leaves = [t.detach().clone().requires_grad_(True) for t in (origins, directions, base_x, base_y)]
        x01, s = _zip_points(tdist, leaves[0], leaves[1], radii, leaves[2], leaves[3], deg_jitter, n, m, std_scale)
        f = GridFn.apply(x01.reshape(-1, 3), table.float(), offsets.numpy(), Sl, H).reshape(list(x01.shape[:-1]) + [L, C])
        w = torch.erf(1 / torch.sqrt(8 * s[..., None] ** 2 * grid_sizes.float() ** 2))
        feat = (f * w[..., None]).mean(dim=-3).flatten(-2, -1).reshape(-1, L * C)
        (feat * grad_feat[:, :L * C].float()).sum().backward()
    for dst, leaf in zip((g_o, g_d, g_bx, g_by), leaves):
        dst += leaf.grad


def zip_composite_bwd(raw_rgb, raw_density, tdist, dirs, opaque, bg, rgb_padding, density_bias, weights, acc, depth, g_rgb, g_depth, g_acc, g_w,
                      d_raw_rgb, d_raw_density, g_dirs=None):
    with torch.enable_grad():
        rr = None if raw_rgb is None else raw_rgb.detach().clone().requires_grad_(True)
        rd = raw_density.detach().clone().requires_grad_(True)
        if g_dirs is not None:
            dirs = dirs.detach().clone().requires_grad_(True)
        outs = zip_composite_fwd(rr, rd, tdist, dirs, opaque, bg, rgb_padding, density_bias)
        loss = 0
        for o, g in zip(outs, (g_rgb, g_depth, g_acc, g_w)):
            if g is not None:
                loss = loss + (o * g).sum()
        loss.backward()
    d_raw_density.copy_(rd.grad)
    if rr is not None:
        d_raw_rgb.copy_(rr.grad if rr.grad is not None else torch.zeros_like(rr))
    if g_dirs is not None:
        g_dirs.copy_(dirs.grad if dirs.grad is not None else torch.zeros_like(dirs))


def pinhole_rays(coords, first_pixel, n, W, H, pose, cx, cy, fx, fy, training, near, far, device):
    from oracle import callers as oc
    if coords is None:
        pix = torch.arange(first_pixel, first_pixel + n)
        coords = torch.stack([pix // W, pix % W], -1)
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], dtype=np.float32)
    r = oc.pinhole_rays(coords.cpu().long(), np.asarray(pose, dtype=np.float32), K, H, near, far, training=bool(training), W=W)
    return tuple(r[k].to(device) for k in ("origins", "directions", "viewdirs", "radii", "near", "far"))


def mip_loss_tail(rgb, tgt, dist1, dist0, tdepth, conf, s_f, w_f, s_c, w_c, disparity, depth_lambda, coarse_mult, prop_lambda):
    from oracle import callers as oc
    dev = rgb.device
    rg = rgb.detach().cpu().clone().requires_grad_(True)
    out = torch.zeros(5)
    loss = oc.rgb_loss(rg, tgt.cpu())
    out[1] = loss.detach()
    leaves, g1 = [rg], None
    if tdepth is not None:
        d1, d0 = dist1.detach().cpu().clone().requires_grad_(True), dist0.detach().cpu().clone().requires_grad_(True)
        td = tdepth.cpu()
        out[0] = float((td != 0).sum())
        if out[0] > 0:
            ld = oc.depth_loss(d1, d0, td, None if conf is None else conf.cpu(), coarse_mult, bool(disparity)) * depth_lambda
            out[2] = ld.detach()
            loss = loss + ld
        leaves += [d1, d0]
    if s_c is not None:
        wc = w_c.detach().cpu().clone().requires_grad_(True)
        lp = oc.proposal_loss(s_f.cpu(), w_f.cpu(), s_c.cpu(), wc, prop_lambda)
        out[3] = lp.detach()
        loss = loss + lp
        leaves.append(wc)
    out[4] = loss.detach()
    gs = torch.autograd.grad(loss, leaves, allow_unused=True)
    gs = [torch.zeros_like(l) if g is None else g for g, l in zip(gs, leaves)]
    it = iter(gs)
    g_rgb = next(it).to(dev)
    gd1 = gd0 = gw = None
    if tdepth is not None:
        gd1, gd0 = next(it).to(dev), next(it).to(dev)
    if s_c is not None:
        gw = next(it).to(dev)
    return out.to(dev), g_rgb, gd1, gd0, gw


def semantic_composite_fwd(weights, logits, C, softmax, row_index=None):
    if row_index is not None:
        ri = row_index.reshape(-1).long()
        logits = torch.where((ri >= 0)[:, None], logits[ri.clamp(min=0), :C].float(), torch.zeros(1, C))
        weights = weights * (ri >= 0).reshape(weights.shape)
    v = logits[:, :C].float().reshape(weights.shape[0], weights.shape[1], C)
    if softmax:
        v = torch.softmax(v, -1)
    return (weights[..., None] * v).sum(-2)


def semantic_composite_bwd(weights, logits, g_sem, C, softmax, d_logits, want_g_w=False):
    R, S = weights.shape
    v = logits[:, :C].float().reshape(R, S, C)
    if softmax:
        p = torch.softmax(v, -1)
        dot = (p * g_sem[:, None, :]).sum(-1, keepdim=True)
        d_logits[:, :C] = (weights[..., None] * p * (g_sem[:, None, :] - dot)).reshape(R * S, C)
        return torch.zeros(R, S) if want_g_w else None
    d_logits[:, :C] = (weights[..., None] * g_sem[:, None, :]).expand(R, S, C).reshape(R * S, C)
    return (v * g_sem[:, None, :]).sum(-1) if want_g_w else None


def zip_pixels_to_rays(pix_x, pix_y, cam_idx, pixtocams, camtoworlds, want_imageplane=False):
    from oracle import callers as oc
    ci = np.zeros(pix_x.shape[0], np.int64) if cam_idx is None else cam_idx.cpu().numpy().astype(np.int64)
    r = oc.zip_pixels_to_rays(pix_x.cpu().numpy(), pix_y.cpu().numpy(), ci, pixtocams.cpu().numpy().reshape(-1, 3, 3), camtoworlds.cpu().numpy())
    out = {k: torch.from_numpy(v).to(pix_x.device) for k, v in r.items()}
    if not want_imageplane:
        out.pop("imageplane")
    return out


def zip_loss_tail(rgb, tgt, lossmult=None, depth=None, tdepth=None, dmask=None, cmask=None, sem=None, labels=None, smask=None, hist=None,
                  mse=False, charb_padding=0.001, data_mult=1.0, depth_lambda=0.5, com_mult=0.2, sem_mult=0.04, pulse_width=(0.03, 0.003),
                  interlevel_mult=0.01, distortion_mult=0.005):
    from oracle import callers as oc
    assert not mse
    n = lambda t: None if t is None else t.detach().cpu().numpy()
    R, dev = rgb.shape[0], rgb.device
    sd = [n(h[0]) for h in hist] if hist is not None else None
    wt = [n(h[1]) for h in hist] if hist is not None else None
    if hist is None:   # the oracle always takes histograms: feed a dummy one with both regularisers off
        sd, wt, interlevel_mult, distortion_mult = [np.array([[0., 1.]])] * 3, [np.array([[1.]])] * 3, 0.0, 0.0
    sm = smask if smask is not None else (torch.ones(R) if sem is not None else None)
    L, G = oc.zip_loss_tail(n(rgb), n(tgt), n(lossmult), n(depth), n(tdepth), n(dmask), n(cmask), n(sem), n(labels), n(sm), sd, wt,
                            charb_padding, data_mult, depth_lambda, com_mult, sem_mult, pulse_width, interlevel_mult, distortion_mult)
    lm = np.ones(R) if lossmult is None else n(lossmult)
    out = torch.zeros(11)
    out[0] = 3 * float(lm.sum())
    out[1] = 0.0 if dmask is None else float(n(dmask).sum())
    out[2] = 0.0 if cmask is None else float(n(cmask).sum())
    out[3] = 0.0 if sem is None else float(n(sm).sum())
    for i, k in enumerate(("data", "mse", "depth", "d_complete", "sem", "interlevel", "distortion")):
        out[4 + i] = float(L.get(k, 0.0))
    t = lambda k: torch.from_numpy(np.asarray(G[k], np.float32)).to(dev) if k in G else None
    return out.to(dev), dict(rgb=t("rgb"), depth=t("depth"), semantic=t("semantic"), w0=t("w0"), w1=t("w1"), w2=t("w2"))


def zip_percentiles(tdist, weights, t_far, ps=(5, 50, 95)):
    from oracle import zip as oz
    bg = (1 - weights.sum(-1, keepdim=True)).clamp_min(0.)
    return oz.weighted_percentile(torch.cat([tdist, t_far.reshape(-1, 1)], -1), torch.cat([weights, bg], -1), list(ps))


def hash_decay(table, grad, offsets, L, C, mult, loss=None, grad_mult=1.0):
    off = offsets.cpu().numpy()
    for l in range(L):
        rows = int(off[l + 1] - off[l])
        k = mult / (rows * L * C)
        grad[off[l]:off[l + 1]] += 2 * k * grad_mult * table[off[l]:off[l + 1]]
        if loss is not None:
            loss += k * (table[off[l]:off[l + 1]].double() ** 2).sum().float()


def zip_encode_prop_fwd(tdist, origins, directions, radii, base_x, base_y, deg_jitter, table, offsets, grid_sizes, L, n, m, Sl, H, std_scale,
                        w1, b1, w2, b2, round_bf16):
    P = tdist.shape[0] * (tdist.shape[1] - 1)
    rb = lambda t: _zpm_rb(t, round_bf16)
    feat = torch.zeros(P, L)
    zip_encode_fwd(tdist, origins, directions, radii, base_x, base_y, deg_jitter, table, offsets, grid_sizes, feat, L, 1, n, m, Sl, H, std_scale)
    h = rb(torch.relu(rb(feat) @ rb(w1.reshape(-1, L)).t() + b1))
    return h @ rb(w2.reshape(-1, 1)) + b2


def classic_get_rays(H, W, focal, c2w, cx, cy, device):
    o, d = oc.get_rays(H, W, focal, torch.as_tensor(c2w).float().cpu(), [cx, cy])
    return o.contiguous(), d.contiguous()


def classic_ndc_rays(H, W, focal, near, rays_o, rays_d):
    return oc.ndc_rays(H, W, focal, near, rays_o, rays_d)


def classic_ray_batch(H, W, focal, cx, cy, c2w, c2w_static, rays_o, rays_d, n, ndc, near, far, depths, use_viewdirs, device):
    cpu = lambda t: None if t is None else torch.as_tensor(t).float().cpu()
    rows, _ = oc.ray_batch(H, W, focal, rays=None if rays_o is None else (rays_o, rays_d), c2w=cpu(c2w), ndc=ndc, near=near, far=far,
                           use_viewdirs=use_viewdirs, c2w_staticcam=cpu(c2w_static), depths=depths, ori_points=[cx, cy])
    return rows.contiguous()


# ---- fused register-resident MLPs (csrc/fmlp.hip): a model of the kernel's fragment data flow ---------------------------------
# An activation is a list of k-steps [M, 16] whose 16 positions are the MFMA's reduction slots (lane half h, element e -> 8 h + e).
# Inputs loaded from memory fill them in natural order; a layer's accumulator block (32 outputs, natural order n) becomes two
# k-steps whose position p carries output P[p] / 16 + P[p] -- exactly what the lanes hold (see the kernel header).
_P = torch.tensor([0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15])


class _FStream:
    def __init__(self, stream, bias):
        self.s, self.b, self.f, self.nb = stream.float(), bias, 0, 0
        self.dt = stream.dtype if stream.dtype in (torch.bfloat16, torch.float16) else torch.bfloat16   # the kernel's 16-bit rounding type

    def frag(self):                                  # [32 n, 16 positions]: lane = half * 32 + n holds positions 8 half .. +7
        w = self.s[self.f].view(2, 32, 8).permute(1, 0, 2).reshape(32, 16)
        self.f += 1
        return w

    def block(self, segs, relu, to_frags=True, store=None, j=0):
        acc = self.b[self.nb * 32:(self.nb + 1) * 32].clone()[None, :].expand(segs[0][0].shape[0], 32).clone()
        self.nb += 1
        for seg in segs:
            for x in seg:
                acc = acc + x @ self.frag().t()
        if not to_frags:
            return acc
        y = acc.to(self.dt).float()
        if relu:
            y = torch.relu(y)
        if store is not None:                        # training forward: the block's 32 outputs in natural order, row-major
            store[:, 32 * j:32 * j + 32] = y.to(store.dtype)
        return [y[:, _P], y[:, 16 + _P]]

    def dense(self, segs, nblocks, relu, store=None):
        out = []
        for j in range(nblocks):
            out += self.block(segs, relu, store=store, j=j)
        return out


def _rows_to_ksteps(X, nk):
    return [X[:, 16 * s:16 * s + 16].float() for s in range(nk)]


def fmlp_classic_fwd(E, VE, stream, bias, raw, acts=None):
    assert stream.shape[0] == 1184 and bias.numel() == 78 * 32
    st = _FStream(stream, bias)
    a = acts if acts is not None else [None] * 10
    e, ve = _rows_to_ksteps(E, 4), _rows_to_ksteps(VE, 2)
    p = st.dense([e], 8, True, a[0])
    for i in range(4):
        p = st.dense([p], 8, True, a[1 + i])
    p = st.dense([e, p], 8, True, a[5])
    p = st.dense([p], 8, True, a[6])
    q = st.dense([p], 8, True, a[7])
    sigma = st.block([q], False, to_frags=False)[:, 0]
    feat = st.dense([q], 8, False, a[8])
    hv = st.dense([feat, ve], 4, True, a[9])
    rgb = st.block([hv], False, to_frags=False)[:, :3]
    assert st.f == 1184 and st.nb == 78
    raw[:, :3] = rgb
    raw[:, 3] = sigma


def fmlp_zip_train_fwd(Fb, D, stream, bias, raw_rgb, raw_d, acts, bits):
    """model of fzip_fwd_kernel<.., STORE>: the same pass, every layer output stored in natural order, ReLU masks of h and H3 registered"""
    assert stream.shape[0] == 464 and bias.numel() == 35 * 32 and len(acts) == 4 and len(bits) == 3
    st = _FStream(stream, bias)
    f, dv = _rows_to_ksteps(Fb, 4), _rows_to_ksteps(D, 1)
    h1 = st.dense([f], 2, True, acts[0])
    x = st.dense([h1], 8, False, acts[1])
    raw_d.view(-1)[:] = st.block([h1], False, to_frags=False)[:, 0]
    h2 = st.dense([x, dv], 8, True, acts[2])
    rgb = None
    for j in range(8):
        blk = st.block([h2, x, dv], True, store=acts[3], j=j)
        piece = st.block([blk], False, to_frags=False)
        rgb = piece if rgb is None else rgb + piece
    assert st.f == 460 and st.nb == 35
    raw_rgb[:, :3] = rgb[:, :3]
    _BITS[bits[0].data_ptr()] = acts[0][:, :64].float() > 0
    _BITS[bits[1].data_ptr()] = acts[2][:, :256].float() > 0
    _BITS[bits[2].data_ptr()] = acts[3][:, :256].float() > 0


def fmlp_zip_chain_bwd(d_rgb, d_den, stream, bits, dz, g_bias):
    """model of fzip_chain_bwd_kernel (no biases in this stream: every block starts from zero): the transposed layers in chain order, masks from
    the registered bit masks, bias gradients = column sums of the fp32 (masked) pre-rounding values, dx consumed block by block by dH1"""
    assert stream.shape[0] == 448 and len(bits) == 3 and len(dz) == 5 and len(g_bias) == 4
    M = d_rgb.shape[0]
    st = _FStream(stream, torch.zeros(64 * 32))
    dt = st.dt
    mH1, mh, mH3 = (_BITS[b.data_ptr()] for b in bits)

    def finish(acc, mask, store, j, gb):
        if mask is not None:
            acc = acc * mask[:, 32 * j:32 * j + 32].float()
        if gb is not None:
            gb[32 * j:32 * j + 32] += acc.sum(0)
        y = acc.to(dt).float()
        store[:, 32 * j:32 * j + 32] = y.to(store.dtype)
        return [y[:, _P], y[:, 16 + _P]]

    def step(segs, nblocks, mask, store, gb):
        out = []
        for j in range(nblocks):
            out += finish(st.block(segs, False, to_frags=False), mask, store, j, gb)
        return out
    g = torch.zeros(M, 16); g[:, :3] = d_rgb[:, :3]
    g = [g.to(dt).float()]
    e = torch.zeros(M, 32); e[:, :d_den.shape[1]] = d_den
    e = e.to(dt).float()
    e = [e[:, :16], e[:, 16:]]
    p = step([g], 8, mH3, dz[0], g_bias[0])
    q = step([p], 8, mh, dz[1], g_bias[1])
    h0 = h1 = None
    for j in range(8):
        lh = finish(st.block([q, p, e], False, to_frags=False), None, dz[2], j, g_bias[2])
        a0 = st.block([lh], False, to_frags=False); a1 = st.block([lh], False, to_frags=False)
        h0 = a0 if h0 is None else h0 + a0
        h1 = a1 if h1 is None else h1 + a1
    dh1 = finish(h0, mH1, dz[3], 0, g_bias[3]) + finish(h1, mH1, dz[3], 1, g_bias[3])
    step([dh1], 2, None, dz[4], None)
    assert st.f == 448


def fmlp_zip_fwd(Fb, D, stream, bias, raw_rgb, raw_d, x32=None):
    """model of fzip_fwd_kernel: density_layer.0 (2 blocks), density_layer.2 (8 blocks: x), its row 0 once more in fp32, lin_second_stage_0
    on [x | dir], then lin_second_stage_1 block by block, each block's two k-steps multiplied into the rgb accumulator right away"""
    assert stream.shape[0] == 464 and bias.numel() == 35 * 32 and stream.dtype == Fb.dtype
    st = _FStream(stream, bias)
    f, dv = _rows_to_ksteps(Fb, 4), _rows_to_ksteps(D, 1)
    h1 = st.dense([f], 2, True)
    x = st.dense([h1], 8, False)
    if x32 is not None:                                         # block 0 of x back in natural order
        inv = torch.argsort(_P)
        x32[:, :16] = x[0][:, inv].to(x32.dtype); x32[:, 16:32] = x[1][:, inv].to(x32.dtype)
    raw_d.view(-1)[:] = st.block([h1], False, to_frags=False)[:, 0]
    h2 = st.dense([x, dv], 8, True)
    rgb = None
    for j in range(8):
        blk = st.block([h2, x, dv], True)
        piece = st.block([blk], False, to_frags=False)            # (the piece's bias block: rgb_layer's bias in the first, zeros after)
        rgb = piece if rgb is None else rgb + piece
    assert st.f == 460 and st.nb == 35
    raw_rgb[:, :3] = rgb[:, :3]


def fmlp_classic_pts_fwd(pts, viewdirs, S, stream, bias, raw):
    M = pts.shape[0]
    pad = lambda t, w: torch.cat([t, torch.zeros(M, w - t.shape[1])], -1).to(torch.bfloat16)
    fmlp_classic_fwd(pad(oc.embed(pts, 10), 64), pad(oc.embed(viewdirs[:, None].expand(-1, S, -1).reshape(-1, 3), 4), 64), stream, bias, raw)


def fmlp_classic_train_fwd(E, VE, stream, bias, raw, acts, bits):
    assert len(acts) == 10 and len(bits) == 9
    fmlp_classic_fwd(E, VE, stream, bias, raw, acts)
    for y, w in zip(acts[:8] + [acts[9]], bits):     # what ACT_MASK_BITS consumers look up (see linear_fwd above); bits[8]: views_linears.0
        _BITS[w.data_ptr()] = y.float() > 0


def fmlp_proposal_fwd(E, stream, bias, raw_density, acts=None):
    assert stream.shape[0] == 448 and bias.numel() == 33 * 32
    st = _FStream(stream, bias)
    a = acts if acts is not None else [None] * 4
    p = st.dense([_rows_to_ksteps(E, 6)], 8, True, a[0])
    for i in range(3):
        p = st.dense([p], 8, True, a[1 + i])
    raw_density.view(-1)[:] = st.block([p], False, to_frags=False)[:, 0]
    assert st.f == 448 and st.nb == 33


def fmlp_proposal_train_fwd(E, stream, bias, raw_density, acts, bits):
    assert len(acts) == 4 and len(bits) == 4
    fmlp_proposal_fwd(E, stream, bias, raw_density, acts)
    for y, w in zip(acts, bits):
        _BITS[w.data_ptr()] = y.float() > 0


def fcolour_fwd(CB, stream, bias, raw_rgb, acts=None, bits=None, variant=0):
    """model of fcolour_fwd_kernel: cond_layers.0 K-MAJOR (66 k-steps x 4 blocks), then two 128-wide layers and the rgb head"""
    assert stream.shape[0] == 336 and bias.numel() == 13 * 32
    st = _FStream(stream, bias)
    M = CB.shape[0]
    acc = [bias[32 * j:32 * j + 32].clone()[None, :].expand(M, 32).clone() for j in range(4)]
    st.nb = 4
    for x in _rows_to_ksteps(CB, 66):
        for j in range(4):
            acc[j] = acc[j] + x @ st.frag().t()
    p = []
    for j in range(4):
        y = torch.relu(acc[j].to(torch.bfloat16).float())
        if acts is not None:
            acts[0][:, 32 * j:32 * j + 32] = y.to(acts[0].dtype)
        p += [y[:, _P], y[:, 16 + _P]]
    q = st.dense([p], 4, True, None if acts is None else acts[1])
    p = st.dense([q], 4, True, None if acts is None else acts[2])
    raw_rgb[:, :3] = st.block([p], False, to_frags=False)[:, :3]
    assert st.f == 336 and st.nb == 13
    if acts is not None:
        for y, w in zip(acts, bits):
            _BITS[w.data_ptr()] = y[:, :128].float() > 0


def fcolour_bwd(d_raw_rgb, stream, bits, dC, dB, g_bias):
    """model of fcolour_bwd_kernel: the data-gradient chain on the transposed weights; masks, bias gradients (of the masked fp32
    accumulators), bf16 stores"""
    assert stream.shape[0] == 336
    st = _FStream(stream, torch.zeros(44 * 32))
    M = d_raw_rgb.shape[0]
    g = torch.zeros(M, 16)
    g[:, :3] = d_raw_rgb.to(torch.bfloat16).float()

    def layer(inp, nblocks, mask, out, gb):
        frs = []
        for j in range(nblocks):
            a = st.block([inp], False, to_frags=False) * mask[:, 32 * j:32 * j + 32]
            gb[32 * j:32 * j + 32] += a.sum(0)
            y = a.to(torch.bfloat16).float()
            out[:, 32 * j:32 * j + 32] = y.to(out.dtype)
            frs += [y[:, _P], y[:, 16 + _P]]
        return frs
    m = [_BITS[b.data_ptr()].float() for b in bits]
    p = layer([g], 4, m[0], dC[0], g_bias[0])
    p = layer(p, 4, m[1], dC[1], g_bias[1])
    p = layer(p, 4, m[2], dC[2], g_bias[2])
    layer(p, 32, m[3], dB, g_bias[3])
    assert st.f == 324


def fchain_bwd(net, d_raw, stream, bits, dz, g_bias):
    """model of fchain_bwd_kernel: the data-gradient chains of the 256-wide networks on the transposed weights (masks, bf16
    stores, bias gradients = column sums of the stored gradients)"""
    classic = net == 0
    assert stream.shape[0] == (1104 if classic else 400)
    st = _FStream(stream, torch.zeros(80 * 32))
    M = d_raw.shape[0]
    d_raw = d_raw.reshape(M, -1)

    def head(cols):
        g = torch.zeros(M, 16)
        g[:, :len(cols)] = d_raw[:, cols].to(torch.bfloat16).float()
        return g

    def layer(segs, nblocks, mask, out, gb):
        frs = []
        for j in range(nblocks):
            a = st.block(segs, False, to_frags=False)
            if mask is not None:
                a = a * mask[:, 32 * j:32 * j + 32]
            y = a.to(torch.bfloat16).float()
            gb[32 * j:32 * j + 32] += y.sum(0)                    # (the kernel sums the bf16 gradients it stores, on the matrix cores)
            out[:, 32 * j:32 * j + 32] = y.to(out.dtype)
            frs += [y[:, _P], y[:, 16 + _P]]
        return frs
    m = [_BITS[b.data_ptr()].float() for b in bits]
    if classic:
        p = layer([[head([0, 1, 2])]], 4, m[8][:, :128], dz[0], g_bias[0])
        p = layer([p], 8, None, dz[1], g_bias[1])
        p = layer([p, [head([3])]], 8, m[7], dz[2], g_bias[2])
        for i in range(6, -1, -1):
            p = layer([p], 8, m[i], dz[9 - i], g_bias[9 - i])
        assert st.f == 1100
    else:
        p = layer([[head([0])]], 8, m[3], dz[0], g_bias[0])
        for i in range(2, -1, -1):
            p = layer([p], 8, m[i], dz[3 - i], g_bias[3 - i])
        assert st.f == 392


def gather_pack(flat, idx, dst, tiles=None):
    if tiles is not None and tiles.numel():
        t = tiles.long()
        i = torch.arange(16).view(1, 16, 1)
        j = torch.arange(64).view(1, 1, 64)
        d = (t[:, 0].view(-1, 1, 1) + i * t[:, 3].view(-1, 1, 1) + j).reshape(-1)
        s = (t[:, 1].view(-1, 1, 1) + j * t[:, 2].view(-1, 1, 1) + i).reshape(-1)
        assert bool((idx.long()[d] == -3).all()) and int((idx == -3).sum()) == d.numel()
        keep = dst.view(-1).clone()
    k = idx.long()
    lo = (k >= 0) & ((k & (1 << 30)) != 0)                     # split-bf16 weights: the low part bf16(x - bf16(x))
    x = flat[torch.where(k >= 0, k & 0x3fffffff, torch.zeros_like(k))]
    x = torch.where(lo, x - x.to(torch.bfloat16).float(), x)
    v = torch.where(k >= 0, x, torch.where(k == -2, torch.ones_like(flat[:1]), torch.zeros_like(flat[:1])).expand_as(k))
    dst.view(-1)[:] = v.to(dst.dtype)
    if tiles is not None and tiles.numel():
        skip = idx == -3
        dst.view(-1)[skip] = keep[skip]
        dst.view(-1)[d] = flat[s].to(dst.dtype)


def gather_pack_pair(flat, idx, dst, tiles, idx32, dst32):
    gather_pack(flat, idx, dst, tiles)
    gather_pack(flat, idx32, dst32)


def adam_step_dev(p, g, m, v, lr, b1, b2, eps, step_dev, grad_scale=1.0, zero_grad=True, **kw):
    adam_step(p, g, m, v, lr, b1, b2, eps, 0, grad_scale, zero_grad, step_dev=step_dev, **kw)


_NAMES = ["nonfinite_flag", "fmlp_zip_fwd", "fmlp_zip_train_fwd", "fmlp_zip_chain_bwd", "zip_prop_mlp_fwd", "zip_prop_mlp_bwd", "colsum_wide_f32", "zip_glo_modulate", "zip_glo_modulate_bwd", "fchain_bwd", "app_embed", "app_embed_bwd", "split_cast", "split8_cast", "fcolour_fwd", "fcolour_bwd", "gather_pack", "adam_step_dev", "grad_clip_coef", "fmlp_classic_fwd", "fmlp_classic_pts_fwd", "fmlp_proposal_fwd", "fmlp_classic_train_fwd", "fmlp_proposal_train_fwd", "classic_get_rays", "classic_ndc_rays", "classic_ray_batch", "zip_encode_prop_fwd", "mip_encode_bwd", "mip_viewenc_bwd", "hash_decay", "zip_percentiles", "zip_pixels_to_rays", "zip_loss_tail", "semantic_composite_fwd", "semantic_composite_bwd", "zip_resample", "zip_encode_fwd", "zip_encode_fwd_count", "zip_encode_bwd", "zip_encode_bwd_binned", "zip_encode_ray_bwd", "zip_composite_fwd", "zip_composite_bwd",
          "linear_fwd", "linear_wgrad", "classic_embed", "mip_encode", "mip_viewenc", "classic_sample_pdf", "classic_points",
          "classic_merge_sort", "mip_resample", "stratified", "mip_composite_fwd", "mip_composite_bwd", "classic_composite_fwd",
          "classic_composite_bwd", "adam_step", "colsum_f32", "cast_pad", "pinhole_rays", "mip_loss_tail", "jitter_u", "gather_pack_pair"]


@contextlib.contextmanager
def emulate_ops():
    """Swap the C-ABI wrappers of snerf_amd.ops for the CPU emulation inside a `with` block."""
    from snerf_amd import ops
    saved = {n: getattr(ops, n) for n in _NAMES}
    try:
        for n in _NAMES:
            setattr(ops, n, globals()[n])
        yield ops
    finally:
        for n, f in saved.items():
            setattr(ops, n, f)
