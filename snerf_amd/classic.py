"""Drop-in Python operator API of the classic NeRF renderer (path B), backed by
the HIP kernels of libsnerf_hip.so.

Mirrors, with the same names, argument meaning and return layout:
  get_embedder   s-nerf/model/run_nerf_helpers.py:55-70
  NeRF           s-nerf/model/run_nerf_helpers.py:74-126   (same state_dict keys and shapes)
  run_network    s-nerf/model/run_nerf_helpers.py:460-474
  raw2outputs    s-nerf/model/run_nerf_helpers.py:381-424
  sample_pdf     s-nerf/model/run_nerf_helpers.py:336-379
  render_rays    s-nerf/model/render.py:281-409
  batchify_rays  s-nerf/model/render.py:8-19

Differences that are deliberate and documented in DESIGN.md:
  * tensors must live on the GPU; there is no CPU path (the library raises);
  * ``NeRF(..., compute="bf16"|"f32")`` selects the GEMM arithmetic (bf16 MFMA
    with fp32 accumulation, or exact-fp32 MFMA for parity work);
  * random draws are taken with ``torch.rand`` on the device like the
    reference, but may also be passed in (``t_rand=``, ``u=``) for parity tests;
  * ``pytest=True`` (numpy-seeded draws) is honoured the same way the
    reference does it.
"""
from typing import Optional

import numpy as np
import torch
from torch import nn

from . import ops
from .mlp import ClassicNeRFNet, ParamArena


class Embedder:
    """Descriptor of the reference's positional encoder; the encoding itself is fused into
    ``run_network``'s first kernel.  Calling it encodes stand-alone (used by tests / NeRF.forward)."""

    def __init__(self, multires: int, input_dims: int = 3):
        self.multires = multires
        self.input_dims = input_dims
        self.out_dim = input_dims * (1 + 2 * multires)

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        flat = x.reshape(-1, 3).contiguous().float()
        out = torch.empty(flat.shape[0], self.out_dim, dtype=torch.float32, device=x.device)
        ops.classic_embed(flat, None, 1, self.multires, 0, out, None, self.out_dim, None, 0, ops.F32)
        return out.reshape(list(x.shape[:-1]) + [self.out_dim])


def get_embedder(multires, i=0):
    if i == -1:
        raise NotImplementedError("identity embedding (i_embed=-1) is not on the accelerated path")
    e = Embedder(multires)
    return e, e.out_dim


def _register_tree(root: nn.Module, dotted: str, param: nn.Parameter):
    """Register `param` under nested container modules so that state_dict() keys equal `dotted`."""
    parts = dotted.split(".")
    mod = root
    for p in parts[:-1]:
        if p not in mod._modules:
            mod.add_module(p, nn.Module())
        mod = mod._modules[p]
    mod.register_parameter(parts[-1], param)


class _ArenaModule(nn.Module):
    """nn.Module whose parameters are views into one flat fp32 arena."""

    def _setup_arena(self, shapes, device):
        self.arena = ParamArena(shapes, device)
        self._pnames = []
        for n, _ in shapes:
            prm = nn.Parameter(self.arena.p[n])
            _register_tree(self, n, prm)
            self._pnames.append(n)

    def param_list(self):
        d = dict(self.named_parameters())
        return [d[n] for n in self._pnames]

    def _param_version(self):
        return sum(p._version for p in self.param_list()) + self.arena.epoch

    def _check_arena(self):
        # .to()/.cuda()/load_state_dict(assign=True) would silently detach parameters from the arena
        for n, p in zip(self._pnames, self.param_list()):
            if p.data_ptr() != self.arena.p[n].data_ptr():
                raise RuntimeError(f"parameter {n} no longer aliases the flat arena; construct the module on its final device")


def _dt(compute: str) -> int:
    return {"bf16": ops.BF16, "f32": ops.F32, "fp32": ops.F32}[compute]


class NeRF(_ArenaModule):
    """Same constructor signature and parameter names as the reference's ``NeRF`` (use_viewdirs=True
    is the accelerated configuration)."""

    def __init__(self, D=8, W=256, input_ch=3, input_ch_views=3, output_ch=4, skips=[4], use_viewdirs=False,
                 compute: str = "bf16", device="cuda", variant: int = 8):
        super().__init__()
        if not use_viewdirs:
            raise NotImplementedError("the accelerated NeRF requires use_viewdirs=True (the S-NeRF configuration)")
        self.D, self.W, self.input_ch, self.input_ch_views, self.skips, self.use_viewdirs = D, W, input_ch, input_ch_views, list(skips), True
        self.compute = compute
        shapes = ClassicNeRFNet.param_shapes(D, W, input_ch, input_ch_views, tuple(skips))
        self._setup_arena(shapes, torch.device(device))
        self.net = ClassicNeRFNet(self.arena, "", _dt(compute), D, W, input_ch, input_ch_views, tuple(skips), variant)
        self.net.version_fn = self._param_version
        with torch.no_grad():  # nn.Linear default init, like the reference module
            for i in range(D):
                self._init_linear(f"pts_linears.{i}")
            for n in ("views_linears.0", "feature_linear", "alpha_linear", "rgb_linear"):
                self._init_linear(n)

    def _init_linear(self, name):
        w, b = self.arena.p[name + ".weight"], self.arena.p[name + ".bias"]
        nn.init.kaiming_uniform_(w, a=5 ** 0.5)
        bound = 1.0 / (w.shape[1] ** 0.5)
        nn.init.uniform_(b, -bound, bound)

    def forward(self, x):
        raise NotImplementedError("call run_network(inputs, viewdirs, model, embed_fn, embeddirs_fn): the encoding is fused "
                                  "with the first layer, pre-embedded inputs never exist on this path")


class _RunNetworkFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model: NeRF, pts, viewdirs, S, keep, *params):
        raw, saved = model.net.forward(pts, viewdirs, S, keep)
        ctx.model, ctx.saved, ctx.keep = model, saved, keep
        return raw

    @staticmethod
    def backward(ctx, d_raw):
        if not ctx.keep:
            raise RuntimeError("run_network was executed without saved activations")
        m = ctx.model
        m.arena.grad.zero_()
        m.net.backward(d_raw.contiguous(), ctx.saved)
        ctx.saved = None
        grads = tuple(m.arena.g[n].clone() for n in m._pnames)
        m.arena.grad.zero_()      # leave the arena clean for whoever accumulates into it next
        return (None, None, None, None, None) + grads


def run_network(inputs, viewdirs, fn, embed_fn, embeddirs_fn, netchunk=1024 * 64):
    """inputs [N,S,3], viewdirs [N,3] -> [N,S,4] (rgb, sigma).  `fn` must be a snerf_amd NeRF module;
    the embedders only describe the encoding (it is fused into the first kernel).  `netchunk` is
    accepted for signature compatibility: the kernels tile internally."""
    if not isinstance(fn, NeRF):
        raise TypeError("run_network: `fn` must be a snerf_amd.classic.NeRF (no eager fallback for foreign modules)")
    if not isinstance(embed_fn, Embedder) or embed_fn.out_dim != fn.input_ch:
        raise TypeError("run_network: embed_fn must come from snerf_amd.classic.get_embedder and match the model")
    if viewdirs is None or not isinstance(embeddirs_fn, Embedder) or embeddirs_fn.out_dim != fn.input_ch_views:
        raise TypeError("run_network: viewdirs + matching embeddirs_fn are required (use_viewdirs=True)")
    fn._check_arena()
    N, S = inputs.shape[0], inputs.shape[1]
    pts = inputs.reshape(-1, 3).contiguous().float()
    vd = viewdirs.float()
    if vd.stride(-1) != 1:
        vd = vd.contiguous()
    params = fn.param_list()
    keep = torch.is_grad_enabled() and any(p.requires_grad for p in params)   # grad mode is off inside Function.forward
    raw = _RunNetworkFn.apply(fn, pts, vd, S, keep, *params)
    return raw.reshape(N, S, raw.shape[-1])


class _Raw2OutputsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw, z_vals, rays_d, noise, white):
        raw2 = raw.reshape(-1, raw.shape[-1])
        rgb, disp, acc, w, depth = ops.classic_composite_fwd(raw2, noise, z_vals, rays_d, white)
        ctx.save_for_backward(raw2, z_vals, rays_d, noise if noise is not None else torch.empty(0, device=raw.device), w, acc, depth)
        ctx.has_noise, ctx.white, ctx.shape = noise is not None, white, raw.shape
        return rgb, disp, acc, w, depth

    @staticmethod
    def backward(ctx, g_rgb, g_disp, g_acc, g_w, g_depth):
        raw2, z_vals, rays_d, noise, w, acc, depth = ctx.saved_tensors
        d_raw = torch.zeros_like(raw2)
        c = lambda t: None if t is None else t.contiguous()
        ops.classic_composite_bwd(raw2, noise if ctx.has_noise else None, z_vals, rays_d, ctx.white, w, acc, depth,
                                  c(g_rgb), c(g_disp), c(g_acc), c(g_depth), c(g_w), d_raw)
        return d_raw.reshape(ctx.shape), None, None, None, None


def raw2outputs(raw, z_vals, rays_d, raw_noise_std=0, white_bkgd=False, pytest=False, noise=None):
    """-> (rgb_map, disp_map, acc_map, weights, depth_map); differentiable w.r.t. `raw`."""
    if noise is None and raw_noise_std > 0.:
        if pytest:
            np.random.seed(0)
            noise = torch.tensor(np.random.rand(*list(raw[..., 3].shape)) * raw_noise_std, dtype=torch.float32, device=raw.device)
        else:
            noise = torch.randn(raw[..., 3].shape, device=raw.device) * raw_noise_std
    z = z_vals.contiguous().float()
    rd = rays_d.float()
    if rd.stride(-1) != 1:
        rd = rd.contiguous()
    return _Raw2OutputsFn.apply(raw.float().contiguous(), z, rd, None if noise is None else noise.contiguous().float(), bool(white_bkgd))


def _draw_u(N, N_samples, det, pytest, u, dev):
    if u is None:
        if pytest:  # run_nerf_helpers.py:350-359: numpy-seeded draws
            np.random.seed(0)
            un = np.broadcast_to(np.linspace(0., 1., N_samples), (N, N_samples)) if det else np.random.rand(N, N_samples)
            u = torch.tensor(np.ascontiguousarray(un), dtype=torch.float32, device=dev)
        elif det:
            u = torch.linspace(0., 1., steps=N_samples).to(dev)
        else:
            u = torch.rand(N, N_samples, device=dev)
    return u.to(dev).float().contiguous()


def sample_pdf(bins, weights, N_samples, det=False, pytest=False, u=None, return_inds=False):
    """Reference signature: bins [N,nb], weights [N,nb-1] -> samples [N,N_samples] (no gradient, like the
    reference's use under .detach()).  `u`/`return_inds` are parity-test extras."""
    u = _draw_u(bins.shape[0], N_samples, det, pytest, u, bins.device)
    s, inds, _ = ops.classic_sample_pdf(bins.contiguous().float(), weights.detach().contiguous().float(), u, False, return_inds)
    return (s, inds.long()) if return_inds else s


def _sample_pdf_z(z_vals, weights, N_samples, det, pytest, u, return_inds):
    """sample_pdf(z_vals_mid, weights[...,1:-1], ...) as render_rays calls it (render.py:378-380), fused:
    mids, the slice and z_std are formed inside the kernel."""
    u = _draw_u(z_vals.shape[0], N_samples, det, pytest, u, z_vals.device)
    return ops.classic_sample_pdf(z_vals, weights.detach().contiguous(), u, True, return_inds, True)


def render_rays(ray_batch, network_fn, network_query_fn, N_samples, retraw=False, lindisp=False, perturb=0.,
                N_importance=0, network_fine=None, white_bkgd=False, raw_noise_std=0., verbose=False, pytest=False,
                t_rand: Optional[torch.Tensor] = None, u: Optional[torch.Tensor] = None, return_inds: bool = False):
    """Volumetric rendering of a ray batch [N, 8|11] = [o3, d3, near, far, (viewdir3)]; returns the
    reference's dict (render.py:394-401).  Extra keyword-only inputs `t_rand` [N,S] / `u` [N,Nimp]
    replace the internal torch.rand draws (parity tests)."""
    N_rays = ray_batch.shape[0]
    dev = ray_batch.device
    rb = ray_batch.float()
    if rb.stride(-1) != 1:
        rb = rb.contiguous()
    rays_o, rays_d = rb[:, 0:3], rb[:, 3:6]
    viewdirs = rb[:, -3:] if rb.shape[-1] > 9 else None
    base = torch.linspace(0., 1., steps=N_samples).to(dev)
    if perturb > 0. and t_rand is None:
        if pytest:
            np.random.seed(0)
            t_rand = torch.tensor(np.random.rand(N_rays, N_samples), dtype=torch.float32, device=dev)
        else:
            t_rand = torch.rand(N_rays, N_samples, device=dev)
    z_vals = ops.stratified(base, None if t_rand is None else t_rand.contiguous().float(), rb[:, 6], rb[:, 7], N_rays, 0, lindisp)
    pts = ops.classic_points(rb, z_vals)
    if network_fn is None:
        raise NotImplementedError("alpha_model variants (network_fn=None) are outside the accelerated path")
    raw = network_query_fn(pts, viewdirs, network_fn)
    rgb_map, disp_map, acc_map, weights, depth_map = raw2outputs(raw, z_vals, rays_d, raw_noise_std, white_bkgd, pytest=pytest)
    inds = None
    if N_importance > 0:
        rgb_map_0, disp_map_0, acc_map_0 = rgb_map, disp_map, acc_map
        z_samples, inds, z_std = _sample_pdf_z(z_vals, weights, N_importance, perturb == 0., pytest, u, return_inds)
        z_all = ops.classic_merge_sort(z_vals, z_samples)
        pts = ops.classic_points(rb, z_all)
        run_fn = network_fn if network_fine is None else network_fine
        raw = network_query_fn(pts, viewdirs, run_fn)
        rgb_map, disp_map, acc_map, weights_, depth_map = raw2outputs(raw, z_all, rays_d, raw_noise_std, white_bkgd, pytest=pytest)
    ret = {'rgb_map': rgb_map, 'disp_map': disp_map, 'acc_map': acc_map, 'depth_map': depth_map, 'z_vals_map': z_vals,
           'weights': weights}
    if retraw:
        ret['raw'] = raw
    if N_importance > 0:
        ret['rgb0'] = rgb_map_0
        ret['disp0'] = disp_map_0
        ret['acc0'] = acc_map_0
        ret['z_std'] = z_std
        if return_inds:
            ret['inds'] = inds.long()
            ret['z_samples'] = z_samples
            ret['z_vals_fine'] = z_all
    return ret


def batchify_rays(rays_flat, chunk=1024 * 32, **kwargs):
    """Render rays in chunks (render.py:8-19)."""
    all_ret = {}
    for i in range(0, rays_flat.shape[0], chunk):
        ret = render_rays(rays_flat[i:i + chunk], **kwargs)
        for k in ret:
            all_ret.setdefault(k, []).append(ret[k])
    return {k: torch.cat(v, 0) for k, v in all_ret.items()}


def make_network_query_fn(embed_fn, embeddirs_fn, netchunk=1024 * 64):
    """The lambda create_nerf builds (render.py:215-218)."""
    return lambda inputs, viewdirs, network_fn: run_network(inputs, viewdirs, network_fn, embed_fn=embed_fn,
                                                            embeddirs_fn=embeddirs_fn, netchunk=netchunk)
