"""Drop-in Python operator API of the classic NeRF renderer (path B), backed by
the HIP kernels of libsnerf_hip.so.

Mirrors, with the same names, argument meaning and return layout:
  get_embedder   s-nerf/model/run_nerf_helpers.py:55-70
  NeRF           s-nerf/model/run_nerf_helpers.py:74-126   (same state_dict keys and shapes)
  run_network    s-nerf/model/run_nerf_helpers.py:460-474
  raw2outputs    s-nerf/model/run_nerf_helpers.py:381-424
  sample_pdf     s-nerf/model/run_nerf_helpers.py:336-379
  NeRF_RGB       s-nerf/model/run_nerf_helpers.py:157-212  (colour network over a frozen alpha model)
  render_rays    s-nerf/model/render.py:281-409
  batchify_rays  s-nerf/model/render.py:8-19
  get_rays       s-nerf/model/run_nerf_helpers.py:247-258
  ndc_rays       s-nerf/model/run_nerf_helpers.py:314-332
  render         s-nerf/model/render.py:22-91
  render_path    s-nerf/model/render.py:94-135
  create_nerf    s-nerf/model/render.py:165-278

Differences that are deliberate and documented in DESIGN.md:
  * tensors must live on the GPU; there is no CPU path (the library raises);
  * ``NeRF(..., compute="bf16"|"f32")`` selects the GEMM arithmetic (bf16 MFMA
    with fp32 accumulation, or exact-fp32 MFMA for parity work);
  * random draws are taken with ``torch.rand`` on the device like the
    reference, but may also be passed in (``t_rand=``, ``u=``) for parity tests;
  * ``pytest=True`` (numpy-seeded draws) is honoured the same way the
    reference does it.
"""
import os
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
        if self.multires == 0:
            return x                                     # nn.Identity()
        flat = x.reshape(-1, 3).contiguous().float()
        out = torch.empty(flat.shape[0], self.out_dim, dtype=torch.float32, device=x.device)
        ops.classic_embed(flat, None, 1, self.multires, 0, out, None, self.out_dim, None, 0, ops.F32)
        return out.reshape(list(x.shape[:-1]) + [self.out_dim])


def get_embedder(multires, i=0):
    """(embedder, out_dim) (run_nerf_helpers.py:55-70).  i = -1: the identity (nn.Identity(), 3 in the reference, :56-57) -- here an
    Embedder with zero frequencies, whose fused encoding is the input itself."""
    e = Embedder(0 if i == -1 else multires)
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

    def load_state_dict(self, state_dict, strict=True, assign=False):
        """Also accepts the ``module.``-prefixed keys of a checkpoint saved from an nn.DataParallel / DDP wrapper (the reference
        saves the wrapped model: s-nerf/train.py:268, utils/device_utils.py:33,38, eval.py:72-74)."""
        if assign:
            raise ValueError("assign=True would detach the parameters from the flat arena")
        return super().load_state_dict(strip_module_prefix(state_dict), strict=strict)

    def set_deterministic(self, flag: bool = True):
        """Bit-reproducible parameter gradients for parity runs: the MLP executors fold weight- and bias-gradient
        partials in a fixed order instead of adding them with fp32 atomics.  Costs a workspace round trip per weight gradient."""
        for v in vars(self).values():
            for net in (v if isinstance(v, (list, tuple)) else [v]):
                if hasattr(net, "deterministic") and hasattr(net, "ensure_packed"):
                    net.deterministic = bool(flag)
        self._deterministic = bool(flag)                 # (embedding-table gradients: fixed-order accumulation, ops.app_embed_bwd)
        return self

    def _param_version(self):
        return sum(p._version for p in self.param_list()) + self.arena.epoch

    def _check_arena(self):
        # .to()/.cuda()/load_state_dict(assign=True) would silently detach parameters from the arena
        for n, p in zip(self._pnames, self.param_list()):
            if p.data_ptr() != self.arena.p[n].data_ptr():
                raise RuntimeError(f"parameter {n} no longer aliases the flat arena; construct the module on its final device")


def _dt(compute: str) -> int:
    """compute mode -> GEMM dtype code.  "bf16x3_fwd" (the mip path): the split-bf16 forward with a single-pass bf16 backward (mlp._Net.bwd_plain);
    "f16f8" (the mip path): fp16 tiles + e4m3 correction tiles in the forward (two pass-equivalents, the same 1e-4 contract), scaled fp16 backward"""
    return {"bf16": ops.BF16, "f32": ops.F32, "fp32": ops.F32, "bf16x3": ops.BF16X3, "bf16x3_fwd": ops.BF16X3, "f16f8": ops.F16F8, "fp16": ops.F16, "f16": ops.F16}[compute]


class NeRF(_ArenaModule):
    """Same constructor signature and parameter names as the reference's ``NeRF`` (use_viewdirs=True
    is the accelerated configuration)."""
    _alpha_head = True

    def __init__(self, D=8, W=256, input_ch=3, input_ch_views=3, output_ch=4, skips=[4], use_viewdirs=False,
                 compute: str = "bf16", device="cuda", variant: int = 8):
        super().__init__()
        # use_viewdirs=False (run_nerf_helpers.py:100-101, 122-124; create_nerf passes output_ch = 5 with N_importance > 0, render.py:180-183):
        # the trunk + output_linear(W -> output_ch) on the per-layer MFMA GEMMs; the fused register-resident kernels cover use_viewdirs=True
        oc = 0 if use_viewdirs else int(output_ch)
        if not use_viewdirs and not (self._alpha_head and 4 <= oc <= 16):
            raise ValueError("NeRF(use_viewdirs=False): output_ch in 4..16 (raw2outputs reads columns 0..3)")
        self.D, self.W, self.input_ch, self.input_ch_views, self.skips, self.use_viewdirs = D, W, input_ch, input_ch_views, list(skips), bool(use_viewdirs)
        self.output_ch = int(output_ch)
        self.compute = compute
        shapes = ClassicNeRFNet.param_shapes(D, W, input_ch, input_ch_views, tuple(skips), alpha_head=self._alpha_head, output_ch=oc)
        self._setup_arena(shapes, torch.device(device))
        self.net = ClassicNeRFNet(self.arena, "", _dt(compute), D, W, input_ch, input_ch_views, tuple(skips), variant,
                                  alpha_head=self._alpha_head, output_ch=oc, bwd_plain=compute == "bf16x3_fwd")
        self.net.version_fn = self._param_version
        with torch.no_grad():  # nn.Linear default init, like the reference module
            for i in range(D):
                self._init_linear(f"pts_linears.{i}")
            heads = ("views_linears.0", "output_linear") if oc else ("views_linears.0", "feature_linear", "rgb_linear") + (("alpha_linear",) if self._alpha_head else ())
            for n in heads:
                self._init_linear(n)

    def _init_linear(self, name):
        w, b = self.arena.p[name + ".weight"], self.arena.p[name + ".bias"]
        nn.init.kaiming_uniform_(w, a=5 ** 0.5)
        bound = 1.0 / (w.shape[1] ** 0.5)
        nn.init.uniform_(b, -bound, bound)

    def forward(self, x):
        raise NotImplementedError("call run_network(inputs, viewdirs, model, embed_fn, embeddirs_fn): the encoding is fused "
                                  "with the first layer, pre-embedded inputs never exist on this path")


class NeRF_RGB(NeRF):
    """``NeRF_RGB`` (run_nerf_helpers.py:157-212): the colour network of the two-stage variant -- same trunk / feature / views / rgb
    layers, NO alpha head; the density column of the output is the frozen ``alpha_model``'s (a ``NeRF``), evaluated under no_grad
    (:198-199).  ``alpha_model`` is a registered sub-module like in the reference, so ``state_dict()`` carries its parameters under
    ``alpha_model.`` and ``parameters()`` lists them (they never receive a gradient)."""
    _alpha_head = False

    def __init__(self, D=8, W=256, input_ch=3, input_ch_views=3, output_ch=4, skips=[4], use_viewdirs=False, alpha_model=None,
                 compute: str = "bf16", device="cuda", variant: int = 8):
        super().__init__(D, W, input_ch, input_ch_views, output_ch, skips, use_viewdirs, compute, device, variant)
        if alpha_model is not None and not (isinstance(alpha_model, NeRF) and alpha_model._alpha_head):
            raise TypeError("NeRF_RGB: alpha_model must be a snerf_amd.classic.NeRF")
        self.alpha_model = alpha_model


class _RunNetworkFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model: NeRF, pts, viewdirs, S, keep, *params):
        raw, saved = model.net.forward(pts, viewdirs, S, keep)
        if not model._alpha_head:
            # NeRF_RGB: column 3 is the frozen alpha model's density (run_nerf_helpers.py:198-199, under no_grad)
            if model.alpha_model is None:
                raise RuntimeError("NeRF_RGB without an alpha_model cannot be evaluated (the reference raises TypeError here)")
            model.alpha_model._check_arena()
            raw_a, _ = model.alpha_model.net.forward(pts, viewdirs, S, False)
            raw[:, 3] = raw_a[:, 3]
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
    if fn.use_viewdirs:
        if viewdirs is None or not isinstance(embeddirs_fn, Embedder) or embeddirs_fn.out_dim != fn.input_ch_views:
            raise TypeError("run_network: viewdirs + matching embeddirs_fn are required (use_viewdirs=True)")
    elif viewdirs is not None:
        # (the reference would concatenate the encoded directions and fail in torch.split on a network built without them)
        raise TypeError("run_network: this network was built with use_viewdirs=False; pass viewdirs=None")
    fn._check_arena()
    N, S = inputs.shape[0], inputs.shape[1]
    pts = inputs.reshape(-1, 3).contiguous().float()
    vd = None
    if viewdirs is not None:
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


def _fine_pass_front_to_back(rb, z_all, viewdirs, run_fn, network_query_fn, eps_t, G):
    """The fine network front to back in groups of G samples (inference extension, NOT in the reference): after every group the rays
    whose transmittance -- from the fine network's own densities, raw2outputs' alpha (run_nerf_helpers.py:394-414) -- has fallen to
    <= eps_t leave, and the rows of the next group are compacted to the survivors.  Unevaluated samples keep raw = 0 (alpha = 0,
    weight 0): the weights they would have had sum to <= eps_t per ray, which bounds the error of acc_map and (x 1) of rgb_map.
    The 64 uniform coarse positions the fine pass re-evaluates (render.py:380-389) are what lies behind an opaque hit: measured on a
    fitted street scene 26 % of the fine evaluations at eps_t = 1e-4 (tools/ert_classic_analysis.py).
    -> (raw [N, S, C], fine-network evaluations done): the count is returned per call -- render_rays hands it on as ret['ert_evals'] --,
    not kept in module state (two models rendering from two threads would race on it)."""
    N, S = z_all.shape
    dev = z_all.device
    dn = rb[:, 3:6].norm(dim=-1)
    alive = None                       # None = every ray; else the surviving rays' indices, with their rows / transmittances kept compact
    rbs, T, dns = rb, torch.ones(N, device=dev), dn
    raw, evaluated = None, 0
    # G: one group size, or a schedule of sizes (the last one repeats) -- e.g. (96, 16): hardly a ray ends within its first 96 sorted
    # samples (the coarse positions in front of its first surface + the front half of the 128 importance samples drawn around it), so
    # those go in one piece and the tail in fine steps (measured on the fitted street scene: 1.23x vs 1.19x for uniform groups of 48)
    sizes = [int(G)] if isinstance(G, (int, float)) else [int(x) for x in G]
    bounds, k = [0], 0
    while bounds[-1] < S:
        bounds.append(min(S, bounds[-1] + sizes[min(k, len(sizes) - 1)]))
        k += 1
    if z_all.is_cuda:
        # the HIP form: positions of the compacted rows, then ONE step (scatter + transmittance + compaction, snerf_classic_ert_step) and
        # one count read-back per group -- no index_select / masked copies of the ray rows
        z_all = z_all.contiguous()
        scratch = (torch.empty(N, dtype=torch.int32, device=dev), torch.empty(N, dtype=torch.int32, device=dev),
                   torch.empty(N, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int64, device=dev))
        for g0, g1 in zip(bounds[:-1], bounds[1:]):
            n = N if alive is None else alive.numel()
            if n == 0:
                break
            pts, vd = ops.classic_ert_points(rb, z_all, alive, g0, g1 - g0)
            rg = network_query_fn(pts, vd if viewdirs is not None else None, run_fn)
            if raw is None:
                raw = torch.zeros(N, S, rg.shape[-1], dtype=torch.float32, device=dev)
            evaluated += n * (g1 - g0)
            alive = ops.classic_ert_step(rg.float().contiguous(), alive, z_all, rb, g0, g1 - g0, eps_t, T, raw, scratch)
        return raw, evaluated
    for g0, g1 in zip(bounds[:-1], bounds[1:]):
        if rbs.shape[0] == 0:
            break
        zr = (z_all[:, g0:g1] if alive is None else z_all[alive, g0:g1]).contiguous()
        pts = ops.classic_points(rbs, zr)                                              # render.py:354 on the surviving rows
        rg = network_query_fn(pts, None if viewdirs is None else rbs[:, -3:], run_fn)
        if raw is None:
            raw = torch.zeros(N, S, rg.shape[-1], dtype=rg.dtype, device=dev)
        if alive is None:
            raw[:, g0:g1] = rg
        else:
            raw[alive, g0:g1] = rg
        evaluated += int(rg.shape[0]) * (g1 - g0)
        if g1 < S:
            znext = z_all[:, g0 + 1:g1 + 1] if alive is None else z_all[alive, g0 + 1:g1 + 1]
            T = T * torch.exp(-(torch.relu(rg[..., 3]) * ((znext - zr) * dns[:, None])).sum(-1))       # x prod (1 - alpha) over the group
            keep = T > eps_t
            if int(keep.sum()) < keep.numel():                                         # (the one device->host read per group)
                alive = torch.nonzero(keep).reshape(-1) if alive is None else alive[keep]
                rbs, T, dns = rbs[keep], T[keep], dns[keep]
    return raw, evaluated


def render_rays(ray_batch, network_fn, network_query_fn, N_samples, retraw=False, lindisp=False, perturb=0.,
                N_importance=0, network_fine=None, white_bkgd=False, raw_noise_std=0., verbose=False, pytest=False,
                t_rand: Optional[torch.Tensor] = None, u: Optional[torch.Tensor] = None, return_inds: bool = False, ert=None):
    """Volumetric rendering of a ray batch [N, 8|11] = [o3, d3, near, far, (viewdir3)]; returns the
    reference's dict (render.py:394-401).  Extra keyword-only inputs `t_rand` [N,S] / `u` [N,Nimp]
    replace the internal torch.rand draws (parity tests).  `ert=(eps_t, G | (G0, G1, ...))` (inference only, under torch.no_grad(); NOT in the
    reference): early ray termination + row compaction in the fine pass -- see _fine_pass_front_to_back."""
    N_rays = ray_batch.shape[0]
    if N_rays == 0:   # as the reference (run_network's torch.cat of no chunks): an error, not an empty dict
        raise ValueError("render_rays: empty ray batch")
    if ert is not None:           # validated at entry, whatever branch runs below
        if torch.is_grad_enabled() or raw_noise_std > 0.:
            raise NotImplementedError("ert is an inference mode: call under torch.no_grad() with raw_noise_std = 0")
        if not (float(ert[0]) < 1.0 and min([ert[1]] if isinstance(ert[1], (int, float)) else list(ert[1])) >= 1):
            raise ValueError("ert = (eps_t < 1, G >= 1 or a schedule of group sizes >= 1)")
        if N_importance <= 0:
            raise ValueError("ert terminates rays in the FINE pass (render.py:380-389): it needs N_importance > 0")
    ert_evals = None
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
    if network_fn is not None:
        raw = network_query_fn(pts, viewdirs, network_fn)
    elif getattr(network_fine, "alpha_model", None) is not None:
        raw = network_query_fn(pts, viewdirs, network_fine.alpha_model)        # render.py:364-366: coarse pass by the alpha model
    else:
        raw = network_query_fn(pts, viewdirs, network_fine)                    # render.py:368-369
    rgb_map, disp_map, acc_map, weights, depth_map = raw2outputs(raw, z_vals, rays_d, raw_noise_std, white_bkgd, pytest=pytest)
    inds = None
    if N_importance > 0:
        rgb_map_0, disp_map_0, acc_map_0 = rgb_map, disp_map, acc_map
        z_samples, inds, z_std = _sample_pdf_z(z_vals, weights, N_importance, perturb == 0., pytest, u, return_inds)
        z_all = ops.classic_merge_sort(z_vals, z_samples)
        run_fn = network_fn if network_fine is None else network_fine
        if ert is not None:
            raw, ert_evals = _fine_pass_front_to_back(rb, z_all, viewdirs, run_fn, network_query_fn, float(ert[0]), ert[1])
        else:
            pts = ops.classic_points(rb, z_all)
            raw = network_query_fn(pts, viewdirs, run_fn)
        rgb_map, disp_map, acc_map, weights_, depth_map = raw2outputs(raw, z_all, rays_d, raw_noise_std, white_bkgd, pytest=pytest)
    ret = {'rgb_map': rgb_map, 'disp_map': disp_map, 'acc_map': acc_map, 'depth_map': depth_map, 'z_vals_map': z_vals,
           'weights': weights}
    if retraw:
        ret['raw'] = raw
    if ert_evals is not None:     # [1, 2] = (fine-network evaluations done, evaluations of the plain render); batchify_rays concatenates the chunks' rows
        ret['ert_evals'] = torch.tensor([[ert_evals, N_rays * int(z_all.shape[1])]], dtype=torch.int64)
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


# ------------------------------------------------------------------------------------------------------------------
# Driver functions of the classic path
# ------------------------------------------------------------------------------------------------------------------
def _device_of(*ts):
    """device of the first tensor argument; host inputs (numpy poses) land on the current GPU"""
    for t in ts:
        if torch.is_tensor(t):
            return t.device
    return torch.device("cuda")


def get_rays(H, W, focal, c2w, ori_points=None):
    """Pinhole rays of an H x W frame (run_nerf_helpers.py:247-258) -> rays_o, rays_d [H,W,3] on the GPU.  `c2w` [3,4] (or
    [4,4]): tensor (any device) or array; `ori_points` = principal point (x, y), default the image centre."""
    cx, cy = (W * 0.5, H * 0.5) if not ori_points else (ori_points[0], ori_points[1])
    return ops.classic_get_rays(H, W, float(focal), c2w, float(cx), float(cy), _device_of(c2w))


def ndc_rays(H, W, focal, near, rays_o, rays_d):
    """NDC warp of forward-facing rays (run_nerf_helpers.py:314-332)."""
    sh = rays_d.shape
    o, d = ops.classic_ndc_rays(H, W, float(focal), float(near), rays_o.float().expand(sh).reshape(-1, 3).contiguous(),
                                rays_d.float().reshape(-1, 3).contiguous())
    return o.reshape(sh), d.reshape(sh)


def render(H, W, focal, chunk=1024 * 32, rays=None, c2w=None, ndc=True, near=0., far=1., use_viewdirs=False, c2w_staticcam=None,
           depths=None, ori_points=None, **kwargs):
    """render() of the reference (render.py:22-91): build the ray batch -- from `c2w` (whole frame) or the given `rays` =
    (rays_o, rays_d) -- and render it in chunks.  -> [rgb_map, disp_map, acc_map, depth_map, extras] shaped like the ray grid.
    The batch assembly (pinhole rays, unit view directions, static camera, NDC warp, row layout) is one kernel launch."""
    if c2w is not None:
        n, sh = H * W, (H, W, 3)
        cx, cy = (W * 0.5, H * 0.5) if not ori_points else (ori_points[0], ori_points[1])
        ro = rd = None
        dev = _device_of(c2w, depths)
    else:
        rays_o, rays_d = rays
        sh = tuple(rays_d.shape)
        dev = _device_of(rays_d, rays_o)
        rd = rays_d.to(dev, torch.float32).reshape(-1, 3).contiguous()
        ro = rays_o.to(dev, torch.float32).expand(sh).reshape(-1, 3).contiguous()
        n, cx, cy = rd.shape[0], W * 0.5, H * 0.5
        if c2w_staticcam is not None and n != H * W:
            raise ValueError("c2w_staticcam needs the rays of the whole H x W frame")
    dep = None if depths is None else depths.to(dev, torch.float32).reshape(-1).contiguous()
    per_ray = torch.is_tensor(near) or torch.is_tensor(far)
    rows = ops.classic_ray_batch(H, W, float(focal), float(cx), float(cy), c2w, c2w_staticcam if use_viewdirs else None, ro, rd, n,
                                 bool(ndc), 0.0 if per_ray else float(near), 1.0 if per_ray else float(far), dep, bool(use_viewdirs), dev)
    if per_ray:      # near * ones_like(rays_d[..., :1]) (render.py:74): anything that broadcasts against [N, 1]
        ones = torch.ones(n, 1, dtype=torch.float32, device=dev)
        rows[:, 6:7] = torch.as_tensor(near, dtype=torch.float32, device=dev) * ones
        rows[:, 7:8] = torch.as_tensor(far, dtype=torch.float32, device=dev) * ones
    all_ret = batchify_rays(rows, chunk, **kwargs)
    for k in all_ret:
        all_ret[k] = torch.reshape(all_ret[k], list(sh[:-1]) + list(all_ret[k].shape[1:]))
    k_extract = ['rgb_map', 'disp_map', 'acc_map', 'depth_map']
    return [all_ret[k] for k in k_extract] + [{k: all_ret[k] for k in all_ret if k not in k_extract}]


to8b = lambda x: (255 * np.clip(x, 0, 1)).astype(np.uint8)


def render_path(render_poses, hwf, chunk, render_kwargs, gt_imgs=None, savedir=None, render_factor=0, ori_points=None, render_masks=None):
    """Render a camera path (render.py:94-135) -> (rgbs [P,H,W,3], disps [P,H,W]) as numpy arrays."""
    H, W, focal = hwf
    H, W = int(H), int(W)
    focal = float(np.array(focal).mean())
    if render_factor != 0:
        H, W, focal = H // render_factor, W // render_factor, focal / render_factor
    rgbs, disps = [], []
    for i, c2w in enumerate(render_poses):
        extra = dict(ori_points=ori_points[i]) if ori_points else {}
        rgb, disp, acc, depth, extras = render(H, W, focal, chunk=chunk, c2w=c2w[:3, :4], retraw=True, **extra, **render_kwargs)
        if render_masks:
            rgb[render_masks[i]] = 0
        rgbs.append(rgb.cpu().numpy())
        disps.append(disp.cpu().numpy())
        # (`savedir`: the reference converts the frame to 8 bits here and writes nothing, render.py:129-132 -- its imageio call is commented out)
    return np.stack(rgbs, 0), np.stack(disps, 0)


def create_nerf(args, compute: str = "bf16", device="cuda"):
    """Instantiate the classic models, their optimiser and the render kwargs (render.py:165-278); `args` carries the reference's
    flags (multires, multires_views, i_embed, use_viewdirs, N_importance, netdepth(_fine), netwidth(_fine), alpha_model_path,
    no_coarse, weighted_loss, netchunk, lrate, basedir, expname, ft_path, no_reload, perturb, N_samples, white_bkgd, raw_noise_std,
    dataset_type, no_ndc, lindisp).  -> (render_kwargs_train, render_kwargs_test, start, grad_vars, optimizer, model_confidence)"""
    embed_fn, input_ch = get_embedder(args.multires, args.i_embed)
    input_ch_views, embeddirs_fn, model_confidence = 0, None, None
    if args.use_viewdirs:
        embeddirs_fn, input_ch_views = get_embedder(args.multires_views, args.i_embed)
    output_ch = 5 if args.N_importance > 0 else 4
    skips = [4]
    mk = lambda cls, D, W, **kw: cls(D=D, W=W, input_ch=input_ch, output_ch=output_ch, skips=skips, input_ch_views=input_ch_views,
                                     use_viewdirs=args.use_viewdirs, compute=compute, device=device, **kw)
    alpha_model = None
    if getattr(args, "alpha_model_path", None) is None:
        model = mk(NeRF, args.netdepth, args.netwidth)
        grad_vars = list(model.parameters())
    else:
        alpha_model = mk(NeRF, args.netdepth_fine, args.netwidth_fine)
        ckpt = torch.load(args.alpha_model_path, map_location="cpu")
        alpha_model.load_state_dict(strip_module_prefix(ckpt['network_fine_state_dict']))
        if not args.no_coarse:
            model = mk(NeRF_RGB, args.netdepth, args.netwidth, alpha_model=alpha_model)
            grad_vars = list(model.parameters())
        else:
            model, grad_vars = None, []
    model_fine = None
    if args.N_importance > 0:
        if alpha_model is None:
            model_fine = mk(NeRF, args.netdepth_fine, args.netwidth_fine)
        else:
            model_fine = mk(NeRF_RGB, args.netdepth_fine, args.netwidth_fine, alpha_model=alpha_model)
        grad_vars += list(model_fine.parameters())
    if getattr(args, "weighted_loss", False):
        raise NotImplementedError("weighted_loss: the reference instantiates DepthConfNet here, a class it never defines (NameError, render.py:207)")
    network_query_fn = make_network_query_fn(embed_fn, embeddirs_fn, args.netchunk)
    optimizer = torch.optim.Adam(params=grad_vars, lr=args.lrate, betas=(0.9, 0.999))
    start = 0
    if getattr(args, "ft_path", None) is not None and args.ft_path != 'None':
        ckpts = [args.ft_path]
    else:
        exp = os.path.join(args.basedir, args.expname)
        ckpts = [os.path.join(exp, f) for f in sorted(os.listdir(exp)) if 'tar' in f] if os.path.isdir(exp) else []
    if len(ckpts) > 0 and not args.no_reload:
        ckpt = torch.load(ckpts[-1], map_location="cpu")
        start = ckpt['global_step']
        optimizer.load_state_dict(ckpt['optimizer_state_dict'])
        model.load_state_dict(strip_module_prefix(ckpt['network_fn_state_dict']))
        if model_fine is not None:
            model_fine.load_state_dict(strip_module_prefix(ckpt['network_fine_state_dict']))
    render_kwargs_train = {'network_query_fn': network_query_fn, 'perturb': args.perturb, 'N_importance': args.N_importance,
                           'network_fine': model_fine, 'N_samples': args.N_samples, 'network_fn': model,
                           'use_viewdirs': args.use_viewdirs, 'white_bkgd': args.white_bkgd, 'raw_noise_std': args.raw_noise_std}
    if args.dataset_type != 'llff' or args.no_ndc:          # NDC only for LLFF-style forward-facing data
        render_kwargs_train['ndc'] = False
        render_kwargs_train['lindisp'] = args.lindisp
    else:
        render_kwargs_train['ndc'] = True
    render_kwargs_test = dict(render_kwargs_train)
    render_kwargs_test['perturb'] = False
    render_kwargs_test['raw_noise_std'] = 0.
    return render_kwargs_train, render_kwargs_test, start, grad_vars, optimizer, model_confidence


def strip_module_prefix(state_dict):
    """Checkpoints saved from an nn.DataParallel / DistributedDataParallel wrapper carry a ``module.`` prefix on every key
    (s-nerf/utils/device_utils.py:33,38; s-nerf/eval.py:72-74 loads them into a wrapped model).  One process drives one GPU here, so
    the modules are never wrapped: accept both spellings."""
    if state_dict and all(k.startswith("module.") for k in state_dict):
        return type(state_dict)((k[len("module."):], v) for k, v in state_dict.items())
    return state_dict
