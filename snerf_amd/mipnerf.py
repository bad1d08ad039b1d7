"""Drop-in ``MipNerfModel`` (path A, the renderer s-nerf/train.py and eval.py
actually run) backed by libsnerf_hip.so.

Mirrors s-nerf/model/models.py:
  MipNerfModel   :10-187   same constructor keywords, same ``state_dict`` keys/shapes
                           (``mlp.layers.N.layers.0.weight`` ... ``proposal.density_layer.bias``),
                           same forward signature and return layout
  make_mipnerf   :190-197
  render_image   :328-360

Only the configuration the reference can actually run is accelerated (SURVEY.md
section 0): warp sampling (``no_warp_sample=0``; the other branch raises NameError
in the reference, models.py:82/178), contraction ``fn=1`` with radius 3, two levels,
view directions on (optionally with the per-image appearance embedding and the semantic head).  Anything else
raises NotImplementedError -- there is no eager fallback.

Per level the whole chain  sample -> encode -> MLP -> activations -> composite  runs
as HIP kernels; one ``torch.autograd.Function`` spans both levels so that
``loss.backward()`` in an unmodified training loop reaches the parameters.
"""
from collections import namedtuple

import torch
from torch import nn

from . import ops
from .classic import _ArenaModule, _dt
from .mlp import MipNerfNet, MipProposalNet

Rays = namedtuple("Rays", ("origins", "directions", "viewdirs", "radii", "lossmult", "near", "far", "app"))
EPS32 = float(torch.finfo(torch.float32).eps)


class MipNerfModel(_ArenaModule):
    def __init__(self, n_samples: int = 128, n_levels: int = 2, resample_padding: float = 0.01, stop_level_grad: bool = True,
                 use_viewdirs: bool = True, lindisp: bool = False, ray_shape: str = "cylinder", min_deg_point: int = 0,
                 max_deg_point: int = 16, deg_view: int = 4, density_noise: float = 1., density_bias: float = -1.,
                 rgb_padding: float = 0.001, disable_integration: bool = False, no_warp_sample=True, fn=None, radius=None,
                 real=False, transform_idx=0, rgb_layer=1, hidden_layer=256, encode_appearance=False, N_vocab=100,
                 proposal_hidden_layer=256, proposal_loss=False, N_fine=128, semantic=False, semantic_class_num=0,
                 compute: str = "bf16", device="cuda", variant: int = 8):
        super().__init__()
        if no_warp_sample:
            raise NotImplementedError("no_warp_sample=1 is broken in the reference itself (models.py:82 vs :178); only the warp branch exists")
        if not use_viewdirs:
            raise NotImplementedError("use_viewdirs=False is broken in the reference itself: models.py:164 calls self.mlp(samples_enc) without the `condition` "
                                      "argument MLP.forward requires (models.py:265) and unpacks its three results into two")
        if n_levels != 2 or min_deg_point != 0 or not stop_level_grad:
            raise NotImplementedError("accelerated MipNerfModel: n_levels=2, stop_level_grad, min_deg_point=0 -- the values make_mipnerf (models.py:190-214) "
                                      "and every shipped config construct it with; constructor defaults nothing sets")
        if semantic and not (0 < semantic_class_num <= 32):
            raise NotImplementedError("semantic head: 1..32 classes")
        if fn not in (0, 1):
            raise ValueError("fn: 1 = contraction (the shipped nuScenes config), 0 = view-centred warp (mip.py:367-378)")
        self._viewc = (0.0, 0.0, 0.0)         # fn = 0: the mean camera centre (forward's `viewc` argument / set_viewc)
        self.n_levels, self.n_samples, self.N_fine = n_levels, n_samples, N_fine
        self.disable_integration = bool(disable_integration)          # --disable_integration (arg_parser.py:188; models.py:132-133)
        self.resample_padding, self.ray_shape, self.max_deg_point, self.deg_view = resample_padding, ray_shape, max_deg_point, deg_view
        self.density_noise, self.density_bias, self.rgb_padding = density_noise, density_bias, rgb_padding
        self.transform_idx, self.proposal_loss, self.lindisp = int(transform_idx), proposal_loss, lindisp
        self.radius, self.fn, self.real, self.semantic, self.no_warp_sample, self.use_viewdirs = radius, fn, real, bool(semantic), 0, True
        self.sem_classes = int(semantic_class_num) if semantic else 0
        self.compute = compute
        fd = max_deg_point * 6
        # --encode_appearance (models.py:57,63-64,153-159): a per-image embedding row (48 values, selected by rays.app) appended to the
        # view condition; the table is the parameter `emb.weight`, registered between the two MLPs as in the reference
        self.encode_appearance, self.N_vocab, self.app_dim = bool(encode_appearance), int(N_vocab), 48
        self.view_dim = 3 + 6 * deg_view
        cd = self.view_dim + (self.app_dim if self.encode_appearance else 0)
        shapes = [("mlp." + n, s) for n, s in MipNerfNet.param_shapes(hidden_layer, 8, 4, fd, cd, rgb_layer, 128, self.sem_classes)]
        if self.encode_appearance:
            shapes += [("emb.weight", (self.N_vocab, self.app_dim))]
        shapes += [("proposal." + n, s) for n, s in MipProposalNet.param_shapes(proposal_hidden_layer, 4, fd)]
        self._setup_arena(shapes, torch.device(device))
        dt = _dt(compute)
        self.dt = dt
        bp = compute == "bf16x3_fwd"          # three-pass split-bf16 forward (renders inside the fp32 contract), one-pass bf16 backward
        self.nerf = MipNerfNet(self.arena, "mlp.", dt, hidden_layer, 8, 4, fd, cd, rgb_layer, 128, variant, semantic_classes=self.sem_classes, bwd_plain=bp)
        self.prop = MipProposalNet(self.arena, "proposal.", dt, proposal_hidden_layer, 4, fd, variant, bwd_plain=bp)
        self.nerf.version_fn = self._param_version
        self.prop.version_fn = self._param_version
        with torch.no_grad():  # DenseBlock / heads: xavier-uniform weights (models.py:208,256-257), default-Linear biases
            for n in self.arena.names:
                p = self.arena.p[n]
                if n == "emb.weight":
                    nn.init.normal_(p)                    # torch.nn.Embedding's default
                elif n.endswith(".weight"):
                    nn.init.xavier_uniform_(p if p.dim() == 2 else p.view(1, -1))
                else:
                    fan_in = self.arena.p[n[:-4] + "weight"].shape[-1]
                    nn.init.uniform_(p, -1.0 / fan_in ** 0.5, 1.0 / fan_in ** 0.5)

    def _const(self, key, make, dev):
        """host-computed constants (torch.linspace on the CPU, as the reference and the oracle evaluate it), uploaded once: no
        host->device copy inside the step (which also keeps the step capturable in a hipGraph)"""
        c = self.__dict__.setdefault("_consts", {})
        k = (key, str(dev), self.n_samples, self.N_fine)
        if k not in c:
            c[k] = make().to(dev)
        return c[k]

    def set_viewc(self, viewc):
        """the centre of the fn = 0 warp (train.py:36 / eval.py:50: mean of the camera positions), a number or 3 values; kept as host
        floats (kernel arguments), converted once per distinct object"""
        ver = viewc._version if torch.is_tensor(viewc) else None              # (an in-place update of the tensor is a new value)
        if getattr(self, "_viewc_src", None) is viewc and getattr(self, "_viewc_ver", None) == ver and torch.is_tensor(viewc):
            return
        v = torch.as_tensor(viewc, dtype=torch.float32).detach().reshape(-1).cpu()
        v = v.expand(3) if v.numel() == 1 else v
        if v.numel() != 3:
            raise ValueError("viewc: a scalar or 3 values")
        self._viewc, self._viewc_src, self._viewc_ver = tuple(float(x) for x in v), viewc, ver
        self._viewc_given = True

    # ------------------------------------------------------------------ core ----
    def _run(self, rays: Rays, keep: bool, white_bg: bool, s_rand, u, noise0, noise1):
        """Both levels.  Returns (outs, ctx) with outs = (dist0, acc0, s0, w0, rgb1, dist1, acc1, s1, w1)."""
        dev = self.arena.flat.device
        f = lambda t: t.detach().to(dev, torch.float32).contiguous()
        o, d, vd = f(rays.origins), f(rays.directions), f(rays.viewdirs)
        radii, near, far = f(rays.radii).reshape(-1), f(rays.near).reshape(-1), f(rays.far).reshape(-1)
        n = o.shape[0]
        S0, P1 = self.n_samples, self.N_fine
        S1 = P1 - 1
        cone = (1 if self.ray_shape == "cone" else 0) | (2 if self.disable_integration else 0)      # (bit 1: the encoders zero the covariances)
        if self.ray_shape not in ("cone", "cylinder"):
            raise ValueError(self.ray_shape)
        # ---- level 0: stratified s, encode, proposal MLP, composite
        base = self._const("base", lambda: torch.linspace(0., 1., S0 + 1), dev)
        s0 = ops.stratified(base, s_rand, None, None, n, 1)
        E0 = self.prop.buf(n * S0, self.prop.Ew)
        # fn = 0: Jacobi_f divides by sqrt(far.max()) of the batch (mip.py:340) -- a device scalar, no host round trip
        warp = None if self.fn == 1 else (self._viewc, far.max().reshape(1))
        ops.mip_encode(s0, o, d, radii, near, far, cone, self.transform_idx, self.max_deg_point, E0, None, self.prop.Ew, self.dt, warp=warp)
        raw_d0, acts0 = self.prop.forward(E0, keep)
        _, dist0, acc0, w0 = ops.mip_composite_fwd(None, raw_d0, noise0, s0, d, near, far, self.transform_idx, white_bg,
                                                   self.rgb_padding, self.density_bias)
        # ---- level 1: resample (no gradient: stop_level_grad), encode, NeRF MLP, composite
        s1, _ = ops.mip_resample(s0, w0, u, self.resample_padding)
        ert = getattr(self, "_ert", None)
        row_index = sample_id = None
        app = f(rays.app).reshape(-1) if self.encode_appearance else None
        if ert is not None and not keep and ert[2] > 0:
            # front-to-back termination on the fine network's own densities (csrc/ert.hip, snerf_ert_f2b_step): groups of ert[2]
            # consecutive samples; after every group the rays whose transmittance fell below eps_t leave
            G = int(ert[2])
            state = ops.ert_f2b_state(n, S1, G, dev)
            row_index = state[3]
            parts, prev_d, prev_base, base, g0, Gp, evaluated = [], None, 0, 0, 0, 0, 0
            for n0 in range(0, S1, G):
                Gn = min(G, S1 - n0)
                ids = ops.ert_f2b_step(prev_d, prev_base, s1, d, near, far, g0, Gp, n0, Gn, self.transform_idx, self.density_bias, ert[0], state, base)
                if ids.shape[0] == 0:
                    break
                rr, rd, rs = self._eval_fine(ids.shape[0], ids, s1, o, d, vd, radii, near, far, cone, app, warp, S1, False)[:3]
                parts.append((rr, rd, rs))
                prev_d, prev_base, g0, Gp = rd, base, n0, Gn
                base += ids.shape[0]
                evaluated += ids.shape[0]
            self.last_ert_rows = (evaluated, n * S1)
            if not parts:       # (unreachable with eps_t < 1 -- the first group always has every ray -- kept so that the cat below never sees [])
                parts.append(self._eval_fine(0, torch.zeros(0, dtype=torch.int32, device=dev), s1, o, d, vd, radii, near, far, cone, app, warp, S1, False)[:3])
            raw_rgb = torch.cat([p_[0] for p_ in parts], 0)
            raw_d1 = torch.cat([p_[1] for p_ in parts], 0)
            raw_sem_rows = torch.cat([p_[2] for p_ in parts], 0) if self.semantic else None
            saved1 = None
        else:
            if ert is not None and not keep:
                # wave ballot + prefix sums over the proposal histogram pick the fine samples worth evaluating (csrc/ert.hip)
                row_index, sample_id = ops.ert_compact(s0, w0, s1, ert[0], ert[1])
                self.last_ert_rows = (int(sample_id.shape[0]), n * S1)
            rows = n * S1 if sample_id is None else max(int(sample_id.shape[0]), 1)
            raw_rgb, raw_d1, raw_sem_rows, saved1 = self._eval_fine(rows, sample_id, s1, o, d, vd, radii, near, far, cone, app, warp, S1, keep)
        rgb1, dist1, acc1, w1 = ops.mip_composite_fwd(raw_rgb, raw_d1, noise1, s1, d, near, far, self.transform_idx, white_bg,
                                                      self.rgb_padding, self.density_bias, row_index=row_index)
        sem1 = raw_sem = None
        if self.semantic:
            raw_sem = raw_sem_rows                                       # [rows, C] fp32: semantic = sum_i w_i raw_semantic_i (mip.py:175-176)
            sem1 = ops.semantic_composite_fwd(w1, raw_sem, self.sem_classes, False, row_index=row_index)   # compacted rows under ert
        ctx = None
        if keep:
            # detached aliases of the output tensors: the originals become outputs of the autograd Function
            ctx = dict(o=o, vd=vd, radii=radii, cone=cone, d=d, near=near, far=far, s0=s0.detach(), s1=s1.detach(), raw_d0=raw_d0, acts0=acts0, w0=w0.detach(), dist0=dist0.detach(),
                       raw_rgb=raw_rgb, raw_d1=raw_d1, saved1=saved1, w1=w1.detach(), dist1=dist1.detach(), noise0=noise0, noise1=noise1,
                       white=white_bg, raw_sem=raw_sem, app=app)
        return (dist0, acc0, s0, w0, rgb1, dist1, acc1, s1, w1) + ((sem1,) if self.semantic else ()), ctx

    def _eval_fine(self, rows, sample_id, s1, o, d, vd, radii, near, far, cone, app, warp, S1, keep):
        """IPE + view / appearance condition + NeRF MLP on `rows` fine samples (all n * S1 of them, or the compacted list `sample_id`).
        -> (raw_rgb, raw_density, raw_semantic or None, saved activations)"""
        dev = self.arena.flat.device
        H = self.nerf.H
        SKIP, CB = self.nerf.alloc_inputs(max(rows, 1))
        if sample_id is not None and sample_id.shape[0] == 0:
            SKIP.zero_(); CB.zero_()                       # nothing survives: one dummy row nobody reads
            enc_ids = torch.zeros(1, dtype=torch.int32, device=dev)
        else:
            enc_ids = sample_id
        ops.mip_encode(s1, o, d, radii, near, far, cone, self.transform_idx, self.max_deg_point, self.nerf.cs(SKIP, H), None, self.nerf.Ew, self.dt,
                       sample_id=enc_ids, warp=warp)
        if self.encode_appearance:
            if self.dt in ops.SPLIT_DTS:     # the split layout is written from an fp32 image of the whole condition block
                cond = torch.empty(max(rows, 1), self.nerf.Cw, dtype=torch.float32, device=dev)
                ops.mip_viewenc(vd, S1, self.deg_view, cond, self.nerf.Cw, ops.F32, sample_id=enc_ids)
                ops.app_embed(self.arena.p["emb.weight"], app, S1, cond[:, self.view_dim:], ops.F32, sample_id=enc_ids)
                ops.cast_pad(cond, self.nerf.Cw, self.nerf.cs(CB, H), self.nerf.Cw, self.dt)
            else:
                ops.mip_viewenc(vd, S1, self.deg_view, CB[:, H:], self.nerf.Cw, self.dt, sample_id=enc_ids)
                ops.app_embed(self.arena.p["emb.weight"], app, S1, CB[:, H + self.view_dim:], self.dt, sample_id=enc_ids)
        else:
            ops.mip_viewenc(vd, S1, self.deg_view, self.nerf.cs(CB, H), self.nerf.Cw, self.dt, sample_id=enc_ids)
        raw_rgb, raw_d1, saved1 = self.nerf.forward(SKIP, CB, keep)
        return raw_rgb, raw_d1, (self.nerf.raw_sem if self.semantic else None), saved1

    def _backward(self, ctx, g_dist0, g_acc0, g_w0, g_rgb1, g_dist1, g_acc1, g_w1, g_sem1=None, on_done=None, ray_grads=False):
        """Accumulates parameter gradients into the arena.  `on_done(prefix or [prefixes])` is called as soon as the gradients of the
        parameters whose names start with a prefix are final -- the NeRF MLP's heads, then each of its trunk layers as the backward
        reaches it, then the proposal network -- and the trainer starts that block's all-reduce while the rest of the backward runs.
        `ray_grads`: also return d loss / d (origins, directions, viewdirs) [n,3] each -- the reference's pose refinement
        (utils/sample_utils.py:410-435) back-propagates through the encoders into the camera pose: the data gradient is carried one GEMM
        further to the IPE / view encodings, then through integrated_pos_enc, the contraction and its Jacobian, lift_gaussian and the
        interval lengths (t1 - t0)|d| of the compositing.  Fence posts carry no ray gradient (level 1 is detached, mip.py:318)."""
        c = ctx
        warp = None if self.fn == 1 else (self._viewc, c["far"].max().reshape(1))      # (the forward's warp argument, _run)
        g_o = g_d = g_vd = None
        if ray_grads:
            g_o, g_d, g_vd = (torch.zeros_like(c["o"]) for _ in range(3))
        n = c["s0"].shape[0]
        S0, S1 = c["s0"].shape[1] - 1, c["s1"].shape[1] - 1
        dev = c["s0"].device
        cc = lambda t: None if t is None else t.contiguous().float()
        d_raw_sem = None
        if g_sem1 is not None and c.get("raw_sem") is not None:
            # semantic = sum_i w_i raw_i: d raw = w g, and the weights get sum_c g_c raw_ic on top of their other gradients
            d_raw_sem = torch.empty_like(c["raw_sem"])
            gw_sem = ops.semantic_composite_bwd(c["w1"], c["raw_sem"], cc(g_sem1), self.sem_classes, False, d_raw_sem, want_g_w=True)
            g_w1 = gw_sem if g_w1 is None else cc(g_w1) + gw_sem
        if any(t is not None for t in (g_rgb1, g_dist1, g_acc1, g_w1)):
            d_rgb = torch.empty(n * S1, 3, dtype=torch.float32, device=dev)
            d_den = torch.empty(n * S1, 1, dtype=torch.float32, device=dev)
            gdir = torch.empty_like(c["d"]) if ray_grads else None
            ops.mip_composite_bwd(c["raw_rgb"], c["raw_d1"], c["noise1"], c["s1"], c["d"], c["near"], c["far"], self.transform_idx,
                                  c["white"], self.rgb_padding, self.density_bias, c["w1"], c["dist1"], cc(g_rgb1), cc(g_dist1),
                                  cc(g_acc1), cc(g_w1), d_rgb, d_den, g_dirs=gdir)
            layer_done = None if on_done is None else (lambda names: on_done(["mlp." + n for n in names]))
            ig = self.nerf.backward(d_rgb, d_den, c["saved1"], d_raw_sem, want_input_grad=ray_grads, want_cond_grad=self.encode_appearance,
                                    on_done=layer_done)
            if self.encode_appearance:     # d loss / d emb.weight from the condition block's gradient (its columns right of the view encoding)
                dVc = ig[1] if ray_grads else ig
                ops.app_embed_bwd(dVc[:, self.view_dim:], c["app"], S1, self.arena.g["emb.weight"], deterministic=getattr(self, "_deterministic", False))
            if ray_grads:
                dE, dV = ig
                eo, ed = ops.mip_encode_bwd(c["s1"], c["o"], c["d"], c["radii"], c["near"], c["far"], c["cone"], self.transform_idx, self.max_deg_point, dE, warp=warp)
                g_o += eo; g_d += ed + gdir
                g_vd += ops.mip_viewenc_bwd(c["vd"], S1, self.deg_view, dV)
        if on_done is not None:
            on_done("mlp.")           # (whatever of the block was not announced layer by layer; a no-op otherwise)
        if any(t is not None for t in (g_dist0, g_acc0, g_w0)):
            d_den0 = torch.empty(n * S0, 1, dtype=torch.float32, device=dev)
            gdir = torch.empty_like(c["d"]) if ray_grads else None
            ops.mip_composite_bwd(None, c["raw_d0"], c["noise0"], c["s0"], c["d"], c["near"], c["far"], self.transform_idx,
                                  c["white"], self.rgb_padding, self.density_bias, c["w0"], c["dist0"], None, cc(g_dist0),
                                  cc(g_acc0), cc(g_w0), None, d_den0, g_dirs=gdir)
            dE0 = self.prop.backward(d_den0, c["acts0"], want_input_grad=ray_grads)
            if ray_grads:
                eo, ed = ops.mip_encode_bwd(c["s0"], c["o"], c["d"], c["radii"], c["near"], c["far"], c["cone"], self.transform_idx, self.max_deg_point, dE0, warp=warp)
                g_o += eo; g_d += ed + gdir
        if on_done is not None:
            on_done("proposal.")
        return (g_o, g_d, g_vd) if ray_grads else None

    def _draws(self, n, randomized, dev):
        """The reference's three torch RNG draws (mip.py:283, math_ops.py:52, models.py:163-165), taken in its order."""
        s_rand = torch.rand(n, self.n_samples + 1, device=dev) if randomized else None
        noise0 = noise1 = None
        if randomized and self.density_noise > 0:
            noise0 = self.density_noise * torch.randn(n, self.n_samples, device=dev)
        if randomized:
            P1 = self.N_fine
            s = 1 / P1
            jit = torch.empty(n, P1, device=dev).uniform_(to=s - EPS32)
            u = ops.jitter_u(jit, s)               # = torch.minimum(torch.arange(P1) * s + jit, torch.ones_like(jit) - EPS32), bit for bit
            if self.density_noise > 0:
                noise1 = self.density_noise * torch.randn(n, P1 - 1, device=dev)
        else:
            u = self._const("u_det", lambda: torch.linspace(0., 1. - EPS32, self.N_fine), dev)
        return s_rand, u, noise0, noise1

    # ---------------------------------------------------------------- public ----
    def forward(self, rays, randomized, white_bg, viewc=0., s_rand=None, u=None, ert=None):
        """-> [[None, distance, acc(, s_vals, weights)], [rgb, distance, acc, None(, s_vals, weights)]]
        (models.py:178-187).  `s_rand` / `u` override the internal draws (parity tests).
        `ert=(eps_t, eps_w)` (inference only; NOT in the reference): early ray termination + sample compaction -- fine samples whose
        proposal-predicted transmittance is <= eps_t or whose proposal-predicted weight is <= eps_w are not evaluated.
        `ert=(eps_t, eps_w, G)`: the fine level front to back in groups of G samples; a ray stops once its transmittance, from the fine
        network's own densities, is <= eps_t -- the skipped samples' weights then sum to <= eps_t (an exact bound on acc / rgb); eps_w is
        not used in this mode, and every group costs one device->host read of the survivor count (small G = sync-bound)."""
        if white_bg:
            raise NotImplementedError("white_bg=True crashes the reference at the proposal level (mip.py:188, rgb is None)")
        self._check_arena()
        # pose refinement (sample_utils.py:410-435): origins / directions / viewdirs may carry gradients back to a camera pose
        ray_grad = torch.is_grad_enabled() and any(torch.is_tensor(r) and r.requires_grad for r in (rays.origins, rays.directions, rays.viewdirs))
        if torch.is_grad_enabled() and any(torch.is_tensor(r) and r.requires_grad for r in (rays.radii, rays.near, rays.far)):
            raise NotImplementedError("gradients w.r.t. radii / near / far are not propagated (the reference's pose refinement leaves them constant)")
        if self.fn == 0:
            self.set_viewc(viewc)
        dev = self.arena.flat.device
        n = rays.origins.shape[0]
        if n == 0:    # the reference raises on an empty batch too (models.py: reshape of 0 elements with an inferred dimension)
            raise RuntimeError("MipNerfModel.forward: empty ray batch")
        ds_rand, du, noise0, noise1 = self._draws(n, randomized, dev)
        if s_rand is None:
            s_rand = ds_rand
        if u is None:
            u = du
        s_rand = None if s_rand is None else s_rand.to(dev).float().contiguous()
        u = u.to(dev).float().contiguous()
        params = self.param_list()
        keep = torch.is_grad_enabled() and (ray_grad or any(p.requires_grad for p in params))   # grad mode is off inside Function.forward
        if ert is not None and keep:
            raise NotImplementedError("ert (sample compaction) is an inference mode: call under torch.no_grad()")
        # ert = (eps_t, eps_w): selection from the proposal histogram; ert = (eps_t, eps_w, G): front to back in groups of G fine samples,
        # rays leave when their transmittance (the fine network's own densities) falls to eps_t (eps_w unused: the bound is exact)
        self._ert = None if ert is None else (float(ert[0]), float(ert[1]), int(ert[2]) if len(ert) > 2 else 0)
        if self._ert is not None and not (self._ert[0] < 1.0 and self._ert[2] >= 0):     # (NaN fails the comparison too)
            self._ert = None
            raise ValueError("ert: eps_t < 1 (a transmittance; >= 1 would terminate every ray before its first sample; < 0 = never), G >= 0")
        rt = (rays.origins, rays.directions, rays.viewdirs) if ray_grad else (None, None, None)
        outs = _MipFn.apply(self, rays, bool(white_bg), s_rand, u, noise0, noise1, keep, *rt, *params)
        self._ert = None
        dist0, acc0, s0, w0, rgb1, dist1, acc1, s1, w1 = outs[:9]
        ret = [[None, dist0, acc0], [rgb1, dist1, acc1, outs[9] if self.semantic else None]]
        if self.proposal_loss:
            ret[0] += [s0, w0]
            ret[1] += [s1, w1]
        return ret


class _MipFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, rays, white_bg, s_rand, u, noise0, noise1, keep, ray_o, ray_d, ray_vd, *params):
        ctx.set_materialize_grads(False)
        outs, c = model._run(rays, keep, white_bg, s_rand, u, noise0, noise1)
        ctx.model, ctx.c = model, c
        ctx.ray_meta = None if ray_o is None else [(t.shape, t.dtype, t.device) for t in (ray_o, ray_d, ray_vd)]
        ctx.mark_non_differentiable(outs[2], outs[7])
        return outs

    @staticmethod
    def backward(ctx, g_dist0, g_acc0, g_s0, g_w0, g_rgb1, g_dist1, g_acc1, g_s1, g_w1, g_sem1=None):
        if ctx.c is None:
            raise RuntimeError("MipNerfModel.forward ran without saved activations")
        m = ctx.model
        m.arena.grad.zero_()
        rg = m._backward(ctx.c, g_dist0, g_acc0, g_w0, g_rgb1, g_dist1, g_acc1, g_w1, g_sem1, ray_grads=ctx.ray_meta is not None)
        ctx.c = None
        grads = tuple(m.arena.g[n].clone() for n in m._pnames)
        m.arena.grad.zero_()      # the trainers accumulate into the arena and expect it clean at step start
        if rg is None:
            return (None,) * 11 + grads
        rays_g = tuple(g.reshape(sh).to(device=dev, dtype=dt) for g, (sh, dt, dev) in zip(rg, ctx.ray_meta))
        return (None,) * 8 + rays_g + grads


def make_mipnerf(args, device="cuda", compute="bf16"):
    """models.py:190-197."""
    return MipNerfModel(no_warp_sample=args.no_warp_sample, disable_integration=args.disable_integration, ray_shape=args.ray_shape,
                        fn=args.fn, max_deg_point=args.max_degree, radius=args.radius, transform_idx=args.transform_idx,
                        real=args.real, rgb_layer=args.rgb_layer, hidden_layer=args.hidden_layer, density_noise=args.density_noise,
                        encode_appearance=args.encode_appearance, n_samples=args.N_samples, proposal_loss=args.proposal_loss,
                        N_fine=args.N_fine, semantic=args.semantic, semantic_class_num=args.semantic_class_num,
                        compute=compute, device=device)


def render_image(render_fn, rays, rank=0, chunk=8192, world=1, group=None):
    """Chunked full-frame inference (models.py:328-360; eval.py:146): rays fields [H,W,.] -> (rgb [H,W,3], distance [H,W],
    acc [H,W], semantic [H,W,C] or None).  One process drives one GPU, so the reference's reflect-padding of a chunk to the
    DataParallel device count is not needed.  `world` > 1 (one process per GPU, torch.distributed initialised): rank `rank`
    renders one contiguous block of the frame's rays and ONE all-gather per output buffer assembles the frame on every rank
     -- instead of scattering every chunk across the devices as nn.DataParallel does."""
    height, width = rays[0].shape[:2]
    num_rays = height * width
    flat = Rays(*[r.reshape(num_rays, -1) for r in rays])
    if world > 1 and num_rays < world:
        # a rank with an empty block would skip the all-gathers the others enter (and would not even know the number of output columns)
        raise ValueError(f"render_image: {num_rays} rays cannot be sharded over {world} ranks")
    per, rem = divmod(num_rays, world)
    lo = rank * per + min(rank, rem)
    hi = lo + per + (1 if rank < rem else 0)
    res = []
    for i in range(lo, hi, chunk):
        out = render_fn(Rays(*[r[i:min(i + chunk, hi)] for r in flat]))[-1]
        res.append(out[:4] if len(out) > 3 and out[3] is not None else out[:3])
    cols = [torch.cat(r, 0) for r in zip(*res)]
    if world > 1:
        import torch.distributed as dist
        block = per + (1 if rem else 0)                      # every rank contributes a block of the same size (the tail is padding)
        full = []
        for c in cols:
            c2 = c.reshape(c.shape[0], -1)
            pad = torch.zeros(block, c2.shape[1], dtype=c2.dtype, device=c2.device)
            pad[:c2.shape[0]] = c2
            parts = [torch.empty_like(pad) for _ in range(world)]
            dist.all_gather(parts, pad, group=group)
            full.append(torch.cat([p[:per + (1 if r < rem else 0)] for r, p in enumerate(parts)], 0))
        cols = full
    rgb, dist_, acc = cols[:3]
    sem = cols[3].reshape(height, width, -1) if len(cols) > 3 else None
    return rgb.reshape(height, width, -1), dist_.reshape(height, width), acc.reshape(height, width), sem
