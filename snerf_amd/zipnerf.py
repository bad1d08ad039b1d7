"""Drop-in zipnerf ``Model`` (S-NeRF++ background, path C) backed by libsnerf_hip.so.

Mirrors s-nerfpp/zipnerf/internal/models.py: ``Model`` (:28-349) with the same class attributes, the same
``forward(rand, batch, train_frac, compute_extras, zero_glo, sample_n, sample_m, step, max_step, cal_input_grad)``
signature, the same ``(renderings, ray_history)`` return layout (keys ``rgb``/``depth``/``acc`` and
``sdist``/``weights``/``tdist``/``density``/``rgb``) and the same ``state_dict`` keys / shapes (``nerf_mlp.encoder.embeddings``,
``nerf_mlp.density_layer.0.weight``, ``nerf_mlp.lin_second_stage_1.weight``, ``prop_mlp_0.encoder.offsets`` ...).

``cal_input_grad=True`` (pose refinement, zipnerf/train.py:187-224) back-propagates to ``batch['origins' / 'directions' / 'viewdirs' /
'base_x' / 'base_y']`` when they require grad.

Accelerated branch = what ``configs/waymo.gin`` + class defaults run: ``raydist_fn='power_transformation'``, distinct proposal
MLPs with C = 1 grids (desired resolution 512 / 2048), NeRF grid C = 4 (8192), ``disable_density_normals``, ``deg_view = 1``,
optional GLO vectors and semantic head, no exposure scaling, ``single_jitter``.  Everything else raises NotImplementedError (no eager fallback).

Per level: ONE resample launch (dilation + annealed logits + inverse-CDF intervals + s->t warp), ONE fused featurisation launch
(7-multisample cone casting + contraction + hash-grid gather + erf down-weighting + mean, written into the MLP operand buffer),
the small MLPs on the MFMA GEMMs, ONE compositing launch.  The reference's ``[R,S,7,3]`` means, ``[R*S*7, L*C]`` features and
the ``[R,3S+2,S]`` / ``[R,S+1,n]`` sampling masks never exist.
"""
import numpy as np
import torch
from torch import nn

from . import ops
from .classic import _ArenaModule, _dt
from .mlp import ZipNerfNet, ZipPropNet

EPS32 = float(torch.finfo(torch.float32).eps)


def _level_layout(num_levels, base_resolution, desired_resolution, log2_hashmap_size):
    """gridencoder/grid.py:104-141 (input_dim 3, align_corners False) -> (offsets, grid sizes, per-level scale)"""
    scale = ops.grid_per_level_scale(base_resolution, desired_resolution, num_levels)
    offsets, res = ops.grid_level_layout(3, num_levels, scale, base_resolution, log2_hashmap_size, False)
    return offsets, res, scale


class _Encoder:
    def __init__(self, level_dim, desired_resolution, base_resolution=16, log2_hashmap_size=21, level_interval=2):
        self.L = int(np.log(desired_resolution / base_resolution) / np.log(level_interval)) + 1      # models.py:413
        self.C, self.H = level_dim, base_resolution
        self.offsets, self.res, self.scale = _level_layout(self.L, base_resolution, desired_resolution, log2_hashmap_size)
        self.Sl = float(np.log2(self.scale))
        self.rows = int(self.offsets[-1])
        # leading (small, heavily contended) levels take the LDS-privatised backward, in slabs of <= 36 K cells (144 KB fp32).  Every
        # slab pass walks all intervals; on a DENSE level a slab is a range of z-layers and the walk rejects an interval with one
        # contracted point, so such a level qualifies up to 32 slabs, a hashed one (no spatial order in its rows) up to 8
        sizes = np.diff(self.offsets)
        self.lds_cells = (144 * 1024) // (4 * level_dim)
        self.lds_levels, self.lds_slabs = 0, 0
        while self.lds_levels < self.L:
            l = self.lds_levels
            slabs = -(-int(sizes[l]) // self.lds_cells)
            dense = int(self.res[l]) ** 3 <= int(sizes[l])
            if slabs > (32 if dense else 8):
                break
            self.lds_slabs += slabs
            self.lds_levels += 1


class Model(_ArenaModule):
    num_prop_samples = (64, 64)
    num_nerf_samples: int = 32
    num_levels: int = 3
    bg_intensity_range = (1., 1.)
    anneal_slope: float = 10
    stop_level_grad: bool = True
    use_viewdirs: bool = True
    raydist_fn = 'contract'
    single_jitter: bool = True
    dilation_multiplier: float = 0.5
    dilation_bias: float = 0.0025
    num_glo_features: int = 0
    num_glo_embeddings: int = 1000
    near_anneal_rate = None
    near_anneal_init: float = 0.95
    resample_padding: float = 0.0
    opaque_background: bool = False
    power_lambda: float = -1.5
    std_scale: float = 0.35
    prop_desired_grid_size = [512, 2048]
    distinct_prop: bool = True
    single_mlp: bool = False

    def __init__(self, config=None, compute: str = "bf16", table_dtype: str = "ref", device="cuda", grid_log2_hashmap_size: int = 21,
                 nerf_desired_resolution: int = 8192, init_std: float = 1e-4, table_grad_dtype: str = "auto", use_semantic: bool = False,
                 class_num: int = 19, table_grad_mode: str = "binned", **kwargs):
        super().__init__()
        for k, v in kwargs.items():
            setattr(self, k, v)
        self.config = config
        if self.raydist_fn != 'power_transformation':
            # (the class default 'contract' of the reference, models.py:40, cannot run there either: coord.construct_ray_warps falls
            # through to `fn.__name__` on the string -- AttributeError; configs/waymo.gin binds 'power_transformation')
            raise NotImplementedError("accelerated zipnerf Model: raydist_fn='power_transformation' (configs/waymo.gin)")
        if (self.num_levels != 3 or len(self.num_prop_samples) != 2 or not self.distinct_prop or self.single_mlp
                or not self.stop_level_grad or not self.use_viewdirs or self.bg_intensity_range[0] != self.bg_intensity_range[1]):
            raise NotImplementedError("accelerated zipnerf Model: 2 distinct proposal MLPs + NeRF MLP")
        if getattr(self, "learned_exposure_scaling", False):
            raise NotImplementedError("learned exposure scaling (RawNeRF, configs/llff_raw.gin) is not built")
        # GLO (models.py:44-45, 75-77, 131-139; configs/360_glo*.gin): a learned vector per training image modulates the NeRF MLP's
        # bottleneck; the embedding table exists unless config.zero_glo (then the MLP always sees zeros)
        self.num_glo_features = int(self.num_glo_features)
        self.glo_embedding = self.num_glo_features > 0 and not (config is not None and getattr(config, "zero_glo", False))
        # semantic head (Config.use_semantic -> NerfMLP.use_semantic, models.py:66,297-305,594-597): 19-class softmax of x[..., 1:20]
        self.scattered_rays = False      # set True when inference batches are random pixels rather than image rows (a locality hint only)
        self.use_semantic = bool(use_semantic or (config is not None and getattr(config, "use_semantic", False)))
        self.class_num = int(class_num)
        dev = torch.device(device)
        self.encs = [_Encoder(1, self.prop_desired_grid_size[0], 16, grid_log2_hashmap_size),
                     _Encoder(1, self.prop_desired_grid_size[1], 16, grid_log2_hashmap_size),
                     _Encoder(4, nerf_desired_resolution, 16, grid_log2_hashmap_size)]
        self.names = ["prop_mlp_0.", "prop_mlp_1.", "nerf_mlp."]
        shapes = [("nerf_mlp.encoder.embeddings", (self.encs[2].rows, 4))]
        shapes += [("nerf_mlp." + n, s) for n, s in ZipNerfNet.param_shapes(self.encs[2].L * 4, glo_dim=self.num_glo_features)]
        for i in range(2):
            shapes += [(f"prop_mlp_{i}.encoder.embeddings", (self.encs[i].rows, 1))]
            shapes += [(f"prop_mlp_{i}." + n, s) for n, s in ZipPropNet.param_shapes(self.encs[i].L)]
        if self.glo_embedding:
            shapes += [("glo_vecs.weight", (int(self.num_glo_embeddings), self.num_glo_features))]      # registered last, as in the reference
        self._setup_arena(shapes, dev)
        self.dt = _dt(compute)
        self.compute = compute
        # compute="fp16": fp16 features / activations / gradient buffers and the fp16 MFMA (BASELINE config 4's "fp16 MLP", the tcnn-style
        # half-precision network); gradients flow through the networks multiplied by a static loss scale (here for loss.backward(), in
        # ZipTrainer folded into the Adam launch)
        self.autograd_loss_scale = 4096.0 if self.dt == ops.F16 else 1.0
        # gather tables: "ref" (default) = what the reference does under autocast (gridencoder/grid.py:41-44): embeddings are halved only
        # when C is even, i.e. the C = 4 NeRF table in fp16 and the two C = 1 proposal tables in fp32; "f16" halves all three (narrower
        # storage than the reference's on the proposal levels: an extension, benchmarked separately); "f32" none
        self.table_mode = {"ref": "ref", "f16": "f16", "fp16": "f16", "f32": "f32", "fp32": "f32"}[table_dtype]
        self.table_half = self.table_mode != "f32"        # (any table halved)
        # precision of the table gradient's CONTRIBUTIONS (the sums are exact in every binned mode).  A contribution is (interpolation
        # weight) x (d loss / d feature), and the second factor arrives in the compute dtype: with 16-bit compute it carries 8 (bf16) or 11
        # (fp16) significant bits already.  "auto" (default): fp16 records (10 / 4 instead of 18 / 8 bytes at C = 4 / 1; one rounding to 11
        # bits per contribution) wherever the table is halved -- the reference scatters __half2 atomics there, gridencoder.cu:300-330, every
        # ADDITION rounding to 11 bits -- or the networks compute in 16 bits; fp32 records with fp32 compute on an fp32 table (the parity
        # mode).  "table": by the table's storage only; "f32" / "f16": every level; "bf16": the atomic scatter with packed bf16 pairs
        # (round 1, kept for A/B runs)
        if table_grad_dtype not in ("auto", "table", "f32", "fp32", "f16", "fp16", "bf16"):
            raise ValueError(table_grad_dtype)
        self.table_grad_bf16 = table_grad_dtype == "bf16"
        self.table_grad_records = {"auto": "auto", "table": "table", "f32": "f32", "fp32": "f32", "f16": "f16", "fp16": "f16", "bf16": "f32"}[table_grad_dtype]
        # "binned" (default): contributions are binned by destination and accumulated per bin in LDS with fixed-point integer atomics --
        # no L2 atomics on the hashed levels, fp32-exact sums, BIT-REPRODUCIBLE gradients (csrc/zip.hip, snerf_zip_encode_bwd_binned);
        # "atomic": the reference's scatter (gridencoder.cu:248-340) with fp32 (or packed bf16, table_grad_dtype) global atomics
        if table_grad_mode not in ("binned", "atomic"):
            raise ValueError(table_grad_mode)
        self.table_grad_mode = "atomic" if self.table_grad_bf16 else table_grad_mode
        # binned mode: the training forward's featurisation sweep also counts the gradient's records (snerf_zip_encode_fwd_count) instead
        # of a separate count sweep in the backward; False (or SNERF_ZIP_COUNT_IN_BWD=1, for A/B runs) keeps the round-2 order
        import os
        self.count_in_forward = os.environ.get("SNERF_ZIP_COUNT_IN_BWD", "") == ""
        if self.table_grad_mode == "binned":
            # the binned kernels hold ZB_NBMAX bins per level: a table whose level has more row ranges (2^22 rows at C = 4, 2^24 at
            # C = 1, i.e. grid_log2_hashmap_size >= 23 / 25) does not fit them -- the plan raises, and construction fails HERE rather
            # than in the middle of the first backward pass
            for e in self.encs:
                ops.zip_bin_plan(e.offsets, e.C, 1)
        self.nets = [ZipPropNet(self.arena, "prop_mlp_0.", self.dt, self.encs[0].L), ZipPropNet(self.arena, "prop_mlp_1.", self.dt, self.encs[1].L),
                     ZipNerfNet(self.arena, "nerf_mlp.", self.dt, self.encs[2].L * 4, glo_dim=self.num_glo_features)]
        for n in self.nets:
            n.version_fn = self._param_version
        self._tables, self._tables_version = [None] * 3, -1
        self.dev_offsets, self.dev_sizes = [], []
        for i, pre in enumerate(self.names):
            e = self.encs[i]
            enc_mod = self._modules[pre[:-1]]._modules["encoder"]
            off = torch.from_numpy(e.offsets).to(dev)
            enc_mod.register_buffer("offsets", off)
            idx = torch.empty(e.rows, dtype=torch.long, device=dev)
            for l in range(e.L):
                idx[e.offsets[l]:e.offsets[l + 1]] = l
            enc_mod.register_buffer("idx", idx)
            enc_mod.register_buffer("grid_sizes", torch.from_numpy(e.res).to(dev))
            self.dev_offsets.append(off); self.dev_sizes.append(enc_mod.grid_sizes)
        with torch.no_grad():
            for n in self.arena.names:
                p = self.arena.p[n]
                if n.endswith("embeddings"):
                    p.uniform_(-init_std, init_std)                                  # grid.py:151-153
                elif n == "glo_vecs.weight":
                    p.normal_()                                                      # nn.Embedding
                elif n.endswith(".weight"):
                    if "lin_second_stage" in n:
                        nn.init.kaiming_uniform_(p)                                  # models.py:464
                    else:
                        nn.init.kaiming_uniform_(p if p.dim() == 2 else p.view(1, -1), a=5 ** 0.5)
                else:
                    fan_in = self.arena.p[n[:-4] + "weight"].shape[-1]
                    nn.init.uniform_(p, -1.0 / fan_in ** 0.5, 1.0 / fan_in ** 0.5)

    # --------------------------------------------------------------------------------------------------------------
    def _table(self, lvl):
        """the gather table: fp16 copy of the fp32 master embeddings where `table_dtype` says so (default: like the reference under
        autocast, grid.py:41-44 -- C = 4 halved, the C = 1 tables stay fp32); the gradient always accumulates in fp32"""
        v = self._param_version()
        if v != self._tables_version:
            self._tables, self._tables_version = [None] * 3, v
        if self._tables[lvl] is None:
            emb = self.arena.p[self.names[lvl] + "encoder.embeddings"]
            half = self.table_mode == "f16" or (self.table_mode == "ref" and self.encs[lvl].C % 2 == 0)
            self._tables[lvl] = emb.detach().half().contiguous() if half else emb.detach()
        return self._tables[lvl]

    def _anneal(self, train_frac):
        if self.anneal_slope > 0:
            return (self.anneal_slope * train_frac) / ((self.anneal_slope - 1) * train_frac + 1)   # Schlick bias, models.py:191-194
        return 1.0

    def _run(self, batch, keep, train_frac, draws, sample_n, sample_m):
        dev = self.arena.flat.device
        f = lambda t: t.detach().to(dev, torch.float32).contiguous()
        o, d, vd = f(batch['origins']), f(batch['directions']), f(batch['viewdirs'])
        bx, by = f(batch['base_x']), f(batch['base_y'])
        radii, near, far = f(batch['radii']).reshape(-1), f(batch['near']).reshape(-1), f(batch['far']).reshape(-1)
        R = o.shape[0]
        # near-bound annealing (models.py:47-48, 147-158): the first interval is [clip(1 - train_frac / rate, 0, init), 1]; its length
        # scales the dilation (:170-171) and it is the domain of the dilation / resampling clamps (:182-186, :207-213)
        s_near = 0.0 if self.near_anneal_rate is None else float(min(max(1.0 - train_frac / self.near_anneal_rate, 0.0), self.near_anneal_init))
        sdist = torch.cat([torch.full((R, 1), s_near, device=dev), torch.ones(R, 1, device=dev)], -1)
        weights = torch.ones(R, 1, device=dev)
        anneal = self._anneal(train_frac)
        bg = float(self.bg_intensity_range[0])
        prod = 1
        levels = []
        for lvl in range(3):
            is_prop = lvl < 2
            ns = self.num_prop_samples[lvl] if is_prop else self.num_nerf_samples
            dilation = self.dilation_bias + self.dilation_multiplier * (1.0 - s_near) / prod
            prod *= ns
            use_dilation = (self.dilation_bias > 0 or self.dilation_multiplier > 0) and lvl > 0
            u, degj = draws[lvl]
            sdist, tdist = ops.zip_resample(sdist, weights.detach(), u, ns, near, far, dilation, use_dilation, anneal, self.resample_padding,
                                            self.power_lambda, dom=(s_near, 1.0))
            e, net = self.encs[lvl], self.nets[lvl]
            P = R * ns
            if is_prop and not keep and sample_n <= 8 and e.L <= 16:
                # inference: featurisation and the proposal MLP in one kernel (no feature buffer, no HBM-bound N = 64 / N = 1 GEMMs)
                pre = self.names[lvl]
                raw_d = ops.zip_encode_prop_fwd(tdist, o, d, radii, bx, by, degj, self._table(lvl), self.dev_offsets[lvl], self.dev_sizes[lvl], e.L,
                                                sample_n, sample_m, e.Sl, e.H, self.std_scale, self.arena.p[pre + "density_layer.0.weight"],
                                                self.arena.p[pre + "density_layer.0.bias"], self.arena.p[pre + "density_layer.2.weight"],
                                                self.arena.p[pre + "density_layer.2.bias"], self.dt)
                rgb, depth, acc, weights = ops.zip_composite_fwd(None, raw_d, tdist, d, self.opaque_background, bg, 0.001, -1.0)
                levels.append(dict(sdist=sdist, tdist=tdist, weights=weights, rgb=rgb, depth=depth, acc=acc, raw_rgb=None, raw_d=raw_d, saved=None,
                                   degj=degj, ns=ns, semantic=None, logits=None))
                continue
            fused_nerf = (not is_prop) and not keep and not self.num_glo_features and net.fused_infer_ok() and net.Fw == 64 and self.class_num < 32
            if is_prop or fused_nerf:
                Fb = torch.zeros(P, net.Fw, dtype=net.tdt, device=dev); SB = None
            else:
                Fb, SB = net.alloc(P)
            # inference renders frames (coherent rays: evaluate the multisamples once per interval for all levels); training draws
            # scattered pixels (one thread per level keeps the most gathers in flight)
            precount = None
            if keep and self.table_grad_mode == "binned" and self.count_in_forward and sample_n <= 8 and e.C in (1, 4):
                # training with the binned table gradient: the featurisation sweep is also that gradient's count pass
                ks, _, lrows = ops.zip_bin_plan(e.offsets, e.C, P * sample_n * 8)
                precount = ops.zip_encode_fwd_count(tdist, o, d, radii, bx, by, degj, self._table(lvl), self.dev_offsets[lvl], self.dev_sizes[lvl], Fb,
                                                    e.L, e.C, sample_n, sample_m, e.Sl, e.H, self.std_scale, ks, lrows)
            else:
                ops.zip_encode_fwd(tdist, o, d, radii, bx, by, degj, self._table(lvl), self.dev_offsets[lvl], self.dev_sizes[lvl], Fb, e.L, e.C,
                                   sample_n, sample_m, e.Sl, e.H, self.std_scale,
                                   levels_per_thread=1 if (keep or self.scattered_rays) else e.L)
            if is_prop:
                raw_d, saved = net.forward(Fb, keep)
                raw_rgb = None
            elif fused_nerf:
                # inference: the whole NeRF MLP in one launch, activations in registers (csrc/fmlp.hip, fzip_fwd_kernel)
                Dn = torch.empty(P, 16, dtype=net.tdt, device=dev)
                ops.mip_viewenc(vd, ns, 1, Dn, 16, self.dt)
                raw_rgb, raw_d = net.forward_fused(Fb, Dn, want_x=self.use_semantic)
                saved = cam = None
            else:
                ops.mip_viewenc(vd, ns, 1, SB[:, net.Wd + net.Bw:], net.Dw, self.dt)
                glo = cam = None
                if self.num_glo_features:
                    # the rays' GLO vectors: the embedding row of the ray's camera, or zeros (`zero_glo`, Model.forward's default)
                    glo = torch.zeros(R, net.Gw, dtype=net.tdt, device=dev)
                    if self.glo_embedding and not getattr(self, "_zero_glo", True):
                        cam = batch['cam_idx'].detach().to(dev, torch.float32).reshape(R, -1)[:, 0].contiguous()
                        ops.app_embed(self.arena.p["glo_vecs.weight"], cam, 1, glo, self.dt)
                raw_rgb, raw_d, saved = net.forward(Fb, SB, keep, glo=glo, S=ns)
            rgb, depth, acc, weights = ops.zip_composite_fwd(raw_rgb, raw_d, tdist, d, self.opaque_background, bg, 0.001, -1.0)
            sem = logits = None
            if self.use_semantic and not is_prop:
                logits = net.last_x[:, 1:1 + self.class_num]                       # x[..., 1:1+C] of the density network's output
                sem = ops.semantic_composite_fwd(weights, logits, self.class_num, True)
            levels.append(dict(sdist=sdist, tdist=tdist, weights=weights, rgb=rgb, depth=depth, acc=acc, raw_rgb=raw_rgb, raw_d=raw_d, saved=saved,
                               degj=degj, ns=ns, semantic=sem, logits=logits, cam=None if is_prop else cam, precount=precount))
        ctx = None
        if keep:   # detached aliases: the originals become outputs of the autograd Function (no graph / reference cycle through ctx)
            det = [{k: (v.detach() if torch.is_tensor(v) else v) for k, v in L.items()} for L in levels]
            ctx = dict(o=o, d=d, vd=vd, radii=radii, bx=bx, by=by, levels=det, bg=bg, n=sample_n, m=sample_m)
        return levels, ctx

    def _half_records(self, lvl):
        """whether level lvl's binned table gradient travels as fp16 records (table_grad_dtype)"""
        if self.table_grad_records in ("table", "auto"):
            halved = self.table_mode == "f16" or (self.table_mode == "ref" and self.encs[lvl].C % 2 == 0)
            return halved or (self.table_grad_records == "auto" and self.dt in (ops.BF16, ops.F16))
        return self.table_grad_records == "f16"

    def _backward(self, ctx, grads, on_done=None, ray_grads=False):
        """grads[lvl] = (g_rgb, g_depth, g_acc, g_w); accumulates parameter gradients into the arena.  `on_done(prefix)` is called as
        soon as a level's gradients (MLP + hash table) are final, NeRF level first.
        `ray_grads` (pose refinement, the reference's `cal_input_grad`: zipnerf/train.py:187-197): also returns d loss / d (origins,
        directions, viewdirs, base_x, base_y) [R,3] -- through the hash-grid positions and the erf down-weighting of every level
        (`snerf_zip_encode_ray_bwd`), the view-direction encoding and the interval lengths (t1 - t0)|d| of the compositing.  Fence
        posts carry no gradient (stop_level_grad)."""
        dev = ctx["o"].device
        cc = lambda t: None if t is None else t.contiguous().float()
        rg = [torch.zeros_like(ctx["o"]) for _ in range(5)] if ray_grads else None      # o, d, vd, bx, by
        for lvl in (2, 1, 0):
            g_rgb, g_depth, g_acc, g_w = grads[lvl][:4]
            g_sem = grads[lvl][4] if len(grads[lvl]) > 4 else None
            if all(t is None for t in (g_rgb, g_depth, g_acc, g_w, g_sem)):
                if on_done is not None:
                    on_done(self.names[lvl])
                continue
            L = ctx["levels"][lvl]
            e, net = self.encs[lvl], self.nets[lvl]
            P = L["weights"].numel()
            sem_on = g_sem is not None and L.get("logits") is not None
            # [d raw density | d semantic logits]: the semantic head shares the density network's output (one row of gradients)
            d_dl = torch.zeros(P, 1 + self.class_num, dtype=torch.float32, device=dev) if sem_on else torch.empty(P, 1, dtype=torch.float32, device=dev)
            d_den = d_dl[:, :1]
            d_rgb = torch.empty(P, 3, dtype=torch.float32, device=dev) if L["raw_rgb"] is not None else None
            if all(t is None for t in (g_rgb, g_depth, g_acc, g_w)):
                d_den.zero_()
                if d_rgb is not None:
                    d_rgb.zero_()
            else:
                gdir = torch.empty_like(ctx["d"]) if ray_grads else None
                ops.zip_composite_bwd(L["raw_rgb"], L["raw_d"], L["tdist"], ctx["d"], self.opaque_background, ctx["bg"], 0.001, -1.0, L["weights"],
                                      L["acc"], L["depth"], cc(g_rgb), cc(g_depth), cc(g_acc), cc(g_w), d_rgb, d_den, g_dirs=gdir)
                if ray_grads:
                    rg[1] += gdir
            if sem_on:
                ops.semantic_composite_bwd(L["weights"], L["logits"], cc(g_sem), self.class_num, True, d_dl[:, 1:])
            d_den = d_dl if sem_on else d_den
            if lvl < 2:
                dF = net.backward(d_den, L["saved"])
            else:
                want_glo = L.get("cam") is not None                                     # the embedding rows were looked up: they get a gradient
                res = net.backward(d_rgb, d_den, L["saved"], want_dir_grad=ray_grads, want_glo_grad=want_glo)
                res = res if isinstance(res, tuple) else (res,)
                dF = res[0]
                if ray_grads:
                    rg[2] += ops.mip_viewenc_bwd(ctx["vd"], L["ns"], 1, res[1])         # dir_enc = pos_enc(viewdirs, 0, deg_view = 1)
                if want_glo:
                    ops.app_embed_bwd(res[-1], L["cam"], 1, self.arena.g["glo_vecs.weight"], deterministic=getattr(self, "_deterministic", False))
            if ray_grads:
                ops.zip_encode_ray_bwd(L["tdist"], ctx["o"], ctx["d"], ctx["radii"], ctx["bx"], ctx["by"], L["degj"], self._table(lvl),
                                       self.dev_offsets[lvl], self.dev_sizes[lvl], dF, e.L, e.C, ctx["n"], ctx["m"], e.Sl, e.H, self.std_scale,
                                       rg[0], rg[1], rg[3], rg[4])
            gtab = self.arena.g[self.names[lvl] + "encoder.embeddings"]
            if self.table_grad_mode == "binned":
                ks, g64_rows, lrows = ops.zip_bin_plan(e.offsets, e.C, P * ctx["n"] * 8)
                ops.zip_encode_bwd_binned(L["tdist"], ctx["o"], ctx["d"], ctx["radii"], ctx["bx"], ctx["by"], L["degj"], self.dev_offsets[lvl],
                                          self.dev_sizes[lvl], dF, gtab, e.L, e.C, ctx["n"], ctx["m"], e.Sl, e.H, self.std_scale, ks, g64_rows, lrows,
                                          precounted=L.get("precount"), half_records=self._half_records(lvl))
                L["precount"] = None                                                   # (its workgroup-offset buffer is large: release it now)
                if on_done is not None:
                    on_done(self.names[lvl])
                continue
            g16 = None
            if self.table_grad_bf16 and (e.C % 2 == 0 or e.C == 1):
                # hashed levels scatter packed bf16 pairs (C = 4: half the atomics; C = 1: x-neighbour corners of even cells share one
                # atomic, a quarter fewer); folded into the fp32 gradient right after
                g16 = torch.zeros(gtab.shape, dtype=torch.bfloat16, device=dev)
            ops.zip_encode_bwd(L["tdist"], ctx["o"], ctx["d"], ctx["radii"], ctx["bx"], ctx["by"], L["degj"], self.dev_offsets[lvl],
                               self.dev_sizes[lvl], dF, gtab, e.L, e.C, ctx["n"], ctx["m"], e.Sl,
                               e.H, self.std_scale, e.lds_levels, e.lds_cells, e.lds_slabs, grad_table_bf16=g16)
            if g16 is not None:
                gtab.add_(g16)
            if on_done is not None:
                on_done(self.names[lvl])
        return rg

    def _draws(self, R, rand, dev, sample_n):
        """the reference's RNG draws in its order: per level one single-jitter draw (stepfun.py:216) then the helix phase
        jitter (render.py:152)"""
        out = []
        for lvl in range(3):
            ns = self.num_prop_samples[lvl] if lvl < 2 else self.num_nerf_samples
            if not rand:
                pad = 1 / (2 * ns)
                out.append((torch.linspace(pad, 1. - pad - EPS32, ns).to(dev), None))
            else:
                u_max = EPS32 + (1 - EPS32) / ns
                max_jitter = (1 - u_max) / (ns - 1) - EPS32
                jit = torch.rand(R, 1 if self.single_jitter else ns, device=dev)
                u = (torch.linspace(0, 1 - u_max, ns).to(dev) + jit * max_jitter).contiguous()
                out.append((u, torch.rand(R, ns, sample_n, device=dev)))
        return out

    def _forward_extras(self, batch, train_frac, draws, sample_n, sample_m):
        """compute_extras=True, the rendering scripts' mode (random_render_waymo_seq.py:197; render.py:243-267, models.py:316-346):
        besides rgb / depth / acc every level carries distance_mean (the log-space expectation = depth), the 5 / 50 / 95 % distance
        percentiles, and the first `vis_num_rays` rays' histograms (ray_sdist, ray_weights, ray_rgbs; the proposal levels get the final
        level's average colour).  Inference only: nothing is saved for backward."""
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.param_list()):
            raise NotImplementedError("compute_extras is the inference / rendering mode: call it under torch.no_grad()")
        n = int(getattr(self.config, "vis_num_rays", 16)) if self.config is not None else 16
        levels, _ = self._run(batch, False, train_frac, draws, sample_n, sample_m)
        t_far = batch['far'].detach().to(levels[0]["tdist"].device, torch.float32).reshape(-1).contiguous()
        renderings, history = [], []
        for L in levels:
            pct = ops.zip_percentiles(L["tdist"], L["weights"], t_far)
            r = dict(rgb=L["rgb"], depth=L["depth"], acc=L["acc"], distance_mean=L["depth"], distance_percentile_5=pct[:, 0],
                     distance_median=pct[:, 1], distance_percentile_95=pct[:, 2], ray_sdist=L["sdist"][:n], ray_weights=L["weights"][:n])
            if L["semantic"] is not None:
                r["semantic"] = L["semantic"]
            renderings.append(r)
            history.append(dict(sdist=L["sdist"], weights=L["weights"], tdist=L["tdist"]))
        fin = levels[2]
        S = fin["weights"].shape[1]
        pad = 0.001                                                                     # rgb_padding, models.py:372
        rgbs = torch.sigmoid(fin["raw_rgb"][:min(n, fin["weights"].shape[0]) * S].float().reshape(-1, S, 3)) * (1 + 2 * pad) - pad
        renderings[2]["ray_rgbs"] = rgbs
        final = (rgbs * renderings[2]["ray_weights"][..., None]).sum(-2)
        for r in renderings[:2]:
            r["ray_rgbs"] = final[:, None, :].expand(-1, r["ray_weights"].shape[1], -1)
        return renderings, history

    def forward(self, rand, batch, train_frac, compute_extras, zero_glo=True, sample_n=7, sample_m=3, step=0, max_step=25000,
                cal_input_grad=False, draws=None):
        """-> (renderings, ray_history) like models.py:98-349.  `draws` = [(u, deg_jitter)] * 3 overrides the RNG (parity tests)."""
        self._check_arena()
        rkeys = ('origins', 'directions', 'viewdirs', 'base_x', 'base_y')
        ray_grad = torch.is_grad_enabled() and any(torch.is_tensor(batch[k]) and batch[k].requires_grad for k in rkeys)
        if ray_grad and not cal_input_grad:
            raise NotImplementedError("rays that require grad need cal_input_grad=True (the pose-refinement mode of zipnerf/train.py:187-224); "
                                      "the partial gradient the reference would form without the encoder's input gradient is not provided")
        if ray_grad and compute_extras:
            raise NotImplementedError("compute_extras is the rendering mode: no gradients")
        self._zero_glo = bool(zero_glo)          # (models.py:104,131-139: Model.forward defaults to zeros instead of the embedding rows)
        if self.num_glo_features and self.glo_embedding and not zero_glo and 'cam_idx' not in batch:
            raise KeyError("zero_glo=False needs batch['cam_idx']")
        dev = self.arena.flat.device
        R = batch['origins'].shape[0]
        if draws is None:
            draws = self._draws(R, rand, dev, sample_n)
        if compute_extras:
            return self._forward_extras(batch, float(train_frac), draws, sample_n, sample_m)
        params = self.param_list()
        keep = torch.is_grad_enabled() and (ray_grad or any(p.requires_grad for p in params))
        rt = tuple(batch[k] for k in rkeys) if ray_grad else (None,) * 5
        outs = _ZipFn.apply(self, batch, keep, float(train_frac), draws, sample_n, sample_m, *rt, *params)
        renderings, history = [], []
        for lvl in range(3):
            rgb, depth, acc, w, sd, td = outs[6 * lvl:6 * lvl + 6]
            renderings.append(dict(rgb=rgb, depth=depth, acc=acc))
            history.append(dict(sdist=sd, weights=w, tdist=td))
        if self.use_semantic:
            renderings[-1]["semantic"] = outs[18]
        return renderings, history




def _dist_info(accelerator):
    """(rank, world, gather) from an accelerate.Accelerator-like object or, when None, from torch.distributed."""
    import torch.distributed as dist
    if accelerator is not None:
        return accelerator.local_process_index, accelerator.num_processes, None
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size(), None
    return 0, 1, None


@torch.no_grad()
def render_image(render_fn, accelerator, batch, rand, config):
    """models.py:727-813: render all the pixels of an image (test mode).  `batch` = dict of [H,W,.] ray fields, `render_fn(rand, chunk)
    -> (renderings, ray_history)`, `config.render_chunk_size` rays per call; returns the final level's 2-D buffers reshaped to
    [H,W,...] plus, for `ray_*` keys, one tensor per level subsampled to `config.vis_num_rays` rays.

    Sharding: every rank renders ONE contiguous block of ceil(H*W / world) rays in its own chunks, and the frame is
    assembled with ONE all-gather per output buffer -- instead of the reference's split of every chunk across the processes with a
    gather per chunk (:761-770).  Rays are independent, so the assembled frame is the same.  `accelerator` may be an
    accelerate.Accelerator (only its process index / count are read) or None (torch.distributed if initialised, else one process)."""
    import torch.distributed as dist
    height, width = batch['origins'].shape[:2]
    num_rays = height * width
    flat = {k: v.reshape((num_rays, -1)) for k, v in batch.items() if v is not None}
    rank, world, _ = _dist_info(accelerator)
    per = (num_rays + world - 1) // world
    lo, hi = min(rank * per, num_rays), min((rank + 1) * per, num_rays)
    chunk = int(config.render_chunk_size)
    chunks = []
    for i0 in range(lo, hi, chunk):
        cb = {k: v[i0:min(i0 + chunk, hi)] for k, v in flat.items()}
        rend, _ = render_fn(rand, cb)
        cr = dict(rend[-1])
        for k in rend[0]:
            if k.startswith('ray_'):
                cr[k] = [r[k] for r in rend]
        chunks.append(cr)
    keys = list(chunks[0].keys()) if chunks else []
    if world > 1:      # a rank whose block is empty (more ranks than rays) still has to know the keys
        obj = [keys if rank == 0 else None]
        dist.broadcast_object_list(obj, src=0)
        keys = obj[0]
    rendering = {}
    for k in keys:
        if isinstance(chunks[0][k], list):
            rendering[k] = [torch.cat([c[k][i] for c in chunks]) for i in range(len(chunks[0][k]))]
        else:
            rendering[k] = torch.cat([c[k] for c in chunks])
    if world > 1:
        def gather(z, rows):
            pad = torch.zeros((rows,) + tuple(z.shape[1:]), dtype=z.dtype, device=z.device)
            pad[:z.shape[0]] = z
            out = [torch.empty_like(pad) for _ in range(world)]
            dist.all_gather(out, pad)
            return out
        for k in keys:
            if isinstance(rendering[k], list):
                nloc = torch.tensor([rendering[k][0].shape[0]], device=rendering[k][0].device)
                ns = [torch.zeros_like(nloc) for _ in range(world)]
                dist.all_gather(ns, nloc)
                mx = int(max(int(x) for x in ns))
                rendering[k] = [torch.cat([g[:int(c)] for g, c in zip(gather(z, mx), ns)]) for z in rendering[k]]
            else:
                parts = gather(rendering[k], per)
                rendering[k] = torch.cat([parts[r][:max(0, min((r + 1) * per, num_rays) - r * per)] for r in range(world)])
    for k, z in rendering.items():
        if not k.startswith('ray_'):
            rendering[k] = z.reshape((height, width) + tuple(z.shape[1:]))
    rkeys = [k for k in rendering if k.startswith('ray_')]
    if rkeys:
        nr = rendering[rkeys[0]][0].shape[0]
        idx = torch.randperm(nr)[:int(getattr(config, "vis_num_rays", 16))].to(rendering[rkeys[0]][0].device)
        for k in rkeys:
            rendering[k] = [r[idx] for r in rendering[k]]
    return rendering


class _ZipFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, batch, keep, train_frac, draws, sample_n, sample_m, ray_o, ray_d, ray_vd, ray_bx, ray_by, *params):
        ctx.set_materialize_grads(False)
        levels, c = model._run(batch, keep, train_frac, draws, sample_n, sample_m)
        ctx.model, ctx.c = model, c
        ctx.ray_meta = None if ray_o is None else [(t.shape, t.dtype, t.device) for t in (ray_o, ray_d, ray_vd, ray_bx, ray_by)]
        outs, nondiff = [], []
        for L in levels:
            outs += [L["rgb"], L["depth"], L["acc"], L["weights"], L["sdist"], L["tdist"]]
            nondiff += [L["sdist"], L["tdist"]]
        ctx.mark_non_differentiable(*nondiff)
        if model.use_semantic:
            outs.append(levels[2]["semantic"])
        return tuple(outs)

    @staticmethod
    def backward(ctx, *g):
        if ctx.c is None:
            raise RuntimeError("zipnerf Model.forward ran without saved activations")
        m = ctx.model
        m.arena.grad.zero_()
        ls = m.autograd_loss_scale           # fp16 compute: the backward runs on scaled output gradients (ZipTrainer folds the same factor into Adam)
        if ls != 1.0:
            g = tuple(None if t is None else t * ls for t in g)
        grads = [(g[6 * l], g[6 * l + 1], g[6 * l + 2], g[6 * l + 3]) for l in range(3)]
        if m.use_semantic:
            grads[2] = grads[2] + (g[18],)
        rg = m._backward(ctx.c, grads, ray_grads=ctx.ray_meta is not None)
        ctx.c = None
        grads = tuple(m.arena.g[n].clone() if ls == 1.0 else m.arena.g[n] * (1.0 / ls) for n in m._pnames)
        if rg is not None and ls != 1.0:
            rg = [t * (1.0 / ls) for t in rg]
        m.arena.grad.zero_()      # the trainers accumulate into the arena and expect it clean at step start
        if rg is None:
            return (None,) * 12 + grads
        rays_g = tuple(g.reshape(sh).to(device=dev, dtype=dt) for g, (sh, dt, dev) in zip(rg, ctx.ray_meta))
        return (None,) * 7 + rays_g + grads
