#!/usr/bin/env python3
"""Capture golden vectors G10/G11 for path C from the REFERENCE's zipnerf Python (build container only).

The reference's modules are imported from /root/reference/s-nerfpp/zipnerf with pass-through stubs for packages the
image lacks (gin, torch_scatter, skimage, cv2) and with oracle/grid.py injected as its `gridencoder` module (the
reference's CUDA extension cannot be built or run here) -- so these vectors pin everything AROUND the
hash grid: ray warps, dilation, interval sampling, multisample cone casting, contraction, erf down-weighting, the MLPs,
alpha weights, volumetric rendering and the 3-level Model.forward.
"""
import os
import sys
import types

os.environ["TORCHDYNAMO_DISABLE"] = "1"   # coord.py wraps two functions in torch.compile (48 s first call on CPU)
sys.dont_write_bytecode = True

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
if REPO not in sys.path:
    sys.path.insert(0, REPO)
from oracle import common as _oracle_common  # noqa: E402  (save_golden: writes the fixture, or compares under --check)
OUT = os.path.join(REPO, "tests", "golden")
REF = "/root/reference/s-nerfpp/zipnerf"
sys.path.insert(0, REPO)

from oracle import grid as og  # noqa: E402
from oracle import zip as oz  # noqa: E402


class OracleGridEncoder(nn.Module):
    """Stand-in for the reference's GridEncoder (gridencoder/grid.py:96-201): same attributes, forward through oracle/grid.py."""

    def __init__(self, input_dim=3, num_levels=16, level_dim=2, per_level_scale=2, base_resolution=16, log2_hashmap_size=19,
                 desired_resolution=None, gridtype='hash', align_corners=False, interpolation='linear', init_std=1e-4):
        super().__init__()
        off, res, s = og.level_layout(input_dim, num_levels, level_dim, per_level_scale, base_resolution, log2_hashmap_size,
                                      desired_resolution, align_corners)
        self.num_levels, self.level_dim, self.output_dim, self.init_std = num_levels, level_dim, num_levels * level_dim, init_std
        self.per_level_scale, self.base_resolution = s, base_resolution
        self.register_buffer("offsets", torch.from_numpy(off))
        self.register_buffer("grid_sizes", torch.from_numpy(res))
        idx = torch.empty(int(off[-1]), dtype=torch.long)
        for i in range(num_levels):
            idx[off[i]:off[i + 1]] = i
        self.register_buffer("idx", idx)
        self.embeddings = nn.Parameter(torch.empty(int(off[-1]), level_dim).uniform_(-init_std, init_std))

    def forward(self, inputs, bound=1, cal_input_grad=False):
        x = ((inputs + bound) / (2 * bound)).reshape(-1, 3).detach().numpy().astype(np.float32)
        out = og.grid_encode_forward(x, self.embeddings.detach().numpy(), self.offsets.numpy(), float(np.log2(self.per_level_scale)),
                                     self.base_resolution, 0, False, 0)
        return torch.from_numpy(out).permute(1, 0, 2).reshape(list(inputs.shape[:-1]) + [self.output_dim])


def import_reference():
    if not os.path.isdir(REF):
        raise SystemExit("reference tree not present")
    gin = types.ModuleType("gin")

    def configurable(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f
    gin.configurable = configurable
    gin.config = types.SimpleNamespace(external_configurable=lambda f, module=None: f)
    sys.modules["gin"] = gin
    ts = types.ModuleType("torch_scatter"); ts.segment_coo = lambda *a, **k: None; sys.modules["torch_scatter"] = ts
    sk = types.ModuleType("skimage"); skm = types.ModuleType("skimage.metrics")
    skm.structural_similarity = skm.peak_signal_noise_ratio = lambda *a, **k: None
    sys.modules["skimage"] = sk; sys.modules["skimage.metrics"] = skm
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    ge = types.ModuleType("gridencoder"); ge.GridEncoder = OracleGridEncoder; sys.modules["gridencoder"] = ge
    sys.path.insert(0, REF)
    from internal import coord, math as rmath, models, render, stepfun
    return coord, rmath, models, render, stepfun


def small_specs():
    """reduced hash tables (2^14 rows / level) so the fixtures stay small; the level structure (dense + hashed levels,
    L = 6 / 8 / 10, C = 1 / 1 / 4) is the shipped one"""
    return [oz.GridSpec(6, 1, 512, 16, 14), oz.GridSpec(8, 1, 2048, 16, 14), oz.GridSpec(10, 4, 8192, 16, 14)]


def formula_params(shapes):
    """deterministic parameters: tables ~ 0.3 sin(.), linear layers like oracle.common.formula_weight"""
    from oracle import common
    p = {}
    for li, (k, s) in enumerate(shapes):
        if k.endswith("embeddings"):
            i = torch.arange(s[0], dtype=torch.float64)[:, None]; j = torch.arange(s[1], dtype=torch.float64)[None, :]
            p[k] = (0.3 * torch.sin(0.017 * i + 1.3 * j + li)).float()
        elif len(s) == 2:
            p[k] = common.formula_weight(s[0], s[1], li)
        else:
            p[k] = common.formula_bias(s[0], li)
    return p


def make_batch(n, seed):
    g = torch.Generator().manual_seed(seed)
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1) * (1 + 0.1 * torch.rand(n, 1, generator=g))
    vd = torch.nn.functional.normalize(d, dim=-1)
    r = torch.randn(n, 3, generator=g)
    bx = torch.nn.functional.normalize(torch.cross(vd, r, dim=-1), dim=-1)
    by = torch.nn.functional.normalize(torch.cross(vd, bx, dim=-1), dim=-1)
    return dict(origins=torch.randn(n, 3, generator=g) * 0.1, directions=d, viewdirs=vd, radii=2e-3 + 2e-3 * torch.rand(n, 1, generator=g),
                near=torch.full((n, 1), 0.1), far=torch.full((n, 1), 10.0), base_x=bx, base_y=by)


def main():
    coord, rmath, models, render, stepfun = import_reference()
    G = {}
    g = torch.Generator().manual_seed(77)
    R = lambda *s: torch.rand(*s, generator=g)
    # ---- G10: stage-level vectors ----
    near, far = torch.full((5, 1), 0.1), torch.full((5, 1), 10.0)
    _, s2t = coord.construct_ray_warps('power_transformation', near, far, -1.5)
    s = torch.sort(R(5, 9), -1)[0]; s[:, 0] = 0; s[:, -1] = 1
    G["g10_warp"] = dict(s=s, near=near, far=far, t=s2t(s))
    t = torch.sort(R(6, 13), -1)[0]; t[:, 0] = 0; t[:, -1] = 1
    w = R(6, 12) ** 3; w[0] = 0; w[0, 4] = 1; w = w / w.sum(-1, keepdim=True)
    td, wd = stepfun.max_dilate_weights(t, w, 0.0103, domain=(0., 1.), renormalize=True)
    G["g10_dilate"] = dict(t=t, w=w, dilation=np.float32(0.0103), t_dilate=td, w_dilate=wd)
    logits = torch.where(t[..., 1:] > t[..., :-1], 0.7 * torch.log(w + 0.0), torch.full_like(w, -float("inf")))
    sd_det = stepfun.sample_intervals(None, t, logits, 16, single_jitter=True, domain=(0., 1.))
    torch.manual_seed(3)
    sd_rnd = stepfun.sample_intervals(True, t, logits, 16, single_jitter=True, domain=(0., 1.))
    torch.manual_seed(3)
    jit = torch.rand(6, 1)
    G["g10_sample_intervals"] = dict(t=t, logits=logits, sdist_det=sd_det, sdist_rand=sd_rnd, jitter=jit)
    b = make_batch(7, 5)
    tdist = torch.sort(R(7, 6) * 5 + 0.2, -1)[0]
    m0, s0 = render.cast_rays(tdist, b["origins"], b["directions"], b["radii"], rand=False, n=7, m=3, std_scale=0.35, batch=b)
    torch.manual_seed(9)
    m1, s1 = render.cast_rays(tdist, b["origins"], b["directions"], b["radii"], rand=True, n=7, m=3, std_scale=0.35, batch=b)
    torch.manual_seed(9)
    degj = torch.rand(7, 5, 7)
    G["g10_cast_rays"] = dict(tdist=tdist, **{"b_" + k: v for k, v in b.items()}, means_det=m0, stds_det=s0, means_rand=m1, stds_rand=s1, deg_jitter=degj)
    x = torch.randn(40, 3, generator=g) * torch.tensor([0.3, 1.0, 4.0]); st = R(40) * 0.05
    zc, sc = coord.contract_mean_std(x, st)
    G["g10_contract"] = dict(x=x, std=st, z=zc, std_out=sc)
    dens = R(7, 5) * 3; dens[0] = 0
    wts = render.compute_alpha_weights(dens, tdist, b["directions"], opaque_background=True)[0]
    wts_no = render.compute_alpha_weights(dens, tdist, b["directions"], opaque_background=False)[0]
    rgbs = R(7, 5, 3)
    rend = render.volumetric_rendering(rgbs, wts, tdist, 1.0, b["far"], False)
    rend_no = render.volumetric_rendering(rgbs, wts_no, tdist, 0.5, b["far"], False)
    G["g10_render"] = dict(density=dens, tdist=tdist, dirs=b["directions"], weights=wts, weights_noopaque=wts_no, rgbs=rgbs, rgb=rend["rgb"],
                           depth=rend["depth"], rgb_noopaque=rend_no["rgb"], depth_noopaque=rend_no["depth"])

    # ---- G11: Model.forward (3 levels) with the oracle grid encoder injected ----
    specs = small_specs()
    cfg = types.SimpleNamespace(use_semantic=False, vis_num_rays=8, zero_glo=True)
    torch.manual_seed(0)
    model = models.Model(config=cfg, raydist_fn='power_transformation', opaque_background=True)
    # apply the waymo.gin bindings as attributes / rebuilt sub-modules (gin is stubbed)
    model.nerf_mlp = models.NerfMLP(disable_density_normals=True, deg_view=1, grid_log2_hashmap_size=14, use_semantic=False)
    model.prop_mlp_0 = models.PropMLP(disable_density_normals=True, disable_rgb=True, grid_level_dim=1, grid_disired_resolution=512, grid_log2_hashmap_size=14)
    model.prop_mlp_1 = models.PropMLP(disable_density_normals=True, disable_rgb=True, grid_level_dim=1, grid_disired_resolution=2048, grid_log2_hashmap_size=14)
    shapes = oz.param_shapes(specs)
    sd_ref = {k: v for k, v in model.state_dict().items() if k.endswith(("weight", "bias", "embeddings"))}
    assert [k for k, _ in shapes] == list(sd_ref.keys()), (list(sd_ref.keys()), [k for k, _ in shapes])
    assert all(tuple(sd_ref[k].shape) == tuple(s) for k, s in shapes)
    p = formula_params(shapes)
    model.load_state_dict(p, strict=False)
    batch = make_batch(20, 11)
    with torch.no_grad():
        rend, hist = model(None, dict(batch), train_frac=1.0, compute_extras=False)
        torch.manual_seed(21)
        rend_r, hist_r = model(True, dict(batch), train_frac=0.37, compute_extras=False)
    # replay the reference's RNG draws: per level one single-jitter draw [R,1] (stepfun.py:216) then deg jitter [R,S,7] (render.py:152)
    torch.manual_seed(21)
    jit, degj = [], []
    for ns in (64, 64, 32):
        jit.append(torch.rand(20, 1)); degj.append(torch.rand(20, ns, 7))
    d = {"b_" + k: v for k, v in batch.items()}
    for lvl in range(3):
        d[f"det_sdist{lvl}"] = hist[lvl]["sdist"]; d[f"det_weights{lvl}"] = hist[lvl]["weights"]
        d[f"rand_sdist{lvl}"] = hist_r[lvl]["sdist"]; d[f"rand_weights{lvl}"] = hist_r[lvl]["weights"]
        d[f"jitter{lvl}"] = jit[lvl]; d[f"deg_jitter{lvl}"] = degj[lvl]
    d.update(det_rgb=rend[-1]["rgb"], det_depth=rend[-1]["depth"], rand_rgb=rend_r[-1]["rgb"], rand_depth=rend_r[-1]["depth"],
             det_depth0=rend[0]["depth"], rand_train_frac=np.float32(0.37))
    # compute_extras=True (the rendering scripts' mode, random_render_waymo_seq.py:197): acc, distance statistics, visualisation rays
    with torch.no_grad():
        rend_x, _ = model(None, dict(batch), train_frac=1.0, compute_extras=True)
    for lvl in range(3):
        for k in ("acc", "distance_mean", "distance_percentile_5", "distance_median", "distance_percentile_95", "ray_sdist", "ray_weights", "ray_rgbs"):
            d[f"x{lvl}_{k}"] = rend_x[lvl][k]
    # the same model with the semantic head enabled (Config.use_semantic / NerfMLP.use_semantic: models.py:297-305, 594-597)
    cfg.use_semantic = True
    model.config = cfg
    model.nerf_mlp.use_semantic = True
    with torch.no_grad():
        rend_s, _ = model(None, dict(batch), train_frac=1.0, compute_extras=False)
    d.update(sem_rgb=rend_s[-1]["rgb"], sem_semantic=rend_s[-1]["semantic"])
    G["g11_zip_model"] = d
    os.makedirs(OUT, exist_ok=True)
    for name, dd in G.items():
        arr = {k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in dd.items()}
        _oracle_common.save_golden(os.path.join(OUT, name + ".npz"), **arr)
        print("wrote", name, sum(a.nbytes for a in arr.values()), "bytes")


if __name__ == "__main__":
    main()
