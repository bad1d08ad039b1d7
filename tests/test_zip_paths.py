"""zipnerf Model (path C) end to end: the drop-in module vs the reference's own Model.forward outputs (golden G11, captured
with the oracle grid encoder injected) and vs the oracle, forward and backward.  "hip" = real kernels, "emulated" = host logic."""
import importlib.util
import os

import numpy as np
import pytest
import torch

from cpu_ops_emulation import emulate_ops
from oracle import zip as oz

DEV = "cuda"


@pytest.fixture(params=[pytest.param("hip", marks=pytest.mark.gpu), "emulated"])
def backend(request):
    global DEV
    if request.param == "hip":
        DEV = "cuda"
        yield "hip"
    else:
        DEV = "cpu"
        with emulate_ops():
            yield "emulated"
    DEV = "cuda"


def close(a, b, rtol, atol, what=""):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a - b).abs(); tol = atol + rtol * b.abs()
    bad = err > tol
    assert not bool(bad.any()), f"{what}: {int(bad.sum())}/{bad.numel()} out of tol, max err {err.max().item():.3e}"


def zip_setup():
    spec = importlib.util.spec_from_file_location("gen_golden_zip", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "gen_golden_zip.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    specs = m.small_specs()
    return specs, m.formula_params(oz.param_shapes(specs))


def make_model(compute, table, p, use_semantic=False):
    from snerf_amd import zipnerf
    m = zipnerf.Model(config=None, raydist_fn='power_transformation', opaque_background=True, compute=compute, table_dtype=table, device=DEV,
                      grid_log2_hashmap_size=14, use_semantic=use_semantic)
    sd = m.state_dict()
    for k, v in p.items():
        assert k in sd and tuple(sd[k].shape) == tuple(v.shape), k
    for k in ("nerf_mlp.encoder.offsets", "nerf_mlp.encoder.idx", "nerf_mlp.encoder.grid_sizes", "prop_mlp_1.encoder.offsets"):
        assert k in sd
    m.load_state_dict(p, strict=False)
    return m


@pytest.mark.parametrize("compute,table,tol", [("f32", "f32", 2e-4), ("bf16", "f16", 5e-2), ("fp16", "f16", 2e-3)])
def test_zip_model_vs_reference_golden(backend, golden, compute, table, tol):
    g = golden("g11_zip_model")
    specs, p = zip_setup()
    batch = {k[2:]: v.to(DEV) for k, v in g.items() if k.startswith("b_")}
    m = make_model(compute, table, p)
    with torch.no_grad():
        rend, hist = m(None, batch, 1.0, False)
    assert set(rend[-1]) >= {"rgb", "depth", "acc"} and set(hist[-1]) >= {"sdist", "weights", "tdist"}
    close(hist[0]["sdist"], g["det_sdist0"], 1e-5, 1e-6, "sdist level 0 (no network upstream)")
    if compute == "f32":
        for lvl in range(3):
            close(hist[lvl]["sdist"], g[f"det_sdist{lvl}"], tol, tol, f"sdist {lvl}")
            close(hist[lvl]["weights"], g[f"det_weights{lvl}"], 10 * tol, tol, f"weights {lvl}")
        close(rend[0]["depth"], g["det_depth0"], tol, tol, "depth level 0")
    close(rend[-1]["rgb"], g["det_rgb"], tol, tol, "rgb"); close(rend[-1]["depth"], g["det_depth"], tol, 10 * tol, "depth")
    if compute == "f32":
        draws = [(oz.rand_u(ns, g[f"jitter{i}"]).to(DEV).contiguous(), g[f"deg_jitter{i}"].to(DEV).contiguous()) for i, ns in enumerate((64, 64, 32))]
        with torch.no_grad():
            rend, hist = m(True, batch, float(g["rand_train_frac"]), False, draws=draws)
        for lvl in range(3):
            close(hist[lvl]["sdist"], g[f"rand_sdist{lvl}"], tol, tol, f"rand sdist {lvl}")
        close(rend[-1]["rgb"], g["rand_rgb"], tol, tol, "rand rgb"); close(rend[-1]["depth"], g["rand_depth"], tol, tol, "rand depth")


def test_zip_near_bound_annealing_vs_reference_golden(backend, golden):
    """Model.near_anneal_rate (models.py:47-48, 147-158): the sampling domain starts at clip(1 - train_frac / rate, 0, near_anneal_init)
    -- fence posts of all three levels, weights, colour and depth against the reference Model's own (g24), fp32."""
    g = golden("g24_zip_near_anneal")
    specs, p = zip_setup()
    batch = {k[2:]: v.to(DEV) for k, v in g.items() if k.startswith("b_")}
    m = make_model("f32", "f32", p)
    m.near_anneal_rate, m.near_anneal_init = float(g["near_anneal_rate"]), float(g["near_anneal_init"])
    for tag in ("a", "b"):
        with torch.no_grad():
            rend, hist = m(None, batch, float(g[tag + "_train_frac"]), False)
        close(hist[0]["sdist"], g[tag + "_sdist0"], 1e-5, 1e-6, tag + " sdist level 0")
        for lvl in (1, 2):
            close(hist[lvl]["sdist"], g[f"{tag}_sdist{lvl}"], 2e-4, 2e-5, f"{tag} sdist level {lvl}")
        close(hist[2]["weights"], g[tag + "_weights2"], 2e-3, 2e-5, tag + " weights")
        close(rend[-1]["rgb"], g[tag + "_rgb"], 2e-4, 2e-4, tag + " rgb"); close(rend[-1]["depth"], g[tag + "_depth"], 2e-4, 2e-4, tag + " depth")


@pytest.mark.parametrize("compute", ["f32", "bf16", "fp16"])
def test_zip_glo_vectors_vs_reference_golden(backend, golden, compute):
    """Model(num_glo_features=4) (models.py:44-45, 75-77, 131-139, 454-459, 620-630; configs/360_glo4.gin): per-image GLO vectors ->
    lin_glo_0 / lin_glo_1 -> (scale, shift) modulation of the NeRF MLP's bottleneck.  Forward with the embedding rows and with zero_glo,
    state_dict layout, and the parameter gradients the reference's own autograd produced (g25) -- the GLO MLP, the embedding table
    (only the looked-up rows), the layers on both sides of the modulation."""
    g = golden("g25_zip_glo")
    from snerf_amd import zipnerf
    Fg, E = int(g["num_glo_features"]), int(g["num_glo_embeddings"])
    specs, _ = zip_setup()
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("gen_golden_zip", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "gen_golden_zip.py"))
    gz = importlib.util.module_from_spec(spec); spec.loader.exec_module(gz)
    shapes = oz.param_shapes(specs, num_glo_features=Fg, num_glo_embeddings=E, zero_glo=False)
    p = gz.formula_params(shapes)
    p["glo_vecs.weight"], p["nerf_mlp.lin_glo_0.weight"], p["nerf_mlp.lin_glo_1.weight"] = g["glo_vecs"], g["lin_glo_0_weight"], g["lin_glo_1_weight"]
    m = zipnerf.Model(config=None, raydist_fn='power_transformation', opaque_background=True, compute=compute, table_dtype="f32", device=DEV,
                      grid_log2_hashmap_size=14, num_glo_features=Fg, num_glo_embeddings=E)
    keys = [k for k in m.state_dict().keys() if k.endswith(("weight", "bias", "embeddings"))]
    assert keys == [k for k, _ in shapes], keys                                    # lin_glo_* in front of the second stage, glo_vecs last
    m.load_state_dict(p, strict=False)
    batch = {k[2:]: v.to(DEV) for k, v in g.items() if k.startswith("b_")}
    tol = {"f32": 2e-4, "bf16": 3e-2, "fp16": 2e-3}[compute]
    named = dict(m.named_parameters())
    for tag, zero in (("emb", False), ("zero", True)):
        for q in m.parameters():
            q.grad = None
        rend, hist = m(None, batch, 1.0, False, zero_glo=zero)
        loss = ((rend[-1]["rgb"] - g["target"].to(DEV)) ** 2).mean() + 0.01 * rend[-1]["depth"].mean()
        loss.backward()
        close(rend[-1]["rgb"], g[tag + "_rgb"], tol, tol, tag + " rgb"); close(rend[-1]["depth"], g[tag + "_depth"], tol, tol, tag + " depth")
        for k in g:
            if k.startswith(tag + "_grad."):
                n = k[len(tag) + 6:]
                got, want = named[n].grad, g[k]
                if tag == "zero" and n == "glo_vecs.weight":
                    continue
                if compute != "f32" and "glo" not in n:      # (formula weights are rank 2 and amplify bf16 rounding to O(1): fp32 only;
                    continue                                  #  the GLO layers carry full-rank random weights and are held in bf16 too)
                assert got is not None, n
                rel = float((got.detach().cpu() - want).norm() / (want.norm() + 1e-30))
                print(f"MEASURED glo gradient {compute} {tag} {n}: rel L2 {rel:.2e}")
                assert rel < {"f32": 2e-3, "bf16": 0.25, "fp16": 0.05}[compute], (tag, n, rel)
        if zero:
            gv = named["glo_vecs.weight"].grad
            assert gv is None or float(gv.abs().max()) == 0.0                      # zeros went in: the table gets no gradient


@pytest.mark.parametrize("compute,table,tol", [("f32", "f32", 2e-4), ("bf16", "f16", 3e-2), ("fp16", "f16", 2e-3)])
def test_zip_semantic_head_vs_reference_golden(backend, golden, compute, table, tol):
    """Config.use_semantic: the 19-class distribution rendered by the reference Model (golden) and the unchanged colour."""
    g = golden("g11_zip_model")
    specs, p = zip_setup()
    batch = {k[2:]: v.to(DEV) for k, v in g.items() if k.startswith("b_")}
    m = make_model(compute, table, p, use_semantic=True)
    with torch.no_grad():
        rend, _ = m(None, batch, 1.0, False)
    assert rend[-1]["semantic"].shape == (20, 19) and "semantic" not in rend[0]
    close(rend[-1]["semantic"], g["sem_semantic"], tol, tol, "semantic")
    close(rend[-1]["rgb"], g["sem_rgb"], max(tol, 2e-4), max(tol, 2e-4), "rgb")
    close(rend[-1]["semantic"].sum(-1), torch.ones(20), 1e-5 if compute == "f32" else 1e-3, 1e-5 if compute == "f32" else 1e-3, "class probabilities sum to acc = 1")


@pytest.mark.parametrize("use_semantic", [False, True])
def test_zip_model_backward_vs_oracle_autograd(backend, use_semantic):
    """fp32 gradients of every parameter incl. the three hash tables vs torch autograd through the oracle (the oracle's grid
    lookup is made differentiable with an explicit transpose-gather: features are linear in the table).  With the semantic head
    a class-weighted term on the rendered distribution is added to the loss (its gradient reaches the density network only
    through the logits: the compositing weights are detached, render.py:237-241)."""
    specs, p = zip_setup()
    R = 12
    g = torch.Generator().manual_seed(5)
    d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
    rv = torch.randn(R, 3, generator=g)
    bx = torch.nn.functional.normalize(torch.cross(d, rv, dim=-1), dim=-1)
    batch = dict(origins=torch.randn(R, 3, generator=g) * 0.1, directions=d, viewdirs=d, radii=2e-3 + 2e-3 * torch.rand(R, 1, generator=g),
                 near=torch.full((R, 1), 0.1), far=torch.full((R, 1), 10.0), base_x=bx, base_y=torch.nn.functional.normalize(torch.cross(d, bx, dim=-1), dim=-1))
    target = torch.rand(R, 3, generator=g)
    semw = torch.randn(R, 19, generator=g)
    m = make_model("f32", "f32", p, use_semantic=use_semantic)
    rend, hist = m(None, {k: v.to(DEV) for k, v in batch.items()}, 1.0, False)
    wts = [h["weights"] for h in hist]
    loss = ((rend[-1]["rgb"] - target.to(DEV)) ** 2).mean() + 0.1 * rend[-1]["depth"].mean() + 0.05 * sum((w ** 2).sum() for w in wts)
    if use_semantic:
        loss = loss + 0.3 * (rend[-1]["semantic"] * semw.to(DEV)).sum() / R
    loss.backward()
    # oracle: same forward with torch-differentiable table lookups
    pr = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    import oracle.zip as ozm
    orig = ozm.grid_features

    def grid_features_diff(spec, emb, means):
        from oracle import grid as og
        x = ((means.reshape(-1, 3) + 1) / 2).detach().numpy().astype(np.float32)
        B = x.shape[0]
        outs = []
        for l in range(spec.L):                         # features = sum_corners w * emb[row]: build rows/weights with the oracle's own index math
            hs, scale, res = og._level_setup(l, spec.S, spec.H, spec.offsets)
            pg, frac, _ = og._positions(x, scale, False, 0)
            acc = 0
            for idx in range(8):
                w = np.ones(B, dtype=np.float32); pl = pg.copy()
                for dd in range(3):
                    if idx & (1 << dd):
                        w = w * frac[:, dd]; pl[:, dd] = pg[:, dd] + 1
                    else:
                        w = w * (1 - frac[:, dd])
                rows = og.grid_index(0, False, hs, res, pl) + int(spec.offsets[l])
                acc = acc + torch.from_numpy(w)[:, None] * emb[torch.from_numpy(rows)]
            outs.append(acc)
        return torch.stack(outs, 1).reshape(list(means.shape[:-1]) + [spec.L, spec.C])
    ozm.grid_features = grid_features_diff
    try:
        rend_o, hist_o = oz.model_forward(pr, specs, batch, train_frac=1.0, use_semantic=use_semantic)
    finally:
        ozm.grid_features = orig
    loss_o = ((rend_o[-1]["rgb"] - target) ** 2).mean() + 0.1 * rend_o[-1]["depth"].mean() + 0.05 * sum((h["weights"] ** 2).sum() for h in hist_o)
    if use_semantic:
        loss_o = loss_o + 0.3 * (rend_o[-1]["semantic"] * semw).sum() / R
    loss_o.backward()
    close(loss, loss_o, 1e-4, 1e-5, "loss")
    named = dict(m.named_parameters())
    bad = []
    for k in p:
        gref = pr[k].grad
        assert gref is not None, k
        got = named[k].grad.detach().cpu()
        rel = ((got - gref).norm() / (gref.norm() + 1e-20)).item()
        print(f"grad {k}: rel {rel:.3e} |ref| {gref.norm().item():.3e} |got| {got.norm().item():.3e}")
        # fine hash levels (resolution 8193) turn a 1e-7 position difference (sincosf / cbrtf / erff vs torch) into a ~1e-3 change of the
        # trilinear weights, hence the wider bound for the tables on the device
        bad = bad + [(k, rel)] if rel >= (3e-2 if k.endswith('embeddings') else 5e-3) else bad
    assert not bad, bad


@pytest.mark.gpu
def test_zip_train_step_at_scale():
    """The production-size model (waymo.gin grids, 2^21 hash entries) on enough rays that every GEMM takes its large-M kernels
    (persistent 8-phase NT, 8-phase wgrad with operands that are column ranges of wider buffers): runs, finite, and the bf16
    step agrees with the fp32-parity step on the rendered colour."""
    from snerf_amd import ops, zipnerf
    R = 4096
    g = torch.Generator().manual_seed(3)
    pix = torch.randint(0, 1920 * 1280, (R,), generator=g)
    i, j = (pix % 1920).float(), (pix // 1920).float()
    d = torch.stack([(i - 960 + 0.5) / 2050, -(j - 640 + 0.5) / 2050, -torch.ones(R)], -1)
    vd = torch.nn.functional.normalize(d, dim=-1)
    bx = torch.nn.functional.normalize(torch.cross(vd, torch.tensor([0.0, 1.0, 0.0]).expand(R, 3), dim=-1), dim=-1)
    by = torch.nn.functional.normalize(torch.cross(vd, bx, dim=-1), dim=-1)
    batch = {k: v.cuda() for k, v in dict(origins=torch.randn(R, 3, generator=g) * 0.05, directions=d, viewdirs=vd,
                                          radii=torch.full((R, 1), 2.0 / 2050 / 12 ** 0.5), near=torch.full((R, 1), 0.1),
                                          far=torch.full((R, 1), 10.0), base_x=bx, base_y=by).items()}
    tgt = torch.rand(R, 3, generator=g).cuda()
    rgbs = {}
    for compute, table in (("bf16", "f16"), ("fp16", "f16"), ("f32", "f32")):
        torch.manual_seed(0)
        m = zipnerf.Model(config=None, raydist_fn='power_transformation', opaque_background=True, compute=compute, table_dtype=table,
                          grid_log2_hashmap_size=21, init_std=0.1)
        draws = m._draws(R, False, m.arena.flat.device, 7)
        ren, _ = m(False, batch, 1.0, False, draws=draws)
        rgb = ren[2]["rgb"]
        loss = ((rgb - tgt) ** 2).mean()
        loss.backward()
        gr = torch.cat([p.grad.flatten() for p in m.param_list() if p.grad is not None])
        assert bool(torch.isfinite(rgb).all()) and bool(torch.isfinite(gr).all()) and float(gr.abs().max()) > 0
        rgbs[compute] = rgb.detach().float()
    assert float((rgbs["bf16"] - rgbs["f32"]).abs().max()) < 3e-2
    e16 = float((rgbs["fp16"] - rgbs["f32"]).abs().max())
    print(f"MEASURED full-size zip model, fp16 compute vs fp32: max |d rgb| {e16:.2e}")
    assert e16 < 1e-4                     # (measured 1.5e-5; the bf16 mode is bounded by 3e-2 above)


@pytest.mark.gpu
def test_zip_table_gradient_bf16_pairs_match_fp32():
    """table_grad_dtype="bf16" (half the atomics on the hashed levels) against the fp32 scatter on the same step."""
    from snerf_amd import zipnerf
    R = 2048
    g = torch.Generator().manual_seed(4)
    d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
    bx = torch.nn.functional.normalize(torch.cross(d, torch.tensor([0.0, 1.0, 0.0]).expand(R, 3), dim=-1), dim=-1)
    by = torch.nn.functional.normalize(torch.cross(d, bx, dim=-1), dim=-1)
    batch = {k: v.cuda() for k, v in dict(origins=torch.randn(R, 3, generator=g) * 0.05, directions=d, viewdirs=d, radii=torch.full((R, 1), 5e-4),
                                          near=torch.full((R, 1), 0.1), far=torch.full((R, 1), 10.0), base_x=bx, base_y=by).items()}
    tgt = torch.rand(R, 3, generator=g).cuda()
    grads = {}
    for mode in ("f32", "bf16"):
        torch.manual_seed(0)
        m = zipnerf.Model(config=None, raydist_fn='power_transformation', opaque_background=True, compute="bf16", table_dtype="f16",
                          grid_log2_hashmap_size=19, init_std=0.1, table_grad_dtype=mode)
        draws = m._draws(R, False, m.arena.flat.device, 7)
        ren, hist = m(False, batch, 1.0, False, draws=draws)
        # a term on the proposal histograms so that the single-channel grids (paired bf16 atomics on their hashed levels) get gradients
        (((ren[2]["rgb"] - tgt) ** 2).mean() + 0.05 * sum((h["weights"] ** 2).sum() for h in hist[:2]) / R).backward()
        grads[mode] = {k: dict(m.named_parameters())[k + ".encoder.embeddings"].grad.float().flatten().clone() for k in ("nerf_mlp", "prop_mlp_0", "prop_mlp_1")}
    for k in ("nerf_mlp", "prop_mlp_0", "prop_mlp_1"):
        a, b = grads["f32"][k], grads["bf16"][k]
        assert float(a.abs().max()) > 0, k
        cos = float((a * b).sum() / (a.norm() * b.norm()))
        assert cos > 0.999, (k, cos)          # measured 0.9994-0.9999 (the fp32 side is the exact binned accumulation since round 2)
        assert float((a - b).norm() / a.norm()) < 4e-2, (k, float((a - b).norm() / a.norm()))   # bf16 pairs: 8-bit mantissa per addend (measured <= 3.5e-2)


@pytest.mark.gpu
def test_zip_fp16_compute_gradients_and_loss_scale():
    """compute="fp16" (fp16 features, activations, gradient buffers; the fp16 MFMA): parameter gradients of loss.backward() against the
    fp32 mode on the same step -- the networks' backward runs on output gradients multiplied by the static loss scale, which the
    autograd bridge divides back -- and ZipTrainer's loss-scale plumbing checked EXACTLY in fp32 compute: a step with loss_scale = 4096
    updates the parameters like a step with loss_scale = 1 (every gradient source -- loss tail, hash decay -- carries the factor, the
    Adam launch undoes it)."""
    from snerf_amd import zipnerf
    from snerf_amd.trainer import ZipTrainer
    R = 2048
    g = torch.Generator().manual_seed(4)
    d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
    bx = torch.nn.functional.normalize(torch.cross(d, torch.tensor([0.0, 1.0, 0.0]).expand(R, 3), dim=-1), dim=-1)
    by = torch.nn.functional.normalize(torch.cross(d, bx, dim=-1), dim=-1)
    batch = {k: v.cuda() for k, v in dict(origins=torch.randn(R, 3, generator=g) * 0.05, directions=d, viewdirs=d, radii=torch.full((R, 1), 5e-4),
                                          near=torch.full((R, 1), 0.1), far=torch.full((R, 1), 10.0), base_x=bx, base_y=by).items()}
    tgt = torch.rand(R, 3, generator=g).cuda()
    grads = {}
    for compute, table in (("f32", "f32"), ("fp16", "f16"), ("bf16", "f16")):
        torch.manual_seed(0)
        m = zipnerf.Model(config=None, raydist_fn='power_transformation', opaque_background=True, compute=compute, table_dtype=table,
                          grid_log2_hashmap_size=19, init_std=0.1)
        assert m.autograd_loss_scale == (4096.0 if compute == "fp16" else 1.0)
        draws = m._draws(R, False, m.arena.flat.device, 7)
        ren, hist = m(False, batch, 1.0, False, draws=draws)
        (((ren[2]["rgb"] - tgt) ** 2).mean() + 0.05 * sum((h["weights"] ** 2).sum() for h in hist[:2]) / R).backward()
        grads[compute] = {k: p.grad.float().flatten().clone() for k, p in m.named_parameters() if p.grad is not None}
    worst = {"fp16": 0.0, "bf16": 0.0}
    for k, a in grads["f32"].items():
        if float(a.norm()) == 0:
            continue
        for c in ("fp16", "bf16"):
            rel = float((grads[c][k] - a).norm() / a.norm())
            worst[c] = max(worst[c], rel)
            if c == "fp16":
                print(f"MEASURED fp16-compute gradient vs fp32, {k}: rel L2 {rel:.2e}")
    print(f"MEASURED worst parameter-gradient rel L2 vs fp32: fp16 compute {worst['fp16']:.2e}, bf16 compute {worst['bf16']:.2e}")
    assert worst["fp16"] < 2e-2 and worst["fp16"] < worst["bf16"]
    # the trainer's loss scale, exactly (fp32 compute: the only difference between the two runs is the factor and its inverse)
    new = {}
    for ls in (1.0, 4096.0):
        torch.manual_seed(0)
        m = zipnerf.Model(config=None, raydist_fn='power_transformation', opaque_background=True, compute="f32", table_dtype="f32",
                          grid_log2_hashmap_size=14, init_std=0.1)
        tr = ZipTrainer(m, lr=1e-2, eps=1e-8, loss_scale=ls)
        draws = m._draws(R, False, m.arena.flat.device, 7)
        start = m.arena.flat.clone()
        loss, _ = tr.step(batch, tgt, rand=False, draws=draws)
        new[ls] = (float(loss), (m.arena.flat - start).clone())
    assert abs(new[1.0][0] - new[4096.0][0]) <= 1e-6 * abs(new[1.0][0])
    da, db = new[1.0][1], new[4096.0][1]
    # (Adam's first step is lr * g / (|g| + eps), eps = 1e-8 in the range of the gradients: the update follows their magnitude; what is
    #  left between the two runs is the summation order of the fp32 atomics and the binned accumulation's rounding at another scale)
    moved = da.abs() > 1e-3 * 1e-2
    worst = float((da - db).abs().max())
    print(f"MEASURED loss scale 4096 vs 1 (fp32 compute): max |d update| {worst:.2e} of lr = 1e-2; {float(moved.float().mean()):.2f} of the parameters moved")
    assert float(moved.float().mean()) > 0.01 and worst <= 1e-3 * 1e-2, worst
    # and the fp16 trainer follows the fp32 trainer's loss curve on the same batch
    curves = {}
    for compute, table in (("f32", "f32"), ("fp16", "f16")):
        torch.manual_seed(0)
        m = zipnerf.Model(config=None, raydist_fn='power_transformation', opaque_background=True, compute=compute, table_dtype=table,
                          grid_log2_hashmap_size=14, init_std=0.1)
        tr = ZipTrainer(m, lr=2e-3, eps=1e-8)
        assert tr.loss_scale == (4096.0 if compute == "fp16" else 1.0)
        draws = m._draws(R, False, m.arena.flat.device, 7)
        curves[compute] = [float(tr.step(batch, tgt, rand=False, draws=draws)[0]) for _ in range(15)]
        assert bool(torch.isfinite(m.arena.flat).all())
    a, b = curves["f32"], curves["fp16"]
    print(f"MEASURED 15 training steps: fp32 loss {a[0]:.5f} -> {a[-1]:.5f}, fp16 compute {b[0]:.5f} -> {b[-1]:.5f}")
    assert a[-1] < a[0] and b[-1] < b[0] and max(abs(x - y) for x, y in zip(a, b)) < 0.02 * (a[0] - a[-1]) + 1e-4 * a[0]


def test_zip_fp16_vs_reference_autocast_golden(backend, golden):
    """compute="fp16" against the reference Model run under torch.autocast(dtype=float16) (g26: zipnerf/train.py:215's
    `accelerator.autocast()`; oracle/gen_golden_zip_fp16.py): final colour / depth and the dense parameters' gradients.  The reference's
    autocast rounds every nn.Linear OUTPUT to fp16 (raw density, raw rgb included); this build keeps the heads' outputs in fp32, so it
    sits between the two reference runs: the test bounds its distance to the fp16 golden by the tolerance north_star states for the
    reduced-precision modes, and its distance to the fp32 golden by the reference's own fp16-vs-fp32 deviation."""
    g = golden("g26_zip_fp16")
    specs, p = zip_setup()
    for k in g:
        if k.startswith("w."):
            p[k[2:]] = g[k]
    batch = {k[2:]: v.to(DEV) for k, v in g.items() if k.startswith("b_")}
    tgt = g["target"].to(DEV)
    res = {}
    for compute, table in (("f32", "f32"), ("fp16", "f32")):
        m = make_model(compute, table, p)
        rend, hist = m(None, batch, 1.0, False)
        (((rend[-1]["rgb"] - tgt) ** 2).mean() + 0.01 * rend[-1]["depth"].mean()).backward()
        res[compute] = (rend[-1]["rgb"].detach().cpu(), rend[-1]["depth"].detach().cpu(), {k: v.grad.detach().cpu() for k, v in m.named_parameters() if v.grad is not None})
    close(res["f32"][0], g["f32_rgb"], 2e-4, 2e-4, "fp32 rgb"); close(res["f32"][1], g["f32_depth"], 2e-4, 2e-4, "fp32 depth")
    ref_dev_rgb = float((g["f16_rgb"] - g["f32_rgb"]).abs().max())
    e16, e32 = float((res["fp16"][0] - g["f16_rgb"]).abs().max()), float((res["fp16"][0] - g["f32_rgb"]).abs().max())
    print(f"MEASURED fp16 compute rgb: vs reference autocast {e16:.2e}, vs reference fp32 {e32:.2e} (reference autocast vs its fp32: {ref_dev_rgb:.2e})")
    assert e16 < 2e-3 and e32 <= 1.5 * ref_dev_rgb
    close(res["fp16"][1], g["f16_depth"], 5e-3, 5e-3, "fp16 depth")
    for k in g:
        if k.startswith("f16_grad."):
            n = k[9:]
            r16, r32 = g[k], g["f32_grad." + n]
            ref_dev = float((r16 - r32).norm() / r32.norm())
            got = res["fp16"][2][n]
            d16, d32 = float((got - r16).norm() / r16.norm()), float((got - r32).norm() / r32.norm())
            f32 = float((res["f32"][2][n] - r32).norm() / r32.norm())
            print(f"MEASURED fp16 compute d {n}: rel L2 vs reference autocast {d16:.2e}, vs reference fp32 {d32:.2e} (reference's own deviation {ref_dev:.2e}; fp32 mode {f32:.1e})")
            assert f32 < 5e-3, (n, f32)             # (HIP fp32 mode vs the reference fp32: 2.2e-3 on the first layer, whose operand is the hash-grid feature)
            assert d32 <= ref_dev and d16 <= 1.5 * ref_dev, (n, d16, d32, ref_dev)


def test_zip_record_precision_policy_and_loss_scale_plumbing(backend):
    """Host logic of round 4 (runs on the emulated backend too): which levels' table gradients travel as fp16 records for every
    (table_dtype, compute, table_grad_dtype) combination, the rounding-mode codes of the fused proposal networks, and ZipTrainer's loss
    scale: in fp32 compute a step with loss_scale = 1024 must move the parameters like a step with loss_scale = 1 (every gradient source
    carries the factor -- loss tail, hash decay, an auxiliary loss on the histograms -- and Adam's grad_scale undoes it)."""
    from snerf_amd import ops, zipnerf
    from snerf_amd.trainer import ZipTrainer
    assert [ops.round_mode(x) for x in (False, True, ops.F32, ops.BF16, ops.F16)] == [0, 1, 0, 1, 2]
    mk = lambda **kw: zipnerf.Model(config=None, raydist_fn='power_transformation', opaque_background=True, device=DEV, grid_log2_hashmap_size=12, **kw)
    want = {("ref", "bf16", "auto"): [True, True, True], ("ref", "f32", "auto"): [False, False, True], ("f32", "f32", "auto"): [False, False, False],
            ("f32", "fp16", "auto"): [True, True, True], ("ref", "bf16", "table"): [False, False, True], ("f16", "f32", "table"): [True, True, True],
            ("ref", "bf16", "f32"): [False, False, False], ("f32", "f32", "f16"): [True, True, True]}
    for (table, compute, tg), halves in want.items():
        m = mk(table_dtype=table, compute=compute, table_grad_dtype=tg)
        assert [m._half_records(l) for l in range(3)] == halves, (table, compute, tg)
        assert m.autograd_loss_scale == (4096.0 if compute == "fp16" else 1.0)
    with pytest.raises(ValueError):
        mk(table_grad_dtype="int8")
    assert mk(table_grad_dtype="bf16").table_grad_mode == "atomic"
    specs, p = zip_setup()
    R = 16
    g = torch.Generator().manual_seed(5)
    d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
    bx = torch.nn.functional.normalize(torch.cross(d, torch.randn(R, 3, generator=g), dim=-1), dim=-1)
    batch = {k: v.to(DEV) for k, v in dict(origins=torch.randn(R, 3, generator=g) * 0.1, directions=d, viewdirs=d, radii=2e-3 + 2e-3 * torch.rand(R, 1, generator=g),
                                            near=torch.full((R, 1), 0.1), far=torch.full((R, 1), 10.0), base_x=bx,
                                            base_y=torch.nn.functional.normalize(torch.cross(d, bx, dim=-1), dim=-1)).items()}
    target = torch.rand(R, 3, generator=g).to(DEV)
    aux = lambda hist: 0.01 * sum((h["weights"] ** 2).sum() for h in hist)
    moved = {}
    for ls in (1.0, 1024.0):
        m = make_model("f32", "f32", p)
        tr = ZipTrainer(m, lr=1e-2, eps=1e-8, loss_scale=ls)
        assert tr.loss_scale == ls
        start = m.arena.flat.clone()
        draws = m._draws(R, False, m.arena.flat.device, 7)
        loss, _ = tr.step(batch, target, rand=False, draws=draws, aux_loss_fn=aux)
        moved[ls] = (float(loss), (m.arena.flat - start).clone())
    assert abs(moved[1.0][0] - moved[1024.0][0]) <= 1e-6 * abs(moved[1.0][0])
    da, db = moved[1.0][1], moved[1024.0][1]
    assert float(da.abs().max()) > 1e-4 and float((da - db).abs().max()) <= 2e-3 * 1e-2, float((da - db).abs().max())


@pytest.mark.parametrize("compute,tol", [("bf16", 2e-2), ("fp16", 2e-3)])
def test_zip_fused_inference_mlp_matches_the_per_layer_path(backend, golden, compute, tol):
    """Inference evaluates the NeRF MLP in ONE launch (csrc/fmlp.hip fzip_fwd_kernel: activations in registers, the last hidden layer
    consumed by the rgb layer block by block, the raw density taken from the unrounded accumulator); `fused_infer = False` selects the
    seven per-layer launches of the training forward.  Same rounding points (every stored activation in the compute dtype, fp32
    accumulation, fp32 heads): the two differ by summation order only.  Also against the reference's golden (g11), and ragged row counts."""
    from snerf_amd import ops
    g = golden("g11_zip_model")
    specs, p = zip_setup()
    batch = {k[2:]: v.to(DEV) for k, v in g.items() if k.startswith("b_")}
    m = make_model(compute, "f32", p)
    net = m.nets[2]
    assert net.fused_infer_ok()
    outs = {}
    for fused in (True, False):
        net.fused_infer = fused
        with torch.no_grad():
            rend, hist = m(None, batch, 1.0, False)
        outs[fused] = (rend[-1]["rgb"], rend[-1]["depth"], hist[-1]["weights"])
    net.fused_infer = True
    for a, b, what in zip(outs[True], outs[False], ("rgb", "depth", "weights")):
        close(a, b, tol, tol, what + " fused vs per-layer")
    close(outs[True][0], g["det_rgb"], 5 * tol, 5 * tol, "rgb vs the reference")
    # the network alone on random features: ragged row counts, full-rank weights
    gen = torch.Generator().manual_seed(3)
    for k in list(p):
        if k.startswith("nerf_mlp.") and k.endswith(".weight") and "encoder" not in k:
            p[k] = torch.randn(p[k].shape, generator=gen) * (1.2 / p[k].shape[1] ** 0.5)
    m = make_model(compute, "f32", p)
    net = m.nets[2]
    for M in (1, 255, 1000):
        Fb = torch.zeros(M, 64); Fb[:, :40] = torch.randn(M, 40, generator=gen) * 0.5
        Dn = torch.zeros(M, 16); Dn[:, :9] = torch.randn(M, 9, generator=gen)
        Fb, Dn = Fb.to(net.tdt).to(DEV), Dn.to(net.tdt).to(DEV)
        with torch.no_grad():
            rgb_f, den_f = net.forward_fused(Fb, Dn, want_x=True)
            x32 = net.last_x.clone()
            F2, SB = net.alloc(M)
            F2.copy_(Fb); SB[:, net.Wd + net.Bw:net.Wd + net.Bw + 16] = Dn; SB[:, net.Wd + net.Bw + 16:] = 0
            rgb_l, den_l, _ = net.forward(F2, SB, False)
        close(rgb_f, rgb_l, tol, tol, f"raw rgb M={M}"); close(den_f, den_l, tol, tol, f"raw density M={M}")
        assert x32.shape == (M, 32) and x32.dtype == net.tdt
        close(x32, net.last_x[:, :32], tol, tol, f"x[:, :32] (semantic logits) M={M}")
        W = {k[9:]: v.double() for k, v in p.items() if k.startswith("nerf_mlp.") and "encoder" not in k}
        f64, d64 = Fb.double().cpu()[:, :40], Dn.double().cpu()[:, :9]
        h1 = torch.relu(f64 @ W["density_layer.0.weight"].t() + W["density_layer.0.bias"])
        x = h1 @ W["density_layer.2.weight"].t() + W["density_layer.2.bias"]
        h2 = torch.relu(torch.cat([x, d64], -1) @ W["lin_second_stage_0.weight"].t() + W["lin_second_stage_0.bias"])
        h3 = torch.relu(torch.cat([h2, x, d64], -1) @ W["lin_second_stage_1.weight"].t() + W["lin_second_stage_1.bias"])
        close(rgb_f, h3 @ W["rgb_layer.weight"].t() + W["rgb_layer.bias"], 5 * tol, 5 * tol, f"raw rgb vs fp64 M={M}")
        close(den_f, x[:, :1], 5 * tol, 5 * tol, f"raw density vs fp64 M={M}")


@pytest.mark.parametrize("compute,tol", [("bf16", 3e-2), ("fp16", 4e-3)])
def test_zip_fused_training_forward_matches_the_per_layer_path(backend, golden, compute, tol):
    """The training forward of the NeRF MLP as one launch (fzip_fwd_kernel<.., STORE>: every layer's output stored through the transposition
    slabs, ReLU bit masks of the two 256-wide hidden layers in the GEMMs' layout) against the seven per-layer launches (`fused_train =
    False`): the stored activations, the raw outputs and -- through the unchanged backward, which reads the stored activations and bit masks --
    every parameter gradient, on rows that are not a multiple of the 256-row tile."""
    specs, p = zip_setup()
    gen = torch.Generator().manual_seed(5)
    for k in list(p):
        if k.startswith("nerf_mlp.") and k.endswith(".weight") and "encoder" not in k:
            p[k] = torch.randn(p[k].shape, generator=gen) * (1.2 / p[k].shape[1] ** 0.5)
    m = make_model(compute, "f32", p, use_semantic=True)
    net = m.nets[2]
    M = 1000
    Fb0 = torch.zeros(M, 64); Fb0[:, :40] = torch.randn(M, 40, generator=gen) * 0.5
    Dn = torch.zeros(M, 16); Dn[:, :9] = torch.randn(M, 9, generator=gen)
    d_rgb = (torch.randn(M, 3, generator=gen) * 1e-2).to(DEV)
    d_den = (torch.randn(M, 1 + 19, generator=gen) * 1e-2).to(DEV)       # raw density + 19 semantic logits
    res = {}
    for fused in (True, False):
        net.fused_train = fused
        Fb, SB = net.alloc(M)
        Fb.copy_(Fb0.to(net.tdt)); SB[:, net.Wd + net.Bw:] = 0; SB[:, net.Wd + net.Bw:net.Wd + net.Bw + 16] = Dn.to(net.tdt)
        raw_rgb, raw_d, saved = net.forward(Fb, SB, True)
        assert (len(net._bits) == 2) if fused else True
        m.arena.grad.zero_()
        dF = net.backward(d_rgb, d_den, saved)
        res[fused] = dict(raw_rgb=raw_rgb.clone(), raw_d=raw_d.clone(), H1=saved[1].clone(), SB=saved[2][:, :512].clone(), H3=saved[3].clone(), dF=dF.clone(),
                          x=net.last_x.clone(), grad=m.arena.grad.clone())
    net.fused_train = True
    for k in ("raw_rgb", "raw_d", "H1", "SB", "H3", "x"):
        close(res[True][k], res[False][k], tol, tol, k)
    for k in ("dF", "grad"):
        a, b = res[True][k].float(), res[False][k].float()
        rel = float((a - b).norm() / b.norm())
        assert float(b.norm()) > 0 and rel < 2 * tol, (k, rel)


def test_zip_trainer_fused_loss_tail(backend):
    """ZipTrainer.step: the loss terms it reports are the oracle's loss tail (s-nerfpp/zipnerf/train.py:250-311) evaluated on the
    renderer outputs of that step, the step changes the parameters, and repeated steps on a fixed batch reduce the loss."""
    from oracle import callers as ocl
    from snerf_amd import ops
    from snerf_amd.trainer import ZipTrainer
    specs, p = zip_setup()
    R = 24
    g = torch.Generator().manual_seed(9)
    d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
    bx = torch.nn.functional.normalize(torch.cross(d, torch.randn(R, 3, generator=g), dim=-1), dim=-1)
    batch = dict(origins=torch.randn(R, 3, generator=g) * 0.1, directions=d, viewdirs=d, radii=2e-3 + 2e-3 * torch.rand(R, 1, generator=g),
                 near=torch.full((R, 1), 0.1), far=torch.full((R, 1), 10.0), base_x=bx, base_y=torch.nn.functional.normalize(torch.cross(d, bx, dim=-1), dim=-1))
    batch = {k: v.to(DEV) for k, v in batch.items()}
    target = torch.rand(R, 3, generator=g).to(DEV)
    lm = (torch.rand(R, generator=g) < 0.8).float()
    td = torch.rand(R, generator=g) * 5 + 0.5
    tg = dict(lossmult=lm.to(DEV), depth=td.to(DEV), depth_mask=(lm * (torch.rand(R, generator=g) < 0.6)).to(DEV),
              complete_mask=((1 - lm) * 1.0).to(DEV), semantic=torch.randint(0, 19, (R,), generator=g).int().to(DEV), semantic_mask=lm.to(DEV))
    m = make_model("f32", "f32", p, use_semantic=True)
    tr = ZipTrainer(m, lr=2e-3, eps=1e-8)
    start = m.arena.flat.clone()
    draws = m._draws(R, False, m.arena.flat.device, 7)
    loss0, levels = tr.step(batch, target, rand=False, draws=draws, targets=tg)
    n = lambda t: t.detach().cpu().numpy()
    fin = levels[2]
    Lr, _ = ocl.zip_loss_tail(n(fin["rgb"]), n(target), n(tg["lossmult"]), n(fin["depth"]), n(tg["depth"]), n(tg["depth_mask"]), n(tg["complete_mask"]),
                              n(fin["semantic"]), n(tg["semantic"]), n(tg["semantic_mask"]), [n(levels[l]["sdist"]) for l in range(3)],
                              [n(levels[l]["weights"]) for l in range(3)])
    got = dict(zip(ops.ZIP_LOSS_NAMES + ("hash_decay",), tr.last_losses.cpu().tolist()))
    for k in ops.ZIP_LOSS_NAMES:
        assert abs(got[k] - Lr[k]) <= 2e-5 * abs(Lr[k]) + 1e-9, (k, got[k], Lr[k])
    decay = sum(ocl.hash_decay_loss(p[pre + "encoder.embeddings"].numpy(), m.encs[i].offsets, 0.1)[0] for i, pre in enumerate(m.names))
    assert abs(got["hash_decay"] - decay) <= 1e-4 * decay and decay > 0
    assert abs(float(loss0) - (Lr["total"] + decay)) <= 2e-5 * (Lr["total"] + decay)
    assert all(Lr[k] > 0 for k in ("data", "depth", "d_complete", "sem", "interlevel", "distortion"))
    assert float((m.arena.flat - start).abs().max()) > 1e-4
    for _ in range(14):
        loss, _ = tr.step(batch, target, rand=False, draws=draws, targets=tg)
    assert float(loss) < 0.9 * float(loss0), (float(loss0), float(loss))


def test_zip_compute_extras_vs_reference_golden(backend, golden):
    """compute_extras=True (the rendering scripts' mode): acc, distance_mean, 5 / 50 / 95 % distance percentiles and the
    visualisation rays of all three levels against the reference Model run in that mode (vis_num_rays = 8)."""
    import types
    g = golden("g11_zip_model")
    specs, p = zip_setup()
    batch = {k[2:]: v.to(DEV) for k, v in g.items() if k.startswith("b_")}
    m = make_model("f32", "f32", p)
    m.config = types.SimpleNamespace(vis_num_rays=8)
    with pytest.raises(NotImplementedError):
        m(None, batch, 1.0, True)                       # parameters require grad and grad mode is on
    with torch.no_grad():
        rend, hist = m(None, batch, 1.0, True)
    tol = 2e-4
    for lvl in range(3):
        for k in ("acc", "distance_mean", "distance_percentile_5", "distance_median", "distance_percentile_95", "ray_sdist", "ray_weights", "ray_rgbs"):
            close(rend[lvl][k], g[f"x{lvl}_{k}"], 10 * tol if k in ("ray_weights",) else tol, tol, f"level {lvl} {k}")
        close(hist[lvl]["sdist"], g[f"det_sdist{lvl}"], tol, tol, f"sdist {lvl}")
    close(rend[-1]["rgb"], g["det_rgb"], tol, tol, "rgb")


def test_zip_render_image_chunks_match_one_pass(backend, golden):
    """models.py:727-813 render_image: a [H,W] frame rendered in ragged chunks equals one forward over all rays; 2-D buffers come
    back as [H,W,...], ray_* keys as one tensor per level with vis_num_rays rows."""
    import types
    from snerf_amd import zipnerf
    g = golden("g11_zip_model")
    specs, p = zip_setup()
    batch = {k[2:]: v.to(DEV) for k, v in g.items() if k.startswith("b_")}           # 20 rays -> a 4 x 5 frame
    frame = {k: v.reshape(4, 5, -1) for k, v in batch.items()}
    m = make_model("f32", "f32", p, use_semantic=True)
    cfg = types.SimpleNamespace(render_chunk_size=7, vis_num_rays=3)
    m.config = cfg
    fn = lambda rand, b: m(rand, b, train_frac=1.0, compute_extras=True)
    out = zipnerf.render_image(fn, None, frame, False, cfg)
    with torch.no_grad():
        ref, _ = m(None, batch, 1.0, True)
    assert out["rgb"].shape == (4, 5, 3) and out["depth"].shape == (4, 5) and out["semantic"].shape == (4, 5, 19)
    for k in ("rgb", "depth", "acc", "semantic", "distance_median", "distance_percentile_95"):
        # (a percentile on a flat stretch of the CDF amplifies the last-bit differences of a differently shaped GEMM batch)
        close(out[k].reshape(ref[-1][k].shape), ref[-1][k], 1e-4 if k.startswith("distance") else 1e-6, 1e-6, k)
    assert len(out["ray_sdist"]) == 3 and out["ray_sdist"][0].shape == (3, 65) and out["ray_rgbs"][2].shape == (3, 32, 3)
    # the caller's accelerate.Accelerator is accepted as it is (only its process index / count are read)
    acc = types.SimpleNamespace(local_process_index=0, num_processes=1, is_local_main_process=True)
    out2 = zipnerf.render_image(fn, acc, frame, False, cfg)
    assert torch.equal(out2["rgb"], out["rgb"]) and torch.equal(out2["depth"], out["depth"])


@pytest.mark.gpu
@pytest.mark.parametrize("R,S", [(4099, 32), (3, 1), (257, 64)])
def test_zip_percentiles_kernel_vs_oracle(R, S):
    """weighted_percentile through the C-ABI against oracle/zip.py (pinned by the g11 compute_extras vectors): opaque rays
    (weights summing to 1 -> clamped CDF), empty rays (all weight on the far fence post), zero-weight stretches (tied CDF values)."""
    from snerf_amd import ops
    g = torch.Generator().manual_seed(R + S)
    t = torch.sort(torch.rand(R, S + 1, generator=g) * 9 + 0.1, -1).values
    w = torch.rand(R, S, generator=g) ** 4
    w = w / w.sum(-1, keepdim=True) * torch.rand(R, 1, generator=g)
    w[0] = 0.0
    if R > 2:
        w[1] = w[1] / w[1].sum() * 1.0000005
        w[2, : S // 2] = 0.0
    far = t[:, -1:] + torch.rand(R, 1, generator=g)
    ps = [5, 50, 95, 0.5, 99.9]
    ref = oz.weighted_percentile(torch.cat([t, far], -1), torch.cat([w, (1 - w.sum(-1, keepdim=True)).clamp_min(0)], -1), ps)
    got = ops.zip_percentiles(t.cuda(), w.cuda(), far.cuda(), ps).cpu()
    assert got.shape == (R, 5)
    close(got, ref, 2e-5, 1e-6, "percentiles")
    assert torch.equal(got[0], far[0].expand(5) * 0 + got[0]) and bool((got[0] >= t[0, -1]).all())      # empty ray: between the last post and t_far


def _resample_case(R, S0, n, seed, dilate):
    g = torch.Generator().manual_seed(seed)
    if S0 == 1:
        sd = torch.cat([torch.zeros(R, 1), torch.ones(R, 1)], -1)
        w = torch.ones(R, 1)
    else:
        sd = torch.sort(torch.rand(R, S0 + 1, generator=g), -1).values
        sd[:, 0], sd[:, -1] = 0.0, 1.0
        w = torch.rand(R, S0, generator=g) ** 5
        w = w / w.sum(-1, keepdim=True) * torch.rand(R, 1, generator=g)
        w[0] = 0.0                                              # an empty ray: uniform after the padding
        w[1, : S0 // 2] = 0.0                                   # a dead stretch
        if S0 > 8:
            sd[2, 5] = sd[2, 4]                                 # a zero-width interval (logit -inf)
        if S0 > 8 and R > 3:
            sd[3, 1:4] = sd[3, 1]                               # three coincident posts
    pad = 1 / (2 * n)
    u_det = torch.linspace(pad, 1. - pad - 1.1920929e-07, n)
    u_max = 1.1920929e-07 + (1 - 1.1920929e-07) / n
    u_rnd = (torch.linspace(0, 1 - u_max, n) + torch.rand(R, 1, generator=g) * ((1 - u_max) / (n - 1) - 1.1920929e-07)).contiguous()
    near, far = torch.rand(R, generator=g) * 0.3 + 0.05, torch.rand(R, generator=g) * 20 + 5
    return sd, w, u_det, u_rnd, near, far


@pytest.mark.gpu
@pytest.mark.parametrize("R,S0,n,dilate,dilation", [(777, 1, 64, False, 0.0), (1031, 64, 64, True, 0.0025 + 0.5 / 64), (515, 64, 32, True, 0.0025 + 0.5 / 4096),
                                                      (70, 7, 5, True, 0.05), (64, 33, 64, False, 0.0), (3, 64, 64, True, 0.4)])
def test_zip_resample_kernel_vs_oracle(R, S0, n, dilate, dilation):
    """Stage-isolated C2 + C3 (max_dilate_weights, annealed softmax, inverse-CDF sampling, fence posts, power warp) through the C-ABI
    against oracle/zip.py (pinned by G10): the three level shapes of waymo.gin, odd sizes, a dilation wider than the intervals, with
    empty rays, dead stretches, zero-width intervals and coincident posts; deterministic centres and per-ray jitter."""
    from snerf_amd import ops
    import cpu_ops_emulation as emu
    sd, w, u_det, u_rnd, near, far = _resample_case(R, S0, n, R + S0, dilate)
    for u in (u_det, u_rnd):
        for anneal, pad in ((1.0, 0.0), (0.37, 0.01)):
            ref_s, ref_t = emu.zip_resample(sd, w, u, n, near, far, dilation, dilate, anneal, pad, -1.5)
            got_s, got_t = ops.zip_resample(sd.cuda(), w.cuda(), u.cuda(), n, near.cuda(), far.cuda(), dilation, dilate, anneal, pad, -1.5)
            got_s, got_t = got_s.cpu(), got_t.cpu()
            assert bool(torch.isfinite(got_s).all()) and bool((got_s[:, 1:] >= got_s[:, :-1]).all()) and bool((got_s >= 0).all()) and bool((got_s <= 1).all())
            if pad > 0:
                close(got_s, ref_s, 2e-5, 2e-6, "sdist")
                close(got_t, ref_t, 2e-4, 1e-5, "tdist")
            else:
                # without the padding the inverse CDF is ill-conditioned where the histogram is empty: a centre that falls on a
                # stretch with ~1e-15 of the mass divides two rounding errors ((u - c0) / (c1 - c0)), and an all-zero ray (row 0)
                # has no defined answer in the reference either (softmax of all -inf).  Everything else must agree.
                ok = ((got_s - ref_s).abs() <= 2e-6 + 2e-5 * ref_s.abs())[1:] if S0 > 1 else ((got_s - ref_s).abs() <= 2e-6 + 2e-5 * ref_s.abs())
                assert float(ok.float().mean()) > 0.998, float(ok.float().mean())


@pytest.mark.gpu
def test_zip_encode_thread_mappings_agree(golden):
    """Training and inference take different kernels for the same arithmetic: one thread per (interval, level) + GEMM-kernel MLPs when
    activations are kept, one thread per interval over all levels (and, on the proposal levels, the MLP fused behind it) otherwise."""
    g = golden("g11_zip_model")
    specs, p = zip_setup()
    batch = {k[2:]: v.cuda() for k, v in g.items() if k.startswith("b_")}
    for compute, table in (("f32", "f32"), ("bf16", "f16")):
        m = make_model(compute, table, p)
        with torch.no_grad():
            r_inf, h_inf = m(None, batch, 1.0, False)           # keep = False -> all levels per thread
        r_trn, h_trn = m(None, batch, 1.0, False)               # parameters require grad -> one level per thread
        assert r_trn[-1]["rgb"].requires_grad
        # level 0 fence posts have no network upstream: bit-identical.  The proposal levels' densities come from the fused
        # featurisation + MLP kernel at inference and from the GEMM kernels in training: same roundings, different summation order
        assert torch.equal(h_inf[0]["sdist"], h_trn[0]["sdist"])
        tol = 2e-5 if compute == "f32" else 2e-2
        for lvl in range(3):
            close(h_inf[lvl]["weights"], h_trn[lvl]["weights"].detach(), tol, tol, f"{compute} weights {lvl}")
        close(r_inf[-1]["rgb"], r_trn[-1]["rgb"].detach(), tol, tol, f"{compute} rgb")


@pytest.mark.gpu
@pytest.mark.parametrize("lvl,radius", [(0, 2.0 / 2050 / 12 ** 0.5), (0, 0.05), (2, 2.0 / 2050 / 12 ** 0.5), (2, 0.02)])
def test_zip_encode_bwd_lds_slabs_match_global_atomics(lvl, radius):
    """The LDS-privatised backward of the dense low levels (several z-slabs per level, intervals rejected per slab by a conservative
    bound on their multisamples' z-cells) against the plain global-atomic scatter of the same levels: production grids (2^21 rows),
    scattered rays, thin and very wide cones (wide cones make the bound loose: nothing may be dropped)."""
    from snerf_amd import ops, zipnerf
    m = zipnerf.Model(config=None, raydist_fn='power_transformation', opaque_background=True, compute="bf16", table_dtype="f16", init_std=0.1)
    e = m.encs[lvl]
    assert e.lds_levels >= 2 and e.lds_slabs > e.lds_levels          # at least one level takes several slabs
    R, S = 2048, 16
    g = torch.Generator().manual_seed(7 + lvl)
    d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
    up = torch.tensor([0.0, 1.0, 0.0]).expand(R, 3)
    bx = torch.nn.functional.normalize(torch.cross(d, up, dim=-1), dim=-1)
    by = torch.nn.functional.normalize(torch.cross(d, bx, dim=-1), dim=-1)
    o = torch.randn(R, 3, generator=g) * 0.3
    t = torch.sort(torch.rand(R, S + 1, generator=g) ** 2 * 12 + 0.05, -1).values.contiguous()      # inside and far outside the unit ball
    args = [t.cuda(), o.cuda(), d.cuda(), torch.full((R,), radius).cuda(), bx.cuda(), by.cuda(), None]
    k = e.lds_levels                                                   # only the levels the LDS path owns
    dF = (torch.randn(R * S, 64, generator=g) * 0.01).bfloat16().cuda()
    slabs = sum(-(-int(x) // e.lds_cells) for x in (e.offsets[1:k + 1] - e.offsets[:k]))
    g_lds = torch.zeros(e.rows, e.C, device="cuda")
    ops.zip_encode_bwd(*args, m.dev_offsets[lvl], m.dev_sizes[lvl], dF, g_lds, k, e.C, 7, 3, e.Sl, e.H, 0.35, k, e.lds_cells, slabs)
    g_glb = torch.zeros(e.rows, e.C, device="cuda")
    ops.zip_encode_bwd(*args, m.dev_offsets[lvl], m.dev_sizes[lvl], dF, g_glb, k, e.C, 7, 3, e.Sl, e.H, 0.35, 0, e.lds_cells, 0)
    rows = int(e.offsets[k])
    a, b = g_lds[:rows], g_glb[:rows]
    assert float(b.abs().max()) > 0 and float((b != 0).float().mean()) > 0.05
    err = float((a - b).abs().max())
    assert err <= 2e-5 * float(b.abs().max()), (err, float(b.abs().max()))       # summation order only
    assert float(g_lds[rows:].abs().max()) == 0 and float(g_glb[rows:].abs().max()) == 0


def test_zip_ray_gradients_vs_oracle_autograd(backend):
    """`cal_input_grad=True` (pose refinement, zipnerf/train.py:187-224; internal/models.py:491 -> gridencoder/grid.py:65-89 ->
    gridencoder.cu:199-244, 343-369): d loss / d (origins, directions, viewdirs, base_x, base_y) against torch autograd through the
    oracle, whose grid lookup is made differentiable in the POSITIONS with the reference's own dy_dx formulation (GridFn).  The loss
    touches every level (the proposal histograms too), so the position path, the erf down-weighting's std path, the direction
    encoding and the interval lengths of the compositing are all exercised."""
    from cpu_ops_emulation import GridFn
    import oracle.zip as ozm
    specs, p = zip_setup()
    R = 10
    g = torch.Generator().manual_seed(15)
    d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1) * (1 + 0.1 * torch.rand(R, 1, generator=g))
    rv = torch.randn(R, 3, generator=g)
    bx = torch.nn.functional.normalize(torch.cross(d, rv, dim=-1), dim=-1)
    base = dict(origins=torch.randn(R, 3, generator=g) * 0.1, directions=d, viewdirs=torch.nn.functional.normalize(d, dim=-1),
                radii=2e-3 + 2e-3 * torch.rand(R, 1, generator=g), near=torch.full((R, 1), 0.1), far=torch.full((R, 1), 10.0), base_x=bx,
                base_y=torch.nn.functional.normalize(torch.cross(d, bx, dim=-1), dim=-1))
    target = torch.rand(R, 3, generator=g)
    rkeys = ("origins", "directions", "viewdirs", "base_x", "base_y")

    def loss_of(rend, hist, tgt):
        return ((rend[-1]["rgb"] - tgt) ** 2).mean() + 0.1 * rend[-1]["depth"].mean() + 0.05 * sum((h["weights"] ** 2).sum() for h in hist)

    m = make_model("f32", "f32", p)
    batch = {k: (v.to(DEV).clone().requires_grad_(True) if k in rkeys else v.to(DEV)) for k, v in base.items()}
    with pytest.raises(NotImplementedError):
        m(None, batch, 1.0, False)                                     # rays that require grad need cal_input_grad=True
    rend, hist = m(None, batch, 1.0, False, cal_input_grad=True)
    loss_of(rend, hist, target.to(DEV)).backward()

    ob = {k: (v.clone().requires_grad_(True) if k in rkeys else v) for k, v in base.items()}
    pr = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    orig = ozm.grid_features
    ozm.grid_features = lambda spec, emb, means: GridFn.apply(((means.reshape(-1, 3) + 1) / 2), emb, spec.offsets, spec.S, spec.H).reshape(
        list(means.shape[:-1]) + [spec.L, spec.C])
    try:
        # The oracle is evaluated AT the fence posts the model resampled: a 1e-7 difference of a proposal weight moves a resampled
        # post by ~1e-5 (the inverse-CDF step divides by small cdf differences), and at the finest hash levels (cells of 1e-4) that
        # changes the trilinear DERIVATIVE by a fraction of a per cent -- conditioning of the algorithm, not of an implementation.
        rend_o, hist_o = oz.model_forward(pr, specs, ob, train_frac=1.0, sdist_override=[h["sdist"].detach().cpu() for h in hist])
    finally:
        ozm.grid_features = orig
    loss_of(rend_o, hist_o, target).backward()
    for k in rkeys:
        got, ref = batch[k].grad.detach().cpu(), ob[k].grad
        assert ref is not None and float(ref.abs().max()) > 0, k
        rel = float((got - ref).norm() / ref.norm())
        print(f"MEASURED ray grad {k}: rel L2 {rel:.3e} |ref| {float(ref.norm()):.3e}")
        # on the device sincosf / cbrtf / erff differ from torch's by an ulp, which the finest levels turn into ~1e-3 changes of the
        # trilinear derivative: the bound of the table gradients of those levels
        assert rel < (3e-2 if backend == "hip" else 1e-4), (k, rel)
    named = dict(m.named_parameters())
    for k in ("nerf_mlp.lin_second_stage_0.weight", "prop_mlp_0.density_layer.0.weight"):       # parameter gradients are unaffected
        rel = float((named[k].grad.cpu() - pr[k].grad).norm() / pr[k].grad.norm())
        assert rel < 5e-3, (k, rel)


@pytest.mark.gpu
@pytest.mark.parametrize("lvl,half", [(1, False), (2, False), (2, True)])
def test_zip_encode_ray_bwd_kernel_vs_oracle(lvl, half):
    """snerf_zip_encode_ray_bwd stage-isolated: random feature gradients on random intervals, C = 1 (proposal grid) and C = 4 (NeRF grid),
    fp32 and fp16 tables, against autograd through the oracle's cast_rays / contract_mean_std / grid (dy_dx) / erf weighting on
    IDENTICAL inputs."""
    import cpu_ops_emulation as E
    from snerf_amd import ops
    specs, p = zip_setup()
    spec = specs[lvl]
    emb = p[("prop_mlp_1." if lvl == 1 else "nerf_mlp.") + "encoder.embeddings"] * 30      # trained-like magnitudes
    if half:
        emb = emb.half().float()
    R, S, n = 64, 16, 7
    g = torch.Generator().manual_seed(3)
    d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
    bx = torch.nn.functional.normalize(torch.cross(d, torch.randn(R, 3, generator=g), dim=-1), dim=-1)
    by = torch.nn.functional.normalize(torch.cross(d, bx, dim=-1), dim=-1)
    o = torch.randn(R, 3, generator=g) * 0.1
    radii = 2e-3 + 2e-3 * torch.rand(R, generator=g)
    tdist = torch.sort(torch.rand(R, S + 1, generator=g) * 5 + 0.2, -1)[0]
    degj = torch.rand(R, S, n, generator=g)
    L, C = spec.L, spec.C
    gf = torch.randn(R * S, 64, generator=g)
    if half:
        gf = gf.bfloat16().float()
    offs, gs = torch.from_numpy(spec.offsets.astype(np.int32)), torch.from_numpy(spec.res.astype(np.int32))
    ref = [torch.zeros(R, 3) for _ in range(4)]
    E.zip_encode_ray_bwd(tdist, o, d, radii, bx, by, degj, emb, offs, gs, gf, L, C, n, 3, spec.S, spec.H, 0.35, *ref)
    got = [torch.zeros(R, 3, device="cuda") for _ in range(4)]
    c = lambda t: t.cuda().contiguous()
    ops.zip_encode_ray_bwd(c(tdist), c(o), c(d), c(radii), c(bx), c(by), c(degj), c(emb.half() if half else emb), c(offs), c(gs),
                           c(gf.bfloat16() if half else gf), L, C, n, 3, spec.S, spec.H, 0.35, *got)
    for name, a, b in zip(("origins", "directions", "base_x", "base_y"), got, ref):
        rel = float((a.cpu() - b).norm() / b.norm())
        print(f"MEASURED zip_encode_ray_bwd level {lvl} half {half} d {name}: rel L2 {rel:.3e}")
        assert rel < 2e-2, (name, rel)        # finest levels: 1e-7 position differences (sincosf / cbrtf vs torch) -> 1e-3 of the derivative


def test_loss_scaler_policy_is_gradscalers():
    """LossScaler = torch.cuda.amp.GradScaler's update rule (what accelerate wraps around the reference's fp16 training, zipnerf/train.py:44):
    backoff on an overflow (and the step is skipped), growth after `growth_interval` CONSECUTIVE clean steps; checked against torch's own
    `_amp_update_scale_` on the CPU for a random overflow pattern."""
    from snerf_amd.trainer import LossScaler
    sc = LossScaler(init_scale=1024.0, growth_interval=4)
    scale, tracker = torch.full((1,), 1024.0), torch.zeros(1, dtype=torch.int32)
    rng = np.random.default_rng(3)
    skipped = 0
    for i in range(200):
        bad = bool(rng.random() < 0.15)
        assert sc.update(bad) == bad
        skipped += bad
        torch._amp_update_scale_(scale, tracker, torch.full((1,), float(bad)), 2.0, 0.5, 4)
        assert sc.scale == float(scale) and sc.good_steps == int(tracker), i
    assert sc.skipped_steps == skipped > 0
    other = LossScaler()
    other.load_state_dict(sc.state_dict())
    assert other.state_dict() == sc.state_dict()
    with pytest.raises(ValueError):
        LossScaler(backoff_factor=1.5)


@pytest.mark.gpu
def test_nonfinite_flag_and_the_table_gradients_overflow_mark():
    """snerf_nonfinite_flag (the found-inf pass of the dynamic loss scaler) on unaligned spans with the bad element at the head, in
    the vector body and in the tail; and the binned table gradient of an overflowed feature gradient (an Inf or a NaN in d features,
    which fixed-point records cannot carry) marks itself by a NaN in its first element, so that the same pass sees it."""
    from snerf_amd import ops, zipnerf
    base = torch.randn(100_003, device="cuda")
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    for off in (0, 1, 2, 3):
        for n in (0, 1, 5, 1027, 100_003 - 3):
            v = base[off:off + n]
            flag.zero_()
            ops.nonfinite_flag(v, flag)
            assert int(flag) == 0, (off, n)
            for pos in sorted({0, n // 2, n - 1} - {-1}):
                if n == 0:
                    continue
                for bad in (float("inf"), float("-inf"), float("nan")):
                    w = v.clone() if off == 0 else base.clone()[off:off + n]
                    w[pos] = bad
                    flag.zero_()
                    ops.nonfinite_flag(w, flag)
                    assert int(flag) == 1, (off, n, pos, bad)
    big = torch.full((1 << 22,), 3.0e38, device="cuda")       # large finite values are not flagged
    flag.zero_()
    ops.nonfinite_flag(big, flag)
    assert int(flag) == 0
    m = zipnerf.Model(config=None, raydist_fn='power_transformation', opaque_background=True, compute="fp16", table_dtype="f16", grid_log2_hashmap_size=16)
    lvl = 2
    e = m.encs[lvl]
    R, S, n = 300, 32, 7
    g = torch.Generator().manual_seed(11)
    d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
    bx = torch.nn.functional.normalize(torch.cross(d, torch.randn(R, 3, generator=g), dim=-1), dim=-1)
    by = torch.nn.functional.normalize(torch.cross(d, bx, dim=-1), dim=-1)
    c = lambda t: t.cuda().contiguous()
    o, radii = c(torch.randn(R, 3, generator=g) * 0.3), c(2e-3 + 2e-3 * torch.rand(R, generator=g))
    tdist = c(torch.sort(torch.rand(R, S + 1, generator=g) * 6 + 0.05, -1)[0])
    degj = c(torch.rand(R, S, n, generator=g))
    d, bx, by = c(d), c(bx), c(by)
    ks, g64_rows, lrows = ops.zip_bin_plan(e.offsets, e.C, R * S * n * 8)
    tail = (e.L, e.C, n, 3, e.Sl, e.H, m.std_scale)
    for bad, half in ((None, True), (float("inf"), True), (float("nan"), True), (float("-inf"), False)):
        dF = c(torch.randn(R * S, m.nets[lvl].Fw, generator=g) * 1e-2).half()
        if bad is not None:
            dF[R * S // 3, 5] = bad
        gt = torch.zeros(e.rows, e.C, device="cuda")
        ops.zip_encode_bwd_binned(tdist, o, d, radii, bx, by, degj, m.dev_offsets[lvl], m.dev_sizes[lvl], dF, gt, *tail, ks, g64_rows, lrows, half_records=half)
        flag.zero_()
        ops.nonfinite_flag(gt.view(-1), flag)
        assert int(flag) == (0 if bad is None else 1), (bad, half)
        if bad is not None:
            assert bool(torch.isnan(gt.view(-1)[0]))


@pytest.mark.gpu
def test_zip_trainer_dynamic_loss_scale_skips_overflowed_steps():
    """ZipTrainer(loss_scale="dynamic") on the fp16 compute mode: started from a scale the fp16 gradient buffers cannot hold (2^40),
    every overflowed step is skipped WHOLE -- parameters, Adam moments and the step count bit-for-bit untouched, gradients zeroed --
    and halves the scale until the backward fits; from there the steps update and the loss falls.  With a scale that fits from the
    start the dynamic trainer and the static one are the same arithmetic: identical parameters after 3 steps."""
    from snerf_amd import zipnerf
    from snerf_amd.trainer import ZipTrainer, LossScaler
    R = 1024
    g = torch.Generator().manual_seed(4)
    d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
    bx = torch.nn.functional.normalize(torch.cross(d, torch.tensor([0.0, 1.0, 0.0]).expand(R, 3), dim=-1), dim=-1)
    by = torch.nn.functional.normalize(torch.cross(d, bx, dim=-1), dim=-1)
    batch = {k: v.cuda() for k, v in dict(origins=torch.randn(R, 3, generator=g) * 0.05, directions=d, viewdirs=d, radii=torch.full((R, 1), 5e-4),
                                          near=torch.full((R, 1), 0.1), far=torch.full((R, 1), 10.0), base_x=bx, base_y=by).items()}
    tgt = torch.rand(R, 3, generator=g).cuda()
    mk = lambda: zipnerf.Model(config=None, raydist_fn='power_transformation', opaque_background=True, compute="fp16", table_dtype="f16",
                               grid_log2_hashmap_size=14, init_std=0.1)
    torch.manual_seed(0)
    m = mk()
    tr = ZipTrainer(m, lr=2e-3, eps=1e-8, loss_scale=LossScaler(init_scale=2.0 ** 40, growth_interval=1000))
    draws = m._draws(R, False, m.arena.flat.device, 7)
    start, losses, scales = m.arena.flat.clone(), [], []
    for i in range(40):
        before, t0, skipped0 = m.arena.flat.clone(), tr.t, tr.scaler.skipped_steps
        loss, _ = tr.step(batch, tgt, rand=False, draws=draws)
        scales.append(tr.scaler.scale)
        if tr.scaler.skipped_steps > skipped0:
            assert torch.equal(m.arena.flat, before) and tr.t == t0 and not bool(m.arena.grad.any())
            assert not bool(tr.m.any()) or i > 0
        else:
            assert tr.t == t0 + 1
            losses.append(float(loss))
    sk = tr.scaler.skipped_steps
    print(f"MEASURED dynamic loss scale from 2^40: {sk} skipped steps, settled at 2^{int(np.log2(tr.scaler.scale))}; loss {losses[0]:.5f} -> {losses[-1]:.5f} over {len(losses)} steps")
    assert 5 <= sk < 35 and tr.t == 40 - sk and tr.scaler.scale == 2.0 ** (40 - sk)
    assert bool(torch.isfinite(m.arena.flat).all()) and bool(torch.isfinite(tr.m).all()) and bool(torch.isfinite(tr.v).all())
    assert len(losses) >= 5 and losses[-1] < losses[0] and not torch.equal(m.arena.flat, start)
    # a scale that fits: dynamic == static, bit for bit
    finals = []
    for ls in (4096.0, LossScaler(init_scale=4096.0)):
        torch.manual_seed(0)
        m = mk()
        for net in m.nets:
            net.deterministic = True              # (fixed-order folds instead of fp32 atomics: the two runs are then comparable bit for bit)
        tr = ZipTrainer(m, lr=2e-3, eps=1e-8, loss_scale=ls)
        draws = m._draws(R, False, m.arena.flat.device, 7)
        for _ in range(3):
            tr.step(batch, tgt, rand=False, draws=draws)
        finals.append(m.arena.flat.clone())
        if tr.scaler is not None:
            assert tr.scaler.skipped_steps == 0 and tr.scaler.good_steps == 3
    assert torch.equal(finals[0], finals[1])
    with pytest.raises(ValueError):
        ZipTrainer(mk(), loss_scale="auto")


@pytest.mark.gpu
@pytest.mark.parametrize("lvl,gscale", [(0, 1e-3), (2, 1e-3), (2, 1e-11), (0, 3e3)])
def test_zip_table_gradient_binned_is_exact_and_bit_reproducible(lvl, gscale, monkeypatch):
    """The "binned" table gradient (records partitioned by destination, per-bin LDS accumulation in 64-bit fixed point) against the
    atomic scatter on identical inputs at the production grid sizes (2^21-row hashed levels, several row ranges and -- with a lowered
    records-per-bin target -- replicated dense levels): equal to fp32 rounding, and bit-identical run to run.  `gscale`: magnitude of
    d loss / d features -- the fixed-point grid follows it (snerf_zip_bin_scale), so a loss-scaled 1e-11 gradient (mean over 65 536
    rays x sample weight x 1 / n: what the fine levels of a real step see) is as exact as a 1e-3 one."""
    from snerf_amd import ops, zipnerf
    m = zipnerf.Model(config=None, raydist_fn='power_transformation', opaque_background=True, compute="bf16", table_dtype="f16")
    e = m.encs[lvl]
    R, S, n = 1536 + (5 if gscale > 1 else 0), (64 if lvl < 2 else 32), 7          # (one case with a ragged last workgroup)
    g = torch.Generator().manual_seed(7)
    d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
    bx = torch.nn.functional.normalize(torch.cross(d, torch.randn(R, 3, generator=g), dim=-1), dim=-1)
    by = torch.nn.functional.normalize(torch.cross(d, bx, dim=-1), dim=-1)
    c = lambda t: t.cuda().contiguous()
    o, radii = c(torch.randn(R, 3, generator=g) * 0.3), c(2e-3 + 2e-3 * torch.rand(R, generator=g))
    tdist = c(torch.sort(torch.rand(R, S + 1, generator=g) * 6 + 0.05, -1)[0])
    degj = c(torch.rand(R, S, n, generator=g))
    d, bx, by = c(d), c(bx), c(by)
    Fw = m.nets[lvl].Fw
    dF = c(torch.randn(R * S, Fw, generator=g) * gscale).bfloat16()
    common = (tdist, o, d, radii, bx, by, degj, m.dev_offsets[lvl], m.dev_sizes[lvl], dF)
    tail = (e.L, e.C, n, 3, e.Sl, e.H, m.std_scale)
    ref = torch.zeros(e.rows, e.C, device="cuda")
    ops.zip_encode_bwd(*common, ref, *tail, 0, 0, 0)                                   # fp32 global atomics, every level
    monkeypatch.setattr(ops, "ZB_TARGET", 40_000)                                      # forces replicas (K > 1) on the dense levels
    ks, g64_rows, lrows = ops.zip_bin_plan(e.offsets, e.C, R * S * n * 8)
    assert max(ks) > 1 and g64_rows > 0
    outs = []
    for _ in range(2):
        gt = torch.zeros(e.rows, e.C, device="cuda")
        ops.zip_encode_bwd_binned(*common, gt, *tail, ks, g64_rows, lrows)
        outs.append(gt)
    assert torch.equal(outs[0], outs[1]), "binned table gradient must be bit-reproducible"
    # the LDS-staged record writer (default) and the direct one produce the same records in another order: identical sums
    monkeypatch.setattr(ops, "ZIP_BIN_STAGED", False)
    gd = torch.zeros(e.rows, e.C, device="cuda")
    ops.zip_encode_bwd_binned(*common, gd, *tail, ks, g64_rows, lrows)
    monkeypatch.setattr(ops, "ZIP_BIN_STAGED", True)
    assert torch.equal(outs[0], gd), "staged and direct record writers must give bit-identical gradients"
    # round 3: the training forward counts the records itself (snerf_zip_encode_fwd_count = featurisation + pass 0): same features as
    # the plain forward, same per-bin counts as the backward's own count pass, same gradient bit for bit
    tab = m._table(lvl)
    f0 = torch.zeros(R * S, Fw, device="cuda", dtype=dF.dtype)
    f1 = torch.zeros_like(f0)
    enc_args = (tdist, o, d, radii, bx, by, degj, tab, m.dev_offsets[lvl], m.dev_sizes[lvl])
    ops.zip_encode_fwd(*enc_args, f0, *tail, levels_per_thread=1)
    counts, wgo = ops.zip_encode_fwd_count(*enc_args, f1, *tail, ks, lrows)
    assert torch.equal(f0, f1), "the counting forward must produce the plain forward's features"
    from snerf_amd import _lib
    import numpy as np
    c0 = torch.zeros_like(counts)
    w0 = torch.empty_like(wgo)
    ksa, lra = np.asarray(ks, dtype=np.int32), np.asarray(lrows, dtype=np.int32)
    p_ = lambda t: t.data_ptr()
    _lib.call("snerf_zip_encode_bwd_binned", 0, p_(tdist), p_(o), p_(d), p_(radii), p_(bx), p_(by), p_(degj), p_(m.dev_offsets[lvl]), p_(m.dev_sizes[lvl]),
              p_(dF), dF.stride(0), None, R, S, e.L, e.C, n, 3, float(e.Sl), int(e.H), float(m.std_scale), ops._zip_dt(dF), ksa.ctypes.data, lra.ctypes.data,
              p_(c0), p_(w0), None, None, None, 0, None, 0, None, torch.cuda.current_stream().cuda_stream)
    assert torch.equal(counts, c0), "forward-side record counts differ from the backward's count pass"
    assert int(counts.sum()) > 0
    f2 = torch.zeros(R * S, Fw, device="cuda", dtype=dF.dtype)                          # fp32 table: the kernel's generic gather branch
    c32, _ = ops.zip_encode_fwd_count(tdist, o, d, radii, bx, by, degj, tab.float().contiguous(), m.dev_offsets[lvl], m.dev_sizes[lvl], f2, *tail, ks, lrows)
    assert torch.equal(c32, c0)
    gp = torch.zeros(e.rows, e.C, device="cuda")
    ops.zip_encode_bwd_binned(*common, gp, *tail, ks, g64_rows, lrows, precounted=(counts, wgo))
    assert torch.equal(outs[0], gp), "gradient with the forward's counts must equal the gradient with the backward's own count pass"
    rel = float((outs[0] - ref).norm() / ref.norm())
    print(f"MEASURED binned vs atomic table gradient (grid {lvl}): rel L2 {rel:.3e}, K per level {ks}")
    assert float(ref.norm()) > 0 and rel < 2e-6, rel
    nz = ref != 0
    assert float(((outs[0] - ref).abs()[nz] / ref.abs()[nz]).median()) < 1e-6
    # half records (table_grad_dtype="table" on a halved table / "f16"): fp16 values scaled by the launch's exponent, 10 / 4 bytes per record.
    # Every contribution is rounded to 11 bits once; sums exact: bit-reproducible, staged == direct, and within 2^-11 / sqrt(#addends) of
    # the fp32 records -- whatever the gradient's magnitude (1e-11 and 3e3 are cases of this test)
    hs = []
    for staged in (True, False, True):
        monkeypatch.setattr(ops, "ZIP_BIN_STAGED", staged)
        gh = torch.zeros(e.rows, e.C, device="cuda")
        ops.zip_encode_bwd_binned(*common, gh, *tail, ks, g64_rows, lrows, half_records=True, precounted=(counts, wgo) if len(hs) == 2 else None)
        hs.append(gh)
    assert torch.equal(hs[0], hs[1]) and torch.equal(hs[0], hs[2]), "half records: staged / direct / precounted must be bit-identical"
    relh = float((hs[0] - outs[0]).norm() / outs[0].norm())
    nzh = outs[0] != 0
    med = float(((hs[0] - outs[0]).abs()[nzh] / outs[0].abs()[nzh]).median())
    print(f"MEASURED half-record table gradient vs fp32 records (grid {lvl}, |g| ~ {gscale:g}): rel L2 {relh:.3e}, median entry {med:.3e}")
    assert relh < 3e-4 and med < 2.0 ** -11, (relh, med)           # one rounding to 11 bits per contribution (<= 2^-12 relative each), and they average


def test_binned_table_gradient_refuses_tables_with_more_row_ranges_than_bins():
    """at grid_log2_hashmap_size >= 23 (C = 4) / 25 (C = 1) a level has more than ZB_NBMAX = 1024 row ranges; the kernels'
    1024-entry histograms would be overrun.  The plan raises, and the C entry returns a bad-argument status before launching anything."""
    import numpy as np
    from snerf_amd import _lib, ops
    offs = np.array([0, 4913, 4913 + (1 << 23)], dtype=np.int64)
    with pytest.raises(ValueError, match="row ranges"):
        ops.zip_bin_plan(offs, 4, 1 << 20)
    ks, g64, rows = ops.zip_bin_plan(np.array([0, 4913, 4913 + (1 << 21)]), 4, 1 << 20)
    assert rows == [4913, 1 << 21] and all(((r + 4095) // 4096) * k <= ops.ZB_NBMAX for r, k in zip(rows, ks))
    k2 = np.array([1, 1], dtype=np.int32)
    big = np.array([4913, 1 << 23], dtype=np.int32)
    with pytest.raises(_lib.SnerfHipError, match="bad argument"):
        _lib.call("snerf_zip_encode_bwd_binned", 0, None, None, None, None, None, None, None, None, None, None, 8, None, 16, 4, 2, 4, 7, 3, 0.5, 16, 0.35,
                  1, k2.ctypes.data, big.ctypes.data, 1, 1, None, None, None, 0, None, 0, None, None)


@pytest.mark.gpu
@pytest.mark.parametrize("L,hidden,P,dt", [(6, 64, 70001, "bf16"), (8, 64, 4096, "bf16"), (8, 64, 33333, "f32"), (12, 32, 1000, "bf16"), (16, 64, 777, "f32"),
                                          (6, 64, 50001, "fp16"), (12, 64, 999, "fp16")])
def test_fused_proposal_mlp_train_kernels(L, hidden, P, dt):
    """snerf_zip_prop_mlp_fwd / _bwd (the proposal MLP of a training step in one launch each way, hidden activations recomputed in the
    backward) against the torch restatement with the GEMM route's rounding points (tests/cpu_ops_emulation.py); ragged interval counts,
    L <= 8 and L <= 16 instantiations, fewer than 64 hidden units; parameter gradients accumulate and are bit-reproducible."""
    from snerf_amd import ops
    import cpu_ops_emulation as E
    g = torch.Generator().manual_seed(L * 1000 + hidden)
    tdt = {"bf16": torch.bfloat16, "fp16": torch.float16, "f32": torch.float32}[dt]
    rnd = {"bf16": 1, "fp16": 2, "f32": 0}[dt]                                       # ops.round_mode's codes
    Fw = (L + 7) // 8 * 8
    F = torch.zeros(P, Fw)
    F[:, :L] = torch.randn(P, L, generator=g) * 0.5
    F = F.to(tdt)
    w1 = torch.randn(hidden, L, generator=g) * 0.4
    b1 = torch.randn(hidden, generator=g) * 0.2
    w2 = torch.randn(1, hidden, generator=g) * 0.3
    b2 = torch.randn(1, generator=g)
    d_raw = torch.randn(P, generator=g) * 1e-3
    ref_raw = E.zip_prop_mlp_fwd(F, L, w1, b1, w2, b2, rnd)
    gref = [torch.full_like(t, 0.5) for t in (w1, b1, w2, b2)]                       # (the gradients ACCUMULATE)
    ref_dF = E.zip_prop_mlp_bwd(F, d_raw, L, w1, b1, w2, b2, rnd, *gref)
    c = lambda t: t.cuda().contiguous()
    Fc, pc, dc = c(F), [c(t) for t in (w1, b1, w2, b2)], c(d_raw)
    if Fw > L:
        Fc[:, L:] = float("nan")                                                     # padding columns are not read as data
    raw = ops.zip_prop_mlp_fwd(Fc, L, *pc, rnd)
    err = float((raw.cpu() - ref_raw).abs().max() / ref_raw.abs().max())
    print(f"MEASURED fused proposal MLP L={L} hidden={hidden} {dt}: forward max rel err {err:.2e}")
    assert err < (2e-3 if rnd else 1e-5), err                                       # bf16: a hidden activation may round the other way (fp32 sum order)
    outs = []
    for _ in range(2):
        gg = [torch.full_like(t, 0.5).cuda() for t in (w1, b1, w2, b2)]
        dF = ops.zip_prop_mlp_bwd(Fc, dc, L, *pc, rnd, *gg)
        outs.append((dF, gg))
    assert torch.equal(outs[0][0], outs[1][0]) and all(torch.equal(a, b) for a, b in zip(outs[0][1], outs[1][1])), "must be bit-reproducible"
    dF, gg = outs[0]
    assert dF.dtype == tdt and dF.shape == (P, Fw) and float(dF[:, L:].float().abs().max() if Fw > L else 0.0) == 0.0
    e_df = float((dF.float().cpu() - ref_dF.float()).norm() / ref_dF.float().norm())
    assert e_df < (1e-2 if rnd else 1e-5), e_df
    for name, a, b in zip(("w1", "b1", "w2", "b2"), gg, gref):
        e = float((a.cpu() - b).norm() / (b - 0.5).norm().clamp_min(1e-12))
        print(f"MEASURED fused proposal MLP d {name}: rel L2 {e:.2e}")
        assert e < (1e-2 if rnd else 2e-5), (name, e)
