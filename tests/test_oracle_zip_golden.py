"""Pin oracle/zip.py (path C around the hash grid) against vectors captured from the imported reference
(oracle/gen_golden_zip.py).  CPU only."""
import numpy as np
import torch

from oracle import zip as oz


def close(a, b, rtol=1e-6, atol=1e-6, what=""):
    a = a.double(); b = b.double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a - b).abs(); tol = atol + rtol * b.abs()
    assert bool((err <= tol).all()), f"{what}: max err {err.max().item():.3e}"


def test_g10_stages(golden):
    g = golden("g10_warp")
    close(oz.s_to_t(g["s"], g["near"], g["far"], -1.5), g["t"], 1e-6, 1e-7, "s_to_t")
    g = golden("g10_dilate")
    td, wd = oz.max_dilate_weights(g["t"], g["w"], float(g["dilation"]), (0.0, 1.0))
    close(td, g["t_dilate"], 0, 0, "t_dilate"); close(wd, g["w_dilate"], 1e-6, 1e-8, "w_dilate")
    g = golden("g10_sample_intervals")
    sd, _ = oz.sample_intervals(g["t"], g["logits"], oz.det_centers_u(16))
    close(sd, g["sdist_det"], 1e-6, 1e-7, "sample_intervals det")
    sd, _ = oz.sample_intervals(g["t"], g["logits"], oz.rand_u(16, g["jitter"]))
    close(sd, g["sdist_rand"], 1e-6, 1e-7, "sample_intervals rand")
    g = golden("g10_cast_rays")
    b = {k[2:]: v for k, v in g.items() if k.startswith("b_")}
    m, s = oz.cast_rays(g["tdist"], b["origins"], b["directions"], b["radii"], b["base_x"], b["base_y"], None)
    close(m, g["means_det"], 1e-6, 1e-6, "cast means"); close(s, g["stds_det"], 1e-6, 1e-9, "cast stds")
    m, s = oz.cast_rays(g["tdist"], b["origins"], b["directions"], b["radii"], b["base_x"], b["base_y"], g["deg_jitter"])
    close(m, g["means_rand"], 1e-6, 1e-6, "cast means rand")
    g = golden("g10_contract")
    z, so = oz.contract_mean_std(g["x"], g["std"])
    close(z, g["z"], 1e-6, 1e-7, "contract z"); close(so, g["std_out"], 1e-6, 1e-9, "contract std")
    g = golden("g10_render")
    w = oz.compute_alpha_weights(g["density"], g["tdist"], g["dirs"], True)
    close(w, g["weights"], 1e-6, 1e-7, "alpha weights")
    close(oz.compute_alpha_weights(g["density"], g["tdist"], g["dirs"], False), g["weights_noopaque"], 1e-6, 1e-7, "alpha weights (not opaque)")
    r = oz.volumetric_rendering(g["rgbs"], g["weights"], g["tdist"], 1.0)
    close(r["rgb"], g["rgb"], 1e-6, 1e-6, "rgb"); close(r["depth"], g["depth"], 1e-6, 1e-6, "depth")
    r = oz.volumetric_rendering(g["rgbs"], g["weights_noopaque"], g["tdist"], 0.5)
    close(r["rgb"], g["rgb_noopaque"], 1e-6, 1e-6, "rgb (bg 0.5)"); close(r["depth"], g["depth_noopaque"], 1e-6, 1e-6, "depth (not opaque)")


def zip_setup():
    import importlib.util, os, sys
    spec = importlib.util.spec_from_file_location("gen_golden_zip", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "gen_golden_zip.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    specs = m.small_specs()
    return specs, m.formula_params(oz.param_shapes(specs))


def test_g11_model_forward(golden):
    g = golden("g11_zip_model")
    specs, p = zip_setup()
    batch = {k[2:]: v for k, v in g.items() if k.startswith("b_")}
    rend, hist = oz.model_forward(p, specs, batch, train_frac=1.0)
    for lvl in range(3):
        close(hist[lvl]["sdist"], g[f"det_sdist{lvl}"], 1e-5, 1e-6, f"det sdist {lvl}")
        close(hist[lvl]["weights"], g[f"det_weights{lvl}"], 1e-4, 1e-6, f"det weights {lvl}")
    close(rend[-1]["rgb"], g["det_rgb"], 1e-5, 1e-5, "det rgb"); close(rend[-1]["depth"], g["det_depth"], 1e-5, 1e-5, "det depth")
    close(rend[0]["depth"], g["det_depth0"], 1e-5, 1e-5, "det depth level 0")
    rend, hist = oz.model_forward(p, specs, batch, train_frac=float(g["rand_train_frac"]), jitters=[g[f"jitter{i}"] for i in range(3)],
                                  deg_jitters=[g[f"deg_jitter{i}"] for i in range(3)])
    for lvl in range(3):
        close(hist[lvl]["sdist"], g[f"rand_sdist{lvl}"], 1e-5, 1e-6, f"rand sdist {lvl}")
    close(rend[-1]["rgb"], g["rand_rgb"], 1e-5, 1e-5, "rand rgb"); close(rend[-1]["depth"], g["rand_depth"], 1e-5, 1e-5, "rand depth")


def test_g24_near_bound_annealing(golden):
    """near_anneal_rate (models.py:47-48, 147-158, 170-171, 182-186, 207-213): first interval [clip(1 - train_frac / rate, 0, init), 1]."""
    g = golden("g24_zip_near_anneal")
    specs, p = zip_setup()
    batch = {k[2:]: v for k, v in g.items() if k.startswith("b_")}
    for tag, first in (("a", 0.6), ("b", 0.95)):
        rend, hist = oz.model_forward(p, specs, batch, train_frac=float(g[tag + "_train_frac"]), near_anneal_rate=float(g["near_anneal_rate"]),
                                      near_anneal_init=float(g["near_anneal_init"]))
        assert abs(float(hist[0]["sdist"][0, 0]) - first) < 1e-6
        for lvl in range(3):
            close(hist[lvl]["sdist"], g[f"{tag}_sdist{lvl}"], 1e-5, 1e-6, f"{tag} sdist {lvl}")
            close(hist[lvl]["weights"], g[f"{tag}_weights{lvl}"], 1e-4, 1e-6, f"{tag} weights {lvl}")
        close(rend[-1]["rgb"], g[tag + "_rgb"], 1e-5, 1e-5, tag + " rgb"); close(rend[-1]["depth"], g[tag + "_depth"], 1e-5, 1e-5, tag + " depth")


def test_g25_glo_vectors(golden):
    """num_glo_features > 0 (models.py:44-45, 75-77, 131-139, 454-459, 620-630): the restated GLO branch -- forward with the embedding rows
    and with zero_glo, and torch autograd through the oracle against the reference's own parameter gradients (the GLO MLP, the
    embedding table, the layers around the modulation)."""
    g = golden("g25_zip_glo")
    specs, _ = zip_setup()
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("gen_golden_zip", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "gen_golden_zip.py"))
    gz = importlib.util.module_from_spec(spec); spec.loader.exec_module(gz)
    Fg, E = int(g["num_glo_features"]), int(g["num_glo_embeddings"])
    p = gz.formula_params(oz.param_shapes(specs, num_glo_features=Fg, num_glo_embeddings=E, zero_glo=False))
    p["glo_vecs.weight"], p["nerf_mlp.lin_glo_0.weight"], p["nerf_mlp.lin_glo_1.weight"] = g["glo_vecs"], g["lin_glo_0_weight"], g["lin_glo_1_weight"]
    batch = {k[2:]: v for k, v in g.items() if k.startswith("b_")}
    for tag, zero in (("emb", False), ("zero", True)):
        pp = {k: v.clone().requires_grad_(not k.endswith("embeddings")) for k, v in p.items()}
        rend, hist = oz.model_forward(pp, specs, batch, train_frac=1.0, num_glo_features=Fg, zero_glo=zero)
        loss = ((rend[-1]["rgb"] - g["target"]) ** 2).mean() + 0.01 * rend[-1]["depth"].mean()
        loss.backward()
        close(rend[-1]["rgb"], g[tag + "_rgb"], 1e-5, 1e-6, tag + " rgb"); close(rend[-1]["depth"], g[tag + "_depth"], 1e-5, 1e-5, tag + " depth")
        for k in g:
            if k.startswith(tag + "_grad."):
                n = k[len(tag) + 6:]
                got, want = pp[n].grad, g[k]
                assert float((got - want).norm() / (want.norm() + 1e-30)) < 1e-5, (tag, n)
    assert float((g["emb_rgb"] - g["zero_rgb"]).abs().max()) > 1e-3          # the embedding rows matter


def test_g11_model_forward_with_semantic_head(golden):
    """use_semantic: softmax(x[..., 1:20]) of the density network, composited with the detached weights (models.py:594-597,
    render.py:237-241) -- against the reference Model run with the semantic head enabled."""
    g = golden("g11_zip_model")
    specs, p = zip_setup()
    batch = {k[2:]: v for k, v in g.items() if k.startswith("b_")}
    rend, _ = oz.model_forward(p, specs, batch, train_frac=1.0, use_semantic=True)
    close(rend[-1]["rgb"], g["sem_rgb"], 1e-5, 1e-5, "rgb with the semantic head on")
    close(rend[-1]["semantic"], g["sem_semantic"], 1e-5, 1e-6, "semantic")
    assert "semantic" not in rend[0] and "semantic" not in rend[1]


EXTRA_KEYS = ("acc", "distance_mean", "distance_percentile_5", "distance_median", "distance_percentile_95", "ray_sdist", "ray_weights", "ray_rgbs")


def test_g11_model_forward_compute_extras(golden):
    """compute_extras=True (render.py:243-267, models.py:316-346): acc, log-space distance mean, 5/50/95 % distance percentiles and
    the visualisation rays of every level -- against the reference Model run in that mode (vis_num_rays = 8)."""
    g = golden("g11_zip_model")
    specs, p = zip_setup()
    batch = {k[2:]: v for k, v in g.items() if k.startswith("b_")}
    rend, _ = oz.model_forward(p, specs, batch, train_frac=1.0, compute_extras=True, vis_num_rays=8)
    for lvl in range(3):
        for k in EXTRA_KEYS:
            close(rend[lvl][k], g[f"x{lvl}_{k}"], 2e-5, 2e-6, f"level {lvl} {k}")
    assert g["x0_ray_sdist"].shape == (8, 65) and g["x2_ray_rgbs"].shape == (8, 32, 3) and g["x1_ray_rgbs"].shape == (8, 64, 3)
