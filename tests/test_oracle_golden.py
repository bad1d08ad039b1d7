"""Pin the CPU oracle against golden vectors captured from the imported
reference (oracle/gen_golden.py).  CPU only."""
import os

import numpy as np
import torch

from oracle import classic, common, mip


def close(a, b, rtol=1e-6, atol=1e-6):
    a = a.double(); b = b.double()
    assert torch.equal(torch.isnan(a), torch.isnan(b)), "NaN pattern differs"
    a = torch.nan_to_num(a, nan=0.0); b = torch.nan_to_num(b, nan=0.0)
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    assert bool((err <= tol).all()), f"max err {err.max().item():.3e} (tol {tol.min().item():.1e})"


def test_g1_posenc(golden):
    g = golden("g1_posenc")
    assert torch.equal(classic.embed(g["x"], 10), g["classic_pts"])
    assert torch.equal(classic.embed(g["x"], 4), g["classic_dirs"])
    assert torch.equal(mip.pos_enc(g["x"], 0, 4, True), g["mip_view"])


def test_g2_cast_rays(golden):
    g = golden("g2_cast")
    close(mip.transform(g["s_vals"], g["near"], g["far"], 0), g["t_vals"], 0, 0)
    for shape, key in (("cone", "cone"), ("cylinder", "cyl")):
        m, c = mip.cast_rays(g["t_vals"], g["origins"], g["directions"], g["radii"], shape)
        close(m, g[key + "_mean"], 1e-6, 1e-6)
        close(c, g[key + "_cov"], 1e-5, 1e-9)


def test_g3_sample2enc(golden):
    g = golden("g3_sample2enc")
    fm, fc = mip.sample2enc(g["s_vals"], g["origins"], g["directions"], g["radii"], g["near"], g["far"], "cone", 0,
                            full_cov=True)
    close(fm, g["f_means"], 1e-6, 1e-6)
    close(fc, g["f_covs"], 2e-5, 1e-9)
    _, fd = mip.sample2enc(g["s_vals"], g["origins"], g["directions"], g["radii"], g["near"], g["far"], "cone", 0)
    close(fd, torch.diagonal(g["f_covs"], dim1=-2, dim2=-1), 2e-5, 1e-9)
    b = golden("g3b_contract")
    close(mip.contract(b["x"]), b["fx"], 1e-6, 1e-7)
    close(mip.contract_jacobian(b["x"]), b["J"], 1e-6, 1e-7)


def test_g23_view_centred_warp(golden):
    """`fn = 0` (mip.py:367-378: fn1 + Jacobi_f around viewc): the restated sample2enc against the reference's means and FULL warped
    covariances for both ray shapes, and mipnerf_forward(fn_idx=0) against the reference model's outputs."""
    g = golden("g23_warp0")
    rays = {k[len("rays_"):]: v for k, v in g.items() if k.startswith("rays_")}
    for shape in ("cone", "cylinder"):
        fm, fc = mip.sample2enc(g["s_vals"], rays["origins"], rays["directions"], rays["radii"], rays["near"], rays["far"], shape, 0,
                                full_cov=True, fn_idx=0, viewc=g["viewc"])
        close(fm, g[shape + "_f_means"], 1e-6, 1e-6)
        close(fc, g[shape + "_f_covs"], 1e-5, float(g[shape + "_f_covs"].abs().max()) * 1e-6)
    from oracle import common
    names = [str(k) for k in g["param_names"]]
    sd = common.fill_state_dict_({k: torch.empty(tuple(g["grad." + k].shape)) for k in names})
    ref = mip.mipnerf_forward(sd, rays, 16, 17, fn_idx=0, viewc=g["viewc"])
    close(ref[1][0], g["l1_rgb"], 1e-5, 1e-6); close(ref[1][1], g["l1_distance"], 1e-5, 1e-5)
    close(ref[1][2], g["l1_acc"], 1e-5, 1e-6); close(ref[0][1], g["l0_distance"], 1e-5, 1e-5)
    close(ref[1][4], g["l1_s_vals"], 1e-5, 1e-6)


def test_g31_disable_integration(golden):
    """--disable_integration (arg_parser.py:188; models.py:132-133: zeros instead of the sample covariances): mipnerf_forward's branch
    against the reference model's outputs, parameter gradients and ray gradients."""
    from oracle import common
    g = golden("g31_no_integration")
    S0, P1 = int(g["S0"]), int(g["P1"])
    names = [str(k) for k in g["param_names"]]
    sd = {k: v.requires_grad_(True) for k, v in common.fill_state_dict_({k: torch.empty(tuple(g["grad." + k].shape)) for k in names}).items()}
    rays = {k: g[k].clone() for k in ("origins", "directions", "viewdirs", "radii", "lossmult", "near", "far", "app")}
    for k in ("origins", "directions", "viewdirs"):
        rays[k].requires_grad_(True)
    ref = mip.mipnerf_forward(sd, rays, S0, P1, disable_integration=True)
    # Without the exp(-var / 2) damping the 2^15 x features reach the network at full amplitude: one ulp of a contracted mean moves their
    # phase by 2e-3 rad, so two correct fp32 evaluations (the reference's matmul-built Jacobians vs this restatement's closed forms) agree
    # to ~1e-4 in the outputs and to a few per cent in the gradients, not to the 1e-5 of the damped path (g8).  The bounds below are that
    # noise floor (measured: rgb 6e-5, distance 1.0e-4 relative, first-layer weight gradient 4 %); what they pin is the BRANCH -- the
    # default path is 1e-3 .. 1e-1 away from these values.
    close(ref[1][0], g["rgb"], 0, 5e-4); close(ref[1][1], g["dist1"], 5e-4, 0); close(ref[1][2], g["acc1"], 0, 5e-4)
    close(ref[0][1], g["dist0"], 5e-4, 0); close(ref[1][4], g["s1"], 0, 5e-4)
    plain = mip.mipnerf_forward(sd, rays, S0, P1)
    assert float(((plain[1][1] - ref[1][1]).abs() / ref[1][1].abs()).max()) > 1e-2                                # (the branch moves the distances by 3 %: it is what is being compared)
    loss = (ref[1][0] * g["w_rgb"]).sum() + 0.05 * (ref[1][1] * g["w_d1"]).sum() + 0.05 * (ref[0][1] * g["w_d0"]).sum()
    loss.backward()
    for k in names:
        want = g["grad." + k]
        assert float((sd[k].grad - want).norm() / (want.norm() + 1e-20)) < 0.12, k
    for k in ("origins", "directions", "viewdirs"):
        want = g["g_" + k]
        assert float((rays[k].grad - want).norm() / want.norm()) < (0.02 if k == "viewdirs" else 0.5), k


def test_g4_ipe(golden):
    g = golden("g4_ipe")
    e = mip.integrated_pos_enc(g["means"], g["cov_diag"], 0, 16)
    assert torch.equal(e, g["enc_diag"])
    # full-covariance IPE only sees the diagonal
    e2 = mip.integrated_pos_enc(g["means"], torch.diagonal(g["cov_full"], dim1=-2, dim2=-1), 0, 16)
    assert torch.equal(e2, g["enc_full"])


def test_g5_sorted_pdf(golden):
    for num in (32, 33):
        g = golden(f"g5_pdf_{num}")
        assert torch.equal(mip.det_u(num), g["u_det"])
        u = mip.rand_u(num, g["jitter"])
        # reference summation order: exact pin
        s, idx_t = mip.sorted_piecewise_constant_pdf(g["bins"], g["weights"], g["u_det"], sum_mode="torch")
        close(s, g["det_samples"], 0, 1e-7)
        s, idx_tr = mip.sorted_piecewise_constant_pdf(g["bins"], g["weights"], u, sum_mode="torch")
        close(s, g["rand_samples"], 0, 1e-7)
        # canonical order (what the HIP kernel implements): same intervals; values differ only through
        # a 1-ulp change of the row sum amplified by (u - c0) / (c1 - c0) in nearly-empty intervals
        s, idx = mip.sorted_piecewise_constant_pdf(g["bins"], g["weights"], g["u_det"])
        assert torch.equal(idx, idx_t)
        close(s, g["det_samples"], 1e-4, 1e-4)
        s, idx = mip.sorted_piecewise_constant_pdf(g["bins"], g["weights"], u)
        assert torch.equal(idx, idx_tr)
        close(s, g["rand_samples"], 1e-4, 1e-4)
        assert int(idx.min()) >= 0 and int(idx.max()) <= g["bins"].shape[1] - 2
        # samples lie inside the selected interval: index consistency with the reference's output
        b = g["bins"]
        lo = torch.gather(b, 1, idx.long()); hi = torch.gather(b, 1, idx.long() + 1)
        assert bool(((g["rand_samples"] >= lo - 1e-6) & (g["rand_samples"] <= hi + 1e-6)).all())


def test_g6_volumetric(golden):
    for name, white in (("g6_volrend_white0", False), ("g6_volrend_white1", True)):
        g = golden(name)
        c, d, a, w, s = mip.volumetric_rendering(g["rgb"], g["density"], g["s_vals"], g["dirs"], g["near"], g["far"],
                                                 white, g["semantic_in"], 0)
        close(c, g["comp_rgb"], 1e-6, 1e-6); close(d, g["distance"], 1e-6, 1e-6)
        close(a, g["acc"], 1e-6, 1e-6); close(w, g["weights"], 1e-6, 1e-7); close(s, g["semantic"], 1e-6, 1e-6)
    g = golden("g6_volrend_norgb")
    c, d, a, w, s = mip.volumetric_rendering(None, g["density"], g["s_vals"], g["dirs"], g["near"], g["far"])
    assert c is None and s is None
    close(d, g["distance"], 1e-6, 1e-6); close(w, g["weights"], 1e-6, 1e-7)


def _mip_params(hidden, prop_hidden):
    shapes = mip.mipnerf_param_shapes(hidden=hidden, prop_hidden=prop_hidden)
    return common.fill_state_dict_({k: torch.empty(s) for k, s in shapes})


def test_g7_mlp(golden):
    g = golden("g7_mlp_h64")
    names = [str(x) for x in g["param_names"]]
    shapes = mip.mipnerf_param_shapes(hidden=64, prop_hidden=64)
    assert names == [k for k, _ in shapes], "state_dict key order/names differ from the reference module"
    p = {k: v.clone().requires_grad_(True) for k, v in _mip_params(64, 64).items()}
    rr, rd, _ = mip.nerf_mlp(p, g["enc"], g["cond"])
    pr = mip.proposal_mlp(p, g["enc"])
    close(rr, g["raw_rgb"], 1e-5, 1e-5); close(rd, g["raw_density"], 1e-5, 1e-5); close(pr, g["prop_density"], 1e-5, 1e-5)
    loss = (rr ** 2).sum() + (rd * 0.5).sum() + (pr ** 2).sum()
    loss.backward()
    for k, v in p.items():
        close(v.grad, g["grad." + k], 1e-4, 1e-4)
    g = golden("g7_mlp_h1024")
    shapes = [(k, s) for k, s in mip.mipnerf_param_shapes(hidden=1024) if k.startswith("mlp.")]
    # the stand-alone MLP module has no "mlp." prefix and its own key order: bias/weight order is the same
    sd = common.fill_state_dict_({k[len("mlp."):]: torch.empty(s) for k, s in shapes})
    p = {"mlp." + k: v for k, v in sd.items()}
    rr, rd, _ = mip.nerf_mlp(p, g["enc"], g["cond"])
    close(rr, g["raw_rgb"], 1e-5, 1e-5); close(rd, g["raw_density"], 1e-5, 1e-5)


def test_g8_mipnerf_forward(golden):
    g = golden("g8_mipnerf_det")
    rays = {k[len("rays_"):]: v for k, v in g.items() if k.startswith("rays_")}
    ref_rays = common.synthetic_rays(48, seed=3)
    for k in rays:
        assert torch.equal(rays[k], ref_rays[k])
    p = _mip_params(64, 64)
    ret = mip.mipnerf_forward(p, rays, 16, 17)
    close(ret[0][1], g["l0_distance"], 1e-5, 1e-5); close(ret[0][2], g["l0_acc"], 1e-5, 1e-6)
    close(ret[0][3], g["l0_s_vals"], 0, 0); close(ret[0][4], g["l0_weights"], 1e-5, 1e-7)
    close(ret[1][4], g["l1_s_vals"], 1e-5, 1e-6)
    close(ret[1][0], g["l1_rgb"], 1e-5, 1e-5); close(ret[1][1], g["l1_distance"], 1e-5, 1e-5)
    close(ret[1][2], g["l1_acc"], 1e-5, 1e-6); close(ret[1][5], g["l1_weights"], 1e-5, 1e-6)
    r = golden("g8_mipnerf_rand")
    ret = mip.mipnerf_forward(p, rays, 16, 17, s_rand=r["s_rand"], u=mip.rand_u(17, r["jitter"]))
    close(ret[0][3], r["l0_s_vals"], 0, 0); close(ret[0][4], r["l0_weights"], 1e-5, 1e-7)
    close(ret[1][4], r["l1_s_vals"], 1e-5, 1e-6)
    close(ret[1][0], r["l1_rgb"], 1e-5, 1e-5); close(ret[1][1], r["l1_distance"], 1e-5, 1e-5)


def test_g9_classic(golden):
    for name, white in (("g9_raw2outputs_white0", False), ("g9_raw2outputs_white1", True)):
        g = golden(name)
        out = classic.raw2outputs(g["raw"], g["z_vals"], g["rays_d"], None, white)
        for got, key in zip(out, ("rgb_map", "disp_map", "acc_map", "weights", "depth_map")):
            close(got, g[key], 1e-6, 1e-6)
    g = golden("g9_sample_pdf")
    n = g["bins"].shape[0]
    for mode, tol in (("torch", 1e-7), ("canonical", 1e-5)):
        s, inds = classic.sample_pdf(g["bins"], g["weights"], torch.linspace(0., 1., 24).expand(n, 24), sum_mode=mode)
        close(s, g["det_samples"], tol, tol)
        assert torch.equal(inds, g["det_inds"])
        s, inds = classic.sample_pdf(g["bins"], g["weights"], g["u_rand"], sum_mode=mode)
        close(s, g["rand_samples"], tol, tol)
        assert torch.equal(inds, g["rand_inds"])


def _nerf_params(W, flip=False):
    sd = common.fill_state_dict_({k: torch.empty(s) for k, s in classic.nerf_param_shapes(W=W)})
    return {k: v.flip(0) for k, v in sd.items()} if flip else sd


def test_g9_render_rays(golden):
    g = golden("g9_render_rays")
    pc, pf = _nerf_params(64), _nerf_params(64, flip=True)
    rb = g["ray_batch"]
    pts = rb[:, None, 0:3] + rb[:, None, 3:6] * torch.linspace(2, 6, 8)[None, :, None]
    close(classic.run_network(pts, rb[:, -3:], pc), g["run_network_out"], 1e-5, 1e-5)
    r0 = classic.render_rays(rb, pc, None, 16, 0, white_bkgd=True, retraw=True)
    for k in ("rgb_map", "disp_map", "acc_map", "depth_map", "z_vals_map", "weights", "raw"):
        close(r0[k], g["c_" + k], 1e-5, 1e-5)
    # "torch": the reference's own row-sum order -> tight pin; "canonical": the order the HIP kernel uses.
    # The fine pass amplifies a 1-ulp change of a resampled depth by the 2^9 positional-encoding band.
    # In the deterministic case the last uniform is exactly 1.0 and cdf[-1] is 1.0 or 1 - 2^-24 depending on
    # the summation order, so searchsorted(right=True) flips between len-1 and len for that one sample (5 of 24
    # rays here): a genuine order sensitivity of the reference itself, hence the wider det-case tolerance.
    for mode, tol, dtol in (("torch", 1e-5, 1e-5), ("canonical", 1e-4, 1e-3)):
        r1 = classic.render_rays(rb, pc, pf, 16, 32, retraw=True, sum_mode=mode)
        for k in ("rgb_map", "disp_map", "acc_map", "depth_map", "z_vals_map", "weights", "rgb0", "disp0", "acc0", "z_std"):
            close(r1[k], g["cf_" + k], dtol, dtol)
        close(r1["raw"], g["cf_raw"], 10 * dtol, 10 * dtol)
        r2 = classic.render_rays(rb, pc, pf, 16, 32, t_rand=g["t_rand"], u=g["u_rand"], retraw=True, sum_mode=mode)
        for k in ("rgb_map", "disp_map", "acc_map", "depth_map", "z_vals_map", "weights", "rgb0", "z_std"):
            close(r2[k], g["pt_" + k], tol, tol)
        close(r2["raw"], g["pt_raw"], 10 * tol, 10 * tol)


def test_g14_mipnerf_semantic_head(golden):
    """The optional semantic head (models.py:256-261,279-282; mip.py:175-176): outputs, loss and every parameter gradient of the
    reference MipNerfModel(semantic=True) reproduced by the oracle."""
    g = golden("g14_mipnerf_semantic")
    names = [str(k) for k in g["param_names"]]
    shapes = mip.mipnerf_param_shapes(hidden=64, prop_hidden=64, semantic_class_num=7)
    assert [k for k, _ in shapes] == names
    sd = common.fill_state_dict_({k: torch.empty(s) for k, s in shapes})
    pr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    rays = {k[5:]: torch.as_tensor(v) for k, v in g.items() if k.startswith("rays_")}
    ret = mip.mipnerf_forward(pr, rays, 16, 17)
    sem = ret[1][3]
    assert sem is not None and tuple(sem.shape) == (24, 7)
    loss = ((ret[1][0] - g["target"]) ** 2).mean() + 0.01 * ret[0][1].mean() + 0.2 * (sem * g["semw"]).sum() / 24
    loss.backward()
    close(ret[1][0], g["l1_rgb"], 1e-5, 1e-6); close(ret[1][1], g["l1_distance"], 1e-5, 1e-5)
    close(sem, g["l1_semantic"], 1e-5, 1e-6); close(loss, g["loss"], 1e-6, 1e-7)
    for k in names:
        close(pr[k].grad, g["grad." + k], 2e-4, 1e-7)


def test_g19_ray_gradients_of_the_reference(golden):
    """d loss / d (origins, directions, viewdirs) from the reference MipNerfModel's own autograd (pose refinement path) are reproduced
    by autograd through the oracle: pins the differentiable structure (which terms carry gradient: Jacobian, |d| of the intervals;
    which do not: the detached fine fence posts) that the ray-gradient kernels are tested against."""
    g = golden("g19_mipnerf_raygrad")
    hidden = 64
    sd = common.fill_state_dict_({k: torch.empty(s) for k, s in mip.mipnerf_param_shapes(hidden=hidden, prop_hidden=64)})
    rays = {k[5:]: v for k, v in g.items() if k.startswith("rays_")}
    leaves = {k: rays[k].clone().requires_grad_(True) for k in ("origins", "directions", "viewdirs")}
    ref = mip.mipnerf_forward(sd, {**rays, **leaves}, 16, 17)
    target, td = g["target"], g["target_depth"]
    loss = (((ref[1][0] - target) ** 2).mean() + 0.2 * ((1 / ref[1][1] - 1 / td).abs()).mean() + 0.04 * ((1 / ref[0][1] - 1 / td).abs()).mean()
            + 0.01 * (ref[0][4] ** 2).sum() + 0.01 * ref[1][2].mean())
    loss.backward()
    assert abs(float(loss.detach()) - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    for k in ("origins", "directions", "viewdirs"):
        ref_g = g["grad_" + k]
        err = float((leaves[k].grad - ref_g).abs().max())
        # both sides are fp32 autograd of an ill-conditioned sum (2^15-scaled sines): 1e-3 of the largest component is their common noise
        assert err <= 3e-3 * float(ref_g.abs().max()), (k, err, float(ref_g.abs().max()))


def test_oracle_fuzz_vs_reference_log_and_live_run():
    """oracle/fuzz_vs_reference.py (paths A, B: 25 seeds), oracle/fuzz_zip_vs_reference.py (path C around the grid: 12 seeds) and
    oracle/fuzz_callers_vs_reference.py / fuzz_zip_callers_vs_reference.py (ray generation and loss tails of both code bases: 20 / 12 seeds): the
    committed summaries report no violation; where the reference is present (the build container) two seeds of each are replayed
    live -- random inputs AND random weights / hash tables, every stage."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for name, min_lines, ref_dir in (("fuzz_vs_reference", 40, "/root/reference/s-nerf"), ("fuzz_zip_vs_reference", 20, "/root/reference/s-nerfpp/zipnerf"),
                                     ("fuzz_callers_vs_reference", 18, "/root/reference/s-nerf"),
                                     ("fuzz_zip_callers_vs_reference", 20, "/root/reference/s-nerfpp/zipnerf")):
        log = open(os.path.join(root, "oracle", name + ".log")).read()
        assert "# total violations: 0" in log and log.count("\n") > min_lines, name
        assert all(line.rstrip().endswith("| 0") for line in log.splitlines() if line and not line.startswith("#")), name
        if not os.path.isdir(ref_dir):
            continue
        r = subprocess.run([sys.executable, os.path.join(root, "oracle", name + ".py"), "--seeds", "2"], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and "# total violations: 0" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_every_golden_generator_verifies_against_the_reference():
    """`python oracle/gen_golden*.py --check` for all 17 generators: each imports the reference's own modules (/root/reference, build
    container only), regenerates its fixtures in memory and compares them with the committed files bit for bit -- `check: OK`, nothing
    written.  Skipped where the reference is not mounted (the GPU box)."""
    import glob
    import subprocess
    import sys
    import pytest
    if not os.path.isdir("/root/reference/s-nerf"):
        pytest.skip("/root/reference is not mounted here")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gens = sorted(glob.glob(os.path.join(repo, "oracle", "gen_golden*.py")))
    assert len(gens) == 17
    before = {f: os.path.getmtime(os.path.join(repo, "tests", "golden", f)) for f in os.listdir(os.path.join(repo, "tests", "golden"))}
    procs = [(g, subprocess.Popen([sys.executable, g, "--check"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, cwd=repo)) for g in gens]
    for g, p in procs:
        out = p.communicate(timeout=900)[0]
        assert p.returncode == 0 and out.strip().splitlines()[-1] == "check: OK", (os.path.basename(g), out[-400:])
    assert before == {f: os.path.getmtime(os.path.join(repo, "tests", "golden", f)) for f in before}, "a generator wrote under --check"
