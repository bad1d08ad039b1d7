"""Early ray termination + sample compaction (inference extension, csrc/ert.hip): selection rule against a plain restatement,
index maps, exactness when nothing is skipped, and the error bound when samples are skipped.  pytest -m gpu."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _hist(N, S0, S1, seed):
    g = torch.Generator().manual_seed(seed)
    s0 = torch.sort(torch.rand(N, S0 + 1, generator=g), -1).values
    s0[:, 0], s0[:, -1] = 0.0, 1.0
    s1 = torch.sort(torch.rand(N, S1 + 1, generator=g), -1).values
    w0 = torch.rand(N, S0, generator=g) ** 6
    w0 = w0 / w0.sum(-1, keepdim=True) * torch.rand(N, 1, generator=g)
    w0[0] = 0.0                                             # an empty ray
    w0[1] = 0.0; w0[1, 3] = 0.999                           # one opaque interval
    return s0, w0, s1


def _select_ref(s0, w0, s1, eps_t, eps_w):
    """W = cumulative proposal weight, piecewise linear; keep iff 1 - W(s1[i]) > eps_t and W(s1[i+1]) - W(s1[i]) > eps_w."""
    s0, w0, s1 = s0.double(), w0.double(), s1.double()
    cum = torch.cat([torch.zeros_like(w0[:, :1]), torch.cumsum(w0, -1)], -1)
    k = (torch.searchsorted(s0.contiguous(), s1.contiguous(), right=True) - 1).clamp(0, s0.shape[1] - 2)
    a0, a1 = torch.gather(s0, 1, k), torch.gather(s0, 1, k + 1)
    fr = torch.where(a1 > a0, (s1 - a0) / (a1 - a0), torch.zeros_like(s1)).clamp(0, 1)
    W = torch.gather(cum, 1, k) + fr * (torch.gather(cum, 1, k + 1) - torch.gather(cum, 1, k))
    T, m = 1 - W[:, :-1], W[:, 1:] - W[:, :-1]
    return T, m, (T > eps_t) & (m > eps_w)


@pytest.mark.parametrize("N,S0,S1", [(37, 64, 128), (1000, 128, 127), (5, 7, 200)])
def test_ert_compact_selection_and_index_maps(N, S0, S1):
    from snerf_amd import ops
    s0, w0, s1 = _hist(N, S0, S1, N)
    eps_t, eps_w = 1e-2, 1e-4
    row_index, sample_id = ops.ert_compact(s0.cuda(), w0.cuda(), s1.cuda(), eps_t, eps_w)
    T, m, keep = _select_ref(s0, w0, s1, eps_t, eps_w)
    got = (row_index >= 0).cpu()
    clear = ((T - eps_t).abs() > 1e-5) & ((m - eps_w).abs() > 1e-6)          # away from the thresholds the decision is exact
    assert bool((got == keep)[clear].all()), int((got != keep)[clear].sum())
    assert float(clear.float().mean()) > 0.99
    # index maps: rows are the kept samples in ray-major order, each exactly once
    ri, sid = row_index.cpu().long(), sample_id.cpu().long()
    assert sid.shape[0] == int(got.sum())
    assert torch.equal(sid, torch.nonzero(got.reshape(-1)).flatten())
    assert torch.equal(ri.reshape(-1)[sid], torch.arange(sid.shape[0]))
    assert int(got[0].sum()) == 0                                           # the empty ray evaluates nothing


def _model():
    from snerf_amd.mipnerf import MipNerfModel
    torch.manual_seed(0)
    return MipNerfModel(n_samples=64, N_fine=129, no_warp_sample=0, ray_shape="cone", fn=1, radius=3., transform_idx=0, real=True,
                        rgb_layer=3, hidden_layer=256, density_noise=0., max_deg_point=16, proposal_hidden_layer=256,
                        proposal_loss=True, compute="f32", device="cuda")


def test_ert_render_is_exact_without_skips_and_bounded_with_skips():
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    m = _model()
    rays = bench.synth_rays(1024, 7, torch.device("cuda"))
    with torch.no_grad():
        full = m(rays, False, False, 0.)
        same = m(rays, False, False, 0., ert=(-1.0, -1.0))                  # thresholds below every T and m: all samples kept
        fast = m(rays, False, False, 0., ert=(1e-2, 1e-4))
    kept, total = m.last_ert_rows
    for a, b in zip(full[1], same[1]):
        if a is not None:
            assert torch.equal(a, b)                                         # compaction itself changes nothing
    assert 0 < kept < total
    # the skipped samples' true weights bound the error of every output
    w_full, w_fast = full[1][5], fast[1][5]
    skipped = w_fast == 0
    lost = (w_full * skipped).sum(-1)
    assert float((fast[1][0] - full[1][0]).abs().max()) <= float(lost.max()) * 1.05 + 1e-5
    assert float(lost.max()) < 0.2
    with pytest.raises(NotImplementedError):
        m(rays, False, False, 0., ert=(1e-2, 1e-4))                         # training mode (grad enabled)


def test_ert_with_the_semantic_head():
    """semantic rendering on compacted rows: with nothing skipped identical to the dense evaluation, with skips off by at most the
    skipped samples' true weight times the largest semantic logit"""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from snerf_amd.mipnerf import MipNerfModel
    torch.manual_seed(0)
    m = MipNerfModel(n_samples=64, N_fine=129, no_warp_sample=0, ray_shape="cone", fn=1, radius=3., transform_idx=0, real=True, rgb_layer=3,
                     hidden_layer=256, density_noise=0., max_deg_point=16, proposal_hidden_layer=256, proposal_loss=True, compute="f32",
                     semantic=True, semantic_class_num=7, device="cuda")
    rays = bench.synth_rays(512, 9, torch.device("cuda"))
    with torch.no_grad():
        full = m(rays, False, False, 0.)
        same = m(rays, False, False, 0., ert=(-1.0, -1.0))
        fast = m(rays, False, False, 0., ert=(1e-2, 1e-4))
        raw_sem = m.nerf.raw_sem
    sem_full, sem_same, sem_fast = full[1][3], same[1][3], fast[1][3]
    assert sem_full.shape == (512, 7) and torch.equal(sem_full, sem_same)
    kept, total = m.last_ert_rows
    assert 0 < kept < total
    w_full, w_fast = full[1][5], fast[1][5]
    lost = (w_full * (w_fast == 0)).sum(-1)
    bound = float(lost.max()) * float(raw_sem.abs().max()) * 1.05 + 1e-5
    assert float((sem_fast - sem_full).abs().max()) <= bound


def test_front_to_back_ert_is_exact_without_termination_and_bounded_by_eps():
    """ert=(eps_t, eps_w, G): the fine level in groups of G samples, rays leave once the FINE network's transmittance is <= eps_t
    (snerf_ert_f2b_step).  With eps_t = 0 nothing leaves and the render equals the plain one bit for bit (grouping changes nothing);
    on an opaque medium rays stop early and every output is off by at most the exact bound: acc and rgb by eps_t (x the colour range),
    distance by eps_t x t_far."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    m = _model()
    rays = bench.synth_rays(1024, 7, torch.device("cuda"))
    with torch.no_grad():
        full = m(rays, False, False, 0.)
        same = m(rays, False, False, 0., ert=(0.0, 0.0, 32))
        assert m.last_ert_rows[0] == m.last_ert_rows[1]
        same48 = m(rays, False, False, 0., ert=(0.0, 0.0, 48))              # a ragged last group (128 = 48 + 48 + 32)
    for a, b, c in zip(full[1], same[1], same48[1]):
        if a is not None:
            assert torch.equal(a, b) and torch.equal(a, c)
    with torch.no_grad(), pytest.raises(ValueError, match="eps_t < 1"):      # eps_t >= 1 used to end in torch.cat([])
        m(rays, False, False, 0., ert=(1.0, 0.0, 32))
    with torch.no_grad():                                                   # make the medium opaque: rays now end inside the sampled range
        p = dict(m.named_parameters())
        p["proposal.density_layer.bias"] += 6.0
        p["mlp.density_layer.bias"] += 6.0
    m.arena.bump()
    eps = 1e-3
    with torch.no_grad():
        full = m(rays, False, False, 0.)
        fast = m(rays, False, False, 0., ert=(eps, 0.0, 16))
    kept, total = m.last_ert_rows
    assert 0 < kept < 0.8 * total, (kept, total)
    assert float((fast[1][2] - full[1][2]).abs().max()) <= eps * 1.01 + 1e-6                    # acc
    assert float((fast[1][0] - full[1][0]).abs().max()) <= eps * 1.002 * 1.01 + 1e-6            # rgb in [-0.001, 1.001]
    far = float(rays.far.max())
    assert float((fast[1][1] - full[1][1]).abs().max()) <= eps * far * 1.01 + 1e-4              # distance = sum w t_mid
    # the evaluated samples' weights are the full render's; the others are exactly zero
    w_full, w_fast = full[1][5], fast[1][5]
    ev = w_fast != 0
    assert float((w_full[ev] - w_fast[ev]).abs().max()) < 1e-6
    assert float((w_full * (~ev)).sum(-1).max()) <= eps * 1.01 + 1e-6


def test_classic_fine_pass_front_to_back_ert():
    """render_rays(..., ert=(eps_t, G)): path B's fine network front to back in groups of G of the 192 sorted samples, rays leave once
    the fine transmittance is <= eps_t, the next group's rows are compacted to the survivors.  Nothing leaves (eps_t < 0): bit-identical
    to the plain render whatever the grouping.  Opaque medium: fewer evaluations, acc / rgb / depth inside the exact bound."""
    from snerf_amd import classic
    torch.manual_seed(0)
    mk = lambda: classic.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=4, skips=[4], use_viewdirs=True, compute="bf16", device="cuda")
    coarse, fine = mk(), mk()
    e, _ = classic.get_embedder(10, 0)
    ed, _ = classic.get_embedder(4, 0)
    q = classic.make_network_query_fn(e, ed)
    g = torch.Generator().manual_seed(1)
    N = 3000
    d = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1)
    o = torch.randn(N, 3, generator=g) * 0.1 + torch.tensor([0.0, 0.0, 4.0])
    rays = torch.cat([o, -d, torch.full((N, 1), 2.0), torch.full((N, 1), 6.0), -d], -1).cuda()
    kw = dict(network_fn=coarse, network_query_fn=q, N_samples=64, perturb=0.0, N_importance=128, network_fine=fine, retraw=True)
    with torch.no_grad():
        full = classic.render_rays(rays, **kw)
        for G in (32, 50, (128, 16)):                                         # 192 = 6 x 32 = 3 x 50 + 42 (ragged last group) = 128 + 4 x 16 (a schedule)
            same = classic.render_rays(rays, ert=(-1.0, G), **kw)
            for k in ("rgb_map", "acc_map", "depth_map", "disp_map", "raw"):
                assert torch.equal(torch.nan_to_num(full[k], nan=-7.0), torch.nan_to_num(same[k], nan=-7.0)), (G, k)     # (disp = 1 / (depth / acc): 0 / 0 on an empty ray)
        with pytest.raises(ValueError):
            classic.render_rays(rays, ert=(1.0, 32), **kw)
        with torch.no_grad():
            for net in (coarse, fine):
                dict(net.named_parameters())["alpha_linear.bias"] += 40.0      # an opaque medium: rays end within the first ~130 of the 192 sorted samples
                net.arena.bump()
        full = classic.render_rays(rays, **kw)
        eps = 1e-3
        fast = classic.render_rays(rays, ert=(eps, 16), **kw)
        with pytest.raises(ValueError):
            classic.render_rays(rays, ert=(eps, 16), **dict(kw, N_importance=0))          # no fine pass: refused, not silently ignored
    ev, tot = (int(v) for v in fast["ert_evals"].sum(0))
    assert tot == N * 192 and 0 < ev < 0.85 * tot, (ev, tot)
    assert float((fast["acc_map"] - full["acc_map"]).abs().max()) <= eps * 1.01 + 1e-6
    assert float((fast["rgb_map"] - full["rgb_map"]).abs().max()) <= eps * 1.01 + 1e-6          # sigmoid colours in [0, 1]
    assert float((fast["depth_map"] - full["depth_map"]).abs().max()) <= eps * 6.0 * 1.01 + 1e-5  # sum w z, z <= far = 6
    evaluated = (fast["raw"] != 0).any(-1)
    assert torch.equal(fast["raw"][evaluated], full["raw"][evaluated])
    with pytest.raises(NotImplementedError):
        classic.render_rays(rays, ert=(eps, 32), **kw)                        # (grad mode on)


def test_classic_ert_step_kernels_against_torch():
    """snerf_classic_ert_points / snerf_classic_ert_step on their own: positions bit-equal to o + d z of the compacted rows, raw rows at
    their places, transmittance = T exp(-sum relu(sigma) dz |d|) (raw2outputs, run_nerf_helpers.py:394-414), survivors compacted in order;
    the last group (g0 + G == S) keeps no ray and leaves T alone apart from the factor of its own samples."""
    from snerf_amd import ops
    g = torch.Generator().manual_seed(3)
    N, S, C = 1500, 37, 5
    rays = torch.randn(N, 11, generator=g).cuda()
    z = torch.sort(torch.rand(N, S, generator=g) * 4 + 2, -1).values.cuda()
    alive = torch.sort(torch.randperm(N, generator=g)[:900]).values.to(torch.int32).cuda()
    for al in (None, alive):
        n = N if al is None else al.numel()
        idx = torch.arange(N, device="cuda") if al is None else al.long()
        for g0, G in ((0, 7), (7, 20), (27, 10)):
            pts, vd = ops.classic_ert_points(rays, z, al, g0, G)
            ref = rays[idx, None, 0:3] + rays[idx, None, 3:6] * z[idx, g0:g0 + G, None]
            assert torch.equal(pts, ref) and torch.equal(vd, rays[idx, 8:11])
            raw_g = torch.randn(n, G, C, generator=g).cuda()
            T = (torch.rand(N, generator=g) * 0.5 + 0.5).cuda()
            T0 = T.clone()
            raw_full = torch.zeros(N, S, C, device="cuda")
            scratch = (torch.empty(N, dtype=torch.int32, device="cuda"), torch.empty(N, dtype=torch.int32, device="cuda"),
                       torch.empty(N, dtype=torch.int32, device="cuda"), torch.zeros(1, dtype=torch.int64, device="cuda"))
            eps = 0.3
            nxt = ops.classic_ert_step(raw_g, al, z, rays, g0, G, eps, T, raw_full, scratch)
            want = torch.zeros_like(raw_full)
            want[idx, g0:g0 + G] = raw_g
            assert torch.equal(raw_full, want)
            g1 = min(g0 + G, S - 1)                                             # (the ray's last sample has no successor: no factor)
            dz = (z[idx, g0 + 1:g1 + 1] - z[idx, g0:g1]) * rays[idx, 3:6].norm(dim=-1)[:, None]
            Tw = T0.clone()
            Tw[idx] = T0[idx] * torch.exp(-(torch.relu(raw_g[:, :g1 - g0, 3]) * dz).sum(-1))
            assert float((T - Tw).abs().max()) < 1e-6
            if g0 + G == S:
                assert nxt.numel() == 0
            else:
                sure = (Tw[idx] - eps).abs() > 1e-5
                got = torch.zeros(N, dtype=torch.bool, device="cuda"); got[nxt.long()] = True
                assert torch.equal(got[idx][sure], (Tw[idx] > eps)[sure])
                assert torch.equal(nxt, torch.sort(nxt).values) and nxt.dtype == torch.int32
