"""HIP hash-grid encoder vs oracle/grid.py through the C-ABI (forward, dy_dx, table gradient, input gradient, TV, fp16 tables,
GridEncoder module + autograd + autocast)."""
import numpy as np
import pytest
import torch

from oracle import grid as og

pytestmark = pytest.mark.gpu


def setup(L, C, scale, H, log2T, B, seed, desired=None):
    off, res, s = og.level_layout(3, L, C, scale, H, log2T, desired, False)
    rng = np.random.default_rng(seed)
    E = (rng.standard_normal((int(off[-1]), C)) * 0.5).astype(np.float32)
    x = rng.random((B, 3)).astype(np.float32)
    x[0] = [-0.2, 0.5, 0.5]; x[1] = [0.5, 0.5, 1.5]; x[2] = [0.0, 1.0, 0.0]      # 2 out of bounds, 1 on the boundary
    return off, res, float(np.log2(s)), E, x


@pytest.mark.parametrize("C,interp,gridtype", [(1, 0, 0), (2, 0, 0), (4, 0, 0), (4, 1, 0), (8, 0, 0), (2, 0, 1)])
def test_grid_forward_backward_fp32(C, interp, gridtype):
    from snerf_amd import ops
    L, H = 6, 4
    off, res, S, E, x = setup(L, C, 1.7, H, 11, 3000, C + interp)               # 2^11-entry levels: levels >= 2 are hashed / tiled
    ref, ref_dd = og.grid_encode_forward(x, E, off, S, H, gridtype, False, interp, want_dy_dx=True)
    xt, Et, offt = torch.from_numpy(x).cuda(), torch.from_numpy(E).cuda(), torch.from_numpy(off).cuda()
    out, dd = ops.grid_encode_fwd(xt, Et, offt, L, S, H, gridtype, False, interp, want_dy_dx=True)
    np.testing.assert_allclose(out.cpu().numpy().reshape(-1, L, C), ref.transpose(1, 0, 2), rtol=1e-4, atol=2e-5)  # fma contraction in the kernel vs separate fp32 ops in the oracle
    np.testing.assert_allclose(dd.cpu().numpy(), ref_dd, rtol=5e-3, atol=2e-3)   # scale * (v_right - v_left): fp32 cancellation
    out_lm, _ = ops.grid_encode_fwd(xt, Et, offt, L, S, H, gridtype, False, interp, level_major=True)   # the reference's [L,B,C] layout
    assert torch.equal(out_lm.permute(1, 0, 2).reshape(-1, L * C), out)
    assert float(out[:2].abs().max()) == 0.0, "out-of-bound points must encode to zero"
    # backward
    rng = np.random.default_rng(7)
    G = rng.standard_normal((L, x.shape[0], C)).astype(np.float32)
    gE_ref, gx_ref = og.grid_encode_backward(G, x, off, E.shape[0], S, H, gridtype, False, interp, dy_dx=ref_dd)
    Gt = torch.from_numpy(np.ascontiguousarray(G.transpose(1, 0, 2).reshape(-1, L * C))).cuda()
    gE, gx = ops.grid_encode_bwd(Gt, xt, Et, offt, L, S, H, gridtype, False, interp, dy_dx=dd)
    np.testing.assert_allclose(gE.cpu().numpy(), gE_ref, rtol=1e-4, atol=2e-4)   # fp32 atomics: order-dependent last bits
    np.testing.assert_allclose(gx.cpu().numpy(), gx_ref, rtol=1e-3, atol=1e-3)
    # adjointness on the device results themselves
    lhs = float((out.double() * Gt.double()).sum()); rhs = float((Et.double() * gE.double()).sum())
    assert abs(lhs - rhs) < 1e-3 * max(1.0, abs(lhs))


def test_grid_shipped_layout_fp16_and_tv():
    from snerf_amd import ops
    # zipnerf NeRF grid: L=10, C=4, H=16, desired 8192, T=2^21 (internal/models.py:381-386): 14 995 560 rows
    L, C, H = 10, 4, 16
    off, res, s = og.level_layout(3, L, C, 2.0, H, 21, 8192, False)
    assert int(off[-1]) == 14995560
    S = float(np.log2(s))
    g = torch.Generator().manual_seed(0)
    E = (torch.randn(int(off[-1]), C, generator=g) * 0.1)
    x = torch.rand(20000, 3, generator=g)
    offt = torch.from_numpy(off).cuda()
    out32, _ = ops.grid_encode_fwd(x.cuda(), E.cuda(), offt, L, S, H, 0, False, 0)
    sub = slice(0, 512)
    ref = og.grid_encode_forward(x[sub].numpy(), E.numpy(), off, S, H, 0, False, 0)
    np.testing.assert_allclose(out32[sub].cpu().numpy().reshape(-1, L, C), ref.transpose(1, 0, 2), rtol=1e-4, atol=2e-5)  # fma contraction in the kernel vs separate fp32 ops in the oracle
    out16, _ = ops.grid_encode_fwd(x.cuda(), E.cuda().half(), offt, L, S, H, 0, False, 0)
    assert out16.dtype == torch.float16
    assert float((out16.float() - out32).abs().max()) < 1e-3, "fp16 table vs fp32 table"
    # fp16 gradient table (packed-half atomics) vs fp32
    G = torch.randn(20000, L * C, generator=g).cuda() * 0.01
    g32, _ = ops.grid_encode_bwd(G, x.cuda(), E.cuda(), offt, L, S, H, 0, False, 0)
    g16, _ = ops.grid_encode_bwd(G.half(), x.cuda(), E.cuda().half(), offt, L, S, H, 0, False, 0)
    rel = float((g16.float() - g32).norm() / g32.norm())
    assert rel < 2e-2, rel
    # total variation on a small grid vs the oracle
    off2, res2, s2 = og.level_layout(3, 3, 2, 2.0, 4, 9, None, False)
    rng = np.random.default_rng(3)
    E2 = rng.standard_normal((int(off2[-1]), 2)).astype(np.float32); x2 = rng.random((500, 3)).astype(np.float32)
    want = og.grad_total_variation(x2, E2, off2, 0.3, 1.0, 4, 0, False)
    grad = torch.zeros(E2.shape, device="cuda")
    ops.grid_tv_grad(torch.from_numpy(x2).cuda(), torch.from_numpy(E2).cuda(), grad, torch.from_numpy(off2).cuda(), 0.3, 3, 1.0, 4, 0, False)
    np.testing.assert_allclose(grad.cpu().numpy(), want, rtol=1e-3, atol=1e-5)


def test_gridencoder_module_autograd_and_autocast():
    from snerf_amd.gridencoder import GridEncoder
    enc = GridEncoder(input_dim=3, num_levels=6, level_dim=4, base_resolution=16, log2_hashmap_size=15, desired_resolution=512)
    assert list(enc.state_dict().keys()) == ["embeddings", "offsets", "idx", "grid_sizes"]
    torch.manual_seed(11)                                 # the table is drawn from the global generator: pin it
    with torch.no_grad():
        enc.embeddings.normal_(0, 0.2)
    g = torch.Generator().manual_seed(1)
    x = (torch.rand(777, 3, generator=g) * 2 - 1).cuda().requires_grad_(True)
    y = enc(x, bound=1)
    assert y.shape == (777, 24)
    off, S = enc.offsets.cpu().numpy(), float(np.log2(enc.per_level_scale))
    ref, dd = og.grid_encode_forward(((x.detach().cpu().numpy() + 1) / 2).astype(np.float32), enc.embeddings.detach().cpu().numpy(), off, S, 16, 0, False, 0,
                                     want_dy_dx=True)
    np.testing.assert_allclose(y.detach().cpu().numpy().reshape(-1, 6, 4), ref.transpose(1, 0, 2), rtol=1e-4, atol=2e-5)  # fma contraction in the kernel vs separate fp32 ops in the oracle
    w = torch.randn(777, 24, generator=g).cuda()
    (y * w).sum().backward()
    G = w.cpu().numpy().reshape(-1, 6, 4).transpose(1, 0, 2)
    gE, gx = og.grid_encode_backward(np.ascontiguousarray(G), ((x.detach().cpu().numpy() + 1) / 2).astype(np.float32), off, enc.embeddings.shape[0], S, 16,
                                     0, False, 0, dy_dx=dd)
    np.testing.assert_allclose(enc.embeddings.grad.cpu().numpy(), gE, rtol=1e-4, atol=2e-4)
    np.testing.assert_allclose(x.grad.cpu().numpy(), gx * 0.5, rtol=1e-3, atol=5e-3)     # d/dx of (x + 1) / 2: sums of +-300-sized terms that cancel
    enc.grad_total_variation(weight=1e-3, B=1000)                                        # smoke: adds into .grad
    with torch.autocast("cuda", dtype=torch.float16):
        y16 = enc(x.detach())
    assert y16.dtype == torch.float16 and float((y16.float() - y.detach()).abs().max()) < 2e-3


@pytest.mark.parametrize("D,C,gridtype", [(2, 2, 0), (4, 2, 0), (5, 1, 0), (4, 4, 1)])
def test_grid_other_input_dims(D, C, gridtype):
    """The reference instantiates D = 2..5 (gridencoder.cu:376-399); the hash uses one prime per dimension (:50-63)."""
    from snerf_amd import ops
    L, H = 4, 3
    off, res, s = og.level_layout(D, L, C, 1.6, H, 10, None, False)
    S = float(np.log2(s))
    rng = np.random.default_rng(10 * D + C)
    E = (rng.standard_normal((int(off[-1]), C)) * 0.5).astype(np.float32)
    x = rng.random((700, D)).astype(np.float32)
    x[0, 0] = -0.1; x[1, D - 1] = 1.2; x[2] = 0.0; x[3] = 1.0
    ref, ref_dd = og.grid_encode_forward(x, E, off, S, H, gridtype, False, 0, want_dy_dx=True)
    xt, Et, offt = torch.from_numpy(x).cuda(), torch.from_numpy(E).cuda(), torch.from_numpy(off).cuda()
    out, dd = ops.grid_encode_fwd(xt, Et, offt, L, S, H, gridtype, False, 0, want_dy_dx=True)
    np.testing.assert_allclose(out.cpu().numpy().reshape(-1, L, C), ref.transpose(1, 0, 2), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(dd.cpu().numpy(), ref_dd, rtol=5e-3, atol=2e-3)
    assert float(out[:2].abs().max()) == 0.0
    G = rng.standard_normal((L, x.shape[0], C)).astype(np.float32)
    gE_ref, gx_ref = og.grid_encode_backward(G, x, off, E.shape[0], S, H, gridtype, False, 0, dy_dx=ref_dd)
    Gt = torch.from_numpy(np.ascontiguousarray(G.transpose(1, 0, 2).reshape(-1, L * C))).cuda()
    gE, gx = ops.grid_encode_bwd(Gt, xt, Et, offt, L, S, H, gridtype, False, 0, dy_dx=dd)
    np.testing.assert_allclose(gE.cpu().numpy(), gE_ref, rtol=1e-4, atol=2e-4)
    np.testing.assert_allclose(gx.cpu().numpy(), gx_ref, rtol=1e-3, atol=1e-3)
    # the module accepts the dimension
    from snerf_amd.gridencoder import GridEncoder
    enc = GridEncoder(input_dim=D, num_levels=L, level_dim=C, per_level_scale=1.6, base_resolution=H, log2_hashmap_size=10, gridtype="hash" if gridtype == 0 else "tiled")
    assert enc(torch.rand(33, D, device="cuda"), bound=1).shape == (33, L * C)


def test_grid_double_tables():
    """the double instantiation (AT_DISPATCH_FLOATING_TYPES_AND_HALF, gridencoder.cu:469): float positions / weights, table values and
    their accumulation in double"""
    from snerf_amd import ops
    L, C, H = 5, 2, 4
    off, res, S, E, x = setup(L, C, 1.7, H, 11, 2000, 3)
    E64 = E.astype(np.float64) + 1e-9 * np.random.default_rng(4).standard_normal(E.shape)      # values that do not fit fp32
    ref, ref_dd = og.grid_encode_forward(x, E64, off, S, H, 0, False, 1, want_dy_dx=True, vt=np.float64)
    xt, Et, offt = torch.from_numpy(x).cuda(), torch.from_numpy(E64).cuda(), torch.from_numpy(off).cuda()
    out, dd = ops.grid_encode_fwd(xt, Et, offt, L, S, H, 0, False, 1, want_dy_dx=True)
    assert out.dtype == torch.float64 and dd.dtype == torch.float64
    # positions and weights are fp32 in every instantiation (and the kernel may contract x * scale + 0.5 into one fma): agreement with
    # the oracle is limited by fp32 weight rounding ...
    np.testing.assert_allclose(out.cpu().numpy().reshape(-1, L, C), ref.transpose(1, 0, 2), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(dd.cpu().numpy(), ref_dd, rtol=5e-3, atol=2e-3)
    # ... but the VALUES are carried in double: the encoding is linear in the table, and a 1e-9 perturbation of the entries (invisible
    # to an fp32 accumulation of O(1) values) comes out resolved to 1e-4 of itself
    E32t = torch.from_numpy(E.astype(np.float64)).cuda()
    out32, _ = ops.grid_encode_fwd(xt, E32t, offt, L, S, H, 0, False, 1)
    d_ref = og.grid_encode_forward(x, E64 - E.astype(np.float64), off, S, H, 0, False, 1, vt=np.float64)
    d = (out - out32).cpu().numpy().reshape(-1, L, C)
    assert float(np.abs(d_ref).max()) > 1e-10
    np.testing.assert_allclose(d, d_ref.transpose(1, 0, 2), rtol=1e-5, atol=1e-13)   # (an fp32 accumulation would be off by 1e-7)
    G = np.random.default_rng(5).standard_normal((L, x.shape[0], C))
    gE_ref, gx_ref = og.grid_encode_backward(G, x, off, E.shape[0], S, H, 0, False, 1, dy_dx=ref_dd, vt=np.float64)
    Gt = torch.from_numpy(np.ascontiguousarray(G.transpose(1, 0, 2).reshape(-1, L * C))).cuda()
    gE, gx = ops.grid_encode_bwd(Gt, xt, Et, offt, L, S, H, 0, False, 1, dy_dx=dd)
    assert gE.dtype == torch.float64 and gx.dtype == torch.float64
    np.testing.assert_allclose(gE.cpu().numpy(), gE_ref, rtol=1e-4, atol=2e-4)                  # fp32 weights, see above
    np.testing.assert_allclose(gx.cpu().numpy(), gx_ref, rtol=1e-3, atol=1e-3)
    # adjointness holds to double precision on the device results themselves (same weights on both sides)
    lhs = float((out * Gt).sum()); rhs = float((Et * gE).sum())
    assert abs(lhs - rhs) < 1e-10 * max(1.0, abs(lhs))


# ---------------------------------------------------------------- corner-cached gather + binned table gradient (csrc/zip.hip g3_*) ----
def _clumped_points(B, seed):
    """points as the reference's callers pass them (internal/models.py:488-494: the 7 multisamples of an interval one after the other):
    groups of 7 neighbours around a random centre -- one cell on the coarse levels, several on the fine ones -- plus out-of-bounds and
    boundary points and a ragged tail (B % 8 != 0)"""
    rng = np.random.default_rng(seed)
    centres = rng.random(((B + 6) // 7, 1, 3)).astype(np.float32)
    x = (centres + rng.standard_normal((centres.shape[0], 7, 3)).astype(np.float32) * 2e-3).reshape(-1, 3)[:B]
    x = np.clip(x, 0.0, 1.0)
    x[5] = [-0.2, 0.5, 0.5]; x[6] = [0.5, 0.5, 1.5]; x[7] = [0.0, 1.0, 0.0]; x[11] = [1.0, 1.0, 1.0]
    return np.ascontiguousarray(x)


@pytest.mark.parametrize("C,half", [(4, False), (4, True), (1, False), (1, True), (2, True), (2, False), (8, True)])
def test_grid_fast_forward_is_the_slow_kernel_bit_for_bit(C, half):
    """snerf_grid_encode_fwd runs the corner-cached, pair-loading gather for D = 3 / hash / linear / C in {1, 4}: the same corner
    products in the same order as kernel_grid's restatement (grid.hip) -- identical bits, for random and for clumped points, in both
    output layouts."""
    from snerf_amd import ops
    L, H = 8, 16
    off, res, s = og.level_layout(3, L, C, 2.0, H, 15, 1024, False)
    S = float(np.log2(s))
    rng = np.random.default_rng(C)
    E = torch.from_numpy((rng.standard_normal((int(off[-1]), C)) * 0.5).astype(np.float32)).cuda()
    E = E.half() if half else E
    offt = torch.from_numpy(off).cuda()
    for x in (_clumped_points(20003, 1), np.random.default_rng(2).random((4099, 3)).astype(np.float32)):
        xt = torch.from_numpy(x).cuda()
        slow, _ = ops.grid_encode_fwd(xt, E, offt, L, S, H, 0, False, 0, reference_form=True)          # snerf_grid_encode_fwd_ref
        slow_lm, _ = ops.grid_encode_fwd(xt, E, offt, L, S, H, 0, False, 0, level_major=True, reference_form=True)
        fast, _ = ops.grid_encode_fwd(xt, E, offt, L, S, H, 0, False, 0, reference_form=False)
        fast_lm, _ = ops.grid_encode_fwd(xt, E, offt, L, S, H, 0, False, 0, level_major=True, reference_form=False)
        assert torch.equal(fast, slow) and torch.equal(fast_lm, slow_lm)
        assert float(fast.float().abs().max()) > 0


@pytest.mark.parametrize("C,B", [(4, 20003), (1, 20003), (4, 300007), (2, 20003), (8, 20003), (2, 300007)])
def test_grid_binned_backward_vs_oracle_atomic_kernel_and_itself(C, B):
    """snerf_grid_encode_bwd_binned against oracle/grid.py's kernel_grid_backward restatement, against the atomic kernel, run to run
    (bit-identical), in both gradient layouts, with fp16 gradients / half records, and (B = 300 007: 2.4 M records per level) with the
    dense levels split into replicas that meet in the int64 image."""
    from snerf_amd import ops
    L, H = 7, 8
    off, res, s = og.level_layout(3, L, C, 2.0, H, 14, 512, False)
    S = float(np.log2(s))
    x = _clumped_points(B, 3)
    rng = np.random.default_rng(4)
    G = (rng.standard_normal((B, L * C)) * 0.3).astype(np.float32)
    xt, Gt, offt = torch.from_numpy(x).cuda(), torch.from_numpy(G).cuda(), torch.from_numpy(off).cuda()
    E = torch.zeros(int(off[-1]), C, device="cuda")
    g_at, _ = ops.grid_encode_bwd(Gt, xt, E, offt, L, S, H, 0, False, 0)
    g_bin = ops.grid_encode_bwd_binned(Gt, xt, offt, C, L, S, H)
    assert g_bin.dtype == torch.float32 and g_bin.shape == E.shape
    scale = float(g_at.abs().max())
    assert float((g_bin - g_at).abs().max()) < 2e-5 * scale          # fp32 atomics: order-dependent last bits; the binned sums are exact
    if B <= 50000:
        gE_ref, _ = og.grid_encode_backward(np.ascontiguousarray(G.reshape(B, L, C).transpose(1, 0, 2)), x, off, E.shape[0], S, H, 0, False, 0)
        np.testing.assert_allclose(g_bin.cpu().numpy(), gE_ref, rtol=1e-4, atol=2e-5 * scale)
    assert torch.equal(ops.grid_encode_bwd_binned(Gt, xt, offt, C, L, S, H), g_bin), "not bit-reproducible"
    G_lm = Gt.reshape(B, L, C).permute(1, 0, 2).contiguous()
    assert torch.equal(ops.grid_encode_bwd_binned(G_lm, xt, offt, C, L, S, H, level_major=True), g_bin)
    # fp16 gradients (the reference's autocast case): records rounded once to 11 bits, sums exact; half output like the reference's
    g16 = ops.grid_encode_bwd_binned(Gt.half(), xt, offt, C, L, S, H)
    rel = float((g16 - g_bin).norm() / g_bin.norm())
    assert rel < 1e-3, rel
    g16h = ops.grid_encode_bwd_binned(Gt.half(), xt, offt, C, L, S, H, out_dtype=torch.float16)
    assert g16h.dtype == torch.float16 and float((g16h.float() - g16).abs().max()) <= 1e-3 * scale + 1e-3
    # adjointness with the fast forward
    Et = torch.from_numpy((rng.standard_normal(E.shape) * 0.5).astype(np.float32)).cuda()
    out, _ = ops.grid_encode_fwd(xt, Et, offt, L, S, H, 0, False, 0)
    lhs, rhs = float((out.double() * Gt.double()).sum()), float((Et.double() * g_bin.double()).sum())
    assert abs(lhs - rhs) < 1e-3 * max(1.0, abs(lhs))


@pytest.mark.parametrize("C,half", [(4, True), (4, False), (1, True), (1, False), (2, False), (8, False)])
def test_grid_binned_backward_chunked_is_the_single_pass_bit_for_bit(C, half):
    """The workspace bounds the chunk size, not the result: with a workspace that forces several chunks per level (the level's int64
    image carries the partial sums) and one transposed level at a time, the gradient is bit-identical to the one-chunk run, in both
    gradient layouts; the plan query reports the chunking."""
    from snerf_amd import ops
    L, H, B = 6, 8, 70001
    off, res, s = og.level_layout(3, L, C, 2.0, H, 14, 512, False)
    S = float(np.log2(s))
    x = _clumped_points(B, 5)
    rng = np.random.default_rng(6)
    G = (rng.standard_normal((B, L * C)) * 0.3).astype(np.float32)
    xt, offt = torch.from_numpy(x).cuda(), torch.from_numpy(off).cuda()
    Gt = torch.from_numpy(G).cuda()
    Gt = Gt.half() if half else Gt
    one = ops.grid_encode_bwd_binned(Gt, xt, offt, C, L, S, H)
    p1 = ops.grid_encode_bwd_binned_plan(B, C, L, off, half, Gt.dtype)
    assert p1["chunks"] == 1 and p1["levels_per_transposed_group"] == L and p1["bytes_used"] <= p1["ws_bytes"]
    G_lm = Gt.reshape(B, L, C).permute(1, 0, 2).contiguous()
    seen = set()
    for frac in (0.7, 0.5, 0.35):
        ws = int(p1["bytes_used"] * frac)
        pk = ops.grid_encode_bwd_binned_plan(B, C, L, off, half, Gt.dtype, ws_bytes=ws)
        assert pk["bytes_used"] <= ws
        seen.add((pk["chunks"], pk["levels_per_transposed_group"]))
        assert torch.equal(ops.grid_encode_bwd_binned(Gt, xt, offt, C, L, S, H, ws_bytes=ws), one), pk
        assert torch.equal(ops.grid_encode_bwd_binned(G_lm, xt, offt, C, L, S, H, level_major=True, ws_bytes=ws), one), pk
    assert any(c > 1 for c, _ in seen) and any(lt < L for _, lt in seen), seen
    with pytest.raises(Exception):
        ops.grid_encode_bwd_binned(Gt, xt, offt, C, L, S, H, ws_bytes=4096)       # below the smallest feasible layout: refused, not overrun


@pytest.mark.parametrize("C,half", [(4, True), (4, False), (1, False), (2, False), (8, False)])
def test_grid_binned_backward_many_bins_several_stage_windows(C, half):
    """Hashed levels of 2^19 rows (128-1024 row ranges): a workgroup's 32 768 records spread over hundreds of bins and pass through the
    LDS stage in several windows (the bin that straddles a window boundary, the empty head of the next window), next to dense levels
    that are written directly -- against the atomic scatter (fp32 atomics) and run to run."""
    from snerf_amd import ops
    L, H, B = 5, 16, 40009
    off, res, s = og.level_layout(3, L, C, 2.6, H, 19, None, False)
    S = float(np.log2(s))
    rng = np.random.default_rng(21)
    x = np.concatenate([rng.random((B - 9000, 3)).astype(np.float32), _clumped_points(9000, 2)])      # scattered points: no merging on the fine levels
    G = (rng.standard_normal((B, L * C)) * 0.3).astype(np.float32)
    xt, offt = torch.from_numpy(x).cuda(), torch.from_numpy(off).cuda()
    Gt = torch.from_numpy(G).cuda()
    Gt = Gt.half() if half else Gt
    E = torch.zeros(int(off[-1]), C, device="cuda")
    g_at, _ = ops.grid_encode_bwd(Gt.float(), xt, E, offt, L, S, H, 0, False, 0)
    g_bin = ops.grid_encode_bwd_binned(Gt, xt, offt, C, L, S, H)
    scale = float(g_at.abs().max())
    tol = 2e-3 if half else 2e-5
    assert float((g_bin - g_at).abs().max()) <= tol * scale, float((g_bin - g_at).abs().max()) / scale
    assert torch.equal(ops.grid_encode_bwd_binned(Gt, xt, offt, C, L, S, H), g_bin), "not bit-reproducible"


@pytest.mark.parametrize("C", [4, 1])
def test_grid_binned_backward_merged_runs_do_not_saturate(C):
    """Eight consecutive identical points with identical same-sign gradients merge into ONE record of 8 x w x grad (the reference's
    callers pass an interval's multisamples one after the other, and zipnerf mean-pools them: identical gradients): the record must not
    clamp at the half range / the fixed-point limit -- the scale keeps 3 head bits for the merge.  Against the atomic scatter, half and
    fp32 gradients, with the largest entries among the merged ones."""
    from snerf_amd import ops
    L, H = 4, 8
    off, res, s = og.level_layout(3, L, C, 2.0, H, 14, 512, False)
    S = float(np.log2(s))
    rng = np.random.default_rng(8)
    base = rng.random((512, 3)).astype(np.float32) * 0.9 + 0.05
    x = np.repeat(base, 8, axis=0)                                       # 8 identical points in a row: one cell on every level
    g1 = np.abs(rng.standard_normal((512, L * C))).astype(np.float32) + 0.5
    g1[:4] *= 40.0                                                        # the launch's max |grad| sits in merged runs
    G = np.repeat(g1, 8, axis=0)
    xt, offt = torch.from_numpy(x).cuda(), torch.from_numpy(off).cuda()
    E = torch.zeros(int(off[-1]), C, device="cuda")
    for half in (False, True):
        Gt = torch.from_numpy(G).cuda()
        Gt = Gt.half() if half else Gt
        g_at, _ = ops.grid_encode_bwd(Gt.float(), xt, E, offt, L, S, H, 0, False, 0)      # fp32 atomics on the same (rounded) gradients
        g_bin = ops.grid_encode_bwd_binned(Gt, xt, offt, C, L, S, H)
        scale = float(g_at.abs().max())
        assert scale > 8 * 20.0
        tol = 2e-3 if half else 2e-5                                     # half records: one rounding to 11 bits per contribution
        assert float((g_bin - g_at).abs().max()) <= tol * scale, (half, float((g_bin - g_at).abs().max()), scale)


def test_gridencoder_module_takes_the_fast_path_without_input_gradients():
    """GridEncoder.forward / backward as zipnerf calls it (no gradient to the positions; fp32 and autocast): corner-cached gather +
    binned table gradient behind the reference's module API, against the oracle."""
    from snerf_amd.gridencoder import GridEncoder
    from snerf_amd import ops
    assert ops.GRID_FAST
    for C in (4, 1):
        enc = GridEncoder(input_dim=3, num_levels=6, level_dim=C, base_resolution=16, log2_hashmap_size=15, desired_resolution=512)
        torch.manual_seed(11)
        with torch.no_grad():
            enc.embeddings.normal_(0, 0.2)
        x = torch.from_numpy(_clumped_points(7001, 9) * 2 - 1).cuda()
        g = torch.Generator().manual_seed(1)
        w = torch.randn(7001, 6 * C, generator=g).cuda()
        off, S = enc.offsets.cpu().numpy(), float(np.log2(enc.per_level_scale))
        x01 = ((x.cpu().numpy() + 1) / 2).astype(np.float32)
        ref = og.grid_encode_forward(x01, enc.embeddings.detach().cpu().numpy(), off, S, 16, 0, False, 0)
        gE, _ = og.grid_encode_backward(np.ascontiguousarray(w.cpu().numpy().reshape(-1, 6, C).transpose(1, 0, 2)), x01, off, enc.embeddings.shape[0], S, 16, 0, False, 0)
        y = enc(x, bound=1)
        np.testing.assert_allclose(y.detach().cpu().numpy().reshape(-1, 6, C), ref.transpose(1, 0, 2), rtol=1e-4, atol=2e-5)
        (y * w).sum().backward()
        np.testing.assert_allclose(enc.embeddings.grad.cpu().numpy(), gE, rtol=1e-4, atol=2e-4)
        enc.embeddings.grad = None
        with torch.autocast("cuda", dtype=torch.float16):
            y16 = enc(x)
        (y16.float() * w).sum().backward()
        assert y16.dtype == (torch.float16 if C == 4 else torch.float32) and enc.embeddings.grad.dtype == torch.float32   # (odd C: the table is not halved, grid.py:41-44)
        tol = 2e-2 if C == 4 else 1e-3                          # C = 4: half table and half gradients (grid.py:41-44); C = 1 stays fp32
        assert float((enc.embeddings.grad - torch.from_numpy(gE).cuda()).norm() / np.linalg.norm(gE)) < tol


@pytest.mark.parametrize("B,L,oob", [(1, 3, False), (5, 1, False), (64, 4, True), (2049, 2, False)])
def test_grid_fast_path_edge_cases(B, L, oob):
    """one point, one level, every point out of bounds, one point past a workgroup's 2048: forward and binned backward vs the oracle"""
    from snerf_amd import ops
    C, H = 4, 8
    off, res, s = og.level_layout(3, L, C, 2.0, H, 12, None, False)
    S = float(np.log2(s))
    rng = np.random.default_rng(B + L)
    E = (rng.standard_normal((int(off[-1]), C)) * 0.5).astype(np.float32)
    x = rng.random((B, 3)).astype(np.float32)
    if oob:
        x[:, 0] += 1.5
    G = rng.standard_normal((B, L * C)).astype(np.float32)
    xt, Et, Gt, offt = (torch.from_numpy(a).cuda() for a in (x, E, G, off))
    out, _ = ops.grid_encode_fwd(xt, Et, offt, L, S, H, 0, False, 0)
    ref = og.grid_encode_forward(x, E, off, S, H, 0, False, 0)
    np.testing.assert_allclose(out.cpu().numpy().reshape(B, L, C), ref.transpose(1, 0, 2), rtol=1e-4, atol=2e-5)
    gE = ops.grid_encode_bwd_binned(Gt, xt, offt, C, L, S, H)
    gE_ref, _ = og.grid_encode_backward(np.ascontiguousarray(G.reshape(B, L, C).transpose(1, 0, 2)), x, off, E.shape[0], S, H, 0, False, 0)
    np.testing.assert_allclose(gE.cpu().numpy(), gE_ref, rtol=1e-4, atol=1e-5)
    if oob:
        assert float(out.abs().max()) == 0.0 and float(gE.abs().max()) == 0.0
