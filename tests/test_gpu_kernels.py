"""Stage-isolated parity of every HIP kernel against the CPU oracle (through the C-ABI).
Integer / index outputs are compared bit-exactly; floating point within the tolerance written at each check."""
import numpy as np
import pytest
import torch

from oracle import classic as oc
from oracle import common, mip as om

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from snerf_amd import ops as _ops
    return _ops


def dev(t):
    return t.cuda() if torch.is_tensor(t) else t


def close(a, b, rtol, atol, what=""):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    assert torch.equal(torch.isnan(a), torch.isnan(b)), f"{what}: NaN pattern differs"
    a = torch.nan_to_num(a, nan=0.0); b = torch.nan_to_num(b, nan=0.0)
    err = (a - b).abs(); tol = atol + rtol * b.abs()
    bad = err > tol
    assert not bool(bad.any()), f"{what}: {int(bad.sum())}/{bad.numel()} out of tol, max err {err.max().item():.3e} at {np.unravel_index(int(err.argmax()), err.shape)}"


def gen(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


# ------------------------------------------------------------------ GEMM ----
@pytest.mark.parametrize("dt,M,N,K,variant", [(0, 300, 128, 96, 0), (0, 1000, 256, 1120, 0), (1, 300, 128, 128, 0),
                                              (1, 1000, 256, 1152, 0), (1, 777, 256, 320, 1), (1, 2048, 1024, 1024, 1),
                                              (1, 777, 256, 320, 4), (1, 2048, 1024, 1024, 4), (1, 1000, 512, 64, 4), (1, 1000, 256, 192, 4),
                                              (1, 777, 256, 320, 8), (1, 2048, 1024, 1024, 8), (1, 70000, 512, 128, 8), (1, 100000, 256, 192, 8),
                                              # fp16 (dtype 2): the same kernels with the fp16 MFMA and conversions -- 3 more mantissa bits than bf16
                                              (2, 300, 128, 128, 0), (2, 777, 256, 320, 1), (2, 2048, 1024, 1024, 4), (2, 777, 256, 320, 8),
                                              (2, 70000, 512, 128, 8), (2, 100000, 256, 192, 8)])
def test_linear_fwd(ops, dt, M, N, K, variant):
    tdt = ops.torch_dtype(dt)
    A = gen(M, K + ops.gran(dt), seed=1).to(tdt).cuda()            # wider buffer: lda != K
    W = (gen(N, K, seed=2) / K ** 0.5).to(tdt).cuda()
    bias = gen(N, seed=3).cuda()
    n_store = N - 5
    ref = torch.relu(A[:, :K].double().cpu() @ W.double().cpu().t() + bias.double().cpu())[:, :n_store]
    Y = torch.full((M, N), 7.0, dtype=tdt, device="cuda")
    ops.linear_fwd(A, W, bias, Y, K, n_store, ops.ACT_RELU, dt, variant=variant)
    tol = {0: 1e-5, 1: 1e-2, 2: 1.5e-3}[dt]
    close(Y[:, :n_store], ref, tol, tol, "relu out")
    assert bool((Y[:, n_store:] == 7.0).all()), "columns >= n_store must not be written"
    # fp32 output, no activation, column-offset destination
    Y2 = torch.zeros(M, 4, dtype=torch.float32, device="cuda")
    ops.linear_fwd(A, W, bias, Y2[:, 3:], K, 1, ops.ACT_NONE, dt, out_f32=True, variant=variant)
    ref2 = (A[:, :K].double().cpu() @ W.double().cpu().t() + bias.double().cpu())[:, :1]
    close(Y2[:, 3:], ref2, 1e-5 if dt == 0 else 2e-3, 1e-5 if dt == 0 else 2e-3, "head out")       # (fp32 accumulation of 16-bit products)
    assert bool((Y2[:, :3] == 0).all())


@pytest.mark.parametrize("dt,variant,M", [(0, 0, 515), (1, 0, 515), (1, 1, 515), (1, 4, 515), (1, 8, 515), (1, 8, 70001), (2, 0, 515), (2, 4, 515),
                                          (2, 8, 515), (2, 8, 70001)])
def test_linear_dgrad_mask_colsum(ops, dt, variant, M):
    tdt = ops.torch_dtype(dt)
    Nred, Kout = 192, 256
    dZ = gen(M, Nred, seed=4).to(tdt).cuda()
    Wt = (gen(Kout, Nred, seed=5) / Nred ** 0.5).to(tdt).cuda()
    act = gen(M, Kout, seed=6).to(tdt).cuda()
    dX = torch.zeros(M, Kout, dtype=tdt, device="cuda")
    cs = torch.zeros(Kout, dtype=torch.float32, device="cuda")
    ops.linear_fwd(dZ, Wt, None, dX, Nred, Kout, ops.ACT_MASK, dt, aux=act, colsum=cs, variant=variant)
    ref = (dZ.double().cpu() @ Wt.double().cpu().t()) * (act.double().cpu() > 0)
    tol = {0: 1e-5, 1: 1e-2, 2: 1.5e-3}[dt]
    close(dX, ref, tol, tol, "dgrad")
    close(cs, ref.sum(0), 1e-4 if dt == 0 else 2e-2, (1e-3 if dt == 0 else 5e-2) * max(1.0, (M / 515) ** 0.5), "colsum")


@pytest.mark.parametrize("dt,M,N,K,nv,kv", [(0, 700, 96, 128, 90, 127), (0, 5000, 256, 1120, 256, 1120), (1, 700, 64, 128, 3, 128),
                                            (1, 5000, 256, 320, 256, 283), (1, 3000, 1024, 1152, 1024, 1120), (1, 4100, 256, 128, 256, 96), (1, 2077, 128, 1024, 128, 1024),
                                            (1, 70001, 1024, 1152, 1024, 1120), (1, 9000, 256, 320, 256, 283), (1, 33000, 512, 256, 500, 256),
                                            (2, 700, 64, 128, 3, 128), (2, 5000, 256, 320, 256, 283), (2, 70001, 1024, 1152, 1024, 1120), (2, 33000, 512, 256, 500, 256)])
def test_linear_wgrad(ops, dt, M, N, K, nv, kv):
    tdt = ops.torch_dtype(dt)
    dZ = gen(M, N, seed=7).to(tdt).cuda()
    X = gen(M, K, seed=8).to(tdt).cuda()
    ref = 1.0 + (dZ.double().cpu().t() @ X.double().cpu())[:nv, :kv]
    for variant in (0, 1, 2):                                       # 1 = transposing LDS reads (bf16, whole 128-column tiles), 2 = 8-phase 256x256
        dW = torch.ones(nv, kv, dtype=torch.float32, device="cuda")   # accumulates on top of existing content
        ops.linear_wgrad(dZ, X, dW, nv, kv, dt, variant=variant)
        close(dW, ref, 1e-4, 1e-3 * (M / 1000) ** 0.5, f"wgrad variant {variant}")


def _split_layout(x):
    """fp32 [M, K] (K % 64 == 0) -> the split-bf16 activation layout [M, 2 K]: per 64 logical columns [hi 64 | lo 64]"""
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    M, K = x.shape
    return torch.stack([hi.reshape(M, K // 64, 64), lo.reshape(M, K // 64, 64)], 2).reshape(M, 2 * K).contiguous()


@pytest.mark.parametrize("M,N,K,nv,kv", [(700, 64, 128, 3, 128), (5000, 256, 320, 256, 283), (3000, 1024, 1152, 1024, 1120), (4100, 256, 128, 256, 96),
                                         (70001, 1024, 1152, 1024, 1120), (9000, 256, 320, 256, 283), (33000, 512, 256, 500, 256), (4500, 64, 1024, 1, 1024)])
def test_linear_wgrad_hi_half_of_split_activation(ops, M, N, K, nv, kv):
    """compute="bf16x3_fwd": the single-pass weight gradient reads the hi half of an activation the split-bf16 forward saved (variant bit 14 of
    snerf_linear_wgrad).  Both kernels (128 x 128 and the 8-phase 256 x 256), atomics and the partial-tile fold, X as a column range of a wider
    buffer; the result must equal the plain launch on the de-interleaved hi half BIT FOR BIT where the summation order is fixed (fold)."""
    dZ = gen(M, N, seed=7).to(torch.bfloat16).cuda()
    Kp = ((K + 63) // 64) * 64
    x = torch.zeros(M, Kp)
    x[:, :K] = gen(M, K, seed=8)
    wide = torch.zeros(M, 2 * Kp + 256, dtype=torch.bfloat16)
    wide[:, 128:128 + 2 * Kp] = _split_layout(x)
    wide[:, :128] = 7.0; wide[:, 128 + 2 * Kp:] = -5.0              # neighbours of the column range must not leak in
    Xs = wide.cuda()[:, 128:128 + 2 * Kp]
    Xh = x.to(torch.bfloat16).cuda()
    ref = 1.0 + (dZ.double().cpu().t() @ Xh.double().cpu())[:nv, :kv]
    for variant in (0, 1, 2, 3):
        for det in (False, True):
            dW = torch.ones(nv, kv, dtype=torch.float32, device="cuda")
            ops.linear_wgrad(dZ, Xs, dW, nv, kv, ops.BF16, variant=variant, deterministic=det, x_split_hi=True)
            close(dW, ref, 1e-4, 1e-3 * (M / 1000) ** 0.5, f"hi-half wgrad variant {variant} det {det}")
            if det:
                dW2 = torch.ones(nv, kv, dtype=torch.float32, device="cuda")
                ops.linear_wgrad(dZ, Xh, dW2, nv, kv, ops.BF16, variant=variant, deterministic=True)
                assert torch.equal(dW, dW2), f"variant {variant}: the hi-half launch and the plain launch on the same values differ"


@pytest.mark.parametrize("M,Nred,Kout,variant", [(515, 192, 256, 0), (515, 192, 256, 8), (70001, 64, 256, 8), (3000, 128, 128, 8), (3000, 256, 1024, 4)])
def test_linear_dgrad_mask_from_split_activation(ops, M, Nred, Kout, variant):
    """compute="bf16x3_fwd": a plain bf16 data gradient whose ReLU mask source is a split-bf16 activation (variant bit 14 of snerf_linear_fwd):
    identical to the plain launch masked by the de-interleaved hi half."""
    dZ = gen(M, Nred, seed=4).to(torch.bfloat16).cuda()
    Wt = (gen(Kout, Nred, seed=5) / Nred ** 0.5).to(torch.bfloat16).cuda()
    a = gen(M, Kout, seed=6)
    a = torch.where(a.abs() < 0.3, torch.zeros_like(a), a)        # exact zeros and values whose hi half is zero-adjacent
    act_s = _split_layout(a).cuda()
    act_h = a.to(torch.bfloat16).cuda()
    for n_store in (Kout, Kout - 8):
        out = []
        for aux, sp in ((act_s, True), (act_h, False)):
            dX = torch.full((M, Kout), 3.0, dtype=torch.bfloat16, device="cuda")
            cs = torch.zeros(Kout, dtype=torch.float32, device="cuda")
            ops.linear_fwd(dZ, Wt, None, dX, Nred, n_store, ops.ACT_MASK, ops.BF16, aux=aux, colsum=cs, variant=variant, aux_split=sp)
            out.append((dX, cs))
        assert torch.equal(out[0][0], out[1][0]), "masked data gradient differs from the plain launch on the hi half"
        close(out[0][1], out[1][1], 1e-5, 1e-3 * (M / 515) ** 0.5, "bias gradient")
        ref = (dZ.double().cpu() @ Wt.double().cpu().t()) * (act_h.double().cpu() > 0)
        close(out[0][0][:, :n_store], ref[:, :n_store], 1e-2, 1e-2, "dgrad")


def _s8_unpack(T, K, weight=False):
    """rows in the fp16 + fp8 split layout (fp16-typed [M, >= 2 K]) -> (fp16 part, e4m3 residual / 2^s, e4m3 value / 2^s) as float64"""
    M = T.shape[0]
    t = T[:, :2 * K].contiguous().cpu().view(M, K // 64, 128)
    hi = t[:, :, :64].double().reshape(M, K)
    by = t[:, :, 64:].contiguous().view(torch.uint8).reshape(M, K // 64, 128)
    f, s = by[:, :, :64].contiguous().view(torch.float8_e4m3fn).double().reshape(M, K), by[:, :, 64:].contiguous().view(torch.float8_e4m3fn).double().reshape(M, K)
    return (hi, s / 2.0 ** 20, f / 2.0 ** 9) if weight else (hi, f / 2.0 ** 13, s / 2.0 ** 2)


@pytest.mark.parametrize("M,N,K,variant", [(300, 128, 128, 0), (1000, 256, 1152, 0), (777, 256, 320, 8), (2048, 1024, 1024, 8), (70000, 512, 128, 8), (100000, 256, 192, 8),
                                           (515, 256, 64, 8)])
def test_linear_fwd_f16f8(ops, M, N, K, variant):
    """dtype F16F8 (fp16 tiles + e4m3 correction tiles on the block-scaled MFMA): (1) ops.split8_cast against a host conversion; (2) the product
    against the float64 sum of exactly the three terms the kernels multiply (hi.hi + r8.w8 + x8.wr8 from the unpacked operands): 2e-5 of the row
    scale sum |a||w| -- the block-scaled MFMA sums its 64 e4m3 products with about 6 significant bits of their largest one (measured: 2^-6 of the
    correction's own magnitude, which is 2^-11 of the product's), fp32 accumulation noise elsewhere; (3) against the TRUE product of the fp32 inputs: 2^-14 of sum |a||w| (bf16 alone: 2^-8); (4) the
    interleaved fp16 + fp8 output equals split8_cast of the fp32 output of the same launch, ReLU bit masks included."""
    a32 = gen(M, K, seed=1) * (0.25 + 3.75 * torch.rand(M, 1, generator=torch.Generator().manual_seed(5)))   # (rows of O(1) scale: the e4m3 residual has an ABSOLUTE floor of 2^-23)
    w32 = gen(N, K, seed=2) / K ** 0.5
    A = torch.zeros(M, 2 * K + 128, dtype=torch.float16, device="cuda")            # wider buffer: lda != 2 K
    ops.split8_cast(a32.cuda(), K, A, K)
    W = torch.zeros(N, 2 * K, dtype=torch.float16, device="cuda")
    ops.split8_cast(w32.cuda(), K, W, K, weight=True)
    ah, ar, ax = _s8_unpack(A, K)
    wh, wr, wx = _s8_unpack(W, K, weight=True)
    assert torch.equal(ah.float(), a32.half().float()) and torch.equal(wh.float(), w32.half().float())
    q = lambda x, m: (x * m).clamp(-448, 448).to(torch.float8_e4m3fn).double() / m
    assert torch.equal(ar, q(a32 - a32.half().float(), 2.0 ** 13)) and torch.equal(ax, q(a32, 4.0)), "activation bytes"
    assert torch.equal(wr, q(w32 - w32.half().float(), 2.0 ** 20)) and torch.equal(wx, q(w32, 512.0)), "weight bytes"
    bias = gen(N, seed=3).cuda()
    terms = ah @ wh.t() + ar @ wx.t() + ax @ wr.t() + bias.double().cpu()
    true = a32.double() @ w32.double().t() + bias.double().cpu()
    scale = (a32.double().abs() @ w32.double().abs().t())
    Y32 = torch.zeros(M, N, dtype=torch.float32, device="cuda")
    ops.linear_fwd(A, W, bias, Y32, K, N, ops.ACT_NONE, ops.F16F8, out_f32=True, variant=variant)        # (fp32 outputs: the 128 x 128 kernel)
    e_terms = float(((Y32.double().cpu() - terms).abs() / scale).max())
    e_true = float(((Y32.double().cpu() - true).abs() / scale).max())
    print(f"MEASURED f16f8 M {M} N {N} K {K}: vs its own three terms {e_terms:.2e}, vs the true product {e_true:.2e} of sum |a||w|")
    assert e_terms < 3e-5 and e_true < 2.0 ** -14
    n_store = N - 64
    Y = torch.full((M, 2 * N), 7.0, dtype=torch.float16, device="cuda")
    act = ops.ACT_RELU
    bits = None
    if ops.relu_bits_ok(A, W, Y, K, n_store, ops.F16F8, variant):
        act, bits = ops.ACT_RELU_BITS, torch.zeros(ops.mask_bits_words(M, N), dtype=torch.int32, device="cuda")
    ops.linear_fwd(A, W, bias, Y, K, n_store, act, ops.F16F8, aux=bits, variant=variant)                  # (variant 8: the persistent kernel)
    assert bool((Y[:, 2 * n_store:] == 7.0).all()), "columns >= n_store must not be written"
    yr = torch.relu(Y32).double().cpu()[:, :n_store]
    yh, yr8, yx8 = (t[:, :n_store] for t in _s8_unpack(Y, N))
    # the kernels' own fp32 value differs from Y32 by summation order only: hi part within two fp16 ulps, hi + residual within 2^-13 of the value
    # (+ the residual's absolute floor), the value byte within an e4m3 ulp (2^-3) or its floor
    atol = 8e-6 * (K / 64) ** 0.5                                  # (what two summation orders of K products differ by)
    assert bool(((yh - yr).abs() <= 2.0 ** -10 * yr.abs() + atol).all())
    assert bool(((yh + yr8 - yr).abs() <= 2.0 ** -14 * yr.abs() + atol).all())
    assert bool(((yx8 - yr).abs() <= 2.0 ** -3 * yr.abs() + 2.0 ** -10).all())
    assert bool(((yh > 0) == (yr > 0))[(yr.abs() > 1e-3)].all())
    if variant == 0:                                                 # same kernel as the fp32 launch: the very same accumulators, so the bytes are those of split8_cast
        exp = _cast_ref(ops, torch.relu(Y32), N)
        assert torch.equal(Y[:, :2 * n_store], exp[:, :2 * n_store]), "interleaved output != split8_cast(fp32 output)"
    if bits is not None:                                             # the bit mask reproduces the sign pattern: a plain fp16 data gradient masked by it == masked by the hi halves
        dZ = gen(M, 256, seed=24).half().cuda()
        Wt = (gen(N, 256, seed=25) / 16).half().cuda()
        d1 = torch.zeros(M, N, dtype=torch.float16, device="cuda"); d2 = torch.zeros_like(d1)
        ops.linear_fwd(dZ, Wt, None, d1, 256, n_store, ops.ACT_MASK_BITS, ops.F16, aux=bits, variant=8)
        ops.linear_fwd(dZ, Wt, None, d2, 256, n_store, ops.ACT_MASK, ops.F16, aux=Y, variant=0, aux_split=True)
        assert torch.equal(d1, d2)


def _cast_ref(ops, y32, N):
    out = torch.zeros(y32.shape[0], 2 * N, dtype=torch.float16, device="cuda")
    ops.split8_cast(y32, N, out, N)
    return out


@pytest.mark.parametrize("M,N,K,dt", [(1000, 256, 256, 1), (70001, 512, 320, 1), (4096, 1024, 1024, 1), (1000, 256, 256, 2), (70001, 512, 320, 2)])
def test_relu_bit_mask_roundtrip(ops, M, N, K, dt):
    """ACT_RELU_BITS writes the activation AND a 1-bit mask; ACT_MASK_BITS must reproduce ACT_MASK on that activation exactly (bf16 and fp16)."""
    h16 = ops.torch_dtype(dt)
    A = gen(M, K, seed=21).to(h16).cuda()
    W = (gen(N, K, seed=22) / K ** 0.5).to(h16).cuda()
    bias = gen(N, seed=23).cuda()
    h_ref = torch.empty(M, N, dtype=h16, device="cuda")
    ops.linear_fwd(A, W, bias, h_ref, K, N, ops.ACT_RELU, dt, variant=8)
    h = torch.empty_like(h_ref)
    bits = torch.zeros(ops.mask_bits_words(M, N), dtype=torch.int32, device="cuda")
    assert ops.relu_bits_ok(A, W, h, K, N, dt, 8)
    ops.linear_fwd(A, W, bias, h, K, N, ops.ACT_RELU_BITS, dt, aux=bits, variant=8)
    assert torch.equal(h, h_ref)
    frac = float((h_ref > 0).float().mean())
    assert 0.2 < frac < 0.8
    # data gradient of the next layer: dX[M, N] = dZ[M, K2] . Wt[N, K2]^T masked by h > 0
    K2 = 256 if M != 70001 else 128                              # 128: two k-tiles, the mask DMA is drained explicitly
    dZ = gen(M, K2, seed=24).to(h16).cuda()
    Wt = (gen(N, K2, seed=25) / K2 ** 0.5).to(h16).cuda()
    for with_cs in (True, False):
        cs1 = torch.zeros(N, device="cuda") if with_cs else None
        cs2 = torch.zeros(N, device="cuda") if with_cs else None
        d1 = torch.empty(M, N, dtype=h16, device="cuda")
        d2 = torch.full((M, N), 3.0, dtype=h16, device="cuda")
        ops.linear_fwd(dZ, Wt, None, d1, K2, N, ops.ACT_MASK, dt, aux=h_ref, colsum=cs1, variant=8)
        assert ops.relu_bits_ok(dZ, Wt, d2, K2, N, dt, 8, consumer=True)
        ops.linear_fwd(dZ, Wt, None, d2, K2, N, ops.ACT_MASK_BITS, dt, aux=bits, colsum=cs2, variant=8)
        assert torch.equal(d1, d2)
        if with_cs:
            assert torch.allclose(cs1, cs2, rtol=1e-5, atol=1e-3)


def test_gemm_operands_as_column_ranges(ops):
    """Operands are column ranges of wider buffers (concat-by-columns layout); whatever surrounds them must not leak in."""
    dt, M, N, K = 1, 9000, 256, 320
    big = torch.full((M, 64 + K + 64), float("nan"), dtype=torch.bfloat16, device="cuda")
    X = big[:, 64:64 + K]
    X.copy_(gen(M, K, seed=11).to(torch.bfloat16))
    bigz = torch.full((M, 32 + N), float("nan"), dtype=torch.bfloat16, device="cuda")
    dZ = bigz[:, 32:]
    dZ.copy_(gen(M, N, seed=12).to(torch.bfloat16))
    ref = dZ.double().cpu().t() @ X.double().cpu()
    for variant in (1, 3):
        dW = torch.zeros(N, K, dtype=torch.float32, device="cuda")
        ops.linear_wgrad(dZ, X, dW, N, K, dt, variant=variant)
        close(dW, ref, 1e-4, 3e-3, f"wgrad on column ranges, variant {variant}")
    W = (gen(N, K, seed=13) / K ** 0.5).to(torch.bfloat16).cuda()
    refy = torch.relu(X.double().cpu() @ W.double().cpu().t())
    for variant in (1, 4, 8):
        ybig = torch.full((M, N + 64), 5.0, dtype=torch.bfloat16, device="cuda")
        ops.linear_fwd(X, W, None, ybig[:, 64:], K, N, ops.ACT_RELU, dt, variant=variant)
        close(ybig[:, 64:], refy, 1e-2, 1e-2, f"fwd on column ranges, variant {variant}")
        assert bool((ybig[:, :64] == 5.0).all())


@pytest.mark.parametrize("dt", [0, 1])
def test_zip_glo_modulation_kernels(ops, dt):
    """snerf_zip_glo_modulate / _bwd (internal/models.py:620-630: bottleneck * exp(scale) + shift per ray) and snerf_colsum_wide_f32 against
    torch: forward, d x (incl. the head gradients that enter its leading columns), d (scale | shift) per ray and the per-ray column sums."""
    tdt = ops.torch_dtype(dt)
    R, S, B, nh = 37, 5, 256, 20
    g = torch.Generator().manual_seed(21)
    X = (torch.randn(R * S, B + 16, generator=g)).to(tdt).cuda()
    SS = (torch.randn(R, 2 * B, generator=g) * 0.3).cuda()
    out = torch.zeros(R * S, B + 8, dtype=tdt, device="cuda")
    ops.zip_glo_modulate(X, SS, S, out)
    ss = SS.repeat_interleave(S, 0)
    ref = X[:, :B].float() * torch.exp(ss[:, :B]) + ss[:, B:]
    rt = 2e-6 if dt == 0 else 8e-3
    close(out[:, :B].float(), ref, rt, 1e-6, "modulated bottleneck")
    assert bool((out[:, B:] == 0).all())
    dXm = (torch.randn(R * S, B, generator=g)).to(tdt).cuda()
    dh = torch.randn(R * S, nh, generator=g).cuda()
    dX = torch.empty(R * S, B, dtype=tdt, device="cuda")
    dSS, dxsum = ops.zip_glo_modulate_bwd(dXm, X, SS, dh, S, dX)
    e = torch.exp(ss[:, :B])
    dx_ref = dXm.float() * e
    dx_ref[:, :nh] += dh
    close(dX.float(), dx_ref, rt, 1e-6, "d x")
    close(dSS[:, :B], ((dXm.float() * X[:, :B].float()) * e).view(R, S, B).sum(1), 2e-5, 1e-5, "d scale")
    close(dSS[:, B:], dXm.float().view(R, S, B).sum(1), 2e-5, 1e-5, "d shift")
    close(dxsum, dX.float().view(R, S, B).sum(1), 2e-5, 1e-5, "per-ray column sums of the stored d x")
    for det in (False, True):
        acc = torch.ones(2 * B, device="cuda")
        ops.colsum_wide_f32(dSS, 2 * B, acc, deterministic=det)
        close(acc, 1 + dSS.double().sum(0).float(), 2e-5, 1e-4, "wide column sum")


# -------------------------------------------------------------- encoders ----
@pytest.mark.parametrize("dt", [0, 1])
def test_classic_embed(ops, dt):
    tdt = ops.torch_dtype(dt)
    N, S = 7, 5
    pts = gen(N * S, 3, seed=9, scale=4.0)
    vd = torch.nn.functional.normalize(gen(N, 3, seed=10), dim=-1)
    rb = torch.cat([torch.zeros(N, 8), vd], -1).cuda()             # strided viewdirs like ray_batch[:, -3:]
    g = ops.gran(dt)
    E = torch.full((N * S, 64), 9.0, dtype=tdt, device="cuda"); SK = torch.full((N * S, 64 + 32), 9.0, dtype=tdt, device="cuda")
    V = torch.full((N * S, 16 + g), 9.0, dtype=tdt, device="cuda")
    ops.classic_embed(pts.cuda(), rb[:, -3:], S, 10, 4, E, SK[:, :64], 64, V[:, 16:], g, dt)
    ref_p = oc.embed(pts, 10); ref_v = oc.embed(vd[:, None].expand(N, S, 3).reshape(-1, 3), 4)
    rt, at = (0, 2e-6) if dt == 0 else (8e-3, 1e-5)   # bf16: one rounding of the fp32 value (2^-8 relative)
    close(E[:, :63], ref_p, rt, at, "pts embed"); close(SK[:, :63], ref_p, rt, at, "pts embed copy")
    close(V[:, 16:16 + 27], ref_v, rt, at, "view embed")
    assert bool((E[:, 63] == 0).all()) and bool((V[:, 16 + 27:] == 0).all()) and bool((V[:, :16] == 9).all()) and bool((SK[:, 64:] == 9).all())


@pytest.mark.parametrize("shape", ["cone", "cylinder"])
def test_mip_encode(ops, golden, shape):
    g = golden("g3_sample2enc")
    n, P = g["s_vals"].shape
    S = P - 1
    args = [dev(g[k]).contiguous() for k in ("s_vals", "origins", "directions")] + [dev(g[k]).reshape(-1).contiguous() for k in ("radii", "near", "far")]
    out = torch.full((n * S, 128), 5.0, dtype=torch.float32, device="cuda")
    mo = torch.empty(n * S, 3, device="cuda"); co = torch.empty(n * S, 3, device="cuda")
    ops.mip_encode(*args, shape == "cone", 0, 16, out[:, 16:], None, 100, ops.F32, means_out=mo, covs_out=co)
    fm, fc = om.sample2enc(g["s_vals"], g["origins"], g["directions"], g["radii"], g["near"], g["far"], shape, 0)
    close(mo.reshape(n, S, 3), fm, 2e-6, 2e-6, "contracted means")
    close(co.reshape(n, S, 3), fc, 5e-5, 1e-9, "warped cov diag")
    if shape == "cone":  # also pinned directly against the reference's own output
        close(mo.reshape(n, S, 3), g["f_means"], 2e-6, 2e-6, "means vs reference")
        close(co.reshape(n, S, 3), torch.diagonal(g["f_covs"], dim1=-2, dim2=-1), 5e-5, 1e-9, "cov vs reference")
    enc = om.integrated_pos_enc(fm, fc, 0, 16).reshape(n * S, 96)
    # sin of arguments up to 2^15 |x|: an input ulp (1e-7 relative) moves the phase by up to 2^15 * 2 * 1e-7 ~ 7e-3 rad at the
    # highest degree; those features are damped by exp(-var/2) but not always to zero -> compare per degree
    got = out[:, 16:16 + 96].cpu()
    for deg in range(16):
        cols = [deg * 3 + d for d in range(3)] + [48 + deg * 3 + d for d in range(3)]
        close(got[:, cols], enc[:, cols], 0, 3e-6 * 2 ** deg + 1e-6, f"IPE degree {deg}")
    assert bool((out[:, 16 + 96:16 + 100] == 0).all()) and bool((out[:, :16] == 5).all()) and bool((out[:, 116:] == 5).all())


@pytest.mark.parametrize("shape", ["cone", "cylinder"])
def test_mip_encode_view_centred_warp(ops, golden, shape):
    """snerf_mip_encode_warp, fn_idx = 0 (mip.py:367-378: fn1 + Jacobi_f): warped means / covariance diagonal against the reference's own
    output (g23) and the IPE features against the oracle."""
    g = golden("g23_warp0")
    n, P = g["s_vals"].shape
    S = P - 1
    args = [dev(g[k]).contiguous() for k in ("s_vals", "rays_origins", "rays_directions")] + [dev(g[k]).reshape(-1).contiguous() for k in ("rays_radii", "rays_near", "rays_far")]
    out = torch.full((n * S, 100), 5.0, dtype=torch.float32, device="cuda")
    mo = torch.empty(n * S, 3, device="cuda"); co = torch.empty(n * S, 3, device="cuda")
    warp = (tuple(float(v) for v in g["viewc"]), args[5].max().reshape(1))
    ops.mip_encode(*args, shape == "cone", 0, 16, out, None, 100, ops.F32, means_out=mo, covs_out=co, warp=warp)
    want_c = torch.diagonal(g[shape + "_f_covs"], dim1=-2, dim2=-1)
    close(mo.reshape(n, S, 3), g[shape + "_f_means"], 2e-6, 2e-6, "means vs reference")
    close(co.reshape(n, S, 3), want_c, 5e-5, float(want_c.abs().max()) * 1e-6, "cov diagonal vs reference")
    enc = om.integrated_pos_enc(g[shape + "_f_means"], want_c, 0, 16).reshape(n * S, 96)
    got = out[:, :96].cpu()
    for deg in range(16):
        cols = [deg * 3 + d for d in range(3)] + [48 + deg * 3 + d for d in range(3)]
        close(got[:, cols], enc[:, cols], 0, 3e-6 * 2 ** deg + 1e-6, f"IPE degree {deg}")
    assert bool((out[:, 96:] == 0).all())


def test_mip_encode_row_block_flavours_agree(ops):
    """mip_encode_kernel runs 64 rows per workgroup up to 131 072 rows and 256 above (a training batch of 512 rays vs a frame chunk): the same
    rays in one large launch and in two small ones must give the same bits (bf16 operand and the fp32 image)."""
    n, S = 1100, 128                                           # 140 800 rows in one launch: the 256-row flavour
    g = torch.Generator().manual_seed(31)
    s_vals = torch.sort(torch.rand(n, S + 1, generator=g), -1)[0].cuda()
    o, d = (torch.randn(n, 3, generator=g) * 2).cuda(), torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).cuda()
    rad = (torch.rand(n, generator=g) * 1e-3 + 1e-4).cuda()
    near, far = (torch.rand(n, generator=g) + 0.5).cuda(), (torch.rand(n, generator=g) * 50 + 5).cuda()
    for dt, tdt in ((ops.BF16, torch.bfloat16), (ops.F32, torch.float32)):
        big = torch.full((n * S, 128), 7.0, dtype=tdt, device="cuda")
        ops.mip_encode(s_vals, o, d, rad, near, far, True, 0, 16, big, None, 128, dt)
        parts = []
        for a, b in ((0, 600), (600, n)):                      # 76 800 and 64 000 rows: the 64-row flavour
            out = torch.full(((b - a) * S, 128), 7.0, dtype=tdt, device="cuda")
            ops.mip_encode(s_vals[a:b].contiguous(), o[a:b].contiguous(), d[a:b].contiguous(), rad[a:b].contiguous(), near[a:b].contiguous(),
                           far[a:b].contiguous(), True, 0, 16, out, None, 128, dt)
            parts.append(out)
        assert torch.equal(big, torch.cat(parts, 0)) and float(big[:, 96:].float().abs().max()) == 0.0


def test_mip_encode_ipe_exact_inputs(ops, golden):
    """IPE stage alone: feed means/covs that survive the sampler unchanged (|x| < 3 region is x/3 ... not exact), so instead
    check the bf16 path against the fp32 path of the same kernel."""
    g = golden("g2_cast")
    n, P = g["s_vals"].shape
    S = P - 1
    args = [dev(g[k]).contiguous() for k in ("s_vals", "origins", "directions")] + [dev(g[k]).reshape(-1).contiguous() for k in ("radii", "near", "far")]
    a = torch.empty(n * S, 96, dtype=torch.float32, device="cuda"); b = torch.empty(n * S, 128, dtype=torch.bfloat16, device="cuda")
    ops.mip_encode(*args, True, 0, 16, a, None, 96, ops.F32)
    ops.mip_encode(*args, True, 0, 16, b, None, 128, ops.BF16)
    close(b[:, :96].float(), a, 8e-3, 1e-6, "bf16 IPE vs fp32 IPE")
    assert bool((b[:, 96:] == 0).all())


def test_mip_viewenc(ops, golden):
    x = golden("g1_posenc")["x"]
    S = 3
    out = torch.empty(x.shape[0] * S, 32, dtype=torch.float32, device="cuda")
    ops.mip_viewenc(x.cuda().contiguous(), S, 4, out, 32, ops.F32)
    ref = om.pos_enc(x, 0, 4, True)[:, None].expand(-1, S, -1).reshape(-1, 27)
    close(out[:, :27], ref, 0, 2e-6, "view pos_enc")
    assert bool((out[:, 27:] == 0).all())


# -------------------------------------------------------------- samplers ----
def test_stratified(ops):
    N, P = 9, 17
    near = torch.full((N,), 2.0) + gen(N, seed=11) * 0.1
    far = torch.full((N,), 6.0) + gen(N, seed=12)
    rb = torch.zeros(N, 11); rb[:, 6] = near; rb[:, 7] = far
    rb = rb.cuda()
    base = torch.linspace(0., 1., P)
    rnd = torch.rand(N, P, generator=torch.Generator().manual_seed(13))
    for lindisp in (False, True):
        for r in (None, rnd):
            z = ops.stratified(base.cuda(), dev(r), rb[:, 6], rb[:, 7], N, 0, lindisp)
            ref = oc.stratified_z(near[:, None], far[:, None], P, lindisp, r)
            assert torch.equal(z.cpu(), ref), f"classic stratified lindisp={lindisp} rand={r is not None}"
    s = ops.stratified(torch.linspace(0., 1., P).cuda(), rnd.cuda(), None, None, N, 1)
    assert torch.equal(s.cpu(), om.warp_sample_s(N, P - 1, rnd))


def test_classic_sample_pdf(ops, golden):
    g = golden("g9_sample_pdf")
    bins, w = g["bins"], g["weights"]
    n = bins.shape[0]
    for u in (torch.linspace(0., 1., 24), g["u_rand"]):
        ref_s, ref_i = oc.sample_pdf(bins, w, u if u.dim() == 2 else u.expand(n, 24))
        s, inds, _ = ops.classic_sample_pdf(bins.cuda(), w.cuda(), u.cuda().contiguous(), False, want_inds=True)
        assert torch.equal(inds.cpu().long(), ref_i), "sample_pdf indices must be bit-exact"
        assert torch.equal(s.cpu(), ref_s), "sample_pdf samples must be bit-exact (same op sequence)"
    # fused z_vals form + z_std, larger random case incl. zero weights
    N, S, Nf = 200, 64, 128
    gg = torch.Generator().manual_seed(14)
    z = torch.sort(torch.rand(N, S, generator=gg) * 4 + 2, -1)[0]
    ww = torch.rand(N, S, generator=gg) ** 4
    ww[:5] = 0
    u = torch.rand(N, Nf, generator=gg)
    ref_s, ref_i = oc.sample_pdf(0.5 * (z[:, 1:] + z[:, :-1]), ww[:, 1:-1], u)
    s, inds, std = ops.classic_sample_pdf(z.cuda(), ww.cuda(), u.cuda(), True, want_inds=True, want_std=True)
    assert torch.equal(inds.cpu().long(), ref_i) and torch.equal(s.cpu(), ref_s)
    close(std, torch.std(ref_s, dim=-1, unbiased=False), 1e-5, 1e-6, "z_std")


def test_merge_sort(ops):
    gg = torch.Generator().manual_seed(15)
    a = torch.sort(torch.rand(37, 64, generator=gg), -1)[0]
    b = torch.rand(37, 128, generator=gg)
    b[0, :10] = a[0, :10]  # ties
    out = ops.classic_merge_sort(a.cuda(), b.cuda())
    assert torch.equal(out.cpu(), torch.sort(torch.cat([a, b], -1), -1)[0])


@pytest.mark.parametrize("n,P", [(1, 128), (33, 128), (4096, 128), (517, 65)])
def test_jitter_u_is_the_eager_expression(ops, n, P):
    """snerf_jitter_u = math_ops.py:50-54 (arange(P) * s + uniform_(to = s - eps), clamped below 1), bit for bit, in one launch"""
    eps = float(torch.finfo(torch.float32).eps)
    s = 1 / P
    torch.manual_seed(5)
    jit = torch.empty(n, P, device="cuda").uniform_(to=s - eps)
    jit[0, -1] = s                                     # a draw that must be clamped
    ref = torch.minimum(torch.arange(P, device="cuda") * s + jit, torch.ones_like(jit) - eps)
    got = ops.jitter_u(jit.clone(), s)
    assert torch.equal(got, ref) and float(got.max()) < 1.0


def test_mip_resample(ops, golden):
    for num in (32, 33):
        g = golden(f"g5_pdf_{num}")
        # A14 alone is reached through padding=0 on un-blurrable input; test the fused A13+A14 against the oracle
        for u in (g["u_det"], om.rand_u(num, g["jitter"])):
            ref_s, ref_i = om.warp_resample_s(g["bins"], g["weights"], u, 0.01)
            s, idx = ops.mip_resample(g["bins"].cuda(), g["weights"].cuda(), u.cuda().contiguous(), 0.01, want_idx=True)
            assert torch.equal(idx.cpu(), ref_i), "resample interval indices must be bit-exact"
            assert torch.equal(s.cpu(), ref_s), "resample fence posts must be bit-exact"
    N, S, Nf = 300, 128, 128
    gg = torch.Generator().manual_seed(16)
    sv = torch.sort(torch.rand(N, S + 1, generator=gg), -1)[0]
    w = torch.rand(N, S, generator=gg) ** 6
    w[:3] = 0; w[3] = 0; w[3, 50] = 1.0
    u = om.rand_u(Nf, torch.empty(N, Nf).uniform_(0, 1 / Nf - om.EPS32, generator=gg))
    ref_s, ref_i = om.warp_resample_s(sv, w, u, 0.01)
    s, idx = ops.mip_resample(sv.cuda(), w.cuda(), u.cuda(), 0.01, want_idx=True)
    assert torch.equal(idx.cpu(), ref_i) and torch.equal(s.cpu(), ref_s)
    # batches above 16 384 rays run 64 rays per workgroup instead of 16: the same rows, the same bits (a frame chunk vs a training batch)
    rep = 57                                                      # 300 x 57 = 17 100 rays
    s2, idx2 = ops.mip_resample(sv.cuda().repeat(rep, 1), w.cuda().repeat(rep, 1), u.cuda().repeat(rep, 1), 0.01, want_idx=True)
    assert torch.equal(s2.view(rep, N, Nf), s.expand(rep, -1, -1)) and torch.equal(idx2.view(rep, N, Nf), idx.expand(rep, -1, -1))


# ------------------------------------------------------------ compositing ----
def test_mip_composite_fwd_bwd(ops, golden):
    for name, white in (("g6_volrend_white0", False), ("g6_volrend_white1", True)):
        g = golden(name)
        n, S = g["density"].shape[:2]
        # invert the activations so the kernel's fused sigmoid/softplus reproduce the golden rgb/density
        rgb, den = g["rgb"].double(), g["density"].double()
        raw_rgb = torch.logit(((rgb + 0.001) / 1.002).clamp(1e-6, 1 - 1e-6)).float()
        den_c = den.clamp(1e-6, 50.0)
        raw_den = (torch.log(torch.expm1(den_c)) + 1.0).float()
        rr = raw_rgb.reshape(-1, 3).cuda().contiguous(); rd = raw_den.reshape(-1, 1).cuda().contiguous()
        sv, d = g["s_vals"].cuda(), g["dirs"].cuda()
        near, far = g["near"].reshape(-1).cuda(), g["far"].reshape(-1).cuda()
        comp, dist, acc, w = ops.mip_composite_fwd(rr, rd, None, sv, d, near, far, 0, white, 0.001, -1.0)
        # oracle on the same raw values (rows 0/1 of the golden use density 0 / 1e4 which the inverse clamps)
        raw_rgb_t = raw_rgb.clone().requires_grad_(True); raw_den_t = raw_den.clone().requires_grad_(True)
        rgb_o, den_o = om.activate(raw_rgb_t, raw_den_t)
        c_o, d_o, a_o, w_o, _ = om.volumetric_rendering(rgb_o, den_o, g["s_vals"], g["dirs"], g["near"], g["far"], white)
        close(comp, c_o, 1e-5, 1e-6, "comp_rgb"); close(dist, d_o, 1e-5, 1e-5, "distance"); close(acc, a_o, 1e-5, 1e-6, "acc")
        close(w, w_o, 1e-5, 1e-7, "weights")
        sel = slice(2, None)  # rows with un-clamped densities also match the reference's own numbers
        close(comp[sel], g["comp_rgb"][sel], 2e-5, 2e-6, "comp_rgb vs reference"); close(w[sel], g["weights"][sel], 2e-5, 1e-6, "w vs reference")
        # backward against autograd of the oracle
        gg = torch.Generator().manual_seed(17)
        g_rgb, g_dist, g_acc, g_w = torch.randn(n, 3, generator=gg), torch.randn(n, generator=gg), torch.randn(n, generator=gg), torch.randn(n, S, generator=gg)
        loss = (c_o * g_rgb).sum() + (d_o * g_dist).sum() + (a_o * g_acc).sum() + (w_o * g_w).sum()
        loss.backward()
        d_rgb = torch.empty(n * S, 3, device="cuda"); d_den = torch.empty(n * S, 1, device="cuda")
        ops.mip_composite_bwd(rr, rd, None, sv, d, near, far, 0, white, 0.001, -1.0, w, dist, g_rgb.cuda(), g_dist.cuda(), g_acc.cuda(),
                              g_w.cuda(), d_rgb, d_den)
        close(d_rgb.reshape(n, S, 3), raw_rgb_t.grad, 1e-4, 1e-6, "d raw_rgb")
        close(d_den.reshape(n, S, 1), raw_den_t.grad, 2e-4, 2e-5, "d raw_density")
    g = golden("g6_volrend_norgb")
    den_c = g["density"].double().clamp(1e-6, 50.0)
    rd = (torch.log(torch.expm1(den_c)) + 1.0).float().reshape(-1, 1).cuda()
    comp, dist, acc, w = ops.mip_composite_fwd(None, rd, None, g["s_vals"].cuda(), g["dirs"].cuda(), g["near"].reshape(-1).cuda(),
                                               g["far"].reshape(-1).cuda(), 0, False, 0.001, -1.0)
    assert comp is None
    close(w[2:], g["weights"][2:], 2e-5, 1e-6, "proposal weights vs reference"); close(dist[2:], g["distance"][2:], 2e-5, 1e-5, "distance")
    # sigma == 0 row: distance is clipped up to t_0 (nan/zero path of mip.py:185-186)
    rd0 = torch.full((g["density"].shape[0] * g["density"].shape[1], 1), -80.0, device="cuda")
    _, dist0, acc0, _ = ops.mip_composite_fwd(None, rd0, None, g["s_vals"].cuda(), g["dirs"].cuda(), g["near"].reshape(-1).cuda(),
                                              g["far"].reshape(-1).cuda(), 0, False, 0.001, -1.0)
    t0 = om.transform(g["s_vals"][:, :1], g["near"], g["far"], 0)[:, 0]
    close(dist0, t0, 1e-6, 1e-6, "zero-density distance == t_0"); assert float(acc0.abs().max()) < 1e-30


def test_classic_composite_fwd_bwd(ops, golden):
    for name, white in (("g9_raw2outputs_white0", False), ("g9_raw2outputs_white1", True)):
        g = golden(name)
        raw, z, rd = g["raw"], g["z_vals"], g["rays_d"]
        n, S = z.shape
        out = ops.classic_composite_fwd(raw.reshape(-1, 4).cuda().contiguous(), None, z.cuda(), rd.cuda(), white)
        for got, key in zip(out, ("rgb_map", "disp_map", "acc_map", "weights", "depth_map")):
            close(got, g[key], 2e-5, 2e-6, key + " vs reference")
        raw_t = raw.clone().requires_grad_(True)
        o = oc.raw2outputs(raw_t, z, rd, None, white)
        gg = torch.Generator().manual_seed(18)
        gs = [torch.randn(n, 3, generator=gg), torch.randn(n, generator=gg) * 0.1, torch.randn(n, generator=gg),
              torch.randn(n, S, generator=gg), torch.randn(n, generator=gg)]
        sum((a * b).sum() for a, b in zip(o, gs)).backward()
        d_raw = torch.zeros(n * S, 4, device="cuda")
        ops.classic_composite_bwd(raw.reshape(-1, 4).cuda().contiguous(), None, z.cuda(), rd.cuda(), white, out[3], out[2], out[4],
                                  gs[0].cuda(), gs[1].cuda(), gs[2].cuda(), gs[4].cuda(), gs[3].cuda(), d_raw)
        close(d_raw.reshape(n, S, 4), raw_t.grad, 2e-4, 2e-5, "d raw")
    # many-segment case (S = 192 > 2 wave segments) with noise
    gg = torch.Generator().manual_seed(19)
    n, S = 33, 192
    raw = torch.randn(n, S, 4, generator=gg); z = torch.sort(torch.rand(n, S, generator=gg) * 4 + 2, -1)[0]; rd = torch.randn(n, 3, generator=gg)
    noise = torch.randn(n, S, generator=gg)
    out = ops.classic_composite_fwd(raw.reshape(-1, 4).cuda().contiguous(), noise.cuda(), z.cuda(), rd.cuda(), False)
    ref = oc.raw2outputs(raw, z, rd, noise, False)
    for got, want, key in zip(out, ref, ("rgb_map", "disp_map", "acc_map", "weights", "depth_map")):
        close(got, want, 2e-5, 2e-6, key + " S=192")


# ------------------------------------------------------------ training tail ----
def test_adam_and_small_ops(ops):
    n = 10007
    p = gen(n, seed=20); gr = gen(n, seed=21)
    pt = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([pt], lr=5e-4, betas=(0.9, 0.999), eps=1e-8)
    pc, gc, m, v = p.cuda().clone(), gr.cuda().clone(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    for step in range(1, 4):
        pt.grad = gr.clone() * step
        opt.step()
        gc.copy_((gr * step).cuda() * 2.0)            # grad_scale 0.5 undoes the 2x
        ops.adam_step(pc, gc, m, v, 5e-4, 0.9, 0.999, 1e-8, step, grad_scale=0.5, zero_grad=True)
        assert float(gc.abs().max()) == 0.0
    close(pc, pt.detach(), 1e-5, 1e-6, "adam params")
    x = gen(1234, 4, seed=22).cuda()
    out = torch.ones(3, device="cuda")
    ops.colsum_f32(x, 3, out)
    close(out, 1 + x[:, :3].double().sum(0).cpu(), 1e-5, 1e-4, "colsum")
    dst = torch.full((1234, 64), 3.0, dtype=torch.bfloat16, device="cuda")
    ops.cast_pad(x[:, 3:], 1, dst[:, 32:], 32, ops.BF16)
    close(dst[:, 32].float(), x[:, 3].bfloat16().float(), 0, 0, "cast"); assert bool((dst[:, 33:] == 0).all()) and bool((dst[:, :32] == 3).all())


@pytest.mark.gpu
def test_adam_on_misaligned_slices_matches_aligned():
    """a rank's 1 / world slice of a single-channel hash table starts at an odd float (ZipTrainer's sharded table update at world
    sizes 4 and 8: 6 606 952 / 8 = 825 869): the launch must take p / g / m / v at any 4-byte alignment, each with its own, and give
    exactly the update of the aligned launch (trainer._TableShards)"""
    from snerf_amd import ops
    n = 825869
    base = [gen(n + 8, seed=30 + k).cuda() for k in range(4)]
    base[3].abs_()
    ref = [b[:n].clone() for b in base]
    ops.adam_step(*ref, 1e-2, 0.9, 0.99, 1e-15, 3, grad_scale=0.25, zero_grad=True)
    for offs in ((1, 0, 1, 1), (2, 0, 2, 2), (3, 1, 2, 0), (1, 1, 1, 1)):
        bufs = [torch.zeros(n + 8, device="cuda") for _ in range(4)]
        views = [bufs[k][o:o + n] for k, o in enumerate(offs)]
        for vw, b in zip(views, base):
            vw.copy_(b[:n])
        ops.adam_step(*views, 1e-2, 0.9, 0.99, 1e-15, 3, grad_scale=0.25, zero_grad=True)
        for k in (0, 2, 3):
            assert torch.equal(views[k], ref[k]), (offs, k)
            assert float(bufs[k][:offs[k]].abs().max() if offs[k] else 0.0) == 0.0 and float(bufs[k][offs[k] + n:].abs().max()) == 0.0
        assert float(views[1].abs().max()) == 0.0
    # the dropped-gradient counter (snerf_adam_step_cnt): NaN / +-Inf elements are counted, whatever the policy does with them
    p, g, m, v = (b[:4099].clone() for b in base)
    v.abs_()
    g[5], g[4097], g[4098] = float("inf"), float("nan"), float("-inf")
    p0, cnt = p.clone(), torch.zeros(1, dtype=torch.int64, device="cuda")
    for _ in range(2):
        g[5], g[4097], g[4098] = float("inf"), float("nan"), float("-inf")
        ops.adam_step(p, g, m, v, 1e-2, 0.9, 0.99, 1e-15, 1, nonfinite="zero", dropped=cnt)
    assert int(cnt) == 6 and bool(torch.isfinite(p).all()) and float(p[5]) == float(p0[5]) if float(m[5]) == 0 else True


# ---------------------------------------------------------------- ray gradients (pose refinement) ----
def _pose_rays(n, seed, far_scene=True):
    """rays whose samples straddle the contraction radius (|x| around 3) so that both branches of fn2 / Jacobi_g are exercised"""
    r = common.synthetic_rays(n, seed=seed)
    if far_scene:
        r["near"] = torch.full_like(r["near"], 0.5)
        r["far"] = torch.full_like(r["far"], 30.0)
    return r


@pytest.mark.gpu
@pytest.mark.parametrize("n,S,cone,deg", [(130, 64, True, 16), (37, 127, True, 16), (64, 5, False, 10)])
def test_mip_encode_bwd_vs_oracle_autograd(n, S, cone, deg):
    """d loss / d (origins, directions) through integrated_pos_enc, the contraction with its Jacobian and lift_gaussian: the wave-per-ray
    kernel against torch autograd through oracle/mip.py (which reproduces the reference's forward, G3/G4)."""
    from snerf_amd import ops
    r = _pose_rays(n, 3 + S)
    g = torch.Generator().manual_seed(S)
    s = torch.sort(torch.rand(n, S + 1, generator=g), -1).values
    s[:, 0], s[:, -1] = 0.0, 1.0
    dE = torch.randn(n * S, 6 * deg + 4, generator=g)
    def autograd(dt):
        o, d = r["origins"].to(dt).clone().requires_grad_(True), r["directions"].to(dt).clone().requires_grad_(True)
        fm, fc = om.sample2enc(s.to(dt), o, d, r["radii"].to(dt), r["near"].to(dt), r["far"].to(dt), "cone" if cone else "cylinder", 0)
        enc = om.integrated_pos_enc(fm, fc, 0, deg).reshape(-1, 6 * deg)
        (enc * dE[:, :6 * deg].to(dt)).sum().backward()
        return fm.detach(), o.grad, d.grad
    fm, o32, d32 = autograd(torch.float32)
    _, o64, d64 = autograd(torch.float64)
    nrm = fm.norm(dim=-1)
    assert bool((nrm > 1.0).any()) and bool((nrm < 0.99).any())                  # both sides of the contraction radius
    c = lambda t: t.detach().cuda().contiguous()
    go, gd = ops.mip_encode_bwd(c(s), c(r["origins"]), c(r["directions"]), c(r["radii"]).reshape(-1), c(r["near"]).reshape(-1), c(r["far"]).reshape(-1),
                                cone, 0, deg, c(dE))
    # the sum of 6*deg terms scaled by 2^deg is ill-conditioned in fp32: torch's own fp32 autograd sits 1e-3 away from the float64
    # evaluation.  The kernel must be as close to the float64 gradient as that (per ray), and exact to 1e-5 where fp32 autograd is.
    for got, ref32, ref64, what in ((go, o32, o64, "origins"), (gd, d32, d64, "directions")):
        scale = ref64.abs().max(dim=-1).values
        err = (got.cpu().double() - ref64).abs().max(dim=-1).values / scale
        noise = (ref32.double() - ref64).abs().max(dim=-1).values / scale
        # measured: the kernel's error distribution IS torch's fp32 noise (medians 5.8e-5 / 5.8e-5, maxima 2.4e-3 / 2.4e-3 at deg 16)
        assert float(err.max()) <= 3 * float(noise.max()) + 2e-5 and float(err.median()) <= 3 * float(noise.median()) + 2e-5, what


@pytest.mark.gpu
@pytest.mark.parametrize("n,S,cone,deg", [(130, 64, True, 16), (64, 5, False, 10)])
def test_mip_encode_without_integration_fwd_and_bwd(n, S, cone, deg):
    """--disable_integration (models.py:132-133; bit 1 of the kernels' `cone` argument): the features are safe_sin of the scaled means with
    NO exp(-var / 2) factor -- against integrated_pos_enc(means, zeros) evaluated on the KERNEL's own means (the un-damped 2^15 x phase
    turns one ulp of the mean into 2e-3 rad: a comparison through independently computed means would only test fp32 noise); the covariance
    output is zero; the ray gradient against float64 / float32 autograd through the oracle with the covariance replaced by a constant."""
    from snerf_amd import ops
    r = _pose_rays(n, 11 + S)
    g = torch.Generator().manual_seed(S + 7)
    s = torch.sort(torch.rand(n, S + 1, generator=g), -1).values
    s[:, 0], s[:, -1] = 0.0, 1.0
    c = lambda t: t.detach().cuda().contiguous()
    args = [c(s), c(r["origins"]), c(r["directions"]), c(r["radii"]).reshape(-1), c(r["near"]).reshape(-1), c(r["far"]).reshape(-1)]
    flag = (1 if cone else 0) | 2
    out = torch.empty(n * S, 6 * deg + 4, device="cuda"); mo = torch.empty(n * S, 3, device="cuda"); co = torch.full((n * S, 3), 7.0, device="cuda")
    ops.mip_encode(*args, flag, 0, deg, out, None, 6 * deg + 4, ops.F32, means_out=mo, covs_out=co)
    assert float(co.abs().max()) == 0.0
    want = om.integrated_pos_enc(mo.cpu().reshape(n, S, 3), torch.zeros(n, S, 3), 0, deg).reshape(n * S, 6 * deg)
    assert float((out[:, :6 * deg].cpu() - want).abs().max()) <= 2e-6 and bool((out[:, 6 * deg:] == 0).all())
    plain = torch.empty_like(out)
    ops.mip_encode(*args, flag & 1, 0, deg, plain, None, 6 * deg + 4, ops.F32)
    assert float((plain - out).abs().max()) > 0.1                                      # (the damped features differ)
    dE = torch.randn(n * S, 6 * deg + 4, generator=g)

    def autograd(dt):
        o, d = r["origins"].to(dt).clone().requires_grad_(True), r["directions"].to(dt).clone().requires_grad_(True)
        fm, fc = om.sample2enc(s.to(dt), o, d, r["radii"].to(dt), r["near"].to(dt), r["far"].to(dt), "cone" if cone else "cylinder", 0)
        enc = om.integrated_pos_enc(fm, torch.zeros_like(fc), 0, deg).reshape(-1, 6 * deg)
        (enc * dE[:, :6 * deg].to(dt)).sum().backward()
        return o.grad, d.grad
    (o32, d32), (o64, d64) = autograd(torch.float32), autograd(torch.float64)
    go, gd = ops.mip_encode_bwd(*args, flag, 0, deg, c(dE))
    for got, ref32, ref64, what in ((go, o32, o64, "origins"), (gd, d32, d64, "directions")):
        scale = ref64.abs().max(dim=-1).values
        err = (got.cpu().double() - ref64).abs().max(dim=-1).values / scale
        noise = (ref32.double() - ref64).abs().max(dim=-1).values / scale
        assert float(err.max()) <= 3 * float(noise.max()) + 2e-5 and float(err.median()) <= 3 * float(noise.median()) + 2e-5, (what, float(err.max()), float(noise.max()))


@pytest.mark.gpu
@pytest.mark.parametrize("n,S,cone,deg", [(130, 64, True, 16), (37, 127, False, 10)])
def test_mip_encode_warp_bwd_vs_oracle_autograd(n, S, cone, deg):
    """the same gradient through the view-centred warp of an fn = 0 model (mip.py:367-369 fn1, :323-340 Jacobi_f): snerf_mip_encode_warp_bwd
    against float64 / float32 autograd through oracle/mip.py's sample2enc(fn_idx=0)"""
    from snerf_amd import ops
    r = _pose_rays(n, 5 + S)
    g = torch.Generator().manual_seed(S + 1)
    s = torch.sort(torch.rand(n, S + 1, generator=g), -1).values
    s[:, 0], s[:, -1] = 0.0, 1.0
    dE = torch.randn(n * S, 6 * deg + 4, generator=g)
    viewc = (0.3, -0.2, 0.5)
    def autograd(dt):
        o, d = r["origins"].to(dt).clone().requires_grad_(True), r["directions"].to(dt).clone().requires_grad_(True)
        fm, fc = om.sample2enc(s.to(dt), o, d, r["radii"].to(dt), r["near"].to(dt), r["far"].to(dt), "cone" if cone else "cylinder", 0, fn_idx=0,
                               viewc=torch.tensor(viewc, dtype=dt))
        enc = om.integrated_pos_enc(fm, fc, 0, deg).reshape(-1, 6 * deg)
        (enc * dE[:, :6 * deg].to(dt)).sum().backward()
        return o.grad, d.grad
    o32, d32 = autograd(torch.float32)
    o64, d64 = autograd(torch.float64)
    c = lambda t: t.detach().cuda().contiguous()
    far = c(r["far"]).reshape(-1)
    go, gd = ops.mip_encode_bwd(c(s), c(r["origins"]), c(r["directions"]), c(r["radii"]).reshape(-1), c(r["near"]).reshape(-1), far, cone, 0, deg, c(dE),
                                warp=(viewc, far.max().reshape(1)))
    for got, ref32, ref64, what in ((go, o32, o64, "origins"), (gd, d32, d64, "directions")):
        scale = ref64.abs().max(dim=-1).values
        err = (got.cpu().double() - ref64).abs().max(dim=-1).values / scale
        noise = (ref32.double() - ref64).abs().max(dim=-1).values / scale
        assert float(err.max()) <= 3 * float(noise.max()) + 2e-5 and float(err.median()) <= 3 * float(noise.median()) + 2e-5, (what, float(err.max()), float(noise.max()))


@pytest.mark.gpu
def test_mip_viewenc_bwd_and_composite_direction_gradient_vs_oracle_autograd():
    from snerf_amd import ops
    n, S, deg = 70, 33, 4
    r = _pose_rays(n, 9, far_scene=False)
    g = torch.Generator().manual_seed(1)
    dV = torch.randn(n * S, 3 + 6 * deg + 5, generator=g)
    v = r["viewdirs"].clone().requires_grad_(True)
    e = om.pos_enc(v, 0, deg, True)[:, None].expand(-1, S, -1).reshape(-1, 3 + 6 * deg)
    (e * dV[:, :3 + 6 * deg]).sum().backward()
    got = ops.mip_viewenc_bwd(r["viewdirs"].cuda().contiguous(), S, deg, dV.cuda())
    assert torch.allclose(got.cpu(), v.grad, rtol=1e-4, atol=1e-4 * float(v.grad.abs().max()))
    # compositing: d loss / d directions through delta = (t1 - t0) |d|
    s = torch.sort(torch.rand(n, S + 1, generator=g), -1).values
    raw_rgb, raw_d = torch.randn(n * S, 3, generator=g), torch.randn(n * S, 1, generator=g) * 2
    d = r["directions"].clone().requires_grad_(True)
    rgb, den = om.activate(raw_rgb.reshape(n, S, 3), raw_d.reshape(n, S, 1), 0.001, -1.0)
    c_, dist, acc, w, _ = om.volumetric_rendering(rgb, den, s, d, r["near"], r["far"], False, None, 0)
    g_rgb, g_dist, g_acc, g_w = torch.randn(n, 3, generator=g), torch.randn(n, generator=g) * 0.1, torch.randn(n, generator=g), torch.randn(n, S, generator=g)
    ((c_ * g_rgb).sum() + (dist * g_dist).sum() + (acc * g_acc).sum() + (w * g_w).sum()).backward()
    cu = lambda t: t.detach().cuda().contiguous()
    near, far = cu(r["near"]).reshape(-1), cu(r["far"]).reshape(-1)
    _, dist_k, _, w_k = ops.mip_composite_fwd(cu(raw_rgb), cu(raw_d), None, cu(s), cu(r["directions"]), near, far, 0, False, 0.001, -1.0)
    d_rgb, d_den = torch.empty(n * S, 3, device="cuda"), torch.empty(n * S, 1, device="cuda")
    g_dirs = torch.empty(n, 3, device="cuda")
    ops.mip_composite_bwd(cu(raw_rgb), cu(raw_d), None, cu(s), cu(r["directions"]), near, far, 0, False, 0.001, -1.0, w_k, dist_k, cu(g_rgb), cu(g_dist),
                          cu(g_acc), cu(g_w), d_rgb, d_den, g_dirs=g_dirs)
    err = (g_dirs.cpu() - d.grad).abs().max()
    assert float(err) <= 2e-4 * float(d.grad.abs().max()), (float(err), float(d.grad.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(70001, 768, 256), (131072, 1024, 128), (300, 1024, 128), (66000, 256, 192)])
def test_persistent_kernel_colsum_across_tiles(M, N, K):
    """The persistent kernel keeps the column sums in registers across a workgroup's tiles and flushes once per workgroup (or when the
    column block changes: N = 768 gives 3 column blocks against a tile stride of 256 workgroups, so consecutive tiles of a workgroup do
    change block).  Plain and masked data gradients, fewer tiles than compute units (M = 300), ragged last row tile; twice in a row into
    the same accumulator (the workspace is re-zeroed per launch)."""
    from snerf_amd import ops as O
    dZ = (gen(M, K, seed=1) * 0.5).bfloat16().cuda()
    Wt = (gen(N, K, seed=2) / K ** 0.5).bfloat16().cuda()
    act = gen(M, N, seed=3).bfloat16().cuda()
    ref_full = dZ.double().cpu() @ Wt.double().cpu().t()
    for mode, aux in ((O.ACT_NONE, None), (O.ACT_MASK, act)):
        dX = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
        cs = torch.zeros(N, dtype=torch.float32, device="cuda")
        for _ in range(2):
            O.linear_fwd(dZ, Wt, None, dX, K, N, mode, O.BF16, aux=aux, colsum=cs, variant=8)
        ref = ref_full * (act.double().cpu() > 0) if aux is not None else ref_full
        # the kernel sums the bf16-rounded outputs: compare against the column sums of what it stored, and of the exact product
        stored = dX.double().cpu().sum(0)
        close(cs / 2, stored, 2e-4, 2e-3 * max(1.0, (M / 515) ** 0.5), f"colsum of the stored values, act {mode}")
        close(cs / 2, ref.sum(0), 2e-2, 5e-2 * max(1.0, (M / 515) ** 0.5), f"colsum, act {mode}")


@pytest.mark.gpu
def test_linear_kernels_random_shapes_sweep():
    """A seeded sweep over launch shapes of the two GEMM entry points (every dispatch branch: 128x128, 256x256 block-issue, 8-phase,
    persistent; ragged M, every K granule, column-range operands, fp32 heads, the weight-gradient kernels with padded / partial
    outputs) against float64 matmul.  40 + 25 launches, each with its own tolerance from the reduction length."""
    from snerf_amd import ops as O
    rng = np.random.default_rng(2024)
    for it in range(40):
        N = int(rng.choice([128, 256, 384, 512, 1024]))
        K = int(rng.choice([64, 128, 192, 256, 320, 1024, 1088, 1152]))
        M = int(rng.choice([1, 63, 300, 4097, 33000, int(rng.integers(1, 70000))]))
        variant = int(rng.choice([0, 1, 4, 8]))
        act = int(rng.choice([O.ACT_NONE, O.ACT_RELU]))
        out_f32 = bool(rng.integers(0, 2)) and N == 128
        pad = int(rng.choice([0, 64]))                                  # operands as column ranges of wider buffers
        A = gen(M, K + pad, seed=it).bfloat16().cuda()[:, pad:]
        W = (gen(N, K, seed=100 + it) / K ** 0.5).bfloat16().cuda()
        b = gen(N, seed=200 + it).cuda()
        n_store = N if not out_f32 else int(rng.integers(1, 8))
        Y = torch.zeros(M, N, dtype=torch.float32 if out_f32 else torch.bfloat16, device="cuda")
        O.linear_fwd(A, W, b, Y, K, n_store, act, O.BF16, out_f32=out_f32, variant=variant)
        ref = A.double().cpu() @ W.double().cpu().t() + b.double().cpu()
        if act == O.ACT_RELU:
            ref = ref.clamp_min(0)
        tol = 2e-3 if out_f32 else 1.2e-2
        close(Y[:, :n_store], ref[:, :n_store], tol, tol, f"linear_fwd #{it} M={M} N={N} K={K} v={variant} act={act} f32={out_f32} pad={pad}")
    for it in range(25):
        N = int(rng.choice([64, 128, 256, 1024]))
        K = int(rng.choice([64, 128, 256, 1024, 1152]))
        M = int(rng.choice([64, 515, 4096, 20000, int(rng.integers(64, 40000))]))
        nv, kv = int(rng.integers(1, N + 1)), int(rng.integers(1, K + 1))
        variant = int(rng.choice([0, 1, 2, 3]))
        dZ = (gen(M, N, seed=300 + it) * (gen(M, N, seed=400 + it) > 0)).bfloat16().cuda()
        X = gen(M, K, seed=500 + it).clamp_min(0).bfloat16().cuda()
        dW = torch.zeros(N, K, device="cuda")
        O.linear_wgrad(dZ, X, dW, nv, kv, O.BF16, variant=variant)
        ref = dZ.double().cpu().t() @ X.double().cpu()
        close(dW[:nv, :kv], ref[:nv, :kv], 2e-3, 2e-3 * max(1.0, (M / 512) ** 0.5), f"linear_wgrad #{it} M={M} N={N} K={K} nv={nv} kv={kv} v={variant}")
        assert float(dW[nv:].abs().max() if nv < N else 0) == 0 and float(dW[:, kv:].abs().max() if kv < K else 0) == 0, f"wgrad #{it} wrote outside the valid block"


@pytest.mark.gpu
def test_per_ray_kernels_random_sizes_sweep():
    """Seeded sweep over ray counts and interval counts (not multiples of the wave size, single intervals, more than 64 / 128 / 192
    intervals) for the wave-per-ray and lane-per-ray kernels, against the oracle-backed emulation: mip / zip compositing forward and
    backward, the mip resampler (bit-exact), the classic sample_pdf (bit-exact)."""
    from snerf_amd import ops as O
    import cpu_ops_emulation as E
    rng = np.random.default_rng(77)
    cu = lambda t: None if t is None else t.cuda().contiguous()
    for it in range(14):
        n = int(rng.choice([1, 3, 65, 130, int(rng.integers(1, 400))]))
        S = int(rng.choice([1, 2, 31, 64, 65, 127, 128, 200, int(rng.integers(1, 250))]))
        g = torch.Generator().manual_seed(1000 + it)
        s = torch.sort(torch.rand(n, S + 1, generator=g), -1).values.contiguous()
        dirs = torch.randn(n, 3, generator=g)
        near, far = torch.rand(n, generator=g) + 0.5, torch.rand(n, generator=g) * 50 + 5
        raw_rgb, raw_d = torch.randn(n * S, 3, generator=g), torch.randn(n * S, 1, generator=g) * 3
        noise = torch.randn(n, S, generator=g) * 0.1 if it % 2 else None
        ref = E.mip_composite_fwd(raw_rgb, raw_d, noise, s, dirs, near, far, 0, False, 0.001, -1.0)
        got = O.mip_composite_fwd(cu(raw_rgb), cu(raw_d), cu(noise), cu(s), cu(dirs), cu(near), cu(far), 0, False, 0.001, -1.0)
        for a, b, what in zip(got, ref, ("rgb", "distance", "acc", "weights")):
            close(a, b, 1e-4, 5e-6, f"mip composite fwd #{it} n={n} S={S} {what}")     # t1 - t0 of close posts: both sides round it differently
        gs = [torch.randn(n, 3, generator=g), torch.randn(n, generator=g) * 0.05, torch.randn(n, generator=g), torch.randn(n, S, generator=g)]
        dr_ref, dd_ref = torch.empty(n * S, 3), torch.empty(n * S, 1)
        E.mip_composite_bwd(raw_rgb, raw_d, noise, s, dirs, near, far, 0, False, 0.001, -1.0, ref[3], ref[1], *gs, dr_ref, dd_ref)
        dr, dd = torch.empty(n * S, 3, device="cuda"), torch.empty(n * S, 1, device="cuda")
        O.mip_composite_bwd(cu(raw_rgb), cu(raw_d), cu(noise), cu(s), cu(dirs), cu(near), cu(far), 0, False, 0.001, -1.0, got[3], got[1], *[cu(x) for x in gs], dr, dd)
        close(dr, dr_ref, 5e-4, 1e-5 * float(dr_ref.abs().max() + 1), f"mip composite bwd #{it} n={n} S={S} d_rgb")
        close(dd, dd_ref, 5e-4, 1e-4 * float(dd_ref.abs().max() + 1), f"mip composite bwd #{it} n={n} S={S} d_density")
        # zipnerf compositing on metric fence posts
        td = (s * (far - near)[:, None] + near[:, None]).contiguous()
        zr = E.zip_composite_fwd(raw_rgb, raw_d, td, dirs, True, 1.0, 0.001, -1.0)
        zg = O.zip_composite_fwd(cu(raw_rgb), cu(raw_d), cu(td), cu(dirs), True, 1.0, 0.001, -1.0)
        for a, b, what in zip(zg, zr, ("rgb", "depth", "acc", "weights")):
            close(a, b, 1e-4, 5e-6, f"zip composite fwd #{it} n={n} S={S} {what}")
        # samplers: indices and values bit-exact
        if S >= 2:
            w = torch.rand(n, S, generator=g) ** 3
            P1 = int(rng.choice([2, 17, 64, 129]))
            u = torch.rand(n, P1, generator=g).sort(-1).values.clamp_max(1 - 1.2e-7).contiguous()
            rs, ri = E.mip_resample(s, w, u, 0.01, want_idx=True)
            gs_, gi = O.mip_resample(cu(s), cu(w), cu(u), 0.01, want_idx=True)
            assert torch.equal(gi.cpu().long(), ri.long()) and torch.equal(gs_.cpu(), rs), f"mip_resample #{it} n={n} S={S} P1={P1}"
            bins = (s * 10).contiguous()
            uu = torch.rand(n, P1, generator=g).contiguous()
            wc = torch.rand(n, S, generator=g)
            cs, ci, _ = E.classic_sample_pdf(bins, wc, uu, False, want_inds=True)
            ks, ki, _ = O.classic_sample_pdf(cu(bins), cu(wc), cu(uu), False, want_inds=True)
            assert torch.equal(ki.cpu().long(), ci.long()) and torch.equal(ks.cpu(), cs), f"classic_sample_pdf #{it} n={n} S={S} P1={P1}"


def test_gather_pack_refreshes_operands():
    """snerf_gather_pack: dst[i] = flat[idx[i]] rounded to dst's type, idx -1 -> 0 (padding), -2 -> 1 (identity rows); sizes that are
    not multiples of the 4-wide inner step; and the plan built by mlp._Net from its own packing code reproduces that code's result."""
    from snerf_amd import ops
    g = torch.Generator().manual_seed(5)
    flat = torch.randn(10007, generator=g).cuda()
    for n in (1, 3, 4, 1001, 4096 + 2):
        idx = torch.randint(-2, flat.numel(), (n,), generator=g, dtype=torch.int32)
        idx_d = torch.zeros(((n + 3) // 4) * 4, dtype=torch.int32, device="cuda")[:n]          # 16-byte aligned base
        idx_d.copy_(idx)
        want = torch.where(idx >= 0, flat.cpu()[idx.clamp(min=0).long()], (idx == -2).float())
        for dt in (torch.float32, torch.bfloat16):
            dst = torch.full((n,), 7.0, dtype=dt, device="cuda")
            ops.gather_pack(flat, idx_d, dst)
            assert torch.equal(dst.cpu(), want.to(dt)), (n, dt)
    # end to end: a packed operand equals the slicing code applied to the parameters
    from snerf_amd.mlp import ClassicNeRFNet, ParamArena
    shapes = ClassicNeRFNet.param_shapes(8, 128, 63, 27, (4,))
    arena = ParamArena(shapes, torch.device("cuda"))
    arena.load({k: torch.randn(s, generator=g) for k, s in shapes})
    net = ClassicNeRFNet(arena, "", ops.BF16, 8, 128)
    net.ensure_packed(True)
    W5 = arena.p["pts_linears.5.weight"]
    fw = net.fw["pts_linears.5"]
    assert torch.equal(fw[:128, :63], W5[:, :63].to(torch.bfloat16)) and torch.equal(fw[:128, net.Pw:net.Pw + 128], W5[:, 63:].to(torch.bfloat16))
    assert float(fw[:, 63:net.Pw].abs().max()) == 0.0
    tw = net.tw["pts_linears.5"]                          # data-gradient pack: W^T of the trunk columns
    assert torch.equal(tw[:128, :128], W5[:, 63:].t().to(torch.bfloat16))
    arena.p["pts_linears.5.weight"].mul_(2.0); arena.bump()
    net.ensure_packed(True)
    assert torch.equal(net.fw["pts_linears.5"][:128, :63], (W5[:, :63]).to(torch.bfloat16)) and net.fw["pts_linears.5"].data_ptr() == fw.data_ptr()
    assert torch.equal(net.tw["pts_linears.5"][:128, :128], W5[:, 63:].t().to(torch.bfloat16))
    # the transposed images leave the gather as 16 x 64 tiles (snerf_gather_pack_tiles): same bits as the plain gather over the unmarked map
    plan = net._plans["train"][0]
    tiles = plan.tiles[torch.bfloat16]
    assert tiles is not None and tiles.shape[0] >= 7 * 2 * 8                     # at least seven 128 x 128 trunk transposes (8 x 2 tiles each)
    idx = plan.maps[torch.bfloat16].clone()
    t = tiles.long()
    i, j = torch.arange(16, device="cuda").view(1, 16, 1), torch.arange(64, device="cuda").view(1, 1, 64)
    d = (t[:, 0].view(-1, 1, 1) + i * t[:, 3].view(-1, 1, 1) + j).reshape(-1)
    assert bool((idx[d] == -3).all()) and int((idx == -3).sum()) == d.numel()
    idx[d] = (t[:, 1].view(-1, 1, 1) + j * t[:, 2].view(-1, 1, 1) + i).reshape(-1).to(torch.int32)
    plain = torch.full_like(plan.pools[torch.bfloat16], 3.0)
    ops.gather_pack(arena.flat, idx, plain)
    assert torch.equal(plain, plan.pools[torch.bfloat16])
    # ... and the network's fp32 pool rides in the same launch (snerf_gather_pack_pair): the plain gather of its map gives the same pool
    assert set(plan.maps) == {torch.bfloat16, torch.float32}
    plain32 = torch.full_like(plan.pools[torch.float32], 3.0)
    ops.gather_pack(arena.flat, plan.maps[torch.float32], plain32)
    assert torch.equal(plain32, plan.pools[torch.float32]) and float(plain32.abs().max()) > 0
    both16, both32 = torch.full_like(plan.pools[torch.bfloat16], 5.0), torch.full_like(plan.pools[torch.float32], 5.0)
    ops.gather_pack_pair(arena.flat, plan.maps[torch.bfloat16], both16, plan.tiles[torch.bfloat16], plan.maps[torch.float32], both32)
    assert torch.equal(both16, plan.pools[torch.bfloat16]) and torch.equal(both32, plan.pools[torch.float32])


@pytest.mark.parametrize("M,N,K,nv,kv", [(40000, 256, 256, 256, 256), (70000, 128, 128, 128, 128), (50000, 256, 96, 256, 96), (33000, 128, 104, 128, 99),
                                         (4096, 256, 256, 200, 256)])
def test_wgrad_many_slice_fold(ops, M, N, K, nv, kv):
    """The few-tile weight gradients (hundreds of M slices per output tile) through partial tiles + tn_fold_many_kernel -- the default since round 6 --:
    against a float64 product, bit-identical run to run, accumulating (+=) into dW, untouched outside [:n_valid, :k_valid], and equal to the
    fp32-atomic form within its rounding."""
    g = torch.Generator().manual_seed(M + N)
    dZ = (torch.randn(M, N, generator=g) * 0.1).to(torch.bfloat16).cuda()
    X = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    assert ops.wgrad_uses_fold(M, N, K, ops.BF16, 3, nv, kv)
    ref = dZ.double().t()[:nv] @ X.double()[:, :kv]
    outs = []
    for _ in range(2):
        dW = torch.full((N, K), 0.5, dtype=torch.float32, device="cuda")
        ops.linear_wgrad(dZ, X, dW, nv, kv, ops.BF16, variant=3)
        outs.append(dW)
    assert torch.equal(outs[0], outs[1])
    got = outs[0].double() - 0.5
    assert float((got[:nv, :kv] - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 1e-4
    assert float(got[nv:].abs().max() if nv < N else 0.0) == 0.0 and float(got[:, kv:].abs().max() if kv < K else 0.0) == 0.0
    saved = ops.WGRAD_FOLD
    try:
        ops.WGRAD_FOLD = False
        dA = torch.full((N, K), 0.5, dtype=torch.float32, device="cuda")
        ops.linear_wgrad(dZ, X, dA, nv, kv, ops.BF16, variant=3)
    finally:
        ops.WGRAD_FOLD = saved
    assert float((dA - outs[0]).abs().max()) <= 2e-5 * float(ref.abs().max()) + 1e-4


def test_embedding_index_range_is_reported_and_gradient_is_reproducible(ops):
    """rays.app / cam_idx outside the embedding table -- nn.Embedding raises (models.py:153-159); the kernels clamp, and the
    asynchronous range check reports the launch at the next poll (no sync inside the step).  The deterministic accumulation of the
    table gradient (one workgroup per row, rays in order) is bit-reproducible and equals index_add."""
    V, dim, n, S = 7, 48, 300, 5
    g = torch.Generator().manual_seed(5)
    emb = torch.randn(V, dim, generator=g).to("cuda")
    app = torch.randint(0, V, (n,), generator=g).float().to("cuda")
    dst = torch.zeros(n * S, 64, device="cuda")
    ops.app_embed(emb, app, S, dst, ops.F32)
    ops.poll_index_checks(block=True)                         # in range: nothing raised
    assert torch.equal(dst[:, :dim], emb[app.long()].repeat_interleave(S, 0))
    bad = app.clone(); bad[17] = float(V)
    ops.app_embed(emb, bad, S, dst, ops.F32)
    with pytest.raises(IndexError, match="out of range"):
        ops.poll_index_checks(block=True)
    bad[17] = float("nan")
    ops.app_embed(emb, bad, S, dst, ops.F32)
    with pytest.raises(IndexError):
        ops.poll_index_checks(block=True)
    dV = torch.randn(n * S, 64, generator=g).to("cuda")
    outs = []
    for _ in range(2):
        ge = torch.zeros(V, dim, device="cuda")
        ops.app_embed_bwd(dV, app, S, ge, deterministic=True)
        outs.append(ge)
    assert torch.equal(outs[0], outs[1])
    want = torch.zeros(V, dim, device="cuda").index_add_(0, app.long(), dV[:, :dim].reshape(n, S, dim).sum(1))
    close(outs[0], want.cpu(), 1e-5, 1e-5, "deterministic embedding gradient")
    ga = torch.zeros(V, dim, device="cuda")
    ops.app_embed_bwd(dV, app, S, ga)
    close(ga, want.cpu(), 1e-5, 1e-5, "atomic embedding gradient")


@pytest.mark.gpu
def test_persistent_gemm_two_k_tiles_after_another_layer_is_reproducible():
    """gemm_nt8p_kernel with K = 128 (two k-tiles per output tile: the first layer of the 1024-wide NeRF MLP) launched right after ANOTHER
    persistent GEMM: the next tile's bias vector and the tile's mask words are DMA'd into LDS six to nine DMAs before the epilogue unit that
    reads them -- fewer than the ten the stream's vmcnt(10) leaves in flight.  Until round 4 nothing ordered them and about one launch in
    300 took the previous kernel's bias into one 32 x 64 output block (tools/probes/gemm_k128_bias_race.py: 5-6 events per 2000 launches
    before the fix, none after).  1500 launches here: a regression shows with > 95 % probability."""
    from snerf_amd import ops
    g = torch.Generator(device="cuda").manual_seed(0)
    rnd = lambda *s: torch.rand(*s, device="cuda", generator=g) * 2 - 1
    M, H, K = 524288, 1024, 128
    buf = rnd(M, H + K).bfloat16()
    A = buf[:, H:]                                                  # a column range of a wider buffer, as in the model
    W0, b0 = (rnd(H, K) / K ** 0.5).bfloat16(), rnd(H)
    W1, b1 = (rnd(H, H) / H ** 0.5).bfloat16(), rnd(H)
    Y, Z = torch.empty(M, H, dtype=torch.bfloat16, device="cuda"), torch.empty(M, H, dtype=torch.bfloat16, device="cuda")
    bits = torch.empty(ops.mask_bits_words(M, H), dtype=torch.int32, device="cuda")
    for act, aux, reps in ((ops.ACT_RELU, None, 900), (ops.ACT_RELU_BITS, bits, 600)):
        ops.linear_fwd(A, W0, b0, Y, K, H, act, ops.BF16, aux=aux, variant=8)
        ref = Y.clone()
        want = torch.relu(A.float() @ W0.float().t() + b0)
        assert float((ref.float() - want).abs().max()) < 0.05
        for r in range(reps):
            ops.linear_fwd(Y, W1, b1, Z, H, H, ops.ACT_RELU, ops.BF16, variant=8)        # leaves another layer's bias in the CUs' LDS
            ops.linear_fwd(A, W0, b0, Y, K, H, act, ops.BF16, aux=aux, variant=8)
            assert torch.equal(Y, ref), (act, r, int((Y != ref).sum()))
