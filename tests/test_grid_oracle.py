"""Known-answer tests that pin oracle/grid.py (the reference ships no vectors for its hash-grid kernels and its CUDA source
cannot be built or run here -- SURVEY.md section 8c): level sizing, hash values, affine reproduction, out-of-bound
handling, adjointness of backward, dy_dx vs finite differences, TV gradient vs a dense restatement."""
import numpy as np
import pytest

from oracle import grid as og


def test_level_layout_matches_shipped_configs():
    # (H=16, T=2^21, D=3), desired resolution 8192 / 2048 / 512 with L = 10 / 8 / 6 (internal/models.py:381-386,413-421)
    for L, desired, total in ((10, 8192, 14995560), (8, 2048, 10801256), (6, 512, 6606952)):
        off, res, s = og.level_layout(3, L, 4, 2.0, 16, 21, desired, False)
        assert int(off[-1]) == total, (L, desired, int(off[-1]))
        assert res[0] == 17 and res[-1] == desired + 1
        sizes = np.diff(off)
        assert (sizes % 8 == 0).all() and sizes.max() == 2 ** 21
        hashed = [i for i in range(L) if int(res[i]) ** 3 > 2 ** 21]
        assert hashed[0] == 3, "levels >= 3 are hashed"


def test_fast_hash_known_values():
    # xor of pos * {1, 2654435761, 805459861} in uint32
    def h(x, y, z):
        return ((x * 1) ^ ((y * 2654435761) & 0xFFFFFFFF) ^ ((z * 805459861) & 0xFFFFFFFF)) & 0xFFFFFFFF
    pts = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [5, 7, 11], [8192, 8192, 8192], [123456, 7890, 4242]], dtype=np.uint32)
    got = og.fast_hash(pts)
    want = np.array([h(*map(int, p)) for p in pts], dtype=np.uint32)
    assert (got == want).all()
    assert int(got[1]) == 1 and int(got[2]) == 2654435761 and int(got[3]) == 805459861
    assert int(og.fast_hash(np.array([[3, 2, 0]], dtype=np.uint32))[0]) == (3 ^ ((2 * 2654435761) & 0xFFFFFFFF))


def _affine_table(off, res, C, coef):
    """embeddings of a DENSE level set to an affine function of the cell coordinates"""
    tab = np.zeros((int(off[-1]), C), dtype=np.float32)
    for l in range(len(res)):
        r = int(res[l])
        if r ** 3 > off[l + 1] - off[l]:
            continue
        i = np.arange(r ** 3)
        x, y, z = i % r, (i // r) % r, i // (r * r)
        for c in range(C):
            tab[off[l] + i, c] = coef[c][0] + coef[c][1] * x + coef[c][2] * y + coef[c][3] * z
    return tab


def test_trilinear_reproduces_affine_fields_and_oob_is_zero():
    off, res, s = og.level_layout(3, 3, 2, 1.5, 8, 19, None, False)       # all three levels dense
    coef = [[0.5, 0.25, -0.125, 0.0625], [-1.0, 0.5, 0.75, -0.25]]
    tab = _affine_table(off, res, 2, coef)
    rng = np.random.default_rng(0)
    x = rng.random((64, 3)).astype(np.float32)
    x[0] = [-0.01, 0.5, 0.5]; x[1] = [0.5, 1.01, 0.5]                     # out of bounds
    S = np.log2(s)
    out, dd = og.grid_encode_forward(x, tab, off, S, 8, 0, False, 0, want_dy_dx=True)
    assert (out[:, :2] == 0).all() and (dd[:2] == 0).all()
    for l in range(3):
        scale = np.float32(np.exp2(np.float32(l) * np.float32(S)) * 8 - 1)
        pos = x[2:] * scale + 0.5                                          # cell coordinates (note the +0.5 offset, gridencoder.cu:148)
        for c in range(2):
            want = coef[c][0] + coef[c][1] * pos[:, 0] + coef[c][2] * pos[:, 1] + coef[c][3] * pos[:, 2]
            np.testing.assert_allclose(out[l, 2:, c], want, rtol=2e-5, atol=2e-5)
            grad = dd.reshape(64, 3, 3, 2)[2:, l, :, c]                    # d out / d x = coef * scale
            np.testing.assert_allclose(grad, np.array(coef[c][1:]) * scale * np.ones_like(grad), rtol=1e-4, atol=1e-4)


def test_backward_is_adjoint_of_forward_and_dydx_matches_finite_differences():
    off, res, s = og.level_layout(3, 5, 2, 2.0, 4, 9, None, False)        # levels 3,4 hashed (T = 512)
    rng = np.random.default_rng(1)
    E = rng.standard_normal((int(off[-1]), 2)).astype(np.float32)
    x = rng.random((50, 3)).astype(np.float32)
    S = 1.0
    for interp in (0, 1):
        out, dd = og.grid_encode_forward(x, E, off, S, 4, 0, False, interp, want_dy_dx=True)
        G = rng.standard_normal(out.shape).astype(np.float32)
        gE, gx = og.grid_encode_backward(G, x, off, E.shape[0], S, 4, 0, False, interp, dy_dx=dd)
        lhs = float((out.astype(np.float64) * G).sum()); rhs = float((E.astype(np.float64) * gE).sum())
        assert abs(lhs - rhs) < 1e-3 * max(1.0, abs(lhs)), (lhs, rhs)
        # finite differences of sum(out * G) w.r.t. x (central; the function is piecewise smooth, so use points away from cell borders)
        eps = 1e-3
        fd = np.zeros_like(x, dtype=np.float64)
        for d in range(3):
            xp, xm = x.copy(), x.copy(); xp[:, d] += eps; xm[:, d] -= eps
            fp = (og.grid_encode_forward(np.clip(xp, 0, 1), E, off, S, 4, 0, False, interp).astype(np.float64) * G).sum(axis=(0, 2))
            fm = (og.grid_encode_forward(np.clip(xm, 0, 1), E, off, S, 4, 0, False, interp).astype(np.float64) * G).sum(axis=(0, 2))
            fd[:, d] = (fp - fm) / (2 * eps)
        ok = np.abs(fd - gx) < 0.05 * (np.abs(fd).max() + 1)
        assert ok.mean() > 0.8, ok.mean()                                  # cell-border crossings within +-eps are not differentiable


def test_tv_gradient_dense_restatement():
    off, res, s = og.level_layout(3, 2, 1, 2.0, 4, 19, None, False)
    rng = np.random.default_rng(2)
    E = rng.standard_normal((int(off[-1]), 1)).astype(np.float32)
    x = rng.random((20, 3)).astype(np.float32)
    got = og.grad_total_variation(x, E, off, 0.5, 1.0, 4, 0, False)
    want = np.zeros_like(E, dtype=np.float64)
    for l in range(2):
        r = int(res[l]); scale = np.float32(np.exp2(np.float32(l)) * 4 - 1); rr = int(np.ceil(scale)) + 1
        T = E[off[l]:off[l] + r ** 3, 0].reshape(r, r, r)                  # [z, y, x]
        for p in x:
            c = np.floor(p * scale + 0.5).astype(int)
            v0 = T[c[2], c[1], c[0]]; tot = 0.0; sq = 0.0
            for d in range(3):
                for step in (1, -1):
                    n = c.copy(); n[d] += step
                    if (step == 1 and c[d] < rr) or (step == -1 and c[d] > 0):
                        gv = v0 - T[n[2], n[1], n[0]]; tot += gv; sq += gv * gv
            want[off[l] + c[0] + c[1] * r + c[2] * r * r, 0] += 0.5 / 6 * tot / np.sqrt(sq + 1e-9)
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("D", [2, 4, 5])
def test_other_input_dims_known_answers(D):
    """D = 2, 4, 5 (gridencoder.cu:376-399): the hash takes one prime per dimension (:50-63), dense levels index with stride (res)^d
    (:66-84; the level size rule adds 1 to the resolution when corners are not aligned), and multilinear interpolation reproduces an
    affine field of the cell coordinates exactly -- in double as well as in fp32 tables."""
    primes = [1, 2654435761, 805459861, 3674653429, 2097192037]
    p = np.array([[3, 5, 7, 11, 13][:D]], dtype=np.uint32)
    want = 0
    for d in range(D):
        want ^= (int(p[0, d]) * primes[d]) & 0xFFFFFFFF
    assert int(og.fast_hash(p)[0]) == want
    off, res, s = og.level_layout(D, 2, 1, 1.5, 3, 19, None, False)        # both levels dense
    S = np.log2(s)
    coef = np.array([0.5, 0.25, -0.125, 0.0625, 0.75, -0.375][:D + 1])
    tab = np.zeros((int(off[-1]), 1), dtype=np.float64)
    for l in range(2):
        r = int(res[l])
        i = np.arange(r ** D)
        cell = np.stack([(i // r ** d) % r for d in range(D)], -1)
        tab[off[l] + i, 0] = coef[0] + cell @ coef[1:]
    rng = np.random.default_rng(D)
    x = rng.random((50, D)).astype(np.float32)
    for vt, tol in ((None, 2e-5), (np.float64, 2e-7)):      # double values, but the weights are fp32 products in every instantiation
        out = og.grid_encode_forward(x, tab, off, S, 3, 0, False, 0, vt=vt)
        for l in range(2):
            scale = np.float32(np.exp2(np.float32(l) * np.float32(S)) * 3 - 1)
            pos = (x * scale + np.float32(0.5)).astype(np.float32)
            # the oracle (like the kernel) forms frac = pos - floor(pos) in fp32: evaluate the affine field at floor + frac
            cellpos = np.floor(pos).astype(np.float64) + (pos - np.floor(pos)).astype(np.float64)
            np.testing.assert_allclose(out[l, :, 0], coef[0] + cellpos @ coef[1:], rtol=tol, atol=tol)


def test_eager_restatements_match_the_oracle():
    """oracle/eager.py (the torch-op forms bench.py's baseline legs of paths B / C run on the CPU and on the GPU): GridEncoder forward and
    table gradient against oracle/grid.py, sample_pdf against oracle/classic.py's canonical-order sampler (same indices)."""
    import torch
    from oracle import zip as oz, eager, classic as oc
    spec = oz.GridSpec(6, 4, 512, 16, 12)                       # levels >= 2 hashed
    g = torch.Generator().manual_seed(0)
    emb = torch.randn(spec.rows, 4, generator=g) * 0.3
    m = torch.rand(300, 7, 3, generator=g) * 2.2 - 1.1           # some points outside [-1, 1]: zeros
    assert float((oz.grid_features(spec, emb, m) - eager.grid_features_torch(spec, emb, m)).abs().max()) < 1e-6
    embg = emb.clone().requires_grad_(True)
    G = torch.randn(300, 7, 6, 4, generator=g)
    (eager.grid_features_torch(spec, embg, m) * G).sum().backward()
    x01 = ((m.reshape(-1, 3) + 1) / 2).numpy().astype(np.float32)
    gE, _ = og.grid_encode_backward(np.ascontiguousarray(G.reshape(-1, 6, 4).permute(1, 0, 2).numpy()), x01, spec.offsets, spec.rows, spec.S, spec.H, 0, False, 0)
    assert float((embg.grad - torch.from_numpy(gE)).abs().max()) < 1e-5 * float(np.abs(gE).max())
    bins = torch.sort(torch.rand(50, 63, generator=g), -1).values
    w, u = torch.rand(50, 62, generator=g), torch.rand(50, 128, generator=g)
    s1, i1 = oc.sample_pdf(bins, w, u)
    s2, i2 = eager.sample_pdf_torch(bins, w, u)
    assert float((s1 - s2).abs().max()) < 1e-5 and int((i1 != i2).sum()) <= 2      # (a tie at fp32 rounding may move an index by one)
