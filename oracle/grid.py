"""Oracle for the multiresolution hash-grid encoder of S-NeRF++ / zipnerf (row C6/C10 of SURVEY.md section 8a).
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

PARITY UNPINNED BY THE REFERENCE: the algorithm lives in the reference's own CUDA source
(/root/reference/s-nerfpp/zipnerf/gridencoder/src/gridencoder.cu), which can neither be compiled for this
image (hipify chokes on its __half2 atomics, the build pins -std=c++14 against torch 2.10 headers) nor run
(no GPU in the build container); the reference ships no tests or vectors for it.  This file restates the
published algorithm line by line in numpy (citations below) and is pinned instead by analytic known-answer
tests (tests/test_grid_oracle.py): level sizing/offsets for the shipped configurations (14 995 560 /
10 801 256 / 6 606 952 entries), hand-computed fast_hash values with uint32 wrap, exact reproduction of
affine fields on dense levels, out-of-bound handling, <fwd(x;E), G> == <E, bwd(G)>, dy_dx vs finite
differences.

All arithmetic is fp32 with uint32 index math, like the kernels (gridencoder.cu:50-84, 137-197).
"""
import numpy as np

PRIMES = np.array([1, 2654435761, 805459861, 3674653429, 2097192037, 1434869437, 2165219737], dtype=np.uint64)
F = np.float32


def level_layout(input_dim=3, num_levels=16, level_dim=2, per_level_scale=2.0, base_resolution=16, log2_hashmap_size=19,
                 desired_resolution=None, align_corners=False):
    """grid.py:104-141: per-level resolution = ceil(H * s^i) (+1 unless align_corners), entries =
    min(2^log2T, res^D) rounded up to a multiple of 8.  Returns (offsets int32 [L+1], resolutions int32 [L],
    per_level_scale)."""
    if desired_resolution is not None:
        per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))
    offsets, resolutions, offset = [], [], 0
    max_params = 2 ** log2_hashmap_size
    for i in range(num_levels):
        res = int(np.ceil(base_resolution * per_level_scale ** i))
        res = res if align_corners else res + 1
        n = min(max_params, res ** input_dim)
        n = int(np.ceil(n / 8) * 8)
        resolutions.append(res)
        offsets.append(offset)
        offset += n
    offsets.append(offset)
    return np.array(offsets, dtype=np.int32), np.array(resolutions, dtype=np.int32), float(per_level_scale)


def fast_hash(pos_grid: np.ndarray) -> np.ndarray:
    """gridencoder.cu:50-63: xor_d (pos_grid[d] * primes[d]) in uint32 (wrap-around multiply)."""
    D = pos_grid.shape[-1]
    r = np.zeros(pos_grid.shape[:-1], dtype=np.uint64)
    for d in range(D):
        r ^= (pos_grid[..., d].astype(np.uint64) * PRIMES[d]) & np.uint64(0xFFFFFFFF)
    return r.astype(np.uint32)


def grid_index(gridtype: int, align_corners: bool, hashmap_size: int, resolution: int, pos_grid: np.ndarray) -> np.ndarray:
    """gridencoder.cu:66-84 (without the *C + ch): dense index while the running stride fits the level, else hash."""
    D = pos_grid.shape[-1]
    stride = 1
    index = np.zeros(pos_grid.shape[:-1], dtype=np.uint64)
    d = 0
    while d < D and stride <= hashmap_size:
        index = (index + pos_grid[..., d].astype(np.uint64) * np.uint64(stride)) & np.uint64(0xFFFFFFFF)
        stride = (stride * (resolution if align_corners else resolution + 1)) & 0xFFFFFFFF
        d += 1
    if gridtype == 0 and stride > hashmap_size:
        index = fast_hash(pos_grid).astype(np.uint64)
    return (index % np.uint64(hashmap_size)).astype(np.int64)


def _level_setup(level, S, H, offsets):
    hashmap_size = int(offsets[level + 1] - offsets[level])
    scale = F(F(np.exp2(F(level) * F(S))) * F(H)) - F(1.0)
    resolution = int(np.ceil(scale)) + 1
    return hashmap_size, F(scale), resolution


def _positions(x, scale, align_corners, interp):
    pos = (x * scale + F(0.0 if align_corners else 0.5)).astype(F)
    pg = np.floor(pos).astype(F)
    frac = (pos - pg).astype(F)
    deriv = np.ones_like(frac)
    if interp == 1:  # smoothstep gridencoder.cu:37-46
        deriv = (F(6) * frac * (F(1) - frac)).astype(F)
        frac = (frac * frac * (F(3) - F(2) * frac)).astype(F)
    return pg.astype(np.uint32), frac, deriv


def grid_encode_forward(inputs, embeddings, offsets, S, H, gridtype=0, align_corners=False, interp=0, want_dy_dx=False, vt=None):
    """kernel_grid, gridencoder.cu:87-245.  inputs [B,D] fp32 in [0,1]; embeddings [sO,C]; -> outputs [L,B,C] fp32
    (+ dy_dx [B, L*D*C]).  Out-of-bound points give zeros.  vt = scalar type of the table values and their accumulation (the
    kernel's scalar_t, :137-147; positions and weights are float whatever the table type): default fp32, np.float64 for the
    double instantiation (:469 AT_DISPATCH_FLOATING_TYPES_AND_HALF)."""
    F = np.float32
    V = F if vt is None else vt
    x = np.asarray(inputs, dtype=F)
    E = np.asarray(embeddings).astype(V)
    B, D = x.shape
    C = E.shape[1]
    L = len(offsets) - 1
    out = np.zeros((L, B, C), dtype=V)
    dy_dx = np.zeros((B, L, D, C), dtype=V) if want_dy_dx else None
    oob = ((x < 0) | (x > 1)).any(-1)
    for l in range(L):
        hs, scale, res = _level_setup(l, S, H, offsets)
        tab = E[offsets[l]:offsets[l + 1]]
        pg, frac, deriv = _positions(x, scale, align_corners, interp)
        acc = np.zeros((B, C), dtype=V)
        for idx in range(1 << D):
            w = np.ones(B, dtype=F)
            pl = pg.copy()
            for d in range(D):
                if idx & (1 << d):
                    w = (w * frac[:, d]).astype(F); pl[:, d] = pg[:, d] + 1
                else:
                    w = (w * (F(1) - frac[:, d])).astype(F)
            gi = grid_index(gridtype, align_corners, hs, res, pl)
            acc = (acc + w[:, None] * tab[gi]).astype(V)
        acc[oob] = 0
        out[l] = acc
        if want_dy_dx:
            for gd in range(D):
                g = np.zeros((B, C), dtype=V)
                others = [d for d in range(D) if d != gd]
                for idx in range(1 << (D - 1)):
                    w = np.full(B, scale, dtype=F)
                    pl = pg.copy()
                    for nd, d in enumerate(others):
                        if idx & (1 << nd):
                            w = (w * frac[:, d]).astype(F); pl[:, d] = pg[:, d] + 1
                        else:
                            w = (w * (F(1) - frac[:, d])).astype(F)
                    pl[:, gd] = pg[:, gd]
                    il = grid_index(gridtype, align_corners, hs, res, pl)
                    pl[:, gd] = pg[:, gd] + 1
                    ir = grid_index(gridtype, align_corners, hs, res, pl)
                    g = (g + (w[:, None] * (tab[ir] - tab[il])) * deriv[:, gd:gd + 1]).astype(V)
                g[oob] = 0
                dy_dx[:, l, gd] = g
    return (out, dy_dx.reshape(B, L * D * C)) if want_dy_dx else out


def grid_encode_backward(grad, inputs, offsets, table_rows, S, H, gridtype=0, align_corners=False, interp=0, dy_dx=None, vt=None):
    """kernel_grid_backward (gridencoder.cu:248-340): scatter-add of w * grad into the table (exact fp64 accumulation
    here; the kernels use atomics, order-dependent in the last bits) and kernel_input_backward (:343-369).  vt: see the forward."""
    F = np.float32
    V = F if vt is None else vt
    x = np.asarray(inputs, dtype=F)
    G = np.asarray(grad).astype(V)            # [L,B,C]
    L, B, C = G.shape
    D = x.shape[1]
    gE = np.zeros((table_rows, C), dtype=np.float64)
    oob = ((x < 0) | (x > 1)).any(-1)
    for l in range(L):
        hs, scale, res = _level_setup(l, S, H, offsets)
        pg, frac, _ = _positions(x, scale, align_corners, interp)
        for idx in range(1 << D):
            w = np.ones(B, dtype=F)
            pl = pg.copy()
            for d in range(D):
                if idx & (1 << d):
                    w = (w * frac[:, d]).astype(F); pl[:, d] = pg[:, d] + 1
                else:
                    w = (w * (F(1) - frac[:, d])).astype(F)
            gi = grid_index(gridtype, align_corners, hs, res, pl) + int(offsets[l])
            contrib = (w[:, None] * G[l]).astype(V)
            contrib[oob] = 0
            np.add.at(gE, gi, contrib.astype(np.float64))
    g_in = None
    if dy_dx is not None:
        dd = np.asarray(dy_dx).astype(V).reshape(B, L, D, C)
        g_in = np.einsum("lbc,bldc->bd", G.astype(np.float64), dd.astype(np.float64)).astype(V)
    return gE.astype(V), g_in


def grad_total_variation(inputs, embeddings, offsets, weight, S, H, gridtype=0, align_corners=False):
    """kernel_grad_tv, gridencoder.cu:506-610: normalised TV gradient at the cells of `inputs` -> additive update of the
    table gradient (returned, fp64-accumulated)."""
    x = np.asarray(inputs, dtype=F)
    E = np.asarray(embeddings).astype(F)
    B, D = x.shape
    C = E.shape[1]
    L = len(offsets) - 1
    out = np.zeros(E.shape, dtype=np.float64)
    oob = ((x < 0) | (x > 1)).any(-1)
    wgt = F(weight) / F(2 * D)
    for l in range(L):
        hs, scale, res = _level_setup(l, S, H, offsets)
        tab = E[offsets[l]:offsets[l + 1]]
        pos = (x * scale + F(0.0 if align_corners else 0.5)).astype(F)
        pg = np.floor(pos).astype(np.int64)
        idx0 = grid_index(gridtype, align_corners, hs, res, pg.astype(np.uint32))
        results = np.zeros((B, C), dtype=F)
        idelta = np.zeros((B, C), dtype=F)
        for d in range(D):
            cur = pg[:, d]
            right = cur < res
            pr = pg.copy(); pr[:, d] = cur + 1
            ir = grid_index(gridtype, align_corners, hs, res, np.where(right[:, None], pr, pg).astype(np.uint32))
            gv = np.where(right[:, None], tab[idx0] - tab[ir], 0).astype(F)
            results = (results + gv).astype(F); idelta = (idelta + gv * gv).astype(F)
            left = cur > 0
            plf = pg.copy(); plf[:, d] = cur - 1
            il = grid_index(gridtype, align_corners, hs, res, np.where(left[:, None], plf, pg).astype(np.uint32))
            gv = np.where(left[:, None], tab[idx0] - tab[il], 0).astype(F)
            results = (results + gv).astype(F); idelta = (idelta + gv * gv).astype(F)
        upd = (wgt * results / np.sqrt(idelta + F(1e-9))).astype(F)
        upd[oob] = 0
        np.add.at(out, idx0 + int(offsets[l]), upd.astype(np.float64))
    return out.astype(F)
