"""Shared oracle helpers (test infrastructure only, see oracle/__init__.py).

Canonical accumulation order for the index-producing samplers
------------------------------------------------------------
The reference runs ``torch.cumsum`` / ``torch.sum`` on fp32 tensors.  On the
CPU ``torch.cumsum`` accumulates in fp64 and rounds every prefix to fp32
(checked in the build container: bit-identical to ``np.cumsum(x.astype(f64))
.astype(f32)``), while ``torch.sum`` uses a vectorised fp32 tree whose shape
depends on the CPU's SIMD width.  Sample *indices* must be bit-exact between
the HIP kernels and this oracle, so both use one explicit order:

  * prefix sums and row sums are accumulated sequentially, left to right, in
    fp64, and each emitted value is rounded once to fp32;
  * every other operation is a single correctly-rounded fp32 operation with
    no fused multiply-add contraction.

``seq_cumsum_f32`` / ``seq_sum_f32`` implement that order.
"""
import math

import numpy as np
import torch

F32 = np.float32


def seq_cumsum_f32(x: np.ndarray) -> np.ndarray:
    """Inclusive prefix sum along the last axis, fp64 accumulate, fp32 out."""
    assert x.dtype == np.float32
    return np.cumsum(x.astype(np.float64), axis=-1).astype(np.float32)


def seq_sum_f32(x: np.ndarray, keepdims=True) -> np.ndarray:
    """Row sum along the last axis in the canonical order (== last prefix)."""
    assert x.dtype == np.float32
    c = np.cumsum(x.astype(np.float64), axis=-1)[..., -1:]
    c = c.astype(np.float32)
    return c if keepdims else c[..., 0]


def formula_weight(out_f: int, in_f: int, layer: int, scale: float = None) -> torch.Tensor:
    """Deterministic dense weight W[i, j] = s * sin(0.37 i + 1.13 j + layer).

    Used for golden vectors so that no large weight files are committed
    (SURVEY.md section 8c).  ``s`` defaults to a xavier-like 1.6/sqrt(in_f)
    which keeps activations O(1) through 8+ layers.
    """
    if scale is None:
        scale = 1.6 / math.sqrt(in_f)
    i = torch.arange(out_f, dtype=torch.float64)[:, None]
    j = torch.arange(in_f, dtype=torch.float64)[None, :]
    return (scale * torch.sin(0.37 * i + 1.13 * j + float(layer))).to(torch.float32)


def formula_bias(out_f: int, layer: int, scale: float = 0.05) -> torch.Tensor:
    i = torch.arange(out_f, dtype=torch.float64)
    return (scale * torch.cos(0.71 * i + 0.5 * float(layer))).to(torch.float32)


def fill_state_dict_(sd: dict) -> dict:
    """Overwrite every tensor of a state-dict-like mapping with formula values.

    Layer index = position of the key in the (ordered) mapping, so the same
    call on the reference module and on the build's module yields identical
    parameters without shipping them.
    """
    out = {}
    for idx, (k, v) in enumerate(sd.items()):
        if v.dim() == 2:
            out[k] = formula_weight(v.shape[0], v.shape[1], idx)
        elif v.dim() == 1:
            out[k] = formula_bias(v.shape[0], idx)
        else:
            raise ValueError(f"unexpected parameter rank for {k}: {tuple(v.shape)}")
    return out


def psnr(a: torch.Tensor, b: torch.Tensor) -> float:
    """-10 log10(mean((a-b)^2)); reference s-nerf/model/math_ops.py:78-80."""
    mse = torch.mean((a.double() - b.double()) ** 2).item()
    if mse == 0.0:
        return float("inf")
    return -10.0 * math.log10(mse)


def synthetic_rays(n: int, seed: int = 0, H: int = 900, W: int = 1600, focal: float = 1266.0,
                   near: float = 2.0 * 0.9, far: float = 100.0 * 1.1):
    """nuScenes-like ray batch (workload M2).

    Pinhole camera H x W, focal 1266, principal point at the centre; pixels
    drawn without replacement from a seeded generator; per-ray radii follow the
    reference's neighbour-direction rule (s-nerf/utils/sample_utils.py:119-125:
    |dir(x+1) - dir(x)| * 2/sqrt(12)); origins jittered N(0, 0.1^2).
    Returns a dict of fp32 torch tensors with the field names of the
    reference's ``Rays`` namedtuple (s-nerf/utils/sample_utils.py:11-13).
    """
    rng = np.random.default_rng(seed)
    pix = rng.choice(H * W, size=n, replace=False) if n <= H * W else rng.integers(0, H * W, size=n)
    j = (pix // W).astype(np.float64)  # row
    i = (pix % W).astype(np.float64)   # col
    cx, cy = W * 0.5, H * 0.5
    # camera-to-world: yaw rotation so that directions are not axis aligned
    th = 0.3
    R = np.array([[math.cos(th), 0.0, math.sin(th)], [0.0, 1.0, 0.0], [-math.sin(th), 0.0, math.cos(th)]])

    def dirs_of(ii, jj):
        d = np.stack([(ii - cx + 0.5) / focal, -(jj - cy + 0.5) / focal, -np.ones_like(ii)], -1)
        return d @ R.T

    d = dirs_of(i, j)
    d_next = dirs_of(i, np.minimum(j + 1, H - 1))
    d_prev = dirs_of(i, np.maximum(j - 1, 0))
    dx = np.where((j + 1 <= H - 1)[:, None], d_next - d, d - d_prev)
    radii = np.sqrt((dx ** 2).sum(-1, keepdims=True)) * 2.0 / math.sqrt(12.0)
    o = rng.normal(0.0, 0.1, size=(n, 3))
    viewdirs = d / np.linalg.norm(d, axis=-1, keepdims=True)
    ones = np.ones((n, 1))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    return {
        "origins": t(o), "directions": t(d), "viewdirs": t(viewdirs), "radii": t(radii),
        "lossmult": t(ones), "near": t(ones * near), "far": t(ones * far), "app": t(ones * 0.0),
    }


# ---- golden fixtures: one writer for every oracle/gen_golden*.py, with a verification mode ---------------------------------------------
_GOLDEN_BAD = []


def golden_equal(a: np.ndarray, b: np.ndarray) -> bool:
    """bit-for-bit equality of a stored and a regenerated array (same dtype and shape; NaNs equal; string arrays compared with ==)"""
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape or a.dtype != b.dtype:
        return False
    if a.dtype.kind in "fc":
        return bool(np.array_equal(a, b, equal_nan=True))
    return bool(np.array_equal(a, b))


def save_golden(path: str, **arrays) -> None:
    """np.savez_compressed(path, **arrays) -- or, when the generator runs with `--check`, regenerate-and-compare: the arrays are checked
    against the committed file (keys, dtypes, shapes, every bit) and NOTHING is written; the process prints `check: OK` (exit 0) or the
    mismatching keys (exit 1) when it ends.  tests/test_oracle_golden.py runs every generator this way when /root/reference exists."""
    import atexit
    import os
    import sys
    arrays = {k: np.asarray(v) for k, v in arrays.items()}
    if "--check" not in sys.argv:
        np.savez_compressed(path, **arrays)
        return
    if not _GOLDEN_BAD and not getattr(save_golden, "_hooked", False):
        def report():
            bad = [b for b in _GOLDEN_BAD if b is not None]
            print("check:", "OK" if not bad else f"{len(bad)} mismatches: {bad[:8]}", flush=True)
            if bad:
                os._exit(1)
        atexit.register(report)
        save_golden._hooked = True
    _GOLDEN_BAD.append(None)
    name = os.path.basename(path)
    if not os.path.exists(path):
        _GOLDEN_BAD.append(f"{name}: missing")
        return
    old = np.load(path, allow_pickle=False)
    if set(old.files) != set(arrays):
        _GOLDEN_BAD.append(f"{name}: keys differ ({sorted(set(old.files) ^ set(arrays))[:4]})")
    for k, v in arrays.items():
        if k in old.files and not golden_equal(old[k], v):
            _GOLDEN_BAD.append(f"{name}:{k}")
