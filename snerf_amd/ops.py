"""Thin torch-tensor -> C-ABI wrappers (device pointers, strides, current HIP stream).

PyTorch is plumbing here: it owns device memory and the stream.  Every
function launches hand-written HIP kernels from libsnerf_hip.so through
``_lib.call``; nothing in this file computes on the host or falls back to
torch ops.
"""
import os as _os

import torch

from . import _lib

F32, BF16, F16, F64 = 0, 1, 2, 3
BF16X3 = 4                                 # GEMMs only: split-bf16 operands, the fp32-parity mode at bf16 MFMA rates (csrc/gemm.hip, GemmNT::split)
F16F8 = 5                                  # snerf_linear_fwd only: fp16 + fp8 split operands, the same contract in two pass-equivalents (GemmNT::split == 2)
SPLIT_DTS = (BF16X3, F16F8)                # activation buffers hold 2 physical 2-byte columns per logical column
ACT_NONE, ACT_RELU, ACT_MASK = 0, 1, 2
ACT_RELU_BITS, ACT_MASK_BITS = 3, 4        # ReLU that also writes a 1-bit-per-element mask / ReLU backward from that mask
_TORCH_DT = {F32: torch.float32, BF16: torch.bfloat16, F16: torch.float16, BF16X3: torch.bfloat16, F16F8: torch.float16}


def torch_dtype(dt: int):
    return _TORCH_DT[dt]


def gran(dt: int) -> int:
    """reduction-axis granularity of the GEMM tiles: 128-byte LDS rows."""
    return 32 if dt == F32 else 64                    # (BF16X3: 64 LOGICAL columns = one 128-column physical group)


def roundup(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


def _chk2d(t, dt=None):
    assert t.is_cuda and t.dim() == 2 and t.stride(1) == 1, "expected a row-major 2-D CUDA view"
    if dt is not None:
        assert t.dtype == dt, f"dtype {t.dtype} != {dt}"
    return t


def _f32c(t):
    assert t is None or (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()), "expected contiguous fp32 CUDA tensor"
    return t


_zeros_cache = {}


def lds_scribble(seed):
    """test utility: fill every CU's LDS with a seeded pattern (tests/test_stale_lds.py)"""
    _lib.call("snerf_debug_lds_scribble", int(seed), _stream())


def zero_page(device):
    z = _zeros_cache.get(device)
    if z is None:
        z = torch.zeros(64, dtype=torch.float32, device=device)
        _zeros_cache[device] = z
    return z


# ------------------------------------------------------------------ GEMMs ----
def linear_fwd(A, W, bias, Y, K, n_store, act, dt, out_f32=False, aux=None, colsum=None, variant=0, deterministic=False, aux_split=False):
    """Y[:, :n_store] = act(A[:, :K] @ W[:, :K]^T + bias); A/W/Y are 2-D views (row stride = ld).  `deterministic`: the bias-gradient
    partials (`colsum`) are folded in a fixed order.  `aux_split` (plain bf16 launch, ACT_MASK): the mask source is an activation a
    split-bf16 forward saved (interleaved hi / lo; its hi half is tested)."""
    if deterministic:
        variant |= 256
    if aux_split and act == ACT_MASK:
        assert dt in (BF16, F16) and aux is not None and aux.shape[1] >= 2 * roundup(n_store, 64) - 64
        variant |= 1 << 14
    _chk2d(A, _TORCH_DT[dt]); _chk2d(W, _TORCH_DT[dt]); _chk2d(Y, torch.float32 if out_f32 else _TORCH_DT[dt])
    M = A.shape[0]
    sp = dt in SPLIT_DTS                                  # split-bf16: A [M, 2 K], W [N, 3 K], a bf16 Y [M, 2 n_store] (interleaved layout); fp16 + fp8: W [N, 2 K]
    assert Y.shape[0] == M and A.shape[1] >= K * (2 if sp else 1) and W.shape[1] >= K * ((3 if dt == BF16X3 else 2) if sp else 1)
    assert Y.shape[1] >= n_store * (2 if sp and not out_f32 else 1) or (sp and not out_f32 and Y.shape[1] >= 2 * roundup(n_store, 64) - 64)
    if aux is not None and act < ACT_RELU_BITS:
        _chk2d(aux, _TORCH_DT[dt])
    if act >= ACT_RELU_BITS:
        assert aux is not None and aux.dtype == torch.int32 and aux.is_contiguous() and aux.numel() >= mask_bits_words(M, W.shape[0])
    ws = None
    if colsum is not None:  # partial column sums (no atomics), one row per row slab or per (workgroup, wave row); folded into `colsum` by a second kernel
        tiles = ((M + 255) // 256) * max(W.shape[0] // 256, 1)
        ws = torch.empty(max(2 * ((M + 127) // 128), 2 * min(tiles, 1024)), W.shape[0], dtype=torch.float32, device=A.device)
    _lib.call("snerf_linear_fwd", _p(A), A.stride(0), _p(W), W.stride(0), _p(bias), _p(Y), Y.stride(0),
              _p(aux), 0 if (aux is None or act >= ACT_RELU_BITS) else aux.stride(0), _p(colsum), _p(ws), M, W.shape[0], K, n_store, act, dt,
              1 if out_f32 else 0, variant, _stream())


def fast_epilogue_ok(Y, n_store, dt, out_f32=False, aux=None, act=ACT_NONE):
    """Whether snerf_linear_fwd takes its LDS-transposed 16-byte epilogue for this launch (mirrors the dispatch in gemm.hip): the only
    epilogue whose bias-gradient column sums can be folded deterministically (the direct-store epilogue adds them with atomics)."""
    epc = 4 if dt == F32 else 8
    ok = not (out_f32 and dt in (BF16, F16)) and Y.stride(0) % epc == 0 and Y.data_ptr() % 16 == 0 and n_store % epc == 0
    if act == ACT_MASK and aux is not None:
        ok = ok and aux.stride(0) % epc == 0 and aux.data_ptr() % 16 == 0
    return ok


def mask_bits_words(M, N):
    """int32 words of the ReLU bit mask of an [M, N] activation (N % 64 == 0): one 64-word block per (32 rows, 64 columns)."""
    return 8 * ((M + 255) // 256) * (N // 64) * 64


def relu_bits_ok(A, W, Y, K, n_store, dt, variant, consumer=False):
    """Whether snerf_linear_fwd accepts ACT_RELU_BITS (producer) / ACT_MASK_BITS (consumer) for this launch: the persistent
    8-phase kernel's conditions (mirrors the dispatch in gemm.hip)."""
    N = W.shape[0]
    return (dt in (BF16, BF16X3, F16, F16F8) and (variant & 8) and N % 256 == 0 and K * (3 if dt == BF16X3 else 1) >= 128 and Y.dtype == _TORCH_DT[dt]
            and not (dt == F16F8 and (consumer or K < 128))
            and Y.stride(0) % 8 == 0 and Y.data_ptr() % 16 == 0 and n_store % 8 == 0
            and A.stride(0) * 512 < (1 << 31) and W.stride(0) * 512 < (1 << 31) and mask_bits_words(A.shape[0], N) * 4 < (1 << 31))


# the environment switch (A/B runs, tools/probes/wgrad_fold_ab.sh): "0" = fp32 atomics everywhere, "wide" = the fold for the wide layers only (round 5's
# policy), anything else = the policy below
WGRAD_FOLD = __import__("os").environ.get("SNERF_WGRAD_FOLD", "1") != "0"
WGRAD_FOLD_WIDE_ONLY = __import__("os").environ.get("SNERF_WGRAD_FOLD", "1") == "wide"
WGRAD_FOLD_SMALL_M = __import__("os").environ.get("SNERF_WGRAD_FOLD_SMALL_M", "1") != "0"      # (A/B switch of the small-M rule below)


def wgrad_uses_fold(M, N, K, dt, variant, n_valid=None, k_valid=None):
    """the default cross-slice reduction of linear_wgrad: True = partial tiles + fixed-order fold (see there), False = fp32 atomics.
    A launch runs ~256 (256 x 256 kernel) or ~1024 (128 x 128 kernel) workgroups whatever M is, and every one of them ends with one fp32 atomic per
    valid element of its tile: 16.8 M atomics per launch on full tiles, all issued at the same moment -- 25-70 us per launch that the partial tiles +
    fold replace by 10-25.  So: the wide layers (>= 8 tiles of the 256 x 256 kernel: any M, round 5) and, up to 2^20 rows, its few-tile launches and
    the 128 x 128 kernel's launches with up to four output tiles of >= 8 valid rows (more tiles -- N = 128, K = 1051; N = 1024, K = 96 -- stream
    enough operand bytes per atomic to hide them: measured neutral; a head's 1-3 valid rows issue few atomics).  Beyond 2^20 rows a workgroup's slice
    is long enough for the bursts to drift apart: path C's 2.1 M-row launches are neutral, path B's 8.4 M-row launches LOSE 0.4 ms per step to the
    partial tiles + folds (alternating A/B).  profiles/r6_z_wgrad_fold_narrow_ab.txt"""
    if not WGRAD_FOLD or dt not in (BF16, F16) or M < 4096:
        return False
    few_ok = not WGRAD_FOLD_WIDE_ONLY and M <= (1 << 20)
    if (variant & 2) and N % 256 == 0 and K >= 256:                   # tn_plan's condition for the 256 x 256 kernel
        return (N // 256) * ((K + 255) // 256) >= 8 or few_ok
    if not few_ok:
        return False
    nv = N if n_valid is None else n_valid
    tiles = ((N + 127) // 128) * ((K + 127) // 128)
    # (small M: the 16.8 M atomics are most of a many-tile launch too -- N = 128, K = 1051 and N = 1024, K = 96 of the 512-ray step)
    return nv >= 8 and (tiles <= 4 or (WGRAD_FOLD_SMALL_M and M <= 131072 and tiles <= 16))


def linear_wgrad(dZ, X, dW, n_valid, k_valid, dt, variant=0, deterministic=False, x_split_hi=False):
    """dW[:n_valid, :k_valid] += dZ^T @ X.  dZ [M,N], X [M,K] views, dW fp32 view.  The M slices of a launch either add with fp32 atomics (order varies
    run to run) or store partial tiles into a workspace that a second launch folds in slice order (bit-reproducible): `deterministic`
    forces the fold; it is also the default for the wide layers (below).  `x_split_hi` (dt bf16): X is an activation a split-bf16 forward
    saved ([M, 2 K] interleaved hi / lo) and its hi half is multiplied."""
    _chk2d(dZ, _TORCH_DT[dt]); _chk2d(X, _TORCH_DT[dt]); _chk2d(dW, torch.float32)
    assert dZ.shape[0] == X.shape[0] and dW.shape[0] >= n_valid and dW.shape[1] >= k_valid
    Kx = X.shape[1]
    if x_split_hi:
        assert dt in (BF16, F16) and Kx % 128 == 0
        Kx //= 2
        variant |= 1 << 14
    if dt == BF16X3:
        # dZ / X are the interleaved split operands (physical widths, multiples of 128); the kernels fold the hi / lo combinations
        assert not deterministic, "the split-bf16 weight gradient has no deterministic fold"
        assert dZ.shape[1] % 128 == 0 and X.shape[1] % 128 == 0
    # wide layers on the 256 x 256 kernel (>= 8 output tiles, i.e. <= 32 M slices): the slices store their partial tiles and a second
    # launch folds them in slice order instead of adding 256 x 256 fp32 atomics per slice -- bit-reproducible, and faster: the atomics
    # of 16 slices (16.8 M per launch at N = K = 1024, whatever M is) cost 24-71 us, the stores + fold 7-26 (tools/probes/
    # tn_epilogue_probe.py); 512-ray step 4.55 -> 4.24 ms, 4096-ray step 25.65 -> 25.47 (A/B on one box, tools/probes/wgrad_fold_ab.sh)
    if not deterministic and wgrad_uses_fold(dZ.shape[0], dZ.shape[1], Kx, dt, variant, n_valid, k_valid):
        deterministic = True
    if deterministic:
        nws = _lib.query("snerf_linear_wgrad_ws_floats", dZ.shape[0], dZ.shape[1], Kx, dZ.stride(0), X.stride(0), dt, variant)
        ws = torch.empty(max(int(nws), 1), dtype=torch.float32, device=dZ.device)
        _lib.call("snerf_linear_wgrad_det", _p(dZ), dZ.stride(0), _p(X), X.stride(0), _p(dW), dW.stride(0), _p(zero_page(dZ.device)),
                  dZ.shape[0], dZ.shape[1], Kx, n_valid, k_valid, dt, variant, _p(ws), ws.numel(), _stream())
        return
    _lib.call("snerf_linear_wgrad", _p(dZ), dZ.stride(0), _p(X), X.stride(0), _p(dW), dW.stride(0),
              _p(zero_page(dZ.device)), dZ.shape[0], dZ.shape[1], Kx, n_valid, k_valid, dt, variant, _stream())


# ------------------------------------------------------------ fused MLPs ----
def fmlp_classic_fwd(E, VE, stream, bias, raw):
    """The whole classic NeRF 8 x 256 network in one launch (csrc/fmlp.hip): E [M,>=64] / VE [M,>=32] bf16 embeddings, `stream` /
    `bias` from mlp.fmlp_pack -> raw [M,4] fp32."""
    _chk2d(E, torch.bfloat16); _chk2d(VE, torch.bfloat16); _chk2d(raw, torch.float32)
    assert stream.dtype == torch.bfloat16 and stream.is_contiguous() and bias.dtype == torch.float32 and raw.is_contiguous() and raw.shape[1] == 4
    assert E.shape[1] >= 64 and VE.shape[1] >= 32 and VE.shape[0] == E.shape[0] == raw.shape[0]
    _lib.call("snerf_fmlp_classic_fwd", _p(E), E.stride(0), _p(VE), VE.stride(0), _p(stream), stream.shape[0], _p(bias), bias.numel() // 32,
              _p(raw), E.shape[0], _stream())


def fmlp_classic_pts_fwd(pts, viewdirs, S, stream, bias, raw):
    """fmlp_classic_fwd with the embeddings computed in the kernel: pts [M,3] fp32, viewdirs [M/S,3] fp32 -> raw [M,4] fp32."""
    _f32c(pts); _chk2d(raw, torch.float32)
    assert viewdirs.dtype == torch.float32 and viewdirs.stride(1) == 1 and viewdirs.shape[1] == 3 and pts.shape[0] == viewdirs.shape[0] * S
    assert stream.dtype == torch.bfloat16 and stream.is_contiguous() and bias.dtype == torch.float32 and raw.is_contiguous() and raw.shape == (pts.shape[0], 4)
    _lib.call("snerf_fmlp_classic_pts_fwd", _p(pts), _p(viewdirs), viewdirs.stride(0), int(S), _p(stream), stream.shape[0], _p(bias),
              bias.numel() // 32, _p(raw), pts.shape[0], _stream())


def fmlp_proposal_fwd(E, stream, bias, raw_density):
    """The proposal MLP 96 -> 4 x 256 -> 1 in one launch: E [M,>=96] bf16 IPE rows -> raw density [M,1] fp32."""
    _chk2d(E, torch.bfloat16)
    assert stream.dtype == torch.bfloat16 and stream.is_contiguous() and bias.dtype == torch.float32 and E.shape[1] >= 96
    assert raw_density.dtype == torch.float32 and raw_density.is_contiguous() and raw_density.numel() == E.shape[0]
    _lib.call("snerf_fmlp_proposal_fwd", _p(E), E.stride(0), _p(stream), stream.shape[0], _p(bias), bias.numel() // 32, _p(raw_density),
              E.shape[0], _stream())


def _act_arrays(acts, bits, M, widths):
    import ctypes
    assert len(acts) == len(widths)
    for y, w in zip(acts, widths):
        assert y.dtype == torch.bfloat16 and y.dim() == 2 and y.stride(1) == 1 and y.shape[0] == M and y.shape[1] >= w
        assert y.data_ptr() % 16 == 0 and y.stride(0) % 8 == 0
    for i, b in enumerate(bits):
        assert b.dtype == torch.int32 and b.is_contiguous() and b.numel() >= mask_bits_words(M, 128 if i == 8 else 256)
    ptrs = (ctypes.c_void_p * len(acts))(*[y.data_ptr() for y in acts])
    lds = (ctypes.c_long * len(acts))(*[y.stride(0) for y in acts])
    bp = (ctypes.c_void_p * len(bits))(*[b.data_ptr() for b in bits])
    return ptrs, lds, bp


def fmlp_classic_train_fwd(E, VE, stream, bias, raw, acts, bits):
    """fmlp_classic_fwd that also stores the ten hidden-layer outputs (`acts`: pts_linears.0..7 [M,256], feature [M,256], views
    [M,128], bf16 row-major views) and the ReLU bit masks of the eight trunk layers (`bits[0..7]`: int32 [mask_bits_words(M, 256)]
    each) and of views_linears.0 (`bits[8]`: [mask_bits_words(M, 128)]) for the backward pass."""
    import ctypes
    _chk2d(E, torch.bfloat16); _chk2d(VE, torch.bfloat16); _chk2d(raw, torch.float32)
    assert stream.dtype == torch.bfloat16 and stream.is_contiguous() and bias.dtype == torch.float32 and raw.is_contiguous() and raw.shape[1] == 4
    assert E.shape[1] >= 64 and VE.shape[1] >= 32 and VE.shape[0] == E.shape[0] == raw.shape[0] and len(bits) == 9
    ptrs, lds, bp = _act_arrays(acts, bits, E.shape[0], [256] * 9 + [128])
    _lib.call("snerf_fmlp_classic_train_fwd", _p(E), E.stride(0), _p(VE), VE.stride(0), _p(stream), stream.shape[0], _p(bias),
              bias.numel() // 32, _p(raw), ctypes.addressof(ptrs), ctypes.addressof(lds), ctypes.addressof(bp), E.shape[0], _stream())


def fmlp_proposal_train_fwd(E, stream, bias, raw_density, acts, bits):
    """fmlp_proposal_fwd that also stores the four hidden-layer outputs (`acts`: [M,256] bf16 each) and their ReLU bit masks."""
    import ctypes
    _chk2d(E, torch.bfloat16)
    assert stream.dtype == torch.bfloat16 and stream.is_contiguous() and bias.dtype == torch.float32 and E.shape[1] >= 96
    assert raw_density.dtype == torch.float32 and raw_density.is_contiguous() and raw_density.numel() == E.shape[0] and len(bits) == 4
    ptrs, lds, bp = _act_arrays(acts, bits, E.shape[0], [256] * 4)
    _lib.call("snerf_fmlp_proposal_train_fwd", _p(E), E.stride(0), _p(stream), stream.shape[0], _p(bias), bias.numel() // 32,
              _p(raw_density), ctypes.addressof(ptrs), ctypes.addressof(lds), ctypes.addressof(bp), E.shape[0], _stream())


def fmlp_zip_fwd(Fb, D, stream, bias, raw_rgb, raw_d, x32=None):
    """NeRF MLP of the zipnerf path at inference in ONE launch (csrc/fmlp.hip, fzip_fwd_kernel): Fb [M, >= 64] grid features, D [M, >= 16]
    direction encoding (bf16 or fp16 = the stream's dtype, zero padded) -> raw_rgb [M, 3], raw_d [M, 1] fp32; x32 (optional [M, >= 32],
    compute dtype): the first 32 channels of the bottleneck x (the semantic logits are x[:, 1:1+C])."""
    _chk2d(Fb, Fb.dtype); _chk2d(D, Fb.dtype); _chk2d(raw_rgb, torch.float32); _chk2d(raw_d, torch.float32)
    M = Fb.shape[0]
    assert Fb.dtype in (torch.bfloat16, torch.float16) and stream.dtype == Fb.dtype and stream.is_contiguous() and bias.dtype == torch.float32
    assert D.shape[0] == M and raw_rgb.shape[0] == M and raw_d.shape[0] == M and Fb.shape[1] >= 64 and D.shape[1] >= 16 and raw_rgb.shape[1] >= 3
    if x32 is not None:
        _chk2d(x32, Fb.dtype)
        assert x32.shape[0] == M and x32.shape[1] >= 32
    _lib.call("snerf_fmlp_zip_fwd", _p(Fb), Fb.stride(0), _p(D), D.stride(0), _p(stream), stream.shape[0], _p(bias), bias.numel() // 32,
              _p(raw_rgb), raw_rgb.stride(0), _p(raw_d), raw_d.stride(0), _p(x32), 0 if x32 is None else x32.stride(0), M, _zip_dt(Fb), _stream())


def fmlp_zip_train_fwd(Fb, D, stream, bias, raw_rgb, raw_d, acts, bits):
    """The training forward of the zipnerf NeRF MLP in ONE launch (fzip_fwd_kernel<.., STORE>): as fmlp_zip_fwd, plus acts = [H1 [M, >= 64],
    x, h, H3 [M, >= 256]] (compute dtype 2-D views, written) and bits = 3 x int32: the ReLU bit masks of H1 [mask_bits_words(M, 64)], h and H3
    [mask_bits_words(M, 256)]."""
    import ctypes
    _chk2d(Fb, Fb.dtype); _chk2d(D, Fb.dtype); _chk2d(raw_rgb, torch.float32); _chk2d(raw_d, torch.float32)
    M = Fb.shape[0]
    assert Fb.dtype in (torch.bfloat16, torch.float16) and stream.dtype == Fb.dtype and stream.is_contiguous() and bias.dtype == torch.float32
    assert D.shape[0] == M and raw_rgb.shape[0] == M and raw_d.shape[0] == M and Fb.shape[1] >= 64 and D.shape[1] >= 16 and len(acts) == 4 and len(bits) == 3
    for i, y in enumerate(acts):
        _chk2d(y, Fb.dtype)
        assert y.shape[0] == M and y.shape[1] >= (64 if i == 0 else 256)
    for i, b in enumerate(bits):
        assert b.dtype == torch.int32 and b.is_contiguous() and b.numel() >= mask_bits_words(M, 64 if i == 0 else 256)
    pa = (ctypes.c_void_p * 4)(*[y.data_ptr() for y in acts])
    pl = (ctypes.c_long * 4)(*[y.stride(0) for y in acts])
    pb = (ctypes.c_void_p * 3)(*[b.data_ptr() for b in bits])
    _lib.call("snerf_fmlp_zip_train_fwd", _p(Fb), Fb.stride(0), _p(D), D.stride(0), _p(stream), stream.shape[0], _p(bias), bias.numel() // 32,
              _p(raw_rgb), raw_rgb.stride(0), _p(raw_d), raw_d.stride(0), ctypes.addressof(pa), ctypes.addressof(pl), ctypes.addressof(pb), M,
              _zip_dt(Fb), _stream())


def fmlp_zip_chain_bwd(d_rgb, d_den, stream, bits, dz, g_bias):
    """Data-gradient chain of the zipnerf NeRF MLP in ONE launch (fzip_chain_bwd_kernel): d_rgb [M,3], d_den [M, 1 + C <= 32] fp32 -> dz = [dH3, dh,
    dx ([M, >= 256]), dH1, dF ([M, >= 64])] (compute dtype views, written); bits = the forward's masks of H1, h, H3; the bias gradients of
    lin_second_stage_1 / _0, density_layer.2 / .0 are added to g_bias[0..3] (not bit-reproducible)."""
    import ctypes
    _chk2d(d_rgb, torch.float32); _chk2d(d_den, torch.float32)
    M = d_rgb.shape[0]
    assert d_den.shape[0] == M and 1 <= d_den.shape[1] <= 32 and stream.dtype in (torch.bfloat16, torch.float16) and stream.is_contiguous()
    assert len(bits) == 3 and len(dz) == 5 and len(g_bias) == 4
    for i, b in enumerate(bits):
        assert b.dtype == torch.int32 and b.is_contiguous() and b.numel() >= mask_bits_words(M, 64 if i == 0 else 256)
    for i, y in enumerate(dz):
        _chk2d(y, stream.dtype)
        assert y.shape[0] == M and y.shape[1] >= (256 if i < 3 else 64)
    for gb, w in zip(g_bias, (256, 256, 256, 64)):
        assert gb.dtype == torch.float32 and gb.is_contiguous() and gb.numel() == w
    nws = _lib.query("snerf_fmlp_zip_chain_ws_floats", M)
    ws = torch.empty(max(int(nws), 1), dtype=torch.float32, device=d_rgb.device)
    pb = (ctypes.c_void_p * 3)(*[b.data_ptr() for b in bits])
    pz = (ctypes.c_void_p * 5)(*[y.data_ptr() for y in dz])
    pl = (ctypes.c_long * 5)(*[y.stride(0) for y in dz])
    pg = (ctypes.c_void_p * 4)(*[g.data_ptr() for g in g_bias])
    _lib.call("snerf_fmlp_zip_chain_bwd", _p(d_rgb), d_rgb.stride(0), _p(d_den), d_den.stride(0), d_den.shape[1], _p(stream), stream.shape[0],
              ctypes.addressof(pb), ctypes.addressof(pz), ctypes.addressof(pl), ctypes.addressof(pg), _p(ws), ws.numel(), M, _zip_dt(stream), _stream())


def fcolour_fwd(CB, stream, bias, raw_rgb, acts=None, bits=None, variant=0):
    """Fused colour head of the mip path's NeRF MLP (csrc/fmlp.hip): CB [M, >= 1056] bf16 = [bottleneck 1024 | view encoding 27 | 0]
    -> raw_rgb [M,3] fp32 in ONE launch.  Training: `acts` = 3 x [M, >= 128] bf16 (outputs of cond_layers.0..2), `bits` = 3 x int32
    [mask_bits_words(M, 128)] (their ReLU bit masks), stored for fcolour_bwd and the weight-gradient GEMMs."""
    import ctypes
    _chk2d(CB, torch.bfloat16); _chk2d(raw_rgb, torch.float32)
    M = CB.shape[0]
    assert stream.dtype == torch.bfloat16 and stream.is_contiguous() and bias.dtype == torch.float32 and CB.shape[1] >= 1056
    assert raw_rgb.is_contiguous() and raw_rgb.shape == (M, 3)
    pa = pl = pb = None
    if acts is not None:
        assert len(acts) == 3 and len(bits) == 3
        for y, b in zip(acts, bits):
            assert y.dtype == torch.bfloat16 and y.dim() == 2 and y.stride(1) == 1 and y.shape[0] == M and y.shape[1] >= 128
            assert b.dtype == torch.int32 and b.is_contiguous() and b.numel() >= mask_bits_words(M, 128)
        pa = (ctypes.c_void_p * 3)(*[y.data_ptr() for y in acts])
        pl = (ctypes.c_long * 3)(*[y.stride(0) for y in acts])
        pb = (ctypes.c_void_p * 3)(*[b.data_ptr() for b in bits])
    _lib.call("snerf_fcolour_fwd", _p(CB), CB.stride(0), _p(stream), stream.shape[0], _p(bias), bias.numel() // 32, _p(raw_rgb),
              None if pa is None else ctypes.addressof(pa), None if pl is None else ctypes.addressof(pl),
              None if pb is None else ctypes.addressof(pb), M, int(variant), _stream())


def fcolour_bwd(d_raw_rgb, stream, bits, dC, dB, g_bias):
    """Fused data-gradient chain of the colour head: d_raw_rgb [M,3] fp32 -> dC = [dC2, dC1, dC0] ([M, >= 128] bf16 each) and dB
    ([M, >= 1024] bf16 view); bits = ReLU bit masks of [cond_layers.2, .1, .0, bottleneck]; the bias gradients are added to
    g_bias = [cond_layers.2, .1, .0, bottleneck] (fp32 views of the gradient arena)."""
    import ctypes
    _f32c(d_raw_rgb); _chk2d(dB, torch.bfloat16)
    M = d_raw_rgb.shape[0]
    assert d_raw_rgb.shape[1] == 3 and stream.dtype == torch.bfloat16 and stream.is_contiguous() and len(bits) == 4 and len(dC) == 3 and len(g_bias) == 4
    assert dB.shape[0] == M and dB.shape[1] >= 1024
    for b, n in zip(bits, (128, 128, 128, 1024)):
        assert b.dtype == torch.int32 and b.is_contiguous() and b.numel() >= mask_bits_words(M, n)
    for y in dC:
        assert y.dtype == torch.bfloat16 and y.dim() == 2 and y.stride(1) == 1 and y.shape[0] == M and y.shape[1] >= 128
    for gb, n in zip(g_bias, (128, 128, 128, 1024)):
        assert gb.dtype == torch.float32 and gb.is_contiguous() and gb.numel() == n
    nws = _lib.query("snerf_fcolour_bwd_ws_floats", M)
    ws = torch.empty(max(int(nws), 1), dtype=torch.float32, device=dB.device)
    pb = (ctypes.c_void_p * 4)(*[b.data_ptr() for b in bits])
    pc = (ctypes.c_void_p * 3)(*[y.data_ptr() for y in dC])
    pl = (ctypes.c_long * 3)(*[y.stride(0) for y in dC])
    pg = (ctypes.c_void_p * 4)(*[g.data_ptr() for g in g_bias])
    _lib.call("snerf_fcolour_bwd", _p(d_raw_rgb), _p(stream), stream.shape[0], ctypes.addressof(pb), ctypes.addressof(pc), ctypes.addressof(pl),
              _p(dB), dB.stride(0), ctypes.addressof(pg), _p(ws), ws.numel(), M, _stream())


CHAIN_CLASSIC, CHAIN_PROPOSAL = 0, 1


def fchain_bwd(net, d_raw, stream, bits, dz, g_bias):
    """Fused data-gradient chain of a 256-wide network (csrc/fmlp.hip fchain_bwd_kernel), ONE launch instead of one GEMM per layer.
    CHAIN_CLASSIC: d_raw [M,4] fp32 -> dz = [d views_linears.0 (128), d feature_linear (256), d pts_linears.7 .. .0 (256)]; bits =
    masks of pts_linears.0..7 + views_linears.0.  CHAIN_PROPOSAL: d_raw [M] / [M,1] (d raw density) -> dz = [d layers.3 .. .0]; bits =
    masks of layers.0..3.  The bias gradient of step i is added to g_bias[i] (not bit-reproducible: workgroup-level LDS atomics)."""
    import ctypes
    classic = net == CHAIN_CLASSIC
    d_raw = _f32c(d_raw)
    M = d_raw.shape[0]
    widths = ([128] + [256] * 9) if classic else [256] * 4
    assert d_raw.numel() == M * (4 if classic else 1) and stream.dtype == torch.bfloat16 and stream.is_contiguous()
    assert len(bits) == (9 if classic else 4) and len(dz) == len(widths) == len(g_bias)
    for i, b in enumerate(bits):
        assert b.dtype == torch.int32 and b.is_contiguous() and b.numel() >= mask_bits_words(M, 128 if i == 8 else 256)
    for y, w in zip(dz, widths):
        assert y.dtype == torch.bfloat16 and y.dim() == 2 and y.stride(1) == 1 and y.shape[0] == M and y.shape[1] >= w
    for gb, w in zip(g_bias, widths):
        assert gb.dtype == torch.float32 and gb.is_contiguous() and gb.numel() == w
    nws = _lib.query("snerf_fchain_bwd_ws_floats", net, M)
    ws = torch.empty(max(int(nws), 1), dtype=torch.float32, device=d_raw.device)
    pb = (ctypes.c_void_p * len(bits))(*[b.data_ptr() for b in bits])
    pz = (ctypes.c_void_p * len(dz))(*[y.data_ptr() for y in dz])
    pl = (ctypes.c_long * len(dz))(*[y.stride(0) for y in dz])
    pg = (ctypes.c_void_p * len(dz))(*[g.data_ptr() for g in g_bias])
    _lib.call("snerf_fchain_bwd", net, _p(d_raw), _p(stream), stream.shape[0], ctypes.addressof(pb), ctypes.addressof(pz), ctypes.addressof(pl),
              ctypes.addressof(pg), _p(ws), ws.numel(), M, _stream())


# --------------------------------------------------------------- encoders ----
def classic_embed(pts, viewdirs, S, L, Lv, dst1, dst2, w_pts, dstv, w_views, dt):
    pts = _f32c(pts); M = pts.shape[0]
    assert pts.shape[1] == 3
    if dt in SPLIT_DTS:
        # the exact fp32 embeddings, then the hi / lo (fp16 + fp8) split into each destination's GEMM operand layout
        t1 = torch.empty(M, w_pts, dtype=torch.float32, device=pts.device)
        tv = None if viewdirs is None else torch.empty(M, w_views, dtype=torch.float32, device=pts.device)
        classic_embed(pts, viewdirs, S, L, Lv, t1, None, w_pts, tv, w_views, F32)
        cast_pad(t1, w_pts, dst1, w_pts, dt)
        if dst2 is not None:
            cast_pad(t1, w_pts, dst2, w_pts, dt)
        if tv is not None:
            cast_pad(tv, w_views, dstv, w_views, dt)
        return
    if viewdirs is not None:
        assert viewdirs.dtype == torch.float32 and viewdirs.stride(1) == 1 and viewdirs.shape[1] == 3
        assert M == viewdirs.shape[0] * S
    _lib.call("snerf_classic_embed", _p(pts), _p(viewdirs), 0 if viewdirs is None else viewdirs.stride(0), S, M, L, Lv,
              _p(dst1), dst1.stride(0), _p(dst2), 0 if dst2 is None else dst2.stride(0), w_pts,
              _p(dstv), 0 if dstv is None else dstv.stride(0), w_views, dt, _stream())


def _ids(sample_id):
    if sample_id is None:
        return None, 0
    assert sample_id.dtype == torch.int32 and sample_id.is_contiguous() and sample_id.is_cuda
    return sample_id, sample_id.shape[0]


def mip_encode(s_vals, origins, directions, radii, near, far, cone, transform_idx, max_deg, dst1, dst2, width, dt,
               means_out=None, covs_out=None, sample_id=None, warp=None):
    """cone: ray shape (True / 1 = cone, False / 0 = cylinder); + 2 = --disable_integration (models.py:132-133: the covariances are replaced
    by zeros before the encoding).  sample_id (int32 [rows], optional): encode only the samples ray * S + i listed there, row j of dst <- sample_id[j].
    warp: None = the contraction (model argument fn = 1), or (viewc [3] host floats, far_max device scalar) = the view-centred warp
    fn = 0 (mip.py:367-378)."""
    n, P = s_vals.shape
    for t in (s_vals, origins, directions, radii, near, far):
        _f32c(t)
    ids, rows = _ids(sample_id)
    if dt in SPLIT_DTS:
        # the exact fp32 encoding, then the hi / lo split into the GEMM operand layout
        assert dst2 is None
        tmp = torch.empty(dst1.shape[0], width, dtype=torch.float32, device=dst1.device)
        mip_encode(s_vals, origins, directions, radii, near, far, cone, transform_idx, max_deg, tmp, None, width, F32, means_out, covs_out, sample_id, warp)
        return cast_pad(tmp, width, dst1, width, dt)
    if warp is not None:
        (vx, vy, vz), far_max = warp
        _f32c(far_max)
        assert far_max.numel() == 1 and far_max.device == s_vals.device
        return _lib.call("snerf_mip_encode_warp", _p(s_vals), _p(origins), _p(directions), _p(radii), _p(near), _p(far), n, P - 1,
                         int(cone), transform_idx, max_deg, _p(dst1), dst1.stride(0), _p(dst2),
                         0 if dst2 is None else dst2.stride(0), width, _p(_f32c(means_out)), _p(_f32c(covs_out)), dt, _p(ids), rows,
                         0, float(vx), float(vy), float(vz), _p(far_max), _stream())
    _lib.call("snerf_mip_encode", _p(s_vals), _p(origins), _p(directions), _p(radii), _p(near), _p(far), n, P - 1,
              int(cone), transform_idx, max_deg, _p(dst1), dst1.stride(0), _p(dst2),
              0 if dst2 is None else dst2.stride(0), width, _p(_f32c(means_out)), _p(_f32c(covs_out)), dt, _p(ids), rows, _stream())


def mip_viewenc(viewdirs, S, deg, dst, width, dt, sample_id=None):
    _f32c(viewdirs)
    ids, rows = _ids(sample_id)
    if dt in SPLIT_DTS:
        tmp = torch.empty(dst.shape[0], width, dtype=torch.float32, device=dst.device)
        mip_viewenc(viewdirs, S, deg, tmp, width, F32, sample_id)
        return cast_pad(tmp, width, dst, width, dt)
    _lib.call("snerf_mip_viewenc", _p(viewdirs), viewdirs.shape[0], S, deg, _p(dst), dst.stride(0), width, dt, _p(ids), rows, _stream())


class _IndexCheckSlots:
    """Per-device ring of range-check slots: one int32 device counter, one pinned host mirror and one event per slot, all allocated
    once.  A check zeroes its counter, runs snerf_index_check, copies the counter into the pinned mirror (non_blocking) and records the
    event behind the copy; a poll reads the HOST value of slots whose event has completed -- no device synchronisation, no
    allocation per call (`cnt.item()` was a memcpy + stream sync behind the step's queued kernels)."""
    SLOTS = 16

    def __init__(self, device):
        self.dev = torch.zeros(self.SLOTS, dtype=torch.int32, device=device)
        self.host = torch.zeros(self.SLOTS, dtype=torch.int32).pin_memory()
        self.events = [torch.cuda.Event() for _ in range(self.SLOTS)]
        self.msgs = [None] * self.SLOTS            # None = free
        self.head = 0

    def poll(self, block=False):
        for k in range(self.SLOTS):
            if self.msgs[k] is None:
                continue
            if block:
                self.events[k].synchronize()
            elif not self.events[k].query():
                continue
            msg, self.msgs[k] = self.msgs[k], None
            if int(self.host[k]) != 0:
                raise IndexError(msg)

    def acquire(self):
        for _ in range(self.SLOTS):
            k, self.head = self.head, (self.head + 1) % self.SLOTS
            if self.msgs[k] is None:
                return k
        k = self.head                                # every slot pending (16 checks in flight): wait for the oldest one
        self.events[k].synchronize()
        self.poll()
        return self.acquire()


_index_slots = {}            # device index -> _IndexCheckSlots


def poll_index_checks(block=False):
    """Raise IndexError for an embedding index outside its table seen by an EARLIER launch (torch.nn.Embedding raises on one; the
    kernels clamp).  Without `block` only checks whose result has already arrived in pinned host memory are read: the step never waits
    for the check and never synchronises the device."""
    if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
        return
    for slots in _index_slots.values():
        slots.poll(block)


def index_check(idx, n_rows, what):
    """asynchronous range check of a float index vector against a table of n_rows rows; reported by a later poll_index_checks()"""
    poll_index_checks()
    if torch.cuda.is_current_stream_capturing():
        return                                           # (a captured step replays with fresh data nobody checks: validate before capturing)
    di = idx.device.index if idx.device.index is not None else torch.cuda.current_device()
    slots = _index_slots.get(di)
    if slots is None:
        slots = _index_slots[di] = _IndexCheckSlots(idx.device)
    k = slots.acquire()
    cnt = slots.dev[k:k + 1]
    cnt.zero_()
    _lib.call("snerf_index_check", _p(idx), idx.numel(), int(n_rows), _p(cnt), _stream())
    slots.host[k:k + 1].copy_(cnt, non_blocking=True)
    slots.events[k].record()
    slots.msgs[k] = f"{what}: index out of range for a table of {n_rows} rows (torch.nn.Embedding raises here; the kernel clamped)"


def app_embed(emb, app, S, dst, dt, sample_id=None, check=True):
    """appearance embedding rows into the condition block: dst[ray * S + i, :dim] = emb[int(app[ray])] (models.py:153-159).  `check`:
    indices outside the table are reported (IndexError, as nn.Embedding would) by a later call of this module -- see index_check."""
    _f32c(emb); _f32c(app)
    if check:
        index_check(app, emb.shape[0], "embedding lookup")
    ids, rows = _ids(sample_id)
    _lib.call("snerf_app_embed", _p(emb), _p(app), emb.shape[0], app.numel(), int(S), emb.shape[1], _p(dst), dst.stride(0), dt, _p(ids), rows, _stream())


def app_embed_bwd(dV, app, S, g_emb, deterministic=False):
    """g_emb[int(app[ray])] += sum over the ray's S samples of dV[ray * S + i, :dim] (fp32); `deterministic`: in a fixed order (no atomics)"""
    _chk2d(dV, torch.float32); _f32c(app); _f32c(g_emb)
    assert dV.shape[0] == app.numel() * S and dV.shape[1] >= g_emb.shape[1]
    _lib.call("snerf_app_embed_bwd_det" if deterministic else "snerf_app_embed_bwd", _p(dV), dV.stride(0), _p(app), g_emb.shape[0], app.numel(), int(S),
              g_emb.shape[1], _p(g_emb), _stream())


def ert_compact(s0, w0, s1, eps_t, eps_w):
    """Early ray termination + sample compaction from the proposal histogram (inference): -> (row_index int32 [N,S1] with -1 for
    skipped samples, sample_id int32 [rows] of the kept samples in ray-major order).  One device->host sync for the row count."""
    n, S0, S1 = s0.shape[0], s0.shape[1] - 1, s1.shape[1] - 1
    dev = s0.device
    for t in (s0, w0, s1):
        _f32c(t)
    masks = torch.empty(n, (S1 + 63) // 64, dtype=torch.int64, device=dev)
    counts = torch.empty(n, dtype=torch.int32, device=dev)
    row_index = torch.empty(n, S1, dtype=torch.int32, device=dev)
    sample_id = torch.empty(n * S1, dtype=torch.int32, device=dev)
    total = torch.zeros(1, dtype=torch.int64, device=dev)
    _lib.call("snerf_ert_compact", _p(s0), _p(w0), _p(s1), n, S0, S1, float(eps_t), float(eps_w), _p(masks), _p(counts), _p(row_index),
              _p(sample_id), _p(total), _stream())
    return row_index, sample_id[:int(total.item())]


# --------------------------------------------------------------- samplers ----
def _u_arg(u, n, nf):
    _f32c(u)
    if u.dim() == 1:
        assert u.shape[0] == nf
        return u, 0
    assert u.shape == (n, nf)
    return u, nf


def classic_sample_pdf(bins, weights, u, mid_mode, want_inds=False, want_std=False):
    """mid_mode True: bins = z_vals [N,S], weights [N,S] (the kernel uses mids and weights[:,1:-1]);
    mid_mode False: bins [N,nb], weights [N,nb-1]."""
    _f32c(bins); _f32c(weights)
    n = bins.shape[0]
    if mid_mode:
        assert weights.shape == bins.shape
        nc, wptr = bins.shape[1] - 1, weights.data_ptr() + 4
    else:
        assert weights.shape[1] == bins.shape[1] - 1
        nc, wptr = bins.shape[1], weights.data_ptr()
    nf = u.shape[-1]
    u, us = _u_arg(u, n, nf)
    dev = bins.device
    out = torch.empty(n, nf, dtype=torch.float32, device=dev)
    inds = torch.empty(n, nf, dtype=torch.int32, device=dev) if want_inds else None
    std = torch.empty(n, dtype=torch.float32, device=dev) if want_std else None
    _lib.call("snerf_classic_sample_pdf", _p(bins), bins.stride(0), 1 if mid_mode else 0, wptr, weights.stride(0), nc,
              _p(u), us, n, nf, _p(out), _p(inds), _p(std), _stream())
    return out, inds, std


def classic_points(rays, z_vals):
    assert rays.is_cuda and rays.dtype == torch.float32 and rays.stride(1) == 1
    _f32c(z_vals)
    n, S = z_vals.shape
    pts = torch.empty(n, S, 3, dtype=torch.float32, device=z_vals.device)
    _lib.call("snerf_classic_points", _p(rays), rays.stride(0), _p(z_vals), n, S, _p(pts), _stream())
    return pts


def classic_merge_sort(a, b):
    _f32c(a); _f32c(b)
    n = a.shape[0]
    out = torch.empty(n, a.shape[1] + b.shape[1], dtype=torch.float32, device=a.device)
    _lib.call("snerf_classic_merge_sort", _p(a), a.shape[1], _p(b), b.shape[1], n, _p(out), _stream())
    return out


def mip_resample(s_vals, weights, u, resample_padding=0.01, want_idx=False):
    _f32c(s_vals); _f32c(weights)
    n, P = s_vals.shape
    nf = u.shape[-1]
    u, us = _u_arg(u, n, nf)
    out = torch.empty(n, nf, dtype=torch.float32, device=s_vals.device)
    idx = torch.empty(n, nf, dtype=torch.int32, device=s_vals.device) if want_idx else None
    _lib.call("snerf_mip_resample", _p(s_vals), _p(weights), _p(u), us, n, P - 1, nf, float(resample_padding), _p(out), _p(idx), _stream())
    return out, idx


def jitter_u(jit, s):
    """math_ops.py:50-54: min(arange(P) * s + jit, 1 - eps) IN PLACE over the uniform draw jit [N, P] (one launch, the eager expression's bits)"""
    _f32c(jit)
    _lib.call("snerf_jitter_u", _p(jit), jit.shape[0], jit.shape[1], float(s), _stream())
    return jit


def stratified(base, rnd, near, far, n, mode, lindisp=False):
    """mode 0: classic z-values from near/far [N] views (stride in elements); mode 1: mip s-values."""
    _f32c(base); _f32c(rnd)
    P = base.shape[0]
    out = torch.empty(n, P, dtype=torch.float32, device=base.device)
    nfs = 0
    if mode == 0:
        assert near.dtype == torch.float32 and far.dtype == torch.float32 and near.stride(0) == far.stride(0)
        nfs = near.stride(0)
    _lib.call("snerf_stratified", _p(base), _p(rnd), _p(near), _p(far), nfs, n, P, mode, 1 if lindisp else 0, _p(out), _stream())
    return out


# ------------------------------------------------------------- compositing ----
def mip_composite_fwd(raw_rgb, raw_density, noise, s_vals, dirs, near, far, transform_idx, white, rgb_padding, density_bias,
                      row_index=None):
    """row_index (int32 [n,S], optional): raw_* hold compacted rows; sample (ray, i) reads row row_index[ray, i], -1 = empty."""
    n, P = s_vals.shape
    S = P - 1
    dev = s_vals.device
    comp = torch.empty(n, 3, dtype=torch.float32, device=dev) if raw_rgb is not None else None
    dist = torch.empty(n, dtype=torch.float32, device=dev)
    acc = torch.empty(n, dtype=torch.float32, device=dev)
    w = torch.empty(n, S, dtype=torch.float32, device=dev)
    _lib.call("snerf_mip_composite_fwd", _p(raw_rgb), 0 if raw_rgb is None else raw_rgb.stride(0), _p(raw_density),
              raw_density.stride(0), _p(_f32c(noise)), _p(_f32c(s_vals)), _p(_f32c(dirs)), _p(_f32c(near)), _p(_f32c(far)), n, S,
              transform_idx, 1 if white else 0, float(rgb_padding), float(density_bias), _p(comp), _p(dist), _p(acc), _p(w),
              _p(row_index), _stream())
    return comp, dist, acc, w


def mip_composite_bwd(raw_rgb, raw_density, noise, s_vals, dirs, near, far, transform_idx, white, rgb_padding, density_bias,
                      weights, distance, g_rgb, g_dist, g_acc, g_w, d_raw_rgb, d_raw_density, g_dirs=None):
    """`g_dirs` (optional fp32 [n,3], written): d loss / d directions through the interval lengths (t1 - t0) * |d|."""
    n, P = s_vals.shape
    for t in (g_rgb, g_dist, g_acc, g_w, weights, distance, g_dirs):
        _f32c(t)
    _lib.call("snerf_mip_composite_bwd", _p(raw_rgb), 0 if raw_rgb is None else raw_rgb.stride(0), _p(raw_density),
              raw_density.stride(0), _p(noise), _p(s_vals), _p(dirs), _p(near), _p(far), n, P - 1, transform_idx,
              1 if white else 0, float(rgb_padding), float(density_bias), _p(weights), _p(distance), _p(g_rgb), _p(g_dist),
              _p(g_acc), _p(g_w), _p(d_raw_rgb), 0 if d_raw_rgb is None else d_raw_rgb.stride(0), _p(d_raw_density),
              d_raw_density.stride(0), _p(g_dirs), _stream())


def mip_encode_bwd(s_vals, origins, directions, radii, near, far, cone, transform_idx, max_deg, dE, warp=None):
    """d loss / d (origins, directions) from the IPE feature gradients dE fp32 [n*S, >= 6*max_deg] (pose refinement).  `warp` = the
    forward's ((vx, vy, vz), far_max): the view-centred warp of an fn = 0 model (snerf_mip_encode_warp_bwd)."""
    n, P = s_vals.shape
    _f32c(s_vals); _chk2d(dE, torch.float32)
    assert dE.shape[0] == n * (P - 1)
    g_o = torch.empty(n, 3, dtype=torch.float32, device=s_vals.device)
    g_d = torch.empty_like(g_o)
    if warp is not None:
        (vx, vy, vz), far_max = warp
        _f32c(far_max)
        _lib.call("snerf_mip_encode_warp_bwd", _p(s_vals), _p(origins), _p(directions), _p(radii), _p(near), _p(far), n, P - 1, int(cone),
                  int(transform_idx), int(max_deg), _p(dE), dE.stride(0), _p(g_o), _p(g_d), 0, float(vx), float(vy), float(vz), _p(far_max), _stream())
        return g_o, g_d
    _lib.call("snerf_mip_encode_bwd", _p(s_vals), _p(origins), _p(directions), _p(radii), _p(near), _p(far), n, P - 1, int(cone),
              int(transform_idx), int(max_deg), _p(dE), dE.stride(0), _p(g_o), _p(g_d), _stream())
    return g_o, g_d


def mip_viewenc_bwd(viewdirs, S, deg, dV):
    """d loss / d viewdirs from the view-encoding feature gradients dV fp32 [n*S, >= 3 + 6*deg]."""
    n = viewdirs.shape[0]
    _f32c(viewdirs); _chk2d(dV, torch.float32)
    assert dV.shape[0] == n * S
    g = torch.empty(n, 3, dtype=torch.float32, device=viewdirs.device)
    _lib.call("snerf_mip_viewenc_bwd", _p(viewdirs), n, int(S), int(deg), _p(dV), dV.stride(0), _p(g), _stream())
    return g


def classic_composite_fwd(raw, noise, z_vals, rays_d, white):
    _chk2d(raw, torch.float32); _f32c(z_vals); _f32c(noise)
    n, S = z_vals.shape
    dev = z_vals.device
    assert rays_d.dtype == torch.float32 and rays_d.stride(1) == 1
    rgb = torch.empty(n, 3, dtype=torch.float32, device=dev)
    disp, acc, depth = (torch.empty(n, dtype=torch.float32, device=dev) for _ in range(3))
    w = torch.empty(n, S, dtype=torch.float32, device=dev)
    _lib.call("snerf_classic_composite_fwd", _p(raw), raw.stride(0), _p(noise), _p(z_vals), _p(rays_d), rays_d.stride(0), n, S,
              1 if white else 0, _p(rgb), _p(disp), _p(acc), _p(w), _p(depth), _stream())
    return rgb, disp, acc, w, depth


def classic_composite_bwd(raw, noise, z_vals, rays_d, white, weights, acc, depth, g_rgb, g_disp, g_acc, g_depth, g_w, d_raw):
    n, S = z_vals.shape
    for t in (weights, acc, depth, g_rgb, g_disp, g_acc, g_depth, g_w):
        _f32c(t)
    _lib.call("snerf_classic_composite_bwd", _p(raw), raw.stride(0), _p(noise), _p(z_vals), _p(rays_d), rays_d.stride(0), n, S,
              1 if white else 0, _p(weights), _p(acc), _p(depth), _p(g_rgb), _p(g_disp), _p(g_acc), _p(g_depth), _p(g_w),
              _p(d_raw), d_raw.stride(0), _stream())


# ------------------------------------------------------------ training tail ----
NONFINITE = {"keep": 0, "zero": 1, "nan_to_num": 2}


def adam_step(p, g, m, v, lr, b1, b2, eps, step, grad_scale=1.0, zero_grad=True, nonfinite="zero", grad_max_val=0.0, clip_coef=None,
              step_dev=None, lr_dev=None, dropped=None):
    """Fused Adam over a flat arena (snerf_adam_step_ex).  Gradient hygiene in the reference's order (train_utils.py:234-243):
    `clip_coef` (device float, see grad_clip_coef) -> value clip `grad_max_val` -> `nonfinite` policy ("zero": NaN/Inf gradients are
    dropped so that they can never poison m, v or the parameters; "nan_to_num": torch's nan_to_num_ exactly; "keep": plain Adam).
    `step_dev` (int32 [1] on the device, incremented by the launch) / `lr_dev` (fp32 [1]) replace the host scalars: graph-capturable."""
    for t in (p, g, m, v):
        _f32c(t)
    if step_dev is not None:
        assert step_dev.dtype == torch.int32 and step_dev.is_cuda and step_dev.numel() == 1
    if lr_dev is not None:
        assert lr_dev.dtype == torch.float32 and lr_dev.is_cuda and lr_dev.numel() == 1
    if clip_coef is not None:
        assert clip_coef.dtype == torch.float32 and clip_coef.is_cuda
    if dropped is not None:     # int64 [1] on the device: += the number of NaN / +-Inf gradient elements of this launch (never reset here)
        assert dropped.dtype == torch.int64 and dropped.is_cuda and dropped.numel() == 1
    _lib.call("snerf_adam_step_cnt", _p(p), _p(g), _p(m), _p(v), p.numel(), float(lr), float(b1), float(b2), float(eps), int(step),
              _p(step_dev), _p(lr_dev), float(grad_scale), 1 if zero_grad else 0, NONFINITE[nonfinite], float(grad_max_val), _p(clip_coef),
              _p(dropped), _stream())


def adam_step_dev(p, g, m, v, lr, b1, b2, eps, step_dev, grad_scale=1.0, zero_grad=True, **kw):
    """adam_step with the step counter in device memory (int32 [1], incremented by the launch): graph-capturable.  (The C-ABI also keeps
    the plain entries snerf_adam_step / snerf_adam_step_dev = snerf_adam_step_ex without the hygiene options.)"""
    adam_step(p, g, m, v, lr, b1, b2, eps, 0, grad_scale, zero_grad, step_dev=step_dev, **kw)


def grad_clip_coef(g, grad_scale, max_norm):
    """-> fp32 [2] on the device = (min(1, max_norm / (|grad_scale| ||g|| + 1e-6)), the norm): clip_grad_norm_'s coefficient for adam_step."""
    _f32c(g)
    ws = torch.empty(1024, dtype=torch.float64, device=g.device)
    out = torch.empty(2, dtype=torch.float32, device=g.device)
    _lib.call("snerf_grad_clip_coef", _p(g), g.numel(), float(grad_scale), float(max_norm), _p(ws), _p(out), _stream())
    return out


def nonfinite_flag(g, flag):
    """flag (int32 [1] on the device, zeroed by the caller) |= 1 when any fp32 element of `g` is NaN / +-Inf: the found-inf pass of a
    dynamic loss scaler (snerf_nonfinite_flag).  No host sync."""
    _f32c(g)
    assert flag.dtype == torch.int32 and flag.is_cuda and flag.numel() >= 1
    _lib.call("snerf_nonfinite_flag", _p(g), g.numel(), _p(flag), _stream())


def colsum_f32(x, C, out, deterministic=False):
    _chk2d(x, torch.float32)
    _lib.call("snerf_colsum_f32_det" if deterministic else "snerf_colsum_f32", _p(x), x.stride(0), x.shape[0], C, _p(out), _stream())


def split_cast(src, C, dst, Cpad):
    """fp32 [M, >= C] -> split-bf16 rows in the GEMMs' interleaved layout: dst [M, >= 2 Cpad] bf16, Cpad % 64 == 0 logical columns
    (hi at physical [128 j, 128 j + 64), lo at [128 j + 64, 128 j + 128) of logical columns [64 j, 64 j + 64); columns >= C zero)."""
    _chk2d(src, torch.float32); _chk2d(dst, torch.bfloat16)
    assert dst.shape[0] == src.shape[0] and dst.shape[1] >= 2 * Cpad and src.shape[1] >= C
    _lib.call("snerf_split_cast", _p(src), src.stride(0), src.shape[0], C, Cpad, _p(dst), dst.stride(0), _stream())


def split8_cast(src, C, dst, Cpad, weight=False):
    """fp32 [M, >= C] -> rows in the fp16 + fp8 split layout (dtype F16F8; csrc/gemm.hip, GemmNT::split == 2): dst [M, >= 2 Cpad] fp16-typed, per 64
    logical columns [fp16(x) x 64 | 128 e4m3 bytes]: the residual x - fp16(x) scaled by 2^13 and x scaled by 2^2 (activations), or -- `weight` --
    w scaled by 2^9 and its residual scaled by 2^20; columns >= C zero."""
    _chk2d(src, torch.float32); _chk2d(dst, torch.float16)
    assert dst.shape[0] == src.shape[0] and dst.shape[1] >= 2 * Cpad and src.shape[1] >= C
    _lib.call("snerf_split8_cast", _p(src), src.stride(0), src.shape[0], C, Cpad, _p(dst), dst.stride(0), 1 if weight else 0, _stream())


def cast_pad(src, C, dst, Cpad, dt):
    if dt == BF16X3:
        return split_cast(src, C, dst, Cpad)
    if dt == F16F8:
        return split8_cast(src, C, dst, Cpad)
    _chk2d(src, torch.float32)
    _lib.call("snerf_cast_pad", _p(src), src.stride(0), src.shape[0], C, Cpad, _p(dst), dst.stride(0), dt, _stream())


def gather_pack(flat, idx, dst, tiles=None):
    """dst[i] = flat[idx[i]] (idx -1 -> 0, -2 -> 1) rounded to dst's dtype: all packed operands of a network in one launch.  `tiles` (int32
    [n, 4] = dst_off, src_base, src_stride, dst_ld: 16 x 64 transposed tiles, their elements marked idx = -3): snerf_gather_pack_tiles."""
    assert flat.dtype == torch.float32 and flat.is_contiguous() and idx.dtype == torch.int32 and idx.is_contiguous() and dst.is_contiguous()
    assert dst.numel() == idx.numel() and dst.dtype in (torch.float32, torch.bfloat16, torch.float16)
    if tiles is None or tiles.numel() == 0:
        _lib.call("snerf_gather_pack", _p(flat), _p(idx), idx.numel(), _p(dst), _zip_dt(dst), _stream())
        return
    assert tiles.dtype == torch.int32 and tiles.is_contiguous() and tiles.dim() == 2 and tiles.shape[1] == 4
    _lib.call("snerf_gather_pack_tiles", _p(flat), _p(idx), idx.numel(), _p(dst), _zip_dt(dst), _p(tiles), tiles.shape[0], _stream())


def gather_pack_pair(flat, idx, dst, tiles, idx32, dst32):
    """gather_pack of a 16-bit pool (with its transposed tiles, or None) and of the network's fp32 pool in ONE launch (snerf_gather_pack_pair)"""
    assert flat.dtype == torch.float32 and flat.is_contiguous() and idx.dtype == torch.int32 and idx.is_contiguous() and dst.is_contiguous()
    assert dst.numel() == idx.numel() and dst.dtype in (torch.bfloat16, torch.float16)
    assert idx32.dtype == torch.int32 and idx32.is_contiguous() and dst32.dtype == torch.float32 and dst32.is_contiguous() and dst32.numel() == idx32.numel()
    nt = 0 if tiles is None else tiles.shape[0]
    assert tiles is None or (tiles.dtype == torch.int32 and tiles.is_contiguous() and tiles.dim() == 2 and tiles.shape[1] == 4)
    _lib.call("snerf_gather_pack_pair", _p(flat), _p(idx), idx.numel(), _p(dst), _zip_dt(dst), _p(tiles), nt, _p(idx32), idx32.numel(), _p(dst32), _stream())


# ------------------------------------------------------------ hash grid ----
def grid_per_level_scale(base_resolution, desired_resolution, num_levels):
    """growth factor that puts level num_levels - 1 at desired_resolution (gridencoder/grid.py:104-106)"""
    import numpy as np
    return float(np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1)))


def grid_level_layout(input_dim, num_levels, per_level_scale, base_resolution, log2_hashmap_size, align_corners=False):
    """Row layout of a multiresolution hash table (gridencoder/grid.py:122-141), all levels at once: level l has
    ceil(base * scale^l) (+ 1 unless align_corners) grid points per axis and min(2^log2T, points^D) rows rounded up to a multiple of 8.
    -> (offsets int32 [L + 1], grid_sizes int32 [L]); shared by the drop-in GridEncoder and the zipnerf Model's fused encoders."""
    import numpy as np
    lv = np.arange(num_levels, dtype=np.float64)
    sizes = np.ceil(base_resolution * np.asarray(per_level_scale, dtype=np.float64) ** lv).astype(np.int64) + (0 if align_corners else 1)
    dense = np.array([int(r) ** input_dim for r in sizes], dtype=object)             # (python ints: 8193^5 does not fit int64)
    rows = np.array([-(-min(int(d), 2 ** log2_hashmap_size) // 8) * 8 for d in dense], dtype=np.int64)
    offsets = np.concatenate([[0], np.cumsum(rows)])
    if offsets[-1] >= 2 ** 31:
        raise ValueError("hash table rows exceed int32 offsets")
    return offsets.astype(np.int32), sizes.astype(np.int32)


_GRID_DT = {torch.float32: F32, torch.float16: F16, torch.float64: F64}    # the reference's three instantiations (gridencoder.cu:469)


def grid_encode_fwd(inputs, embeddings, offsets, L, S, H, gridtype, align_corners, interp, want_dy_dx=False, level_major=False, reference_form=None):
    """inputs fp32 [B,D] in [0,1]; embeddings [sO,C] fp32/fp16 -> outputs [B, L*C] (or [L,B,C] if level_major) in the
    table dtype (+ dy_dx [B, L*D*C]).  reference_form (default: not GRID_FAST): kernel_grid's one-thread-per-(point, level) form for every
    instantiation (snerf_grid_encode_fwd_ref) -- the A/B partner of the measurement legs; same bits."""
    _f32c(inputs)
    assert embeddings.is_cuda and embeddings.is_contiguous() and embeddings.dtype in _GRID_DT
    assert offsets.dtype == torch.int32 and offsets.is_cuda
    B, D = inputs.shape
    C = embeddings.shape[1]
    dt = _GRID_DT[embeddings.dtype]
    if level_major:
        out = torch.empty(L, B, C, dtype=embeddings.dtype, device=inputs.device); sl, sb = B * C, C
    else:
        out = torch.empty(B, L * C, dtype=embeddings.dtype, device=inputs.device); sl, sb = C, L * C
    dy_dx = torch.empty(B, L * D * C, dtype=embeddings.dtype, device=inputs.device) if want_dy_dx else None
    ref = (not GRID_FAST) if reference_form is None else bool(reference_form)
    _lib.call("snerf_grid_encode_fwd_ref" if ref else "snerf_grid_encode_fwd", _p(inputs), _p(embeddings), _p(offsets), _p(out), B, D, C, L, float(S), int(H), _p(dy_dx),
              int(gridtype), 1 if align_corners else 0, int(interp), dt, sl, sb, _stream())
    return out, dy_dx


def grid_encode_bwd(grad, inputs, embeddings, offsets, L, S, H, gridtype, align_corners, interp, dy_dx=None, level_major=False):
    """grad [B, L*C] (or [L,B,C]) in the table dtype -> (grad_embeddings like embeddings, grad_inputs [B,D] or None)."""
    _f32c(inputs)
    assert grad.is_contiguous() and grad.dtype == embeddings.dtype
    B, D = inputs.shape
    C = embeddings.shape[1]
    dt = _GRID_DT[embeddings.dtype]
    sl, sb = (B * C, C) if level_major else (C, L * C)
    g_emb = torch.zeros_like(embeddings)
    g_in = torch.zeros(B, D, dtype=embeddings.dtype, device=inputs.device) if dy_dx is not None else None
    _lib.call("snerf_grid_encode_bwd", _p(grad), _p(inputs), _p(embeddings), _p(offsets), _p(g_emb), B, D, C, L, float(S), int(H),
              _p(dy_dx), _p(g_in), int(gridtype), 1 if align_corners else 0, int(interp), dt, sl, sb, _stream())
    return g_emb, g_in


def grid_fast_ok(D, C, gridtype, align_corners, interp, dtype, want_dy_dx=False):
    """the instantiations that run on the pair-loading gather / binned table gradient (csrc/zip.hip g3_*): D = 3, hash, linear -- what
    zipnerf constructs (internal/models.py:413-421: C = 4 / 1), at every channel count.  Everything else -- and SNERF_GRID_FAST=0 -- runs the one-thread-per-(point, level) kernels of csrc/grid.hip."""
    return (GRID_FAST and D == 3 and C in (1, 2, 4, 8) and int(gridtype) == 0 and not align_corners and int(interp) == 0 and not want_dy_dx
            and dtype in (torch.float32, torch.float16))


GRID_FAST = _os.environ.get("SNERF_GRID_FAST", "1") != "0"          # (python-side default of the A/B choice; the library itself holds no switch)


def grid_set_fast_path(on: bool):
    """default form of the stand-alone encoder for callers that do not say (bench.py's grid_encoder leg times both in one process):
    picks the ENTRY POINT grid_encode_fwd calls and whether GridEncoder.backward takes the binned table gradient.  Returns the previous value."""
    global GRID_FAST
    prev = GRID_FAST
    GRID_FAST = bool(on)
    return prev


def grid_host_offsets(offsets):
    """host copy (numpy int32) of a GridEncoder's device `offsets` buffer -- the bin plan of the binned table gradient is made on the
    host.  The copy rides on the tensor OBJECT (with the tensor's version), so a module's buffer is copied once; a foreign or rebuilt
    tensor costs one device->host copy per call.  (Never keyed by address: a freed buffer's address can come back with other contents.)"""
    c = getattr(offsets, "_snerf_host_offsets", None)
    if c is not None and c[0] == offsets._version:
        return c[1]
    import numpy as np
    h = np.ascontiguousarray(offsets.detach().cpu().numpy().astype(np.int32))
    try:
        offsets._snerf_host_offsets = (offsets._version, h)
    except AttributeError:
        pass
    return h


GRID_BWD_WS_BYTES = int(_os.environ.get("SNERF_GRID_BWD_WS", "0"))        # workspace of the binned table gradient; 0: the library's recommendation (<= 1 GiB)


def grid_encode_bwd_binned(grad, inputs, offsets, C, L, S, H, out_dtype=torch.float32, half_records=None, level_major=False, offsets_host=None, ws_bytes=None):
    """Table gradient of the stand-alone GridEncoder without atomics (snerf_grid_encode_bwd_binned; D = 3, hash, linear, C in {1, 2, 4, 8}):
    grad [B, L*C] (or [L,B,C]) fp32 / fp16 -> grad_embeddings [rows, C] in `out_dtype`, bit-reproducible.  `half_records` (default:
    follows the gradient's dtype): contributions travel as fp16 -- the reference adds __half2 atomics into a half table's gradient.
    `ws_bytes`: size of the workspace the call may use (default: snerf_grid_encode_bwd_binned_ws_bytes, at most 1 GiB at any B); the
    call chunks the points to fit it."""
    _f32c(inputs)
    assert grad.is_contiguous() and grad.dtype in (torch.float32, torch.float16) and out_dtype in (torch.float32, torch.float16) and C in (1, 2, 4, 8)
    B = inputs.shape[0]
    oh = grid_host_offsets(offsets) if offsets_host is None else offsets_host
    if half_records is None:
        half_records = grad.dtype == torch.float16 and C in (1, 4)        # (C = 2 / 8 travel as fp32 records)
    g_emb = torch.zeros(int(oh[-1]), C, dtype=out_dtype, device=inputs.device)
    if B == 0:
        return g_emb
    nb = _lib.query("snerf_grid_encode_bwd_binned_ws_bytes", B, C, L, oh.ctypes.data, 1 if half_records else 0)
    if nb < 0:
        raise ValueError("grid_encode_bwd_binned: level layout outside the binned kernels' range (more than 1024 row ranges per level)")
    want = int(ws_bytes if ws_bytes is not None else (GRID_BWD_WS_BYTES or nb))
    ws = _zb_workspace(inputs.device, "g3ws", (want + 255) // 256 * 256, torch.uint8)
    sl, sb = (B * C, C) if level_major else (C, L * C)
    _lib.call("snerf_grid_encode_bwd_binned", _p(grad), _p(inputs), _p(offsets), oh.ctypes.data, _p(g_emb), B, C, L, float(S), int(H),
              _GRID_DT[grad.dtype], _GRID_DT[out_dtype], sl, sb, 1 if half_records else 0, _p(ws), want, _stream())
    return g_emb


def grid_encode_bwd_binned_plan(B, C, L, offsets_host, half_records, grad_dtype=torch.float16, level_major=False, ws_bytes=None):
    """what snerf_grid_encode_bwd_binned would run on a workspace of ws_bytes (default: the recommended size): dict(chunks, chunk_points,
    levels_per_transposed_group, launches, chunk_record_capacity, bytes_used, ws_bytes)"""
    import ctypes
    import numpy as np
    oh = np.ascontiguousarray(np.asarray(offsets_host, dtype=np.int32))
    if ws_bytes is None:
        ws_bytes = _lib.query("snerf_grid_encode_bwd_binned_ws_bytes", B, C, L, oh.ctypes.data, 1 if half_records else 0)
    out = (ctypes.c_long * 6)()
    _lib.call("snerf_grid_encode_bwd_binned_plan", B, C, L, oh.ctypes.data, 1 if half_records else 0, _GRID_DT[grad_dtype], 1 if level_major else 0,
              int(ws_bytes), ctypes.addressof(out))
    return dict(chunks=out[0], chunk_points=out[1], levels_per_transposed_group=out[2], launches=out[3], chunk_record_capacity=out[4], bytes_used=out[5],
                ws_bytes=int(ws_bytes))


def grid_tv_grad(inputs, embeddings, grad, offsets, weight, L, S, H, gridtype, align_corners):
    _f32c(inputs)
    assert grad.dtype == embeddings.dtype and grad.is_contiguous() and embeddings.is_contiguous()
    B, D = inputs.shape
    dt = _GRID_DT[embeddings.dtype]
    _lib.call("snerf_grid_tv_grad", _p(inputs), _p(embeddings), _p(grad), _p(offsets), float(weight), B, D, embeddings.shape[1], L,
              float(S), int(H), int(gridtype), 1 if align_corners else 0, dt, _stream())


# ------------------------------------------------------------ zipnerf path ----
def zip_resample(sdist, weights, u, n, near, far, dilation, dilate, anneal, resample_padding=0.0, lam=-1.5, dom=(0.0, 1.0)):
    _f32c(sdist); _f32c(weights); _f32c(near); _f32c(far)
    R, P0 = sdist.shape
    u, us = _u_arg(u, R, n)
    so = torch.empty(R, n + 1, dtype=torch.float32, device=sdist.device)
    to = torch.empty_like(so)
    _lib.call("snerf_zip_resample", _p(sdist), _p(weights), P0 - 1, _p(u), us, n, _p(near), _p(far), R, float(dilation), 1 if dilate else 0,
              float(anneal), float(resample_padding), float(lam), float(dom[0]), float(dom[1]), _p(so), _p(to), _stream())
    return so, to


def _zip_dt(t):
    return {torch.float32: F32, torch.bfloat16: BF16, torch.float16: F16}[t.dtype]


def round_mode(dt):
    """rounding mode of the fused proposal networks (their activations are rounded like the GEMM route's buffers of that compute dtype):
    0 none (fp32), 1 bf16, 2 fp16.  Accepts a compute dtype code or the older bool (True = bf16)."""
    if isinstance(dt, bool):
        return 1 if dt else 0
    return {F32: 0, BF16: 1, F16: 2}[int(dt)]


def zip_encode_fwd(tdist, origins, directions, radii, base_x, base_y, deg_jitter, table, offsets, grid_sizes, feat, L, C, n, m, Sl, H, std_scale,
                   levels_per_thread=0):
    for t in (tdist, origins, directions, radii, base_x, base_y, deg_jitter):
        _f32c(t)
    R, P = tdist.shape
    assert table.is_contiguous() and offsets.dtype == torch.int32 and grid_sizes.dtype == torch.int32
    _lib.call("snerf_zip_encode_fwd", _p(tdist), _p(origins), _p(directions), _p(radii), _p(base_x), _p(base_y), _p(deg_jitter), _p(table),
              _p(offsets), _p(grid_sizes), _p(feat), feat.stride(0), R, P - 1, L, C, n, m, float(Sl), int(H), float(std_scale), _zip_dt(table),
              _zip_dt(feat), int(levels_per_thread), _stream())


def zip_encode_fwd_count(tdist, origins, directions, radii, base_x, base_y, deg_jitter, table, offsets, grid_sizes, feat, L, C, n, m, Sl, H, std_scale,
                         ksplit, level_rows):
    """zip_encode_fwd (one thread per (interval, level)) that also counts the records of the binned table gradient and reserves the
    workgroups' ranges: pass 0 of zip_encode_bwd_binned in the forward's sweep.  -> (counts, wg_offsets) to hand to zip_encode_bwd_binned
    as `precounted`.  wg_offsets is a buffer of its own per call: it lives until the backward."""
    import numpy as np
    for t in (tdist, origins, directions, radii, base_x, base_y, deg_jitter):
        _f32c(t)
    R, P = tdist.shape
    S = P - 1
    assert table.is_contiguous() and offsets.dtype == torch.int32 and grid_sizes.dtype == torch.int32 and C in (1, 4) and n <= 8 and L <= 16
    ks = np.ascontiguousarray(np.asarray(ksplit, dtype=np.int32))
    lr = np.ascontiguousarray(np.asarray(level_rows, dtype=np.int32))
    assert ks.shape == (L,) and lr.shape == (L,)
    counts = torch.zeros(L, ZB_NBMAX, dtype=torch.int32, device=tdist.device)
    wgo = torch.empty(L * ((R * S + 255) // 256) * ZB_NBMAX, dtype=torch.int32, device=tdist.device)
    _lib.call("snerf_zip_encode_fwd_count", _p(tdist), _p(origins), _p(directions), _p(radii), _p(base_x), _p(base_y), _p(deg_jitter), _p(table),
              _p(offsets), _p(grid_sizes), _p(feat), feat.stride(0), R, S, L, C, n, m, float(Sl), int(H), float(std_scale), _zip_dt(table),
              _zip_dt(feat), ks.ctypes.data, lr.ctypes.data, _p(counts), _p(wgo), _stream())
    return counts, wgo


def zip_prop_mlp_fwd(F, L, w1, b1, w2, b2, round_bf16):
    """Proposal MLP of a training step in one launch: F [P, >= L] (fp32 / bf16, compact) -> raw density [P, 1] fp32."""
    _chk2d(F, F.dtype)
    for t in (w1, b1, w2, b2):
        _f32c(t)
    P, hidden = F.shape[0], b1.numel()
    assert F.dtype in (torch.float32, torch.bfloat16, torch.float16) and w1.numel() == hidden * L and w2.numel() == hidden and b2.numel() == 1 and hidden <= 64 and L <= 16
    raw = torch.empty(P, 1, dtype=torch.float32, device=F.device)
    _lib.call("snerf_zip_prop_mlp_fwd", _p(F), F.stride(0), P, int(L), _p(w1), _p(b1), _p(w2), _p(b2), hidden, round_mode(round_bf16), _zip_dt(F),
              _p(raw), _stream())
    return raw


def zip_prop_mlp_bwd(F, d_raw, L, w1, b1, w2, b2, round_bf16, g_w1, g_b1, g_w2, g_b2):
    """Backward of zip_prop_mlp_fwd: -> dF (F's shape and dtype); the parameter gradients are ADDED into g_* (fp32 views of the
    parameters' shapes), bit-reproducibly."""
    _chk2d(F, F.dtype)
    for t in (w1, b1, w2, b2, g_w1, g_b1, g_w2, g_b2):
        _f32c(t)
    P, hidden = F.shape[0], b1.numel()
    assert d_raw.dtype == torch.float32 and d_raw.is_contiguous() and d_raw.numel() == P and F.shape[1] <= 64
    assert g_w1.numel() == hidden * L and g_b1.numel() == hidden and g_w2.numel() == hidden and g_b2.numel() == 1
    dF = torch.empty(P, F.shape[1], dtype=F.dtype, device=F.device)
    nws = _lib.query("snerf_zip_prop_mlp_ws_floats", int(L), hidden, P)
    ws = torch.empty(max(int(nws), 1), dtype=torch.float32, device=F.device)
    _lib.call("snerf_zip_prop_mlp_bwd", _p(F), F.stride(0), _p(d_raw), P, int(L), _p(w1), _p(b1), _p(w2), _p(b2), hidden, round_mode(round_bf16),
              _zip_dt(F), _p(dF), dF.stride(0), _p(g_w1), _p(g_b1), _p(g_w2), _p(g_b2), _p(ws), ws.numel(), _stream())
    return dF


def zip_encode_prop_fwd(tdist, origins, directions, radii, base_x, base_y, deg_jitter, table, offsets, grid_sizes, L, n, m, Sl, H, std_scale,
                        w1, b1, w2, b2, round_bf16):
    """Fused featurisation + proposal MLP of one proposal level (inference) -> raw density [R*S, 1] fp32."""
    for t in (tdist, origins, directions, radii, base_x, base_y, deg_jitter, w1, b1, w2, b2):
        _f32c(t)
    R, P = tdist.shape
    hidden = b1.numel()
    assert table.is_contiguous() and table.shape[-1] == 1 and w1.numel() == hidden * L and w2.numel() == hidden and b2.numel() == 1
    out = torch.empty(R * (P - 1), 1, dtype=torch.float32, device=tdist.device)
    _lib.call("snerf_zip_encode_prop_fwd", _p(tdist), _p(origins), _p(directions), _p(radii), _p(base_x), _p(base_y), _p(deg_jitter), _p(table),
              _p(offsets), _p(grid_sizes), R, P - 1, L, n, m, float(Sl), int(H), float(std_scale), _zip_dt(table), _p(w1), _p(b1), _p(w2), _p(b2),
              hidden, round_mode(round_bf16), _p(out), _stream())
    return out


def zip_encode_bwd(tdist, origins, directions, radii, base_x, base_y, deg_jitter, offsets, grid_sizes, grad_feat, grad_table, L, C, n, m, Sl, H,
                   std_scale, lds_levels=0, lds_cells=0, lds_slabs=0, grad_table_bf16=None):
    R, P = tdist.shape
    assert grad_table.dtype == torch.float32 and grad_table.is_contiguous()
    if grad_table_bf16 is not None:
        assert grad_table_bf16.dtype == torch.bfloat16 and grad_table_bf16.is_contiguous() and grad_table_bf16.numel() == grad_table.numel()
    _lib.call("snerf_zip_encode_bwd", _p(tdist), _p(origins), _p(directions), _p(radii), _p(base_x), _p(base_y), _p(deg_jitter), _p(offsets),
              _p(grid_sizes), _p(grad_feat), grad_feat.stride(0), _p(grad_table), R, P - 1, L, C, n, m, float(Sl), int(H), float(std_scale),
              _zip_dt(grad_feat), int(lds_levels), int(lds_cells), int(lds_slabs), _p(grad_table_bf16), _stream())


ZB_NBMAX, ZB_TARGET = 1024, 2_000_000


def zip_bin_plan(offsets_host, C, records_per_level):
    """Host-side plan of the binned table gradient: per level the number of replicas K of every row range (levels with few, hot rows
    are split so that a bin holds ~ZB_TARGET records) and the number of leading table rows the int64 meeting image must cover."""
    br = 4096 if C == 4 else 16384
    ks, g64_rows, level_rows = [], 0, []
    for l in range(len(offsets_host) - 1):
        rows = int(offsets_host[l + 1] - offsets_host[l])
        rowbins = max((rows + br - 1) // br, 1)
        if rowbins > ZB_NBMAX:
            raise ValueError(f"binned table gradient: level {l} has {rows} rows = {rowbins} row ranges of {br}, more than the {ZB_NBMAX} bins per "
                             "level of the kernels; use table_grad_mode='atomic' for tables this large")
        k = max(1, min(-(-int(records_per_level) // (rowbins * ZB_TARGET)), ZB_NBMAX // rowbins))
        ks.append(k)
        level_rows.append(rows)
        if k > 1:
            g64_rows = int(offsets_host[l + 1])
    return ks, g64_rows, level_rows


_zb_ws = {}


def _zb_workspace(dev, key, numel, dtype):
    """the binned backward's worst-case buffers, kept across calls (one stream: a call's kernels finish before the next call reuses
    them); grown, never shrunk"""
    t = _zb_ws.get((dev, key))
    if t is None or t.numel() < numel or t.dtype != dtype:
        t = torch.empty(numel, dtype=dtype, device=dev)
        _zb_ws[(dev, key)] = t
    return t


ZIP_BIN_ALL_LEVELS = _os.environ.get("SNERF_ZIP_ALL_LEVELS", "") != ""   # probe switch: A/B runs of tools/bench_zip.py
ZIP_BIN_STAGED = _os.environ.get("SNERF_ZIP_UNSTAGED", "") == ""      # (the environment switch: A/B runs of tools/bench_zip.py)


def zip_encode_bwd_binned(tdist, origins, directions, radii, base_x, base_y, deg_jitter, offsets, grid_sizes, grad_feat, grad_table, L, C, n, m, Sl, H,
                          std_scale, ksplit, g64_rows, level_rows, precounted=None, half_records=False):
    """The table gradient of zip_encode_bwd without L2 atomics on the hashed levels and bit-reproducible: records binned by destination,
    accumulated per bin in LDS with fixed-point integer atomics (snerf_zip_encode_bwd_binned: count, scan on the device, write, accumulate).
    The fixed-point grid follows the gradient's magnitude (snerf_zip_bin_scale: 34 bits below max |grad_feat|).  `precounted` = the
    (counts, wg_offsets) pair zip_encode_fwd_count returned for the same intervals and bin plan: the count pass is then skipped.
    `half_records`: the records carry fp16 values scaled by the launch's exponent (10 / 4 instead of 18 / 8 bytes per record at C = 4 / 1):
    every contribution is rounded to 11 significant bits once, the sums stay exact and order-independent (csrc/zip.hip, ZB_HALF_SHIFT)."""
    import numpy as np
    R, P = tdist.shape
    S = P - 1
    dev = tdist.device
    assert grad_table.dtype == torch.float32 and grad_table.is_contiguous() and C in (1, 4) and L <= 16
    ks = np.ascontiguousarray(np.asarray(ksplit, dtype=np.int32))
    lr = np.ascontiguousarray(np.asarray(level_rows, dtype=np.int32))
    assert ks.shape == (L,) and lr.shape == (L,)
    scale = torch.empty(2, dtype=torch.int32, device=dev)
    _lib.call("snerf_zip_bin_scale", _p(grad_feat), grad_feat.stride(0), R * S, L * C, _zip_dt(grad_feat), _p(scale), _stream())
    args = (_p(tdist), _p(origins), _p(directions), _p(radii), _p(base_x), _p(base_y), _p(deg_jitter), _p(offsets), _p(grid_sizes), _p(grad_feat),
            grad_feat.stride(0), _p(grad_table), R, S, L, C, n, m, float(Sl), int(H), float(std_scale), _zip_dt(grad_feat), ks.ctypes.data,
            lr.ctypes.data)
    if precounted is not None:
        # the training forward (zip_encode_fwd_count) already counted the records and reserved the workgroups' ranges
        counts, wgo = precounted
        assert counts.shape == (L, ZB_NBMAX) and wgo.numel() >= L * ((R * S + 255) // 256) * ZB_NBMAX
    else:
        # pass 0 also reserves each workgroup's record range inside the bins it touches (offsets relative to the bin's start)
        counts = torch.zeros(L, ZB_NBMAX, dtype=torch.int32, device=dev)
        wgo = _zb_workspace(dev, "wgo", L * ((R * S + 255) // 256) * ZB_NBMAX, torch.int32)
        _lib.call("snerf_zip_encode_bwd_binned", 0, *args, _p(counts), _p(wgo), None, None, None, 0, None, 0, None, _stream())
    starts = torch.empty(L * ZB_NBMAX, dtype=torch.int64, device=dev)
    _lib.call("snerf_zip_bin_scan", _p(counts), _p(starts), L, _stream())                  # exclusive scan of the bin counts, on the device
    capacity = R * S * n * 8 * L                               # every (interval, level) emits at most n cells x 8 corners: no host sync
    rec_row = _zb_workspace(dev, "row", capacity if C == 4 else 1, torch.int16)          # (C = 1 records carry their row)
    rec_val = _zb_workspace(dev, "val", capacity * max(C, 2), torch.float32)       # C = 1: {row, value} pairs in one 8-byte record
    # pass 1: the records staged in LDS and written run by run; ZIP_BIN_STAGED = False (A/B probes, tests): pass 3, every thread
    # writes its records where they fall (same records, another order inside a (workgroup, bin) run)
    wpass = (5 if ZIP_BIN_STAGED else 6) if half_records else (1 if ZIP_BIN_STAGED else 3)
    if ZIP_BIN_ALL_LEVELS and ZIP_BIN_STAGED:            # (probe: the all-levels direct writer for the four-channel grid too)
        wpass = 9 if half_records else 8
    _lib.call("snerf_zip_encode_bwd_binned", wpass, *args, _p(counts), _p(wgo), _p(starts), _p(rec_row), _p(rec_val), capacity,
              None, 0, _p(scale) if half_records else None, _stream())
    g64 = torch.zeros(max(g64_rows, 1) * C, dtype=torch.int64, device=dev) if g64_rows > 0 else None
    _lib.call("snerf_zip_encode_bwd_binned", 7 if half_records else 2, *args, _p(counts), _p(wgo), _p(starts), _p(rec_row), _p(rec_val), capacity, _p(g64), int(g64_rows),
              _p(scale), _stream())


def colsum_wide_f32(x, C, out, deterministic=False):
    """out[:C] += x[:, :C].sum(0) for any C (fp32)"""
    _chk2d(x, torch.float32)
    assert out.dtype == torch.float32 and out.is_contiguous() and out.numel() >= C and x.shape[1] >= C
    _lib.call("snerf_colsum_wide_f32", _p(x), x.stride(0), x.shape[0], int(C), _p(out), 1 if deterministic else 0, _stream())


def zip_glo_modulate(X, SS, S, out):
    """out = X * exp(scale) + shift per ray (GLO modulation of the bottleneck, internal/models.py:620-630): X / out [R * S, >= B] in the
    compute dtype, SS [R, 2 B] fp32 = (scale | shift)"""
    B = SS.shape[1] // 2
    _chk2d(X, X.dtype); _chk2d(out, X.dtype); _chk2d(SS, torch.float32)
    R = SS.shape[0]
    assert X.shape[0] == R * S == out.shape[0] and X.shape[1] >= B and out.shape[1] >= B and X.dtype in (torch.float32, torch.bfloat16, torch.float16)
    _lib.call("snerf_zip_glo_modulate", _p(X), X.stride(0), _p(SS), SS.stride(0), R, S, B, _p(out), out.stride(0), _zip_dt(X),
              _stream())


def zip_glo_modulate_bwd(dXm, X, SS, d_head, S, dX):
    """-> (dSS [R, 2 B] fp32, dxsum [R, B] fp32); writes dX = dXm * exp(scale) (+ d_head in its leading columns)."""
    B = SS.shape[1] // 2
    R = SS.shape[0]
    for t in (dXm, X, dX):
        _chk2d(t, X.dtype)
        assert t.shape[0] == R * S and t.shape[1] >= B
    _chk2d(SS, torch.float32)
    nh = 0 if d_head is None else d_head.shape[1]
    if d_head is not None:
        _chk2d(d_head, torch.float32)
        assert d_head.shape[0] == R * S and nh <= B
    dSS = torch.empty(R, 2 * B, dtype=torch.float32, device=X.device)
    dxsum = torch.empty(R, B, dtype=torch.float32, device=X.device)
    _lib.call("snerf_zip_glo_modulate_bwd", _p(dXm), dXm.stride(0), _p(X), X.stride(0), _p(SS), SS.stride(0), _p(d_head), 0 if d_head is None else d_head.stride(0),
              nh, R, S, B, _p(dX), dX.stride(0), _p(dSS), dSS.stride(0), _p(dxsum), dxsum.stride(0), _zip_dt(X), _stream())
    return dSS, dxsum


def zip_composite_fwd(raw_rgb, raw_density, tdist, dirs, opaque, bg, rgb_padding, density_bias):
    _f32c(tdist); _f32c(dirs)
    R, P = tdist.shape
    dev = tdist.device
    rgb = torch.empty(R, 3, dtype=torch.float32, device=dev)
    depth, acc = torch.empty(R, dtype=torch.float32, device=dev), torch.empty(R, dtype=torch.float32, device=dev)
    w = torch.empty(R, P - 1, dtype=torch.float32, device=dev)
    _lib.call("snerf_zip_composite_fwd", _p(raw_rgb), 0 if raw_rgb is None else raw_rgb.stride(0), _p(raw_density), raw_density.stride(0),
              _p(tdist), _p(dirs), R, P - 1, 1 if opaque else 0, float(bg), float(rgb_padding), float(density_bias), _p(rgb), _p(depth), _p(acc),
              _p(w), _stream())
    return rgb, depth, acc, w


def zip_composite_bwd(raw_rgb, raw_density, tdist, dirs, opaque, bg, rgb_padding, density_bias, weights, acc, depth, g_rgb, g_depth, g_acc, g_w,
                      d_raw_rgb, d_raw_density, g_dirs=None):
    """`g_dirs` (optional fp32 [R,3], written): d loss / d directions through the interval lengths (t1 - t0) |d|."""
    R, P = tdist.shape
    for t in (g_rgb, g_depth, g_acc, g_w, weights, acc, depth, g_dirs):
        _f32c(t)
    _lib.call("snerf_zip_composite_bwd", _p(raw_rgb), 0 if raw_rgb is None else raw_rgb.stride(0), _p(raw_density), raw_density.stride(0),
              _p(tdist), _p(dirs), R, P - 1, 1 if opaque else 0, float(bg), float(rgb_padding), float(density_bias), _p(weights), _p(acc),
              _p(depth), _p(g_rgb), _p(g_depth), _p(g_acc), _p(g_w), _p(d_raw_rgb), 0 if d_raw_rgb is None else d_raw_rgb.stride(0),
              _p(d_raw_density), d_raw_density.stride(0), _p(g_dirs), _stream())


def zip_encode_ray_bwd(tdist, origins, directions, radii, base_x, base_y, deg_jitter, table, offsets, grid_sizes, grad_feat, L, C, n, m, Sl, H,
                       std_scale, g_o, g_d, g_bx, g_by):
    """Featurisation backward to the rays (cal_input_grad): accumulates d loss / d (origins, directions, base_x, base_y) [R,3]."""
    for t in (tdist, origins, directions, radii, base_x, base_y, deg_jitter, g_o, g_d, g_bx, g_by):
        _f32c(t)
    R, P = tdist.shape
    assert table.is_contiguous() and grad_feat.stride(1) == 1
    _lib.call("snerf_zip_encode_ray_bwd", _p(tdist), _p(origins), _p(directions), _p(radii), _p(base_x), _p(base_y), _p(deg_jitter), _p(table),
              _p(offsets), _p(grid_sizes), _p(grad_feat), grad_feat.stride(0), R, P - 1, L, C, n, m, float(Sl), int(H), float(std_scale),
              _zip_dt(table), _zip_dt(grad_feat), _p(g_o), _p(g_d), _p(g_bx), _p(g_by), _stream())


# ------------------------------------------------- callers ----
def pinhole_rays(coords, first_pixel, n, W, H, pose, cx, cy, fx, fy, training, near, far, device):
    """Rays of n pixels of a pinhole camera (coords int32 [n,2] = (row, col) on `device`, or None = pixels first_pixel.. row-major).
    `pose`: host [3,4] (or [4,4]) camera-to-world.  -> origins, directions, viewdirs [n,3], radii, near, far [n,1]."""
    import numpy as np
    p = np.ascontiguousarray(np.asarray(pose, dtype=np.float32)[:3, :4])
    if coords is not None:
        assert coords.dtype == torch.int32 and coords.is_contiguous() and coords.shape == (n, 2)
    o, d, v = (torch.empty(n, 3, dtype=torch.float32, device=device) for _ in range(3))
    r, nr, fr = (torch.empty(n, 1, dtype=torch.float32, device=device) for _ in range(3))
    _lib.call("snerf_pinhole_rays", _p(coords), int(first_pixel), int(W), int(H), p.ctypes.data, float(cx), float(cy), float(fx), float(fy),
              int(bool(training)), float(near), float(far), int(n), _p(o), _p(d), _p(v), _p(r), _p(nr), _p(fr), _stream())
    return o, d, v, r, nr, fr


def _host12(c2w):
    """[3,4] (or [4,4]) camera-to-world matrix -> contiguous host float32 [12] (one device->host copy per frame if it lives on the GPU)"""
    import numpy as np
    if c2w is None:
        return None
    a = c2w.detach().float().cpu().numpy() if torch.is_tensor(c2w) else np.asarray(c2w, dtype=np.float32)
    return np.ascontiguousarray(a[:3, :4], dtype=np.float32)


def classic_get_rays(H, W, focal, c2w, cx, cy, device):
    """get_rays (run_nerf_helpers.py:247-258) -> rays_o, rays_d [H,W,3] on `device`."""
    p = _host12(c2w)
    o = torch.empty(H, W, 3, dtype=torch.float32, device=device)
    d = torch.empty_like(o)
    _lib.call("snerf_classic_get_rays", int(H), int(W), float(focal), float(cx), float(cy), p.ctypes.data, _p(o), _p(d), _stream())
    return o, d


def classic_ndc_rays(H, W, focal, near, rays_o, rays_d):
    """ndc_rays (run_nerf_helpers.py:314-332) on [...,3] fp32 tensors."""
    o, d = _f32c(rays_o.contiguous()), _f32c(rays_d.contiguous())
    oo, dd = torch.empty_like(o), torch.empty_like(d)
    _lib.call("snerf_classic_ndc_rays", int(H), int(W), float(focal), float(near), _p(o), _p(d), o.numel() // 3, _p(oo), _p(dd), _stream())
    return oo, dd


def classic_ray_batch(H, W, focal, cx, cy, c2w, c2w_static, rays_o, rays_d, n, ndc, near, far, depths, use_viewdirs, device):
    """The ray-batch assembly of render() (render.py:50-77) in one launch -> rows [n, 8 (+1) (+3)] fp32."""
    p, ps = _host12(c2w), _host12(c2w_static)
    ld = 8 + (1 if depths is not None else 0) + (3 if use_viewdirs else 0)
    rows = torch.empty(n, ld, dtype=torch.float32, device=device)
    if rays_o is not None:
        rays_o, rays_d = _f32c(rays_o), _f32c(rays_d)
        assert rays_o.shape == (n, 3) and rays_d.shape == (n, 3)
    if depths is not None:
        _f32c(depths)
        assert depths.numel() == n
    _lib.call("snerf_classic_ray_batch", int(H), int(W), float(focal), float(cx), float(cy), None if p is None else p.ctypes.data,
              None if ps is None else ps.ctypes.data, _p(rays_o), _p(rays_d), int(n), 1 if ndc else 0, float(near), float(far), _p(depths),
              1 if use_viewdirs else 0, _p(rows), ld, _stream())
    return rows


def ert_f2b_step(prev_raw_d, prev_base, s1, dirs, near, far, g0, G, next_g0, next_G, transform_idx, density_bias, eps_t, state, row_base):
    """One step of the front-to-back early ray termination (snerf_ert_f2b_step): fold the evaluated group [g0, g0 + G) into the rays'
    optical depth and assign the next group to the rays still above eps_t.  `state` = (tau, flags, counts, row_index, sample_id, total)
    from ert_f2b_state.  -> sample_id[:rows] of the next group (one device->host sync for the row count)."""
    tau, flags, counts, row_index, sample_id, total = state
    n, S1 = row_index.shape
    _lib.call("snerf_ert_f2b_step", _p(prev_raw_d), 0 if prev_raw_d is None else prev_raw_d.stride(0), int(prev_base), _p(s1), _p(dirs), _p(near), _p(far),
              n, S1, int(g0), int(G), int(next_g0), int(next_G), int(transform_idx), float(density_bias), float(eps_t), _p(tau), _p(flags), _p(counts),
              _p(row_index), _p(sample_id), int(row_base), _p(total), _stream())
    return sample_id[:int(total.item())]


def classic_ert_points(rays, z_all, alive, g0, G):
    """positions [n, G, 3] and view directions [n, 3] (None for 8-column ray rows) of the rows (alive[i], g0 + k): the next group of the
    classic path's front-to-back fine pass (snerf_classic_ert_points); alive None = every ray"""
    assert rays.is_cuda and rays.dtype == torch.float32 and rays.stride(1) == 1
    _f32c(z_all)
    N, S = z_all.shape
    n = N if alive is None else alive.numel()
    pts = torch.empty(n, G, 3, dtype=torch.float32, device=rays.device)
    vd = torch.empty(n, 3, dtype=torch.float32, device=rays.device) if rays.shape[1] > 9 else None
    _lib.call("snerf_classic_ert_points", _p(rays), rays.stride(0), _p(z_all), S, _p(alive), n, int(g0), int(G), _p(pts), _p(vd), rays.shape[1] - 3, _stream())
    return pts, vd


def classic_ert_step(raw_g, alive, z_all, rays, g0, G, eps_t, T, raw_full, scratch):
    """one step of that pass (snerf_classic_ert_step): scatter the evaluated group, update the transmittances, compact the survivors.
    `scratch` = (keep, offs, alive_next int32 [N], total int64 [1]).  -> alive_next[:count] (one device->host read of the count)"""
    N, S = z_all.shape
    n = N if alive is None else alive.numel()
    keep, offs, nxt, total = scratch
    assert raw_g.dtype == torch.float32 and raw_g.is_contiguous() and raw_full.dtype == torch.float32 and raw_full.is_contiguous() and raw_g.numel() == n * G * raw_full.shape[-1]
    _lib.call("snerf_classic_ert_step", _p(raw_g), raw_full.shape[-1], _p(alive), n, _p(z_all), S, _p(rays), rays.stride(0), int(g0), int(G), float(eps_t),
              _p(T), _p(raw_full), _p(keep), _p(offs), _p(nxt), _p(total), _stream())
    return nxt[:int(total.item())].clone()


def ert_f2b_state(n, S1, group, device):
    return (torch.zeros(n, dtype=torch.float32, device=device), torch.empty(n, dtype=torch.int32, device=device),
            torch.empty(n, dtype=torch.int32, device=device), torch.full((n, S1), -1, dtype=torch.int32, device=device),
            torch.empty(n * group, dtype=torch.int32, device=device), torch.zeros(1, dtype=torch.int64, device=device))


def mip_loss_tail(rgb, tgt, dist1, dist0, tdepth, conf, s_f, w_f, s_c, w_c, disparity, depth_lambda, coarse_mult, prop_lambda):
    """-> (out[5] = {#valid, rgb, depth, proposal loss, total}, g_rgb, g_dist1, g_dist0, g_wc); absent terms return None gradients."""
    n = rgb.shape[0]
    dev = rgb.device
    out = torch.empty(5, dtype=torch.float32, device=dev)
    g_rgb = torch.empty(n, 3, dtype=torch.float32, device=dev)
    g1 = g0 = gw = None
    if tdepth is not None:
        g1, g0 = torch.empty(n, dtype=torch.float32, device=dev), torch.empty(n, dtype=torch.float32, device=dev)
    Pf = Sc = 0
    if s_c is not None:
        Pf, Sc = s_f.shape[1], w_c.shape[1]
        assert w_f.shape[1] == Pf - 1 and s_c.shape[1] == Sc + 1
        gw = torch.empty(n, Sc, dtype=torch.float32, device=dev)
    c = lambda t: None if t is None else _f32c(t)
    _lib.call("snerf_mip_loss_tail", _p(c(rgb)), _p(c(tgt)), _p(c(dist1)), _p(c(dist0)), _p(c(tdepth)), _p(c(conf)), _p(c(s_f)), _p(c(w_f)),
              _p(c(s_c)), _p(c(w_c)), n, Pf, Sc, int(bool(disparity)), float(depth_lambda), float(coarse_mult), float(prop_lambda), _p(out),
              _p(g_rgb), _p(g1), _p(g0), _p(gw), _stream())
    return out, g_rgb, g1, g0, gw


def zip_pixels_to_rays(pix_x, pix_y, cam_idx, pixtocams, camtoworlds, want_imageplane=False):
    """camera_utils.pixels_to_rays on device: int32 pixel coordinates [n] (+ int32 camera index [n] or None) and the stacked
    [ncam,3,3] inverse intrinsics / [ncam,3,4] extrinsics -> dict(origins, directions, viewdirs, radii [n,1], base_x, base_y
    (, imageplane))."""
    n, dev = pix_x.shape[0], pix_x.device
    for t in (pix_x, pix_y) + ((cam_idx,) if cam_idx is not None else ()):
        assert t.dtype == torch.int32 and t.is_contiguous() and t.shape == (n,)
    k, c = _f32c(pixtocams.contiguous()).reshape(-1, 3, 3), _f32c(camtoworlds.contiguous())   # (torch.linalg.inv returns column-major)
    assert c.shape[1:] == (3, 4) and c.shape[0] == k.shape[0]
    o, d, v, bx, by = (torch.empty(n, 3, dtype=torch.float32, device=dev) for _ in range(5))
    r = torch.empty(n, 1, dtype=torch.float32, device=dev)
    ip = torch.empty(n, 2, dtype=torch.float32, device=dev) if want_imageplane else None
    _lib.call("snerf_zip_pixels_to_rays", _p(pix_x), _p(pix_y), _p(cam_idx), _p(k), _p(c), k.shape[0], n, _p(o), _p(d), _p(v), _p(r), _p(ip),
              _p(bx), _p(by), _stream())
    out = dict(origins=o, directions=d, viewdirs=v, radii=r, base_x=bx, base_y=by)
    if ip is not None:
        out["imageplane"] = ip
    return out


ZIP_LOSS_NAMES = ("data", "mse", "depth", "d_complete", "sem", "interlevel", "distortion")


def zip_loss_tail(rgb, tgt, lossmult=None, depth=None, tdepth=None, dmask=None, cmask=None, sem=None, labels=None, smask=None, hist=None,
                  mse=False, charb_padding=0.001, data_mult=1.0, depth_lambda=0.5, com_mult=0.2, sem_mult=0.04, pulse_width=(0.03, 0.003),
                  interlevel_mult=0.01, distortion_mult=0.005):
    """One launch for the zipnerf loss tail.  `hist` = [(sdist, weights)] * 3 (proposal, proposal, NeRF) or None.
    -> (out[11] (see include/snerf_hip.h), dict of gradients: rgb, depth, semantic, w0, w1, w2 -- None where the term is off)."""
    R, dev = rgb.shape[0], rgb.device
    c = lambda t: None if t is None else _f32c(t)
    new = lambda *sh: torch.empty(*sh, dtype=torch.float32, device=dev)
    out, g_rgb = new(11), new(R, 3)
    g_depth = new(R) if depth is not None else None
    C = 0
    g_sem = None
    if sem is not None:
        C = sem.shape[1]
        assert labels.dtype == torch.int32 and labels.is_contiguous() and labels.shape == (R,)
        g_sem = new(R, C)
    s = [None] * 3; w = [None] * 3; S = [0] * 3; gw = [None] * 3
    if hist is not None:
        for i, (sd, wt) in enumerate(hist):
            if sd is None:
                continue
            s[i], w[i], S[i] = c(sd), c(wt), wt.shape[1]
            assert sd.shape == (R, S[i] + 1) and wt.shape[0] == R
        assert s[2] is not None
        if distortion_mult > 0:
            gw[2] = new(R, S[2])
        if interlevel_mult > 0:
            for i in (0, 1):
                if s[i] is not None:
                    gw[i] = new(R, S[i])
    _lib.call("snerf_zip_loss_tail", _p(c(rgb)), _p(c(tgt)), _p(c(lossmult)), _p(c(depth)), _p(c(tdepth)), _p(c(dmask)), _p(c(cmask)), _p(c(sem)),
              _p(labels), _p(c(smask)), C, _p(s[0]), _p(w[0]), S[0], _p(s[1]), _p(w[1]), S[1], _p(s[2]), _p(w[2]), S[2], R, int(bool(mse)),
              float(charb_padding), float(data_mult), float(depth_lambda), float(com_mult), float(sem_mult), float(pulse_width[0]),
              float(pulse_width[1]), float(interlevel_mult), float(distortion_mult), _p(out), _p(g_rgb), _p(g_depth), _p(g_sem), _p(gw[0]),
              _p(gw[1]), _p(gw[2]), _stream())
    return out, dict(rgb=g_rgb, depth=g_depth, semantic=g_sem, w0=gw[0], w1=gw[1], w2=gw[2])


def zip_percentiles(tdist, weights, t_far, ps=(5, 50, 95)):
    """weighted percentiles (in %) of the ray histograms extended to t_far -> [R, len(ps)] (render.py:255-267)."""
    import numpy as np
    R, S = weights.shape
    _f32c(tdist); _f32c(weights)
    tf = _f32c(t_far.reshape(-1).contiguous())
    assert tdist.shape == (R, S + 1) and tf.shape[0] == R
    p = np.ascontiguousarray(np.asarray(ps, dtype=np.float32))
    out = torch.empty(R, len(ps), dtype=torch.float32, device=weights.device)
    _lib.call("snerf_zip_percentiles", _p(tdist), _p(weights), _p(tf), R, S, p.ctypes.data, len(ps), _p(out), _stream())
    return out


def frame_quantize(rgb=None, depth=None, semantic=None, color_map=None, scale_factor=1.0):
    """Quantise rendered frame buffers for the S-NeRF++ wire format in one launch (random_render_waymo_seq.py:214-227):
    rgb [..,3] -> uint8, depth [..] -> uint16 (= depth * 256 / scale_factor), semantic [..,C] -> argmax label uint8 (+ paint uint8 [..,3]
    through `color_map` uint8 [C,3]).  -> dict with the keys of the inputs given ('rgb', 'depth', 'semantic', 'paint'), device tensors."""
    ref = rgb if rgb is not None else (depth if depth is not None else semantic)
    dev = ref.device
    out = {}
    P = (rgb.numel() // 3) if rgb is not None else (depth.numel() if depth is not None else semantic.numel() // semantic.shape[-1])
    r = d = s = cm = r8 = d16 = l8 = p8 = None
    C, ld = 0, 0
    if rgb is not None:
        r = _f32c(rgb.contiguous())
        assert r.shape[-1] == 3 and r.numel() == 3 * P
        r8 = out["rgb"] = torch.empty(r.shape, dtype=torch.uint8, device=dev)
    if depth is not None:
        d = _f32c(depth.contiguous())
        assert d.numel() == P
        d16 = out["depth"] = torch.empty(d.shape, dtype=torch.uint16, device=dev)
    if semantic is not None:
        s = _f32c(semantic.contiguous())
        C = ld = s.shape[-1]
        assert s.numel() == C * P and C <= 256
        l8 = out["semantic"] = torch.empty(s.shape[:-1], dtype=torch.uint8, device=dev)
        if color_map is not None:
            cm = color_map.contiguous()
            assert cm.dtype == torch.uint8 and cm.is_cuda and tuple(cm.shape) == (C, 3)
            p8 = out["paint"] = torch.empty(tuple(s.shape[:-1]) + (3,), dtype=torch.uint8, device=dev)
    _lib.call("snerf_frame_quantize", _p(r), _p(d), _p(s), ld, C, _p(cm), P, float(scale_factor), _p(r8), _p(d16), _p(l8), _p(p8), _stream())
    return out


def hash_decay(table, grad, offsets, L, C, mult, loss=None, grad_mult=1.0):
    """grad += grad_mult * d/d table of mult * mean_{level, channel}(mean over the level's rows of table^2) (train_utils.py:184-203);
    `loss` (1-element fp32 device tensor, optional) accumulates the value.  `grad_mult`: the trainer's loss scale (the term is linear in
    `mult`, so the kernel runs with mult * grad_mult and the value is divided back)."""
    _f32c(table); _f32c(grad)
    assert offsets.dtype == torch.int32 and offsets.is_cuda and table.shape == grad.shape
    if grad_mult != 1.0 and loss is not None:
        tmp = torch.zeros_like(loss)
        _lib.call("snerf_hash_decay", _p(table), _p(grad), _p(offsets), int(L), int(C), float(mult) * float(grad_mult), _p(tmp), _stream())
        loss.add_(tmp, alpha=1.0 / float(grad_mult))
        return
    _lib.call("snerf_hash_decay", _p(table), _p(grad), _p(offsets), int(L), int(C), float(mult) * float(grad_mult), _p(loss), _stream())


def semantic_composite_fwd(weights, logits, C, softmax, row_index=None):
    """semantic [R,C] = sum_i w[r,i] f(logits[r*S+i, :C]), f = softmax (zipnerf) or identity (live mip path); logits: 2-D view.
    `row_index` (int32 [R,S], -1 = skipped): the logits are the rows of a compacted evaluation (ert_compact)."""
    R, S = weights.shape
    _f32c(weights); _chk2d(logits)
    if row_index is not None:
        assert row_index.dtype == torch.int32 and row_index.is_contiguous() and row_index.numel() == R * S
    sem = torch.empty(R, C, dtype=torch.float32, device=weights.device)
    _lib.call("snerf_semantic_composite_fwd", _p(weights), _p(logits), logits.stride(0), _zip_dt(logits), R, S, C, int(bool(softmax)),
              _p(row_index), _p(sem), _stream())
    return sem


def semantic_composite_bwd(weights, logits, g_sem, C, softmax, d_logits, want_g_w=False):
    """d_logits (fp32 2-D view, >= C columns) <- gradient of the logits; returns d weights [R,S] when `want_g_w` (identity flavour)."""
    R, S = weights.shape
    _f32c(weights); _f32c(g_sem); _chk2d(logits); _chk2d(d_logits, torch.float32)
    g_w = torch.empty(R, S, dtype=torch.float32, device=weights.device) if want_g_w else None
    _lib.call("snerf_semantic_composite_bwd", _p(weights), _p(logits), logits.stride(0), _zip_dt(logits), _p(g_sem), R, S, C, int(bool(softmax)),
              _p(d_logits), d_logits.stride(0), _p(g_w), _stream())
    return g_w
