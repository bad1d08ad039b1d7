"""The C-ABI shared library loads without a GPU and exports every symbol include/snerf_hip.h declares."""
import ctypes
import os
import re

import pytest

from snerf_amd import _lib


def test_header_parses_and_lists_entry_points():
    protos = _lib.parse_header()
    src = open(_lib.HEADER_PATH).read()
    declared = set(re.findall(r"\b(?:int|long)\s+(snerf_\w+)\s*\(", re.sub(r"/\*.*?\*/", " ", src, flags=re.S)))
    assert declared == set(protos) and len(protos) >= 18
    for name in ("snerf_linear_fwd", "snerf_linear_wgrad", "snerf_mip_encode", "snerf_mip_resample", "snerf_classic_sample_pdf",
                 "snerf_mip_composite_fwd", "snerf_mip_composite_bwd", "snerf_classic_composite_fwd", "snerf_adam_step"):
        assert name in protos
    # no torch / C++ types in the signatures: only pointers and plain scalars
    for sig in protos.values():
        for ty, _ in sig:
            assert ty in (ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_double)


@pytest.mark.skipif(not os.path.exists(_lib.LIB_PATH), reason="libsnerf_hip.so not built (run __graft_entry__.build())")
def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    for name in _lib.parse_header():
        assert getattr(lib, name) is not None
    assert lib.snerf_version() >= 1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU/PyTorch fallback"):
        _lib.load()


def test_bad_arguments_are_rejected_without_a_gpu():
    """argument validation happens before any launch, so it can be exercised on a CPU-only box"""
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("library not built")
    with pytest.raises(_lib.SnerfHipError, match="bad argument"):
        _lib.call("snerf_linear_fwd", None, 0, None, 0, None, None, 0, None, 0, None, None, 16, 100, 64, 1, 0, 1, 0, 0, None)  # N % 128 != 0
    with pytest.raises(_lib.SnerfHipError, match="bad argument"):
        _lib.call("snerf_mip_resample", None, None, None, 0, 4, 1, 8, 0.01, None, None, None)  # S < 2
    with pytest.raises(_lib.SnerfHipError, match="bad argument"):
        _lib.call("snerf_jitter_u", None, 4, 8, 0.125, None)                                   # no buffer
    with pytest.raises(_lib.SnerfHipError, match="bad argument"):
        _lib.call("snerf_gather_pack_tiles", None, None, 16, None, 1, None, 1, None)           # no arena / map / destination
    assert _lib.call("snerf_jitter_u", None, 0, 8, 0.125, None) is None                        # an empty batch is not an error


def test_no_vector_alu_instruction_hides_in_inline_asm():
    """A VALU result needs wait states before an MFMA reads it; hipcc counts them for the instructions it emits itself but not for the
    text of an inline asm (round 2: an inline `v_pk_max_i16` in front of an MFMA gave rare, timing-dependent wrong results in one
    instantiation of the fused MLP kernel; tools/probes/mfma_war_probe.hip shows the hazard in isolation).  Inline asm in the kernels is
    therefore limited to waits, barriers, scalar / debug register reads, the transposing LDS read and (round 3) the input-row loads of
    the fused colour head (memory instructions, both waited for by hand with the destination registers as operands of the wait), the LDS
    atomic add of the fused gradient chains' bias-gradient table and empty optimisation fences."""
    import glob
    import re
    allowed = ("s_waitcnt", "s_barrier", "s_lshr_b32", "s_getreg_b32", "ds_read_b64_tr_b16", "ds_read_b32", "ds_read_b128", "ds_add_f32", "ds_add_u32", "global_load_dwordx4", "s_nop", "s_sleep", ";")
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "snerf_amd", "csrc")
    for path in sorted(glob.glob(os.path.join(root, "*.hip")) + glob.glob(os.path.join(root, "*.h"))):
        src = open(path).read()
        for m in re.finditer(r'\basm\s*(?:volatile)?\s*\(\s*((?:"[^"]*"\s*)+)', src):
            text = "".join(re.findall(r'"([^"]*)"', m.group(1)))
            for ins in re.split(r"\\n\\t|\\n|\\t", text):
                ins = ins.strip()
                if ins:
                    assert ins.startswith(allowed), f"{os.path.basename(path)}: inline asm instruction {ins!r}"


def test_grid_binned_backward_workspace_query_and_its_limits():
    """snerf_grid_encode_bwd_binned_ws_bytes / _plan are host-only: the recommended workspace is BOUNDED (at most 1 GiB at the bench's 14.7 M
    points, where the round-5 form needed 11-30 GB: at most 1 GB), smaller workspaces plan more chunks, -1 for what the binned kernels do not cover
    (C not in {1, 2, 4, 8}, a level with more than 1024 row ranges) -- GridEncoder.backward then falls back to the atomic scatter."""
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("library not built")
    import numpy as np
    import torch
    from snerf_amd import ops
    q = lambda B, C, off, half: _lib.query("snerf_grid_encode_bwd_binned_ws_bytes", B, C, len(off) - 1, np.ascontiguousarray(np.asarray(off, dtype=np.int32)).ctypes.data, half)
    off, _ = ops.grid_level_layout(3, 10, ops.grid_per_level_scale(16, 8192, 10), 16, 21)
    B = 65536 * 32 * 7
    full = q(B, 4, off, 1)
    assert 0.5e9 < full <= 1e9 and q(B, 4, off, 0) <= 1e9
    plan = ops.grid_encode_bwd_binned_plan(B, 4, 10, off, True, torch.float16)
    assert plan["bytes_used"] <= full and plan["chunks"] >= 2 and 1 <= plan["levels_per_transposed_group"] <= 10 and plan["launches"] < 400
    assert plan["chunk_record_capacity"] >= 8 * plan["chunk_points"]
    lm = ops.grid_encode_bwd_binned_plan(B, 4, 10, off, True, torch.float16, level_major=True)
    assert lm["levels_per_transposed_group"] == 0 and lm["chunks"] <= plan["chunks"]          # the reference's layout needs no transposed copy
    small = ops.grid_encode_bwd_binned_plan(B, 4, 10, off, True, torch.float16, ws_bytes=300 << 20)
    assert small["bytes_used"] <= (300 << 20) and small["chunks"] > plan["chunks"]
    few = ops.grid_encode_bwd_binned_plan(20000, 4, 10, off, True, torch.float16)
    assert few["chunks"] == 1 and few["levels_per_transposed_group"] == 10 and few["bytes_used"] < (64 << 20)
    assert q(1000, 1, off, 1) > 0 and q(0, 4, off, 1) == 0
    assert q(1000, 2, off, 0) > 0 and q(1000, 8, off, 1) > 0 and q(1000, 3, off, 0) == -1
    big = [0, 8, 8 + (1 << 23)]                                                   # 2^23 rows at C = 4: 2048 row ranges of 4096
    assert q(1000, 4, big, 1) == -1 and q(1000, 8, big, 0) == -1 and q(1000, 1, big, 1) > 0    # (C = 1: 512 ranges of 16384)
