"""Row 8f-3, the S-NeRF++ frame writer: quantisation (oracle/callers.frame_quantize pinned to the pixels decoded from the files the
reference's own save path wrote, tests/golden/g17_frame_writer.npz), the native PNG encoder (host-only: runs without a GPU; every
file is decoded again with PIL -- the reader the foreground stages use -- and must give back the exact pixels), and on the GPU
the quantisation kernel and the asynchronous FrameWriter end to end."""
import io
import os

import numpy as np
import pytest
import torch

from oracle import callers as oc
from snerf_amd import _lib

Image = pytest.importorskip("PIL.Image")
gpu = pytest.mark.gpu
needs_io = pytest.mark.skipif(not os.path.exists(_lib.IO_LIB_PATH), reason="libsnerf_io.so not built (run __graft_entry__.build())")


def _np(golden, name):
    return {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in golden(name).items()}


def test_oracle_quantisation_matches_the_reference_files(golden):
    g = _np(golden, "g17_frame_writer")
    q = oc.frame_quantize(g["rgb"], g["depth"], g["semantic"], g["color_map"], float(g["scale_factor"]))
    for k in ("rgb", "depth", "semantic", "paint"):
        assert q[k].dtype == g["png_" + k].dtype and np.array_equal(q[k], g["png_" + k]), k
    assert tuple(q["rgb"][0, 0]) == (0, 255, 0)                     # nan -> 0, +inf -> 1, -inf -> 0
    assert q["semantic"][2, 0] == 0 and q["semantic"][2, 1] == 5    # ties: first maximum
    assert q["depth"][1, 0] == 0 and q["depth"][1, 3] < 65535 and g["depth"][1, 3] * 256 / float(g["scale_factor"]) > 65536   # wrapped


def test_io_header_and_library():
    protos = _lib.parse_header(_lib.IO_HEADER_PATH)
    assert set(protos) == {"snerf_io_version", "snerf_png_write", "snerf_png_encode"}
    if os.path.exists(_lib.IO_LIB_PATH):
        lib = _lib.load_io()
        assert lib.snerf_io_version() >= 1
        for name in protos:
            assert getattr(lib, name) is not None


@needs_io
@pytest.mark.parametrize("shape,dtype", [((1, 1, 3), np.uint8), ((7, 5), np.uint8), ((33, 17, 3), np.uint8), ((64, 48), np.uint16), ((5, 9, 4), np.uint8),
                                         ((3, 3, 2), np.uint8), ((200, 300, 3), np.uint8), ((37, 41), np.uint16), ((1, 500, 3), np.uint8), ((500, 1), np.uint16)])
def test_png_encoder_round_trips_through_pil(shape, dtype):
    from snerf_amd import frame_writer as fw
    rng = np.random.default_rng(sum(shape))
    yy, xx = np.meshgrid(np.arange(shape[0]), np.arange(shape[1]), indexing="ij")
    smooth = (yy * 3 + xx * 2)[..., None] * 7 + np.arange(shape[2] if len(shape) == 3 else 1) * 50
    smooth = smooth.reshape(shape).astype(dtype)
    for a in (rng.integers(0, np.iinfo(dtype).max + 1, shape).astype(dtype), smooth, np.zeros(shape, dtype), np.full(shape, np.iinfo(dtype).max, dtype)):
        for threads in (1, 3, 8):
            for level in (0, 6):
                b = fw.png_encode(a, level=level, threads=threads)
                assert b[:8] == b"\x89PNG\r\n\x1a\n"
                im = np.array(Image.open(io.BytesIO(b)))
                assert im.shape == a.shape and im.dtype == a.dtype and np.array_equal(im, a), (shape, dtype, threads, level)


@needs_io
def test_png_writer_files_and_errors(tmp_path, golden):
    from snerf_amd import frame_writer as fw
    g = _np(golden, "g17_frame_writer")
    for k in ("rgb", "depth", "semantic", "paint"):
        p = str(tmp_path / (k + ".png"))
        fw.png_write(p, g["png_" + k])
        assert np.array_equal(np.array(Image.open(p)), g["png_" + k]), k              # = what the reference's files decode to
    fw.save_img_u8(g["rgb"], str(tmp_path / "u8.png"))
    assert np.array_equal(np.array(Image.open(str(tmp_path / "u8.png"))), g["png_rgb"])
    with pytest.raises(OSError):
        fw.png_write(str(tmp_path / "no_such_dir" / "x.png"), g["png_rgb"])
    with pytest.raises(ValueError):
        fw.png_write(str(tmp_path / "x.png"), g["rgb"])                                # float input is not a wire format
    with pytest.raises(ValueError):
        fw.png_write(str(tmp_path / "x.png"), np.zeros((4, 4, 5), np.uint8))


@gpu
def test_frame_quantize_kernel_vs_reference_files_and_oracle(golden):
    from snerf_amd import ops
    g = _np(golden, "g17_frame_writer")
    c = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    q = ops.frame_quantize(c(g["rgb"]), c(g["depth"]), c(g["semantic"]), c(g["color_map"]), float(g["scale_factor"]))
    for k in ("rgb", "depth", "semantic", "paint"):
        got = q[k].cpu().numpy()
        assert got.dtype == g["png_" + k].dtype and np.array_equal(got, g["png_" + k]), k
    # a whole Waymo frame with the awkward values sprinkled in, bit-exact against the numpy expressions
    rng = np.random.default_rng(1)
    H, W, C = 1280, 1920, 19
    rgb = rng.random((H, W, 3), dtype=np.float32) * 1.2 - 0.1
    rgb.reshape(-1)[rng.integers(0, rgb.size, 1000)] = np.nan
    rgb.reshape(-1)[rng.integers(0, rgb.size, 1000)] = np.inf
    k255 = rng.integers(0, 256, 5000).astype(np.float32) / 255                          # exact multiples of 1/255: truncation boundaries
    rgb.reshape(-1)[rng.integers(0, rgb.size, 5000)] = k255
    depth = rng.random((H, W), dtype=np.float32) * 200
    depth.reshape(-1)[rng.integers(0, depth.size, 100)] = [np.nan, np.inf, -1.0, 1e12] * 25
    sem = rng.random((H, W, C), dtype=np.float32)
    sem.reshape(-1, C)[rng.integers(0, H * W, 500), 7] = np.nan
    cmap = g["color_map"]
    ref = oc.frame_quantize(rgb, depth, sem, cmap, 0.73)
    q = ops.frame_quantize(c(rgb), c(depth), c(sem), c(cmap), 0.73)
    for k in ("rgb", "depth", "semantic", "paint"):
        assert np.array_equal(q[k].cpu().numpy(), ref[k]), (k, int((q[k].cpu().numpy() != ref[k]).sum()))
    only = ops.frame_quantize(depth=c(depth), scale_factor=0.73)
    assert set(only) == {"depth"} and np.array_equal(only["depth"].cpu().numpy(), ref["depth"])


@gpu
@needs_io
def test_frame_writer_end_to_end(tmp_path, golden):
    """FrameWriter: device buffers -> the four PNG files per frame of random_render_waymo_seq.py:214-227, written asynchronously;
    decoding them gives the pixels the reference's files decode to."""
    from snerf_amd import frame_writer as fw
    g = _np(golden, "g17_frame_writer")
    c = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    with fw.FrameWriter(str(tmp_path), scale_factor=float(g["scale_factor"]), color_map=g["color_map"], threads=4) as w:
        for idx in range(3):
            w.write(idx, dict(rgb=c(g["rgb"]), depth=c(g["depth"]), semantic=c(g["semantic"])))
        w.write(7, dict(rgb=c(g["rgb"])))                                              # a frame without depth / semantic
    for idx in range(3):
        for k in ("rgb", "depth", "semantic", "paint"):
            im = np.array(Image.open(str(tmp_path / k / f"{idx:05d}.png")))
            assert np.array_equal(im, g["png_" + k]), (idx, k)
    assert os.path.exists(str(tmp_path / "rgb" / "00007.png")) and not os.path.exists(str(tmp_path / "depth" / "00007.png"))
    # utils.save_img_u8 drop-in on a device tensor (quantised on the GPU) = the same file content as the host route
    fw.save_img_u8(c(g["rgb"]), str(tmp_path / "dev.png"))
    assert np.array_equal(np.array(Image.open(str(tmp_path / "dev.png"))), g["png_rgb"])
    # a writer error (directory removed under it) surfaces at close()
    import shutil
    w = fw.FrameWriter(str(tmp_path / "gone"), threads=2)
    shutil.rmtree(str(tmp_path / "gone"))
    w.write(0, dict(rgb=c(g["rgb"])))
    with pytest.raises(OSError):
        w.close()
