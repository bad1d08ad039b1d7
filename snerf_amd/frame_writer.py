"""Frame writer for the S-NeRF++ background renders: the files the foreground stages consume.

Reference counterpart: the tail of s-nerfpp/zipnerf/random_render_waymo_seq.py:205-227 -- per frame `rgb/<idx>.png` (uint8,
internal/utils.py:111-116 save_img_u8), `depth/<idx>.png` (uint16 = depth * 256 / scale_factor), `semantic/<idx>.png` (argmax
label, uint8) and `paint/<idx>.png` (colour-mapped labels), all through PIL -- read back by stage1_code/utils_render.py:51-73.

Here the rendered buffers are quantised on the GPU in one launch (ops.frame_quantize: 9 bytes per pixel cross PCIe instead of
(3 + 1 + C) floats), copied to pinned host memory on a side stream, and encoded by the native multi-threaded PNG writer
(include/snerf_io.h, libsnerf_io.so) on a background thread, so that the next frame renders while this one is written.  PNG is
lossless: a reader gets exactly the reference's pixels."""
import os
import queue
import threading

import numpy as np
import torch

from . import _lib, ops


def png_write(path, array, level=6, threads=8):
    """Write a uint8 / uint16 array [H,W] or [H,W,C<=4] as PNG through the native encoder (the PIL `Image.fromarray(a).save(path)`
    calls of the reference).  Raises on failure."""
    a = np.ascontiguousarray(array)
    if a.dtype not in (np.uint8, np.uint16) or a.ndim not in (2, 3) or (a.ndim == 3 and not 1 <= a.shape[2] <= 4):
        raise ValueError(f"png_write: expected uint8/uint16 [H,W] or [H,W,1..4], got {a.dtype} {a.shape}")
    h, w = a.shape[:2]
    c = 1 if a.ndim == 2 else a.shape[2]
    rc = _lib.load_io().snerf_png_write(os.fsencode(path), a.ctypes.data, w, h, c, 8 * a.itemsize, int(level), int(threads))
    if rc != 0:
        raise OSError(f"snerf_png_write({path!r}) failed: status {rc}")


def png_encode(array, level=6, threads=8) -> bytes:
    a = np.ascontiguousarray(array)
    h, w = a.shape[:2]
    c = 1 if a.ndim == 2 else a.shape[2]
    cap = int(a.nbytes * 1.01) + 16 * h + (1 << 16)
    out = np.empty(cap, np.uint8)
    n = _lib.load_io().snerf_png_encode(a.ctypes.data, w, h, c, 8 * a.itemsize, int(level), int(threads), out.ctypes.data, cap)
    if n < 0:
        raise OSError(f"snerf_png_encode failed: status {-n}")
    return out[:n].tobytes()


def save_img_u8(img, pth, level=6, threads=8):
    """internal/utils.py:111-116: an image in [0, 1] (torch tensor on any device, or numpy) -> uint8 PNG."""
    if torch.is_tensor(img) and img.is_cuda:
        png_write(pth, ops.frame_quantize(rgb=img)["rgb"].cpu().numpy(), level, threads)
    else:
        a = img.detach().cpu().numpy() if torch.is_tensor(img) else np.asarray(img)
        png_write(pth, (np.clip(np.nan_to_num(a), 0., 1.) * 255.).astype(np.uint8), level, threads)


class FrameWriter:
    """`write(idx, rendering)` returns as soon as the quantisation and the device->host copies are enqueued; the PNG files appear
    asynchronously.  `close()` (or leaving the `with` block) waits for all pending frames and re-raises a writer error."""

    def __init__(self, out_dir, scale_factor=1.0, color_map=None, level=6, threads=None, depth=2, zpad=5):
        threads = min(16, os.cpu_count() or 1) if threads is None else threads
        self.out_dir, self.scale, self.level, self.threads, self.zpad = out_dir, float(scale_factor), level, threads, zpad
        for sub in ("rgb", "depth", "semantic", "paint"):                       # random_render_waymo_seq.py:164-167
            os.makedirs(os.path.join(out_dir, sub), exist_ok=True)
        self.cmap_host = None if color_map is None else np.ascontiguousarray(np.asarray(color_map), dtype=np.uint8)
        self._cmap_dev = None
        self._q = queue.Queue(maxsize=depth)                                    # bounds the frames in flight (pinned memory)
        self._err = None
        self._stream = None
        self._t = threading.Thread(target=self._run, daemon=True)
        self._t.start()

    def _run(self):
        while True:
            item = self._q.get()
            if item is None:
                return
            idx_str, ev, host = item
            try:
                ev.synchronize()
                for k, a in host.items():
                    png_write(os.path.join(self.out_dir, k, idx_str + ".png"), a.numpy(), self.level, self.threads)
            except Exception as e:                                              # surfaced by the next write() / close()
                self._err = e

    def write(self, idx, rendering):
        """rendering: dict with 'rgb' [H,W,3] (and optionally 'depth' [H,W], 'semantic' [H,W,C]) device tensors."""
        if self._err is not None:
            raise self._err
        rgb, dep, sem = rendering.get("rgb"), rendering.get("depth"), rendering.get("semantic")
        dev = rgb.device
        if sem is not None and self.cmap_host is not None and self._cmap_dev is None:
            self._cmap_dev = torch.from_numpy(self.cmap_host).to(dev)
        q = ops.frame_quantize(rgb=rgb, depth=dep, semantic=sem, color_map=self._cmap_dev if sem is not None else None, scale_factor=self.scale)
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=dev)
        self._stream.wait_stream(torch.cuda.current_stream(dev))
        host = {}
        with torch.cuda.stream(self._stream):
            for k, t in q.items():
                h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
                h.copy_(t, non_blocking=True)
                t.record_stream(self._stream)
                host[k] = h
            ev = torch.cuda.Event()
            ev.record(self._stream)
        self._q.put((str(idx).zfill(self.zpad), ev, host))

    def close(self):
        self._q.put(None)
        self._t.join()
        if self._err is not None:
            raise self._err

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
