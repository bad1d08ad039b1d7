#!/usr/bin/env python3
"""Golden vectors for the frame writer (row 8f-3): the reference's own save path -- internal/utils.save_img_u8 for rgb, and the
inline PIL calls of s-nerfpp/zipnerf/random_render_waymo_seq.py:218-227 for depth / semantic / paint (the same expressions, evaluated
here) with its colour map (def_color_map, :26-31) -- writes PNG files from seeded float buffers; the files are decoded again and the
PIXELS are recorded with the inputs as tests/golden/g17_frame_writer.npz.  Build-container only (needs /root/reference)."""
import os
import sys
import tempfile

sys.dont_write_bytecode = True
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
if REPO not in sys.path:
    sys.path.insert(0, REPO)
from oracle import common as _oracle_common  # noqa: E402  (save_golden: writes the fixture, or compares under --check)
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)

from oracle import gen_golden_zip_callers as gz  # noqa: E402


def reference_color_map():
    """num_class / def_color_map / color_map of random_render_waymo_seq.py:25-31, taken by executing those lines of the script (the module itself cannot be
    imported: it parses flags and builds datasets at import time)."""
    src = open("/root/reference/s-nerfpp/zipnerf/random_render_waymo_seq.py").read().split("\n")
    ns = {"np": np}
    exec("\n".join(src[24:31]), ns)
    return np.asarray(ns["color_map"])


def main():
    gz.import_reference()
    from internal import utils
    from PIL import Image
    rng = np.random.default_rng(17)
    H, W, C = 23, 31, 19
    rgb = rng.random((H, W, 3), dtype=np.float32) * 1.3 - 0.15                    # values outside [0, 1] get clipped
    rgb[0, 0] = [np.nan, np.inf, -np.inf]
    rgb[0, 1] = [1.0, 0.0, 0.999999]
    rgb[0, 2] = [1 / 255, 2 / 255 - 1e-7, 254.999 / 255]
    depth = (rng.random((H, W), dtype=np.float32) * 80).astype(np.float32)
    depth[1, :4] = [0.0, 255.99, 256.0, 300.0]                                    # past 65535 after the x256: wraps
    scale_factor = 0.37
    sem = rng.random((H, W, C), dtype=np.float32)
    sem[2, 0] = 0.0                                                                # all equal: first class wins
    sem[2, 1, 5] = sem[2, 1, 11] = 2.0                                             # tie: the lower index wins
    cmap = reference_color_map()
    with tempfile.TemporaryDirectory() as td:
        p = lambda n: os.path.join(td, n)
        utils.save_img_u8(rgb, p("rgb.png"))
        dep = (depth * 256 / scale_factor).astype(np.uint16)
        Image.fromarray(dep).save(p("depth.png"))
        labels = np.argmax(sem, axis=-1)
        Image.fromarray(labels.astype(np.uint8)).save(p("semantic.png"))
        Image.fromarray(cmap[labels].astype(np.uint8)).save(p("paint.png"))
        dec = {k: np.array(Image.open(p(k + ".png"))) for k in ("rgb", "depth", "semantic", "paint")}
    _oracle_common.save_golden(os.path.join(OUT, "g17_frame_writer.npz"), rgb=rgb, depth=depth, semantic=sem, scale_factor=np.float64(scale_factor),
                        color_map=cmap.astype(np.uint8), png_rgb=dec["rgb"], png_depth=dec["depth"].astype(np.uint16), png_semantic=dec["semantic"],
                        png_paint=dec["paint"])
    print("wrote g17_frame_writer.npz", {k: (v.shape, v.dtype) for k, v in dec.items()})


if __name__ == "__main__":
    main()
