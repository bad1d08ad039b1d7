#!/usr/bin/env python3
"""Golden vectors for the foreground composite (row 8f-4) from the REFERENCE's stage-1 code (s-nerfpp/stage1_code/utils_render.py,
ip_utils.py) on seeded images -> tests/golden/g18_foreground.npz.  Build-container only (needs /root/reference).

Packages the image lacks and these functions never call (cv2, matplotlib, pyquaternion, nuscenes, trimesh, nvdiffrast) get empty
import stubs.  handle_occlusion_paste asks a CUDA ray tracer (raytracing.RayTracer, an un-vendored third-party extension) for the
mesh depth along each masked pixel's ray: here a stub `RayTracer.trace` returns the seeded per-pixel depths recorded in the
fixture (`fg_depth`) -- the depth is an INPUT of the paste logic under test -- and `Tensor.cuda()` is the identity (no GPU in the
build container).  The camera files it reads (`target_poses.npy`, `intrinsic.npy`) are written to a temporary directory."""
import os
import sys
import tempfile
import types

sys.dont_write_bytecode = True
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
if REPO not in sys.path:
    sys.path.insert(0, REPO)
from oracle import common as _oracle_common  # noqa: E402  (save_golden: writes the fixture, or compares under --check)
OUT = os.path.join(REPO, "tests", "golden")
REF = "/root/reference/s-nerfpp/stage1_code"


class _Stub(types.ModuleType):
    __path__ = []

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return type(k, (), {}) if k[0].isupper() else (lambda *a, **kw: None)


class _Mesh:
    vertices, faces = np.zeros((3, 3)), np.zeros((1, 3), int)

    def copy(self):
        return self

    def apply_transform(self, m):
        return self


def main():
    if not os.path.isdir(REF):
        raise SystemExit("reference tree not present")
    for name in ("cv2", "matplotlib", "matplotlib.pyplot", "pyquaternion", "nuscenes", "nuscenes.utils", "nuscenes.utils.geometry_utils", "trimesh",
                 "nvdiffrast", "nvdiffrast.torch", "raytracing", "raytracing.raytracing"):
        try:
            __import__(name)
        except Exception:
            sys.modules[name] = _Stub(name)
    sys.path.insert(0, REF)
    import ip_utils
    import utils_render as ur
    from PIL import Image
    rng = np.random.default_rng(18)
    H, W = 40, 56
    yy, xx = np.mgrid[:H, :W]
    blob = lambda cy, cx, ry, rx: (((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1)
    out = {}
    # ---- mask algebra: two overlapping instances
    m1 = (blob(18, 20, 9, 13)[..., None].repeat(3, -1) * 255).astype(np.uint8)
    m2 = (blob(24, 34, 8, 12)[..., None].repeat(3, -1) * 255).astype(np.uint8)
    band = lambda m, k: ((blob(18, 20, 9 + k, 13 + k) if m is m1 else blob(24, 34, 8 + k, 12 + k)) & ~(blob(18, 20, 9 - k, 13 - k) if m is m1 else blob(24, 34, 8 - k, 12 - k)))
    b1 = (band(m1, 2)[..., None].repeat(3, -1) * 255).astype(np.uint8)
    b2 = (band(m2, 2)[..., None].repeat(3, -1) * 255).astype(np.uint8)
    P = Image.fromarray
    m1d = np.array(ip_utils.set_diff(P(m1), P(b1)))
    m2d = np.array(ip_utils.set_diff(P(m2), P(b2)))
    fused = np.array(ur.fuse_bound(P(m1d), P(b1), P(b2), P(m2d)))
    union = (np.array(P(m2d)).astype(bool) | np.array(P(m1d)).astype(bool)).astype(np.uint8) * 255          # generate_images.py:161
    im = rng.integers(0, 256, (H, W, 3)).astype(np.uint8)
    blanked = np.array(ur.fuse_bound_and_im(P(im), P(fused)))
    out.update(mask1=m1, mask2=m2, bound1=b1, bound2=b2, mask1_diff=m1d, mask2_diff=m2d, bound_fused=fused, mask_union=union, image=im, image_blanked=blanked)
    # ---- depth-tested paste
    bg_im = rng.integers(0, 256, (H, W, 3)).astype(np.uint8)
    fg_im = rng.integers(0, 256, (H, W, 3)).astype(np.uint8)
    depth_u16 = rng.integers(256, 60 * 256, (H, W)).astype(np.uint16)
    depth_mat = depth_u16 / 256.                                                                                # get_depth :58-63
    sem = rng.integers(0, 19, (H, W)).astype(np.uint8)
    fg_depth = (rng.random((H, W)) * 60 + 0.5).astype(np.float32)
    fg_depth[rng.random((H, W)) < 0.1] = 0.1                                                                    # "no intersection" value :866
    K = np.array([[50.0, 0, W / 2], [0, 50.0, H / 2], [0, 0, 1]])
    with tempfile.TemporaryDirectory() as td:
        np.save(os.path.join(td, "target_poses.npy"), np.eye(4)[None])
        np.save(os.path.join(td, "intrinsic.npy"), K)

        class RayTracer:
            def __init__(self, v, f):
                pass

            def trace(self, ray_o, ray_d):                       # the masked pixels arrive in np.where order (row-major)
                d = torch.from_numpy(fg_depth[self.ii, self.jj])
                return None, None, d
        ur.raytracing.RayTracer = RayTracer
        torch.Tensor.cuda = lambda self, *a, **k: self
        for cat in ("vehicle", "person", "bicycle"):
            ii, jj = np.where(m1[..., 0] > 0)
            RayTracer.ii, RayTracer.jj = ii, jj
            r = ur.handle_occlusion_paste(0, td, bg_im.copy(), fg_im.copy(), m1.copy(), depth_mat.copy(), sem.copy(), _Mesh(), _Mesh(), cat)
            out.update({f"paste_{cat}_im": r[0], f"paste_{cat}_depth": r[1], f"paste_{cat}_semantic": r[2], f"paste_{cat}_mask": r[3],
                        f"paste_{cat}_occlusion": np.float64(r[4])})
    out.update(bg_im=bg_im, fg_im=fg_im, depth_u16=depth_u16, semantic=sem, fg_depth=fg_depth)
    _oracle_common.save_golden(os.path.join(OUT, "g18_foreground.npz"), **out)
    print("wrote g18_foreground.npz", {k: float(out[k]) for k in out if k.endswith("occlusion")})


if __name__ == "__main__":
    main()
