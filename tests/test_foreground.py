"""Row 8f-4, the image-space foreground composite of S-NeRF++ stage 1: oracle/foreground.py against vectors from the reference's own
functions (tests/golden/g18_foreground.npz: handle_occlusion_paste with an injected mesh depth, set_diff, fuse_bound,
fuse_bound_and_im, mask union), the OpenCV-defined boundary band against hand-computed answers, and on the GPU the kernels behind
snerf_amd/foreground.py bit-exactly against both."""
import numpy as np
import pytest
import torch

from oracle import foreground as of

gpu = pytest.mark.gpu


def _np(golden, name):
    return {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in golden(name).items()}


def test_oracle_paste_and_mask_algebra_vs_reference(golden):
    g = _np(golden, "g18_foreground")
    depth = g["depth_u16"] / 256.
    for cat in ("vehicle", "person", "bicycle"):
        im, d, s, m, occ = of.occlusion_paste(g["bg_im"], g["fg_im"], g["mask1"], depth, g["semantic"], g["fg_depth"], cat)
        assert np.array_equal(im, g[f"paste_{cat}_im"]) and np.array_equal(s, g[f"paste_{cat}_semantic"]) and np.array_equal(m, g[f"paste_{cat}_mask"])
        assert np.array_equal(d, g[f"paste_{cat}_depth"]) and occ == float(g[f"paste_{cat}_occlusion"])
    assert 0.2 < float(g["paste_vehicle_occlusion"]) < 0.6            # the fixture exercises both outcomes of the depth test
    assert np.array_equal(of.set_diff(g["mask1"], g["bound1"]), g["mask1_diff"])
    assert np.array_equal(of.set_diff(g["mask2"], g["bound2"]), g["mask2_diff"])
    assert np.array_equal(of.fuse_bound(g["mask1_diff"], g["bound1"], g["bound2"], g["mask2_diff"]), g["bound_fused"])
    assert np.array_equal(of.mask_union(g["mask1_diff"], g["mask2_diff"]), g["mask_union"])
    assert np.array_equal(of.fuse_bound_and_im(g["image"], g["bound_fused"]), g["image_blanked"])


def test_oracle_boundary_band_known_answers():
    """cv2.dilate / cv2.erode semantics (rect kernel, anchor = size // 2, border never wins), hand-computed: a 4 x 5 block in a 9 x 11
    image.  r = 3 (odd, symmetric window): dilation grows the block by 1 on every side, erosion shrinks it by 1.
    r = 2 (even): the window covers offsets {-1, 0}, so dilation grows the block by one pixel towards +x / +y only and erosion
    removes its first row and column."""
    m = np.zeros((9, 11), np.uint8)
    m[3:7, 2:7] = 255
    d3, e3 = of.rect_morph(m, 3, True), of.rect_morph(m, 3, False)
    exp = np.zeros_like(m); exp[2:8, 1:8] = 255
    assert np.array_equal(d3, exp)
    exp = np.zeros_like(m); exp[4:6, 3:6] = 255
    assert np.array_equal(e3, exp)
    d2, e2 = of.rect_morph(m, 2, True), of.rect_morph(m, 2, False)
    exp = np.zeros_like(m); exp[3:8, 2:8] = 255
    assert np.array_equal(d2, exp)
    exp = np.zeros_like(m); exp[4:7, 3:7] = 255
    assert np.array_equal(e2, exp)
    # a block touching the image border is not eroded from outside
    b = np.zeros((5, 5), np.uint8); b[:3, :3] = 255
    exp = np.zeros_like(b); exp[:2, :2] = 255
    assert np.array_equal(of.rect_morph(b, 3, False), exp)
    band = of.get_bound_im(m[..., None].repeat(3, -1), 3)
    assert band.shape == (9, 11, 3) and band.dtype == np.uint8 and set(np.unique(band)) == {0, 255}
    assert band[..., 0].sum() // 255 == 6 * 7 - 2 * 3
    assert np.array_equal(of.get_bound_im(m[..., None].repeat(3, -1), 0), of.get_bound_im(m[..., None].repeat(3, -1), 1))
    assert of.get_bound_im(m[..., None].repeat(3, -1), 1).sum() == 0
    assert of.bound_radius(np.zeros((4, 4, 3), np.uint8)) == 1 and of.bound_radius(m[..., None].repeat(3, -1), "bicycle") == 3
    wide = np.zeros((4, 400, 3), np.uint8); wide[1, 10:331] = 255
    assert of.bound_radius(wide) == int((320 / 80) ** .82)


def _oracle_frame(bg, depth, sem, instances):
    """generate_images.py:80-185 for one frame with the oracle's pieces (test helper)."""
    bg, depth, sem = bg.copy(), depth.copy(), sem.copy()
    tm = tb = tocc = None
    occ = []
    for inst in instances:
        cat = inst["category"]
        r = of.bound_radius(inst["mask"], cat)
        bg, depth, sem, tested, o = of.occlusion_paste(bg, inst["image"], inst["mask"], depth, sem, inst["fg_depth"], cat)
        occ.append(o)
        bound = of.get_bound_im(inst["mask"], r)
        mask = of.set_diff(inst["mask"], bound)
        tb = bound if tb is None else of.fuse_bound(tm, tb, bound, mask)
        tm = mask if tm is None else of.mask_union(tm, mask)
        if cat == "vehicle":
            tocc = tested if tocc is None else of.mask_union(tocc, tested)
    return dict(fuse=of.fuse_bound_and_im(bg, tb), mask=tm, bound=tb, occluded_mask=tocc, depth=depth, semantic=sem, occlusion=occ)


@gpu
def test_foreground_kernels_vs_reference_vectors(golden):
    from snerf_amd import foreground as fgm
    g = _np(golden, "g18_foreground")
    c = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    depth = (g["depth_u16"] / 256.).astype(np.float32)
    for cat in ("vehicle", "person", "bicycle"):
        bg, mask, d, s = c(g["bg_im"]), c(g["mask1"]), c(depth), c(g["semantic"])
        occ = fgm.handle_occlusion_paste(bg, c(g["fg_im"]), mask, d, s, None if cat == "person" else c(g["fg_depth"]), cat)
        assert np.array_equal(bg.cpu().numpy(), g[f"paste_{cat}_im"]) and np.array_equal(s.cpu().numpy(), g[f"paste_{cat}_semantic"])
        assert np.array_equal(mask.cpu().numpy(), g[f"paste_{cat}_mask"])
        assert np.array_equal(d.cpu().numpy().astype(np.float64), g[f"paste_{cat}_depth"]) and float(occ) == float(g[f"paste_{cat}_occlusion"])
    # mask algebra: set_diff comes out of the band launch, fuse_bound + union out of accumulate, blanking in place
    for k in (1, 2, 3, 4, 7):
        bound, diff = fgm.get_bound_im(c(g["mask1"]), k, return_mask_diff=True)
        ref = of.get_bound_im(g["mask1"], k)
        assert np.array_equal(bound.cpu().numpy(), ref) and np.array_equal(diff.cpu().numpy(), of.set_diff(g["mask1"], ref)), k
    tm, tb = c(g["mask1_diff"]), c(g["bound1"])
    fgm.accumulate(tm, tb, c(g["bound2"]), c(g["mask2_diff"]))
    assert np.array_equal(tb.cpu().numpy(), g["bound_fused"]) and np.array_equal(tm.cpu().numpy(), g["mask_union"])
    im = c(g["image"])
    fgm.fuse_bound_and_im(im, tb)
    assert np.array_equal(im.cpu().numpy(), g["image_blanked"])


@gpu
def test_foreground_frame_loop_full_size_vs_oracle():
    """A 1920 x 1280 frame with three overlapping instances (vehicle, person, bicycle) through composite_frame: every saved image
    bit-exact against the oracle's restatement of the instance loop; masks touching the image border; an empty instance."""
    from snerf_amd import foreground as fgm
    rng = np.random.default_rng(4)
    H, W = 1280, 1920
    yy, xx = np.mgrid[:H, :W]
    bg = rng.integers(0, 256, (H, W, 3)).astype(np.uint8)
    depth = (rng.integers(256, 80 * 256, (H, W)) / 256.).astype(np.float32)
    sem = rng.integers(0, 19, (H, W)).astype(np.uint8)
    inst = []
    for cy, cx, ry, rx, cat in ((700, 500, 260, 420, "vehicle"), (760, 820, 330, 110, "person"), (900, 1890, 150, 210, "bicycle"), (0, 0, 0, 0, "vehicle"),
                                (640, 700, 200, 380, "vehicle")):
        m = (((yy - cy) / max(ry, 1)) ** 2 + ((xx - cx) / max(rx, 1)) ** 2 <= 1) & (ry > 0)
        inst.append(dict(image=rng.integers(0, 256, (H, W, 3)).astype(np.uint8), mask=(m[..., None].repeat(3, -1) * 255).astype(np.uint8),
                         fg_depth=(rng.random((H, W)) * 70 + 1).astype(np.float32), category=cat))
    ref = _oracle_frame(bg, depth.astype(np.float64), sem, inst)
    c = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    out = fgm.composite_frame(c(bg), c(depth), c(sem), [dict(image=c(i["image"]), mask=c(i["mask"]), fg_depth=c(i["fg_depth"]), category=i["category"])
                                                        for i in inst])
    for k in ("fuse", "mask", "bound", "occluded_mask", "semantic"):
        assert np.array_equal(out[k].cpu().numpy(), ref[k]), k
    assert np.array_equal(out["depth"].cpu().numpy().astype(np.float64), ref["depth"])
    assert [float(o) for o in out["occlusion"]] == [float(o) for o in ref["occlusion"]]
    assert float(out["occlusion"][3]) == 1.0 and 0.2 < float(out["occlusion"][0]) < 0.8
