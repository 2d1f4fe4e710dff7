"""Gesture-map rasteriser: the cv2 operators it restates are checked against independent implementations
(scipy correlate with mirror border = BORDER_REFLECT_101; torch bicubic = Keys A=-0.75, half-pixel centres, clamped border),
then the reference's pipeline order and conventions (data_loader/video_this_that_dataset.py:28-130) end to end."""
import os

import numpy as np
import pytest
import torch

from this_and_that_vdm_amd import gesture_map as gm


def test_kernel_is_the_reference_bivariate_gaussian():
    # bivariate_Gaussian(99, 10, ., ., isotropic=True): exp(-0.5 * g^T inv(diag(100,100)) g) / sum  (optical_flow_utils.py:184-219)
    ax = np.arange(-99 // 2 + 1.0, 99 // 2 + 1.0)
    xx, yy = np.meshgrid(ax, ax)
    ref = np.exp(-0.5 * (xx ** 2 + yy ** 2) / 100.0)
    ref /= ref.sum()
    k = gm.gaussian_kernel2d()
    assert k.shape == (99, 99) and ax[0] == -49 and ax[-1] == 49
    np.testing.assert_allclose(k, ref, rtol=1e-12, atol=1e-18)


@pytest.mark.parametrize("shape", [(120, 160), (40, 260), (7, 9)])
def test_filter2d_matches_scipy_mirror(shape):
    from scipy import ndimage
    rng = np.random.default_rng(0)
    img = (rng.random(shape + (3,)) * 255).astype(np.float32)
    got = gm.filter2d_separable(img, gm.gaussian_taps())
    if min(shape) > 49:           # scipy's mirror mode needs the radius inside the image
        k2 = gm.gaussian_kernel2d()
        want = np.stack([ndimage.correlate(img[..., c].astype(np.float64), k2, mode="mirror") for c in range(3)], -1)
        np.testing.assert_allclose(got, want, rtol=0, atol=2e-3)
    # a constant image stays constant under any normalised kernel and reflect border
    flat = np.full(shape + (3,), 255.0, np.float32)
    np.testing.assert_allclose(gm.filter2d_separable(flat, gm.gaussian_taps()), flat, atol=1e-3)


@pytest.mark.parametrize("src,dst", [((480, 640), (256, 448)), ((300, 200), (64, 112)), ((32, 56), (64, 112))])
def test_bicubic_matches_torch(src, dst):
    rng = np.random.default_rng(1)
    img = (rng.random(src + (3,)) * 255).astype(np.float32)
    got = gm.resize_bicubic(img, dst[1], dst[0])
    t = torch.from_numpy(img).permute(2, 0, 1)[None].double()
    want = torch.nn.functional.interpolate(t, size=dst, mode="bicubic", align_corners=False)[0].permute(1, 2, 0).numpy()
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-3)
    assert got.shape == dst + (3,) and got.dtype == np.float32


def test_cubic_weights_partition_of_unity():
    w = gm._cubic_weights(np.linspace(0, 0.999, 50))
    np.testing.assert_allclose(w.sum(-1), 1.0, atol=1e-12)
    np.testing.assert_allclose(gm._cubic_weights(np.array([0.0]))[0], [0, 1, 0, 0], atol=1e-12)


def test_square_colours_and_clipping():
    c = gm.draw_point_canvas(100, 120, 50, 60, first=True)
    assert (c[40:61, 50:71] == np.array([0, 0, 255], np.float32)).all()       # 21 x 21, B,G,R order: red
    assert (c[39, 50:71] == 255).all() and (c[40:61, 49] == 255).all() and c[61, 60, 0] == 255
    g = gm.draw_point_canvas(100, 120, 2, 118, first=False)                   # clipped at the top-right corner
    assert (g[0:13, 108:120] == np.array([0, 255, 0], np.float32)).all() and (g[13, 108:120] == 255).all()
    off = gm.draw_point_canvas(100, 120, -50, 500, first=True)                # entirely outside: untouched canvas
    assert (off == 255).all()


def test_rasterise_pipeline_no_dilate_known_answer():
    # same size in and out, no blur: the resize is the identity, so the frame is the canvas / 255 with channels first
    cond, frames, coords = gm.rasterise_points([(0, 30.7, 20.2), (5, 10, 12)], (64, 112), 64, 112, 14, dilate=False)
    assert cond.shape == (14, 3, 64, 112) and frames == [0, 5] and coords == [(20, 30), (12, 10)]     # int(float(.))
    want0 = gm.draw_point_canvas(64, 112, 20, 30, True).transpose(2, 0, 1) / 255.0
    want5 = gm.draw_point_canvas(64, 112, 12, 10, False).transpose(2, 0, 1) / 255.0
    np.testing.assert_allclose(cond[0], want0, atol=1e-6)
    np.testing.assert_allclose(cond[5], want5, atol=1e-6)
    assert not cond[[1, 2, 3, 4] + list(range(6, 14))].any()                 # every other frame stays zero


def test_rasterise_dilated_properties_and_flip():
    pts = [(0, 200, 150), (13, 500, 300)]
    cond, _, _ = gm.rasterise_points(pts, (480, 640), 256, 448, 14, dilate=True)
    f0, f13 = cond[0], cond[13]
    # far from the dot the canvas stays white; the red dot lowers B and G, never R; the green dot lowers B and R
    assert abs(f0[:, -1, -1] - 1.0).max() < 1e-4
    cy, cx = int(150 * 256 / 480), int(200 * 448 / 640)
    assert f0[2, cy, cx] > 0.999 and f0[0, cy, cx] < 0.75 and f0[1, cy, cx] < 0.75
    cy, cx = int(300 * 256 / 480), int(500 * 448 / 640)
    assert f13[1, cy, cx] > 0.999 and f13[0, cy, cx] < 0.75 and f13[2, cy, cx] < 0.75
    # blur mass: the dot removes 21*21*255 per darkened channel from the source canvas; the resize keeps the mean
    missing = (1.0 - f0[0]).sum() * (480 * 640) / (256 * 448)
    assert abs(missing - 21 * 21) / (21 * 21) < 0.02
    flipped, _, _ = gm.rasterise_points(pts, (480, 640), 256, 448, 14, dilate=True, flip=True)
    np.testing.assert_allclose(flipped, cond[..., ::-1], atol=1e-6)


def test_get_thisthat_sam_reads_the_reference_folder_layout(tmp_path):
    import PIL.Image
    PIL.Image.fromarray(np.zeros((120, 200, 3), np.uint8)).save(os.path.join(tmp_path, "im_0.jpg"))
    with open(os.path.join(tmp_path, "data.txt"), "w") as f:
        f.write("0 50.0 40.0\n7 150 100\n")           # frame horizontal vertical
    cfg = {"video_seq_length": 14, "conditioning_channels": 3, "height": 64, "width": 112, "dilate": True, "motion_bucket_id": None}
    cond, bucket, frames, coords = gm.get_thisthat_sam(cfg, str(tmp_path), store_dir=str(tmp_path), verbose=True)
    assert cond.shape == (14, 3, 64, 112) and cond.dtype == np.float32 and bucket == 200
    assert frames == [0, 7] and coords == [(40, 50), (100, 150)]
    want, _, _ = gm.rasterise_points([(0, 50, 40), (7, 150, 100)], (120, 200), 64, 112, 14)
    np.testing.assert_array_equal(cond, want)
    assert os.path.exists(os.path.join(tmp_path, "condition_TT0.png")) and os.path.exists(os.path.join(tmp_path, "condition_TT1.png"))
    cfg["motion_bucket_id"] = 127
    assert gm.get_thisthat_sam(cfg, str(tmp_path))[1] == 127
    cfg["conditioning_channels"] = 1
    with pytest.raises(NotImplementedError):
        gm.get_thisthat_sam(cfg, str(tmp_path))


def test_reference_example_point_sets_reproduce_their_committed_maps():
    """The four annotated examples the reference ships (__assets__/Bridge_example/*/data.txt; inputs of
    test_code/inference.py), committed as data in tests/golden/gesture_bridge.json together with checksums of the maps
    (tests/golden/make_gesture_golden.py): frame slots, coordinates, value range and the exact rounded image."""
    import hashlib
    import json
    doc = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gesture_bridge.json")))
    assert len(doc["examples"]) == 4
    for name, ex in doc["examples"].items():
        pts = [tuple(p) for p in ex["points"]]
        cond, frames, coords = gm.rasterise_points(pts, tuple(ex["org_hw"]), doc["height"], doc["width"], doc["frames"])
        assert frames == ex["frames"] and [list(c) for c in coords] == ex["coords"], name
        assert cond.shape == (doc["frames"], 3, doc["height"], doc["width"])
        u8 = np.clip(cond * 255.0, 0, 255).round().astype(np.uint8)
        assert hashlib.sha256(u8.tobytes()).hexdigest() == ex["map"]["sha256_u8"], name
        np.testing.assert_allclose(float(cond.astype(np.float64).sum()), ex["map"]["sum"], rtol=1e-9)
        # the first point is drawn red (B, G, R = 0, 0, 255), later ones green: inside the blob the other channels dip
        f0, (v0, h0) = frames[0], coords[0]
        sy, sx = doc["height"] / ex["org_hw"][0], doc["width"] / ex["org_hw"][1]
        y, x = int(v0 * sy), int(h0 * sx)
        assert cond[f0, 2, y, x] > 0.98 and cond[f0, 0, y, x] < 0.9 and cond[f0, 1, y, x] < 0.9, name
        untouched = [f for f in range(doc["frames"]) if f not in frames]
        assert float(np.abs(cond[untouched]).max()) == 0.0
        flipped, _, _ = gm.rasterise_points(pts, tuple(ex["org_hw"]), doc["height"], doc["width"], doc["frames"], flip=True)
        np.testing.assert_array_equal(flipped, cond[..., ::-1])
