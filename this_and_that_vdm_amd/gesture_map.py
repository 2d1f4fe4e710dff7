"""Gesture-map ("this"/"that" points) rasteriser without cv2 -- SURVEY.md 8(f)3.

Mirror of the reference's `get_thisthat_sam` (`data_loader/video_this_that_dataset.py:28-130`; the same code is inlined in
`app.py:282-328`): for every annotated point a white canvas of the ORIGINAL image size gets a 21x21 square (first point red,
later points green, OpenCV channel order B,G,R), is blurred with the 99-tap isotropic Gaussian (sigma 10) the reference
builds with `bivariate_Gaussian(99, 10, 10, 0, isotropic=True)` (`utils/optical_flow_utils.py:197-219`) through
`cv2.filter2D`, resized to the model resolution with `cv2.resize(..., INTER_CUBIC)`, optionally flipped, divided by 255 and
written (channels first) into the frame slot the annotation names; all other frames stay zero.

cv2 is not available in this image, so its two operators are restated from their documented definitions:
  * `filter2D(src, -1, k)`: correlation, anchor at the kernel centre, border BORDER_REFLECT_101 (numpy "reflect").  The
    kernel is an outer product of two normalised 1-D Gaussians, so it is applied as two 1-D passes (float64 accumulate,
    float32 result; OpenCV's DFT path for large kernels differs from this by float32 rounding only).
  * `resize(..., INTER_CUBIC)`: separable 4-tap Keys kernel with A = -0.75, half-pixel centres
    (src = (dst + 0.5) * scale - 0.5), source indices clamped to the border, no antialiasing.
`tests/test_gesture_map_cpu.py` checks the two against independent implementations (scipy.ndimage.correlate mode="mirror",
torch bicubic interpolation, which uses the same A and alignment).  This is host-side request preparation, like the
reference's; it feeds the VAE encode that produces `controlnet_cond` for the denoise loop.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import numpy as np

DOT_RANGE = 10                      # half-width of the square: 21 x 21 pixels        (reference :63)
BLUR_TAPS, BLUR_SIGMA = 99, 10.0    # bivariate_Gaussian(99, 10, 10, 0, isotropic)    (reference :25)
FIRST_POINT_BGR = (0.0, 0.0, 255.0)     # red in OpenCV order                          (reference :69)
OTHER_POINT_BGR = (0.0, 255.0, 0.0)     # green                                        (reference :71)


def gaussian_taps(taps: int = BLUR_TAPS, sigma: float = BLUR_SIGMA) -> np.ndarray:
    """1-D factor of the reference's normalised isotropic kernel: exp(-x^2 / (2 sigma^2)) on the integer grid
    `arange(-taps//2 + 1, taps//2 + 1)` (mesh_grid, optical_flow_utils.py:177), normalised to sum 1.  The outer product of
    two of these equals `bivariate_Gaussian(taps, sigma, ., ., isotropic=True)` exactly (its normaliser factorises)."""
    ax = np.arange(-taps // 2 + 1.0, taps // 2 + 1.0)
    k = np.exp(-0.5 * (ax / sigma) ** 2)
    return k / k.sum()


def gaussian_kernel2d(taps: int = BLUR_TAPS, sigma: float = BLUR_SIGMA) -> np.ndarray:
    k = gaussian_taps(taps, sigma)
    return np.outer(k, k)


def _correlate1d_reflect101(img: np.ndarray, k: np.ndarray, axis: int) -> np.ndarray:
    """1-D correlation along `axis` with BORDER_REFLECT_101; anchor = len(k)//2.  img: [H, W, C] float64."""
    r = len(k) // 2
    n = img.shape[axis]
    # reflect-101 padding also when the radius exceeds the image (index arithmetic instead of np.pad's single bounce)
    idx = np.arange(-r, n + (len(k) - 1 - r))
    if n > 1:
        period = 2 * (n - 1)
        idx = np.abs(idx) % period
        idx = np.where(idx >= n, period - idx, idx)
    else:
        idx = np.zeros_like(idx)
    padded = np.take(img, idx, axis=axis)
    out = np.zeros_like(img)
    for t, w in enumerate(k):       # 99 shifted adds: cheap next to the VAE encode that follows
        out += w * np.take(padded, np.arange(t, t + n), axis=axis)
    return out


def filter2d_separable(img: np.ndarray, k1d: np.ndarray) -> np.ndarray:
    """cv2.filter2D(img, -1, outer(k1d, k1d)) for float32 HxWxC input (see the module docstring)."""
    x = img.astype(np.float64)
    x = _correlate1d_reflect101(x, k1d, axis=1)
    x = _correlate1d_reflect101(x, k1d, axis=0)
    return x.astype(np.float32)


def _cubic_weights(t: np.ndarray, a: float = -0.75) -> np.ndarray:
    """Keys cubic weights for taps at offsets (-1, 0, 1, 2) of a sample at fractional position t in [0, 1)."""
    def near(x):   # |x| <= 1
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1.0

    def far(x):    # 1 < |x| < 2
        return ((a * x - 5.0 * a) * x + 8.0 * a) * x - 4.0 * a

    return np.stack([far(t + 1.0), near(t), near(1.0 - t), far(2.0 - t)], axis=-1)


def _resize_matrix(n_in: int, n_out: int) -> np.ndarray:
    """[n_out, n_in] interpolation matrix of INTER_CUBIC along one axis (border taps clamped, i.e. accumulated)."""
    scale = n_in / n_out
    src = (np.arange(n_out) + 0.5) * scale - 0.5
    base = np.floor(src)
    w = _cubic_weights(src - base)
    m = np.zeros((n_out, n_in), dtype=np.float64)
    for tap in range(4):
        idx = np.clip(base.astype(np.int64) - 1 + tap, 0, n_in - 1)
        np.add.at(m, (np.arange(n_out), idx), w[:, tap])
    return m


def resize_bicubic(img: np.ndarray, width: int, height: int) -> np.ndarray:
    """cv2.resize(img, (width, height), interpolation=cv2.INTER_CUBIC) for float32 HxWxC input."""
    h, w = img.shape[:2]
    my, mx = _resize_matrix(h, height), _resize_matrix(w, width)
    out = np.einsum("yh,hwc->ywc", my, img.astype(np.float64))
    out = np.einsum("xw,ywc->yxc", mx, out)
    return out.astype(np.float32)


def draw_point_canvas(org_height: int, org_width: int, vertical: int, horizontal: int, first: bool) -> np.ndarray:
    """White HxWx3 float32 canvas with the clipped 21x21 square of one point (reference :58-71)."""
    canvas = np.full((org_height, org_width, 3), 255.0, dtype=np.float32)
    y0, y1 = max(0, vertical - DOT_RANGE), min(org_height, vertical + DOT_RANGE + 1)
    x0, x1 = max(0, horizontal - DOT_RANGE), min(org_width, horizontal + DOT_RANGE + 1)
    if y0 < y1 and x0 < x1:
        canvas[y0:y1, x0:x1] = FIRST_POINT_BGR if first else OTHER_POINT_BGR
    return canvas


def rasterise_points(points: Sequence[Tuple[int, float, float]], org_hw: Tuple[int, int], height: int, width: int,
                     num_frames: int, dilate: bool = True, flip: bool = False) -> Tuple[np.ndarray, List[int], List[Tuple[int, int]]]:
    """points: (frame_idx, horizontal, vertical) in original-image pixels, in annotation order (first = "this").
    Returns (condition [num_frames, 3, height, width] float32 in [~0, ~1], frame indices, (vertical, horizontal) pairs)."""
    cond = np.zeros((num_frames, 3, height, width), dtype=np.float32)
    k1d = gaussian_taps()
    frames, coords = [], []
    for i, (frame_idx, horizontal, vertical) in enumerate(points):
        frame_idx, vertical, horizontal = int(frame_idx), int(float(vertical)), int(float(horizontal))
        frames.append(frame_idx)
        coords.append((vertical, horizontal))
        img = draw_point_canvas(org_hw[0], org_hw[1], vertical, horizontal, first=(i == 0))
        if dilate:
            img = filter2d_separable(img, k1d)
        img = resize_bicubic(img, width, height)
        if flip:
            img = img[:, ::-1]
        cond[frame_idx] = (img / 255.0).transpose(2, 0, 1)
    return cond, frames, coords


def read_points_file(path: str) -> List[Tuple[int, float, float]]:
    """`data.txt`: one `frame_idx horizontal vertical` triple per line (reference :35-49)."""
    pts = []
    with open(path, "r") as f:
        for line in f.readlines():
            if not line.strip():
                continue
            frame_idx, horizontal, vertical = line.split(" ")[:3]
            pts.append((int(frame_idx), float(horizontal), float(vertical)))
    return pts


def get_thisthat_sam(config, intput_dir: str, store_dir: Optional[str] = None, flip: bool = False, verbose: bool = False):
    """Drop-in for the reference function of the same name and (mis-spelt) argument: reads `<dir>/data.txt` and the size of
    `<dir>/im_0.jpg`, returns (condition [F, 3, H, W] float32, motion_bucket_id, frame indices, coordinates).
    `config` needs: video_seq_length, conditioning_channels (3), height, width, dilate, motion_bucket_id."""
    import PIL.Image

    if config["conditioning_channels"] != 3:
        raise NotImplementedError("only 3 conditioning channels, as in the reference")
    with PIL.Image.open(os.path.join(intput_dir, "im_0.jpg")) as im:
        org_width, org_height = im.size
    pts = read_points_file(os.path.join(intput_dir, "data.txt"))
    cond, frames, coords = rasterise_points(pts, (org_height, org_width), config["height"], config["width"],
                                            config["video_seq_length"], dilate=bool(config["dilate"]), flip=flip)
    if store_dir is not None and verbose:      # the reference dumps the resized BGR canvases for inspection
        for i, f in enumerate(frames):
            bgr = np.clip(cond[f].transpose(1, 2, 0) * 255.0, 0, 255).round().astype(np.uint8)
            PIL.Image.fromarray(bgr[..., ::-1].copy()).save(os.path.join(store_dir, f"condition_TT{i}.png"))
    bucket = 200 if config["motion_bucket_id"] is None else config["motion_bucket_id"]
    return cond, bucket, frames, coords
