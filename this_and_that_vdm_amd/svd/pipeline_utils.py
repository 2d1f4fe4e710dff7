"""Host-side helpers shared by the two pipelines: image pre/post-processing, CLIP-input antialiased resize,
latent preparation, and a tiny pipeline base.  These sit OUTSIDE the denoise step;
they run stock torch ops on the third-party VAE / CLIP modules the caller supplies."""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Optional, Union

import numpy as np
import PIL.Image
import torch
import torch.nn.functional as F

from .modeling_utils import BaseOutput


@dataclass
class StableVideoDiffusionPipelineOutput(BaseOutput):
    frames: Union[List[List[PIL.Image.Image]], np.ndarray, torch.Tensor] = None


def randn_tensor(shape, generator=None, device=None, dtype=None):
    """Gaussian noise; like diffusers' helper, a CPU generator produces CPU noise that is then moved."""
    gen_device = device
    if generator is not None:
        g = generator[0] if isinstance(generator, (list, tuple)) else generator
        gen_device = g.device
    if isinstance(generator, (list, tuple)):
        parts = [torch.randn((1,) + tuple(shape[1:]), generator=g, device=gen_device, dtype=dtype) for g in generator]
        out = torch.cat(parts, 0)
    else:
        out = torch.randn(tuple(shape), generator=generator, device=gen_device, dtype=dtype)
    return out.to(device)


def append_dims(x: torch.Tensor, target_dims: int) -> torch.Tensor:
    return x[(...,) + (None,) * (target_dims - x.ndim)]


class VaeImageProcessor:
    """PIL / numpy / tensor <-> normalised NCHW tensors (resize to a multiple of the VAE scale factor)."""

    def __init__(self, vae_scale_factor: int = 8, do_normalize: bool = True, do_convert_rgb: bool = False):
        self.vae_scale_factor, self.do_normalize, self.do_convert_rgb = vae_scale_factor, do_normalize, do_convert_rgb

    @staticmethod
    def pil_to_numpy(images) -> np.ndarray:
        images = images if isinstance(images, (list, tuple)) else [images]
        return np.stack([np.asarray(im).astype(np.float32) / 255.0 for im in images], 0)

    @staticmethod
    def numpy_to_pt(images: np.ndarray) -> torch.Tensor:
        if images.ndim == 3:
            images = images[..., None]
        return torch.from_numpy(images.transpose(0, 3, 1, 2))

    @staticmethod
    def numpy_to_pil(images: np.ndarray):
        if images.ndim == 3:
            images = images[None]
        u8 = (images * 255).round().astype("uint8")
        return [PIL.Image.fromarray(im.squeeze()) if im.shape[-1] == 1 else PIL.Image.fromarray(im) for im in u8]

    def preprocess(self, image, height: Optional[int] = None, width: Optional[int] = None) -> torch.Tensor:
        if isinstance(image, PIL.Image.Image) or (isinstance(image, (list, tuple)) and isinstance(image[0], PIL.Image.Image)):
            ims = image if isinstance(image, (list, tuple)) else [image]
            if self.do_convert_rgb:
                ims = [im.convert("RGB") for im in ims]
            if height is not None and width is not None:
                ims = [im.resize((width, height), resample=PIL.Image.LANCZOS) for im in ims]
            t = self.numpy_to_pt(self.pil_to_numpy(ims))
        elif isinstance(image, np.ndarray):
            t = self.numpy_to_pt(image if image.ndim == 4 else image[None])
        elif torch.is_tensor(image):
            t = image if image.ndim == 4 else image[None]
        else:
            raise ValueError(f"unsupported image type {type(image)}")
        if self.do_normalize and (not torch.is_tensor(image) or float(t.min()) >= 0.0):
            t = 2.0 * t - 1.0
        return t

    def postprocess(self, image: torch.Tensor, output_type: str = "pil"):
        image = (image / 2 + 0.5).clamp(0, 1) if self.do_normalize else image
        if output_type == "pt":
            return image
        arr = image.cpu().permute(0, 2, 3, 1).float().numpy()
        return arr if output_type == "np" else self.numpy_to_pil(arr)


def tensor2vid(video: torch.Tensor, processor: VaeImageProcessor, output_type: str = "np"):
    """[B,C,F,H,W] -> list over batch of per-frame images."""
    outs = [processor.postprocess(video[b].permute(1, 0, 2, 3), output_type) for b in range(video.shape[0])]
    if output_type == "np":
        return np.stack(outs)
    if output_type == "pt":
        return torch.stack(outs)
    return outs


def _gauss_kernel1d(size: int, sigma: float, dtype) -> torch.Tensor:
    x = torch.arange(size, dtype=dtype) - size // 2
    if size % 2 == 0:
        x = x + 0.5
    k = torch.exp(-x.pow(2) / (2 * sigma ** 2))
    return k / k.sum()


def resize_with_antialiasing(img: torch.Tensor, size, interpolation: str = "bicubic", align_corners: bool = True):
    """Gaussian pre-blur sized by the down-scale factor (sigma = max((f-1)/2, 1e-3), kernel ~ 4 sigma, odd),
    reflect padding, then interpolate -- what the reference does before CLIP
    (svd/pipeline_stable_video_diffusion_controlnet.py:741-767)."""
    h, w = img.shape[-2:]
    fy, fx = h / size[0], w / size[1]
    sy, sx = max((fy - 1.0) / 2.0, 0.001), max((fx - 1.0) / 2.0, 0.001)
    ky, kx = int(max(4.0 * sy, 3)), int(max(4.0 * sx, 3))
    ky += 1 - ky % 2
    kx += 1 - kx % 2
    b, c = img.shape[:2]
    out = img
    for k1d, k, horizontal in ((_gauss_kernel1d(kx, sx, img.dtype), kx, True), (_gauss_kernel1d(ky, sy, img.dtype), ky, False)):
        front, rear = (k - 1) // 2, (k - 1) - (k - 1) // 2
        pad = (front, rear, 0, 0) if horizontal else (0, 0, front, rear)
        x = F.pad(out, pad, mode="reflect")
        wgt = k1d.to(img.device).view(1, 1, 1, k) if horizontal else k1d.to(img.device).view(1, 1, k, 1)
        out = F.conv2d(x.reshape(b * c, 1, *x.shape[-2:]), wgt).reshape(b, c, h, w)
    return F.interpolate(out, size=size, mode=interpolation, align_corners=align_corners)


class _PixelValues(dict):
    """BatchFeature-like: ``.pixel_values`` and ``["pixel_values"]``."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


class CLIPFeatureExtractor:
    """The slice of transformers.CLIPImageProcessor the pipelines call (reference :145-152): per-channel
    ``(x - image_mean) / image_std`` on an already resized, already [0,1] tensor batch.  Constants default to OpenAI CLIP's
    published preprocessing (what stabilityai/stable-video-diffusion-img2vid ships in feature_extractor/
    preprocessor_config.json); ``from_json_file`` reads a local preprocessor_config.json.  A transformers
    CLIPImageProcessor instance can be passed to the pipelines instead -- the call signature is the same."""

    def __init__(self, image_mean=(0.48145466, 0.4578275, 0.40821073), image_std=(0.26862954, 0.26130258, 0.27577711), **unused):
        self.image_mean, self.image_std = [float(v) for v in image_mean], [float(v) for v in image_std]

    @classmethod
    def from_json_file(cls, path: str) -> "CLIPFeatureExtractor":
        import json
        with open(path) as f:
            cfg = json.load(f)
        return cls(image_mean=cfg.get("image_mean", cls().image_mean), image_std=cfg.get("image_std", cls().image_std))

    def __call__(self, images, do_normalize=True, do_center_crop=False, do_resize=False, do_rescale=False, return_tensors="pt", **unused):
        if do_center_crop or do_resize or do_rescale:
            raise NotImplementedError("the pipelines call the feature extractor with do_center_crop/do_resize/do_rescale=False")
        x = images if torch.is_tensor(images) else torch.as_tensor(np.asarray(images))
        x = x.float()
        if do_normalize:
            mean = torch.tensor(self.image_mean, dtype=x.dtype, device=x.device).view(1, -1, 1, 1)
            std = torch.tensor(self.image_std, dtype=x.dtype, device=x.device).view(1, -1, 1, 1)
            x = (x - mean) / std
        return _PixelValues(pixel_values=x)


class PipelineBase:
    """The sliver of diffusers.DiffusionPipeline the reference scripts touch."""

    def register_modules(self, **modules):
        self._module_names = list(modules)
        for k, v in modules.items():
            setattr(self, k, v)

    @property
    def _execution_device(self):
        return next(self.unet.parameters()).device

    @property
    def device(self):
        return self._execution_device

    def to(self, *args, **kwargs):
        for name in self._module_names:
            m = getattr(self, name)
            if isinstance(m, torch.nn.Module):
                m.to(*args, **kwargs)
        return self

    def progress_bar(self, iterable=None, total=None):
        from tqdm.auto import tqdm
        return tqdm(iterable, total=total, disable=getattr(self, "_progress_disabled", False))

    def set_progress_bar_config(self, **kw):
        self._progress_disabled = bool(kw.get("disable", False))

    def maybe_free_model_hooks(self):
        pass

    def enable_model_cpu_offload(self, *a, **k):
        raise NotImplementedError("the MI355X build keeps all weights resident in HBM (288 GB); offload is not needed")
