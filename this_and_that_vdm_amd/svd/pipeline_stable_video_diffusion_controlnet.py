"""StableVideoDiffusionControlNetPipeline (VGL) -- drop-in for
svd/pipeline_stable_video_diffusion_controlnet.py:371-736 with the 25-step loop executed by the fused,
hipGraph-replayed DenoiseLoop (svd/denoise.py).

Kept: the ``__call__`` surface and argument meaning, CLIP(+text) context with the freshly constructed
LayerNorm((78,1024)) (reference :165-173), un-scaled VAE ``.mode()`` latents (:200), CFG order [uncond, cond]
(:177-185,203-211), fps-1 / motion bucket / noise-aug time ids, per-frame guidance ramp, ``output_type="latent"``,
``latents=``, ``generator=``, ``callback_on_step_end``.
``use_instructpix2pix`` (CFG batch of 3, reference :182-184,208-210,698-702) and ``guess_mode`` without CFG are built;
``guess_mode`` with CFG raises, as the reference's branch (:676-681) cannot run either.
Changed on purpose: the gesture map is VAE-encoded ONCE per request instead of inside every step (reference :652 --
loop-invariant), in the VAE's own dtype after the force_upcast window, exactly where the reference encodes it."""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Union

import numpy as np
import PIL.Image
import torch
import torch.nn as nn

from .denoise import DenoiseLoop
from .pipeline_utils import (CLIPFeatureExtractor, PipelineBase, StableVideoDiffusionPipelineOutput, VaeImageProcessor,
                             append_dims, randn_tensor, resize_with_antialiasing, tensor2vid)
from .temporal_controlnet import ControlNetModel


class _SVDPipelineCore(PipelineBase):
    model_cpu_offload_seq = "image_encoder->unet->vae"
    _callback_tensor_inputs = ["latents"]

    def __init__(self, vae, image_encoder, unet, scheduler, feature_extractor):
        self.register_modules(vae=vae, image_encoder=image_encoder, unet=unet, scheduler=scheduler,
                              feature_extractor=feature_extractor)
        self.vae_scale_factor = 2 ** (len(self.vae.config.block_out_channels) - 1)
        self.image_processor = VaeImageProcessor(vae_scale_factor=self.vae_scale_factor, do_convert_rgb=True)
        self.control_image_processor = VaeImageProcessor(vae_scale_factor=self.vae_scale_factor, do_convert_rgb=True,
                                                         do_normalize=False)
        self._loops: Dict[bool, DenoiseLoop] = {}

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path=None, *, vae=None, image_encoder=None, unet=None, scheduler=None,
                        feature_extractor=None, torch_dtype=None, **kwargs):
        """The reference passes vae / image_encoder / unet explicitly (test_code/inference.py:171-178) and lets diffusers
        load the remaining components from the hub folder.  Same here, from a LOCAL diffusers-format folder (no network):
          feature_extractor/preprocessor_config.json -> CLIP image processor (mean/std used by ``encode_clip``, reference :145-152)
          scheduler/scheduler_config.json            -> EulerDiscreteScheduler
          unet/                                      -> UNetSpatioTemporalConditionModel (only if ``unet`` is not passed)
        vae and image_encoder are third-party models (AutoencoderKLTemporalDecoder / CLIPVisionModelWithProjection) and must be
        passed in.  Without a folder the scheduler defaults to SVD's shipped EulerDiscrete configuration and the feature
        extractor to CLIP's published preprocessing constants; nothing is ever silently skipped."""
        import os
        path = pretrained_model_name_or_path
        have_dir = isinstance(path, str) and os.path.isdir(path)
        if unet is None and have_dir and os.path.isdir(os.path.join(path, "unet")):
            from .unet_spatio_temporal_condition import UNetSpatioTemporalConditionModel
            unet = UNetSpatioTemporalConditionModel.from_pretrained(path, subfolder="unet", torch_dtype=torch_dtype)
        missing = [n for n, v in (("vae", vae), ("image_encoder", image_encoder), ("unet", unet)) if v is None]
        if missing:
            raise ValueError(f"{cls.__name__}.from_pretrained needs these components passed in: {missing}")
        if scheduler is None:
            from .scheduling_euler_discrete import EulerDiscreteScheduler
            cfg = os.path.join(path, "scheduler", "scheduler_config.json") if have_dir else None
            scheduler = EulerDiscreteScheduler.from_config(cfg) if cfg and os.path.isfile(cfg) else EulerDiscreteScheduler()
        if feature_extractor is None:
            cfg = os.path.join(path, "feature_extractor", "preprocessor_config.json") if have_dir else None
            feature_extractor = CLIPFeatureExtractor.from_json_file(cfg) if cfg and os.path.isfile(cfg) else CLIPFeatureExtractor()
        return cls(vae=vae, image_encoder=image_encoder, unet=unet, scheduler=scheduler, feature_extractor=feature_extractor)

    # ---- constants of a request (reference :130-254,305-337)
    def encode_clip(self, image, prompt, use_text, text_encoder, device, num_videos_per_prompt, do_classifier_free_guidance,
                    use_instructpix2pix=False):
        dtype = next(self.image_encoder.parameters()).dtype
        if not isinstance(image, torch.Tensor):
            image = self.image_processor.numpy_to_pt(self.image_processor.pil_to_numpy(image))
            image = (resize_with_antialiasing(image * 2.0 - 1.0, (224, 224)) + 1.0) / 2.0
            if self.feature_extractor is None:
                raise RuntimeError("encode_clip: the pipeline has no feature_extractor, so the CLIP mean/std normalisation the "
                                   "reference always applies (:145-152) cannot be done; build the pipeline with from_pretrained() "
                                   "or pass feature_extractor=CLIPFeatureExtractor()")
            image = self.feature_extractor(images=image, do_normalize=True, do_center_crop=False, do_resize=False,
                                           do_rescale=False, return_tensors="pt").pixel_values
        emb = self.image_encoder(image.to(device=device, dtype=dtype)).image_embeds.unsqueeze(1)
        bs, seq, _ = emb.shape
        ehs = emb.repeat(1, num_videos_per_prompt, 1).view(bs * num_videos_per_prompt, seq, -1)
        if use_text:
            text = text_encoder(prompt)[0]
            ehs = torch.cat((text, ehs), dim=1)
            ln = nn.LayerNorm(tuple(ehs.shape[1:])).to(device=device, dtype=dtype)      # fresh, gamma=1 beta=0 (:172)
            ehs = ln(ehs)
        if do_classifier_free_guidance:
            neg = torch.zeros_like(ehs)
            ehs = torch.cat([ehs, neg, neg]) if use_instructpix2pix else torch.cat([neg, ehs])        # :182-185
        return ehs

    def _encode_vae_image(self, image, device, num_videos_per_prompt, do_classifier_free_guidance, use_instructpix2pix=False):
        lat = self.vae.encode(image.to(device=device)).latent_dist.mode()
        if do_classifier_free_guidance:
            neg = torch.zeros_like(lat)
            lat = torch.cat([lat, lat, neg]) if use_instructpix2pix else torch.cat([neg, lat])        # :208-211
        return lat.repeat(num_videos_per_prompt, 1, 1, 1)

    def _get_add_time_ids(self, fps, motion_bucket_id, noise_aug_strength, dtype, batch_size, num_videos_per_prompt,
                          do_classifier_free_guidance, guess_mode=False, use_instructpix2pix=False):
        ids = [fps, motion_bucket_id, noise_aug_strength]
        passed = self.unet.config.addition_time_embed_dim * len(ids)
        expected = self.unet.add_embedding.linear_1.in_features
        if expected != passed:
            raise ValueError(f"Model expects an added time embedding vector of length {expected}, but a vector of {passed} "
                             "was created. The model has an incorrect config.")
        t = torch.tensor([ids], dtype=dtype).repeat(batch_size * num_videos_per_prompt, 1)
        if not do_classifier_free_guidance:
            return t
        return torch.cat([t, t, t]) if use_instructpix2pix else torch.cat([t, t])

    def decode_latents(self, latents, num_frames, decode_chunk_size=14):
        import inspect
        lat = latents.flatten(0, 1) / self.vae.config.scaling_factor
        takes_frames = "num_frames" in inspect.signature(self.vae.forward).parameters
        out = []
        for i in range(0, lat.shape[0], decode_chunk_size):
            chunk = lat[i:i + decode_chunk_size]
            kw = {"num_frames": chunk.shape[0]} if takes_frames else {}
            out.append(self.vae.decode(chunk, **kw).sample)
        frames = torch.cat(out, 0)
        return frames.reshape(-1, num_frames, *frames.shape[1:]).permute(0, 2, 1, 3, 4).float()

    def check_inputs(self, image, height, width):
        if not isinstance(image, (torch.Tensor, PIL.Image.Image, list)):
            raise ValueError("`image` has to be of type `torch.FloatTensor` or `PIL.Image.Image` or `List[PIL.Image.Image]` "
                             f"but is {type(image)}")
        if height % 8 != 0 or width % 8 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")

    def prepare_latents(self, batch_size, num_frames, num_channels_latents, height, width, dtype, device, generator, latents=None):
        shape = (batch_size, num_frames, num_channels_latents // 2, height // self.vae_scale_factor, width // self.vae_scale_factor)
        if isinstance(generator, list) and len(generator) != batch_size:
            raise ValueError(f"You have passed a list of generators of length {len(generator)}, but requested an effective "
                             f"batch size of {batch_size}. Make sure the batch size matches the length of the generators.")
        latents = randn_tensor(shape, generator=generator, device=device, dtype=dtype) if latents is None else latents.to(device)
        return latents * self.scheduler.init_noise_sigma

    def prepare_condition_image(self, condition_img, device):
        """[0,1] float gesture frames [F,3,H,W] -> fp16 on device (reference :363-364, quirk Q7)."""
        t = torch.from_numpy(condition_img) if isinstance(condition_img, np.ndarray) else condition_img
        return t.to(torch.float16).to(device)

    @property
    def guidance_scale(self):
        return self._guidance_scale

    @property
    def num_timesteps(self):
        return self._num_timesteps

    # ---- the shared generation routine
    @torch.no_grad()
    def _generate(self, image, condition_img, controlnet, prompt, use_text, text_encoder, height, width, num_frames,
                  num_inference_steps, min_guidance_scale, max_guidance_scale, fps, motion_bucket_id, noise_aug_strength,
                  decode_chunk_size, num_videos_per_prompt, generator, latents, output_type, callback_on_step_end,
                  callback_on_step_end_tensor_inputs, return_dict, controlnet_conditioning_scale=1.0,
                  control_guidance_start=0.0, control_guidance_end=1.0, use_instructpix2pix=False, image_guidance_scale=7.5,
                  guess_mode=False):
        height = height or self.unet.config.sample_size * self.vae_scale_factor
        width = width or self.unet.config.sample_size * self.vae_scale_factor
        num_frames = num_frames if num_frames is not None else self.unet.config.num_frames
        decode_chunk_size = decode_chunk_size if decode_chunk_size is not None else num_frames
        self.check_inputs(image, height, width)
        batch_size = 1 if isinstance(image, PIL.Image.Image) else (len(image) if isinstance(image, list) else image.shape[0])
        if batch_size * num_videos_per_prompt != 1:
            raise NotImplementedError("the fused loop serves one request per call (shard requests across GPUs/processes)")
        device = self._execution_device
        do_cfg = max_guidance_scale > 1.0

        ip2p = bool(use_instructpix2pix) and do_cfg
        ehs = self.encode_clip(image, prompt, use_text, text_encoder, device, num_videos_per_prompt, do_cfg, ip2p)
        fps = fps - 1                                                            # SVD was conditioned on fps-1 (:527)
        img = self.image_processor.preprocess(image, height=height, width=width)
        img = img + noise_aug_strength * randn_tensor(img.shape, generator=generator, device=img.device, dtype=img.dtype)
        upcast = self.vae.dtype == torch.float16 and getattr(self.vae.config, "force_upcast", False)
        if upcast:
            self.vae.to(dtype=torch.float32)
        image_latents = self._encode_vae_image(img.to(self.vae.dtype), device, num_videos_per_prompt, do_cfg, ip2p).to(ehs.dtype)
        image_latents = image_latents.unsqueeze(1).repeat(1, num_frames, 1, 1, 1)
        if upcast:
            self.vae.to(dtype=torch.float16)
        gesture_latents = None
        if controlnet is not None:
            # after the cast-back, in the VAE's own dtype: where (and in which precision) the reference encodes it (:652),
            # but once per request instead of once per step (loop-invariant, quirk Q6)
            cond = self.prepare_condition_image(condition_img, device)
            gesture_latents = self.vae.encode(cond.to(self.vae.dtype)).latent_dist.mode()
        added_time_ids = self._get_add_time_ids(fps, motion_bucket_id, noise_aug_strength, ehs.dtype, batch_size,
                                                num_videos_per_prompt, do_cfg, use_instructpix2pix=ip2p).to(device)
        self.scheduler.set_timesteps(num_inference_steps, device=device)
        timesteps = self.scheduler.timesteps
        latents = self.prepare_latents(batch_size * num_videos_per_prompt, num_frames, self.unet.config.in_channels, height,
                                       width, ehs.dtype, device, generator, latents)
        guidance = torch.linspace(min_guidance_scale, max_guidance_scale, num_frames).unsqueeze(0).to(device, latents.dtype)
        guidance = append_dims(guidance.repeat(batch_size * num_videos_per_prompt, 1), latents.ndim)
        self._guidance_scale = guidance
        self._num_timesteps = len(timesteps)
        scale = controlnet_conditioning_scale[0] if isinstance(controlnet_conditioning_scale, list) else controlnet_conditioning_scale

        # which steps keep the ControlNet (reference :611-617): 0.0 outside [start, end], else 1.0
        n = len(timesteps)
        keep = [1.0 - float(i / n < control_guidance_start or (i + 1) / n > control_guidance_end) for i in range(n)]

        key = controlnet is not None
        loop = self._loops.get(key)
        if loop is None or loop.controlnet is not controlnet or loop.unet is not self.unet:
            loop = self._loops[key] = DenoiseLoop(self.unet, controlnet, use_graph=True)
        loop.begin(latents=latents, image_latents=image_latents, encoder_hidden_states=ehs, added_time_ids=added_time_ids,
                   guidance_scale=guidance if do_cfg else None, sigmas=self.scheduler.sigmas, timesteps=timesteps,
                   controlnet_cond=gesture_latents, conditioning_scale=float(scale),
                   controlnet_keep=keep if controlnet is not None else None,
                   image_guidance_scale=float(image_guidance_scale) if ip2p else None,
                   guess_mode=bool(guess_mode) and controlnet is not None)
        with self.progress_bar(total=num_inference_steps) as bar:
            for i, t in enumerate(timesteps):
                loop.step()
                if callback_on_step_end is not None:
                    cur = loop.result().to(latents.dtype)
                    outs = callback_on_step_end(self, i, t, {k: cur for k in callback_on_step_end_tensor_inputs if k == "latents"})
                    new = (outs or {}).pop("latents", None)
                    if new is not None and new is not cur:
                        loop.latents.copy_(new.reshape(loop.latents.shape))
                bar.update()
        latents = loop.result().to(ehs.dtype)
        if output_type != "latent":
            frames = tensor2vid(self.decode_latents(latents.to(self.vae.dtype), num_frames, decode_chunk_size),
                                self.image_processor, output_type=output_type)
        else:
            frames = latents
        self.maybe_free_model_hooks()
        if not return_dict:
            return frames
        return StableVideoDiffusionPipelineOutput(frames=frames)


class StableVideoDiffusionControlNetPipeline(_SVDPipelineCore):
    @torch.no_grad()
    def __call__(
        self,
        image: Union[PIL.Image.Image, List[PIL.Image.Image], torch.FloatTensor],
        condition_img: np.ndarray,
        controlnet: ControlNetModel,
        prompt=None,
        use_text: bool = False,
        text_encoder=None,
        height: int = 576,
        width: int = 1024,
        num_frames: Optional[int] = None,
        num_inference_steps: int = 25,
        min_guidance_scale: float = 1.0,
        max_guidance_scale: float = 3.0,
        fps: int = 7,
        motion_bucket_id: int = 127,
        noise_aug_strength: float = 0.02,
        decode_chunk_size: Optional[int] = None,
        num_videos_per_prompt: Optional[int] = 1,
        generator: Optional[Union[torch.Generator, List[torch.Generator]]] = None,
        latents: Optional[torch.FloatTensor] = None,
        output_type: Optional[str] = "pil",
        callback_on_step_end: Optional[Callable[[int, int, Dict], None]] = None,
        callback_on_step_end_tensor_inputs: List[str] = ["latents"],
        return_dict: bool = True,
        controlnet_conditioning_scale: Union[float, List[float]] = 1.0,
        use_instructpix2pix: bool = False,
        control_guidance_start: Union[float, List[float]] = 0.0,
        control_guidance_end: Union[float, List[float]] = 1.0,
        inner_conditioning_scale: float = 1.0,
        guess_mode: bool = True,
        image_guidance_scale: float = 7.5,
    ):
        if not isinstance(controlnet, ControlNetModel):
            raise TypeError("controlnet must be a ControlNetModel")
        # the reference wraps both in a one-element list (:485-489), so only scalars work there; a one-element list
        # is accepted here as well
        if isinstance(control_guidance_start, (list, tuple)) or isinstance(control_guidance_end, (list, tuple)):
            if not (isinstance(control_guidance_start, (list, tuple)) and isinstance(control_guidance_end, (list, tuple))
                    and len(control_guidance_start) == len(control_guidance_end) == 1):
                raise ValueError("control_guidance_start/end: one value each (a single ControlNetModel)")
            control_guidance_start, control_guidance_end = control_guidance_start[0], control_guidance_end[0]
        if control_guidance_start >= control_guidance_end or control_guidance_start < 0.0 or control_guidance_end > 1.0:
            raise ValueError(f"control guidance window [{control_guidance_start}, {control_guidance_end}] must satisfy "
                             "0 <= start < end <= 1")
        if guess_mode and max_guidance_scale > 1.0:
            # reference :676-681 concatenates zeros onto residuals that already have the CFG batch, so the UNet's skip
            # additions fail on shape; test_code/inference.py always passes guess_mode=False.  Without CFG, guess_mode
            # only switches the 13 residual scales to logspace(-1, 0, 13) (temporal_controlnet.py:626-630): built.
            raise NotImplementedError("guess_mode=True with CFG cannot run in the reference either (:676-681); "
                                      "pass guess_mode=False as test_code/inference.py does")
        return self._generate(image, condition_img, controlnet, prompt, use_text, text_encoder, height, width, num_frames,
                              num_inference_steps, min_guidance_scale, max_guidance_scale, fps, motion_bucket_id,
                              noise_aug_strength, decode_chunk_size, num_videos_per_prompt, generator, latents, output_type,
                              callback_on_step_end, callback_on_step_end_tensor_inputs, return_dict,
                              controlnet_conditioning_scale, float(control_guidance_start), float(control_guidance_end),
                              use_instructpix2pix, image_guidance_scale, guess_mode)
