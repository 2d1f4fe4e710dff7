"""StableVideoDiffusionPipeline (VL: UNet only) -- drop-in for svd/pipeline_stable_video_diffusion.py:323-578; the loop
(:528-562) runs on the fused DenoiseLoop without a ControlNet."""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Union

import PIL.Image
import torch

from .pipeline_stable_video_diffusion_controlnet import _SVDPipelineCore


class StableVideoDiffusionPipeline(_SVDPipelineCore):
    @torch.no_grad()
    def __call__(
        self,
        image: Union[PIL.Image.Image, List[PIL.Image.Image], torch.FloatTensor],
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
    ):
        return self._generate(image, None, None, prompt, use_text, text_encoder, height, width, num_frames,
                              num_inference_steps, min_guidance_scale, max_guidance_scale, fps, motion_bucket_id,
                              noise_aug_strength, decode_chunk_size, num_videos_per_prompt, generator, latents, output_type,
                              callback_on_step_end, callback_on_step_end_tensor_inputs, return_dict)
