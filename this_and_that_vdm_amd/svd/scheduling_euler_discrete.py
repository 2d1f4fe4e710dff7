"""EulerDiscreteScheduler as the SVD pipelines use it (diffusers==0.25.1 class, call sites
svd/pipeline_stable_video_diffusion_controlnet.py:336,583-584,632,709): Karras sigmas, continuous
timesteps 0.25*ln(sigma), v-prediction Euler step.  Host-side schedule logic; the per-step tensor math
used by the fused loop lives in tt_prep_model_input / tt_cfg_euler_step."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from .modeling_utils import BaseOutput, FrozenDict


@dataclass
class EulerDiscreteSchedulerOutput(BaseOutput):
    prev_sample: torch.FloatTensor = None
    pred_original_sample: Optional[torch.FloatTensor] = None


class EulerDiscreteScheduler:
    order = 1

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012,
                 beta_schedule: str = "scaled_linear", prediction_type: str = "v_prediction",
                 interpolation_type: str = "linear", use_karras_sigmas: bool = True, sigma_min: float = 0.002,
                 sigma_max: float = 700.0, timestep_spacing: str = "leading", timestep_type: str = "continuous",
                 steps_offset: int = 1):
        """Defaults = SVD's shipped scheduler_config.json."""
        self.config = FrozenDict(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                                 beta_schedule=beta_schedule, prediction_type=prediction_type,
                                 interpolation_type=interpolation_type, use_karras_sigmas=use_karras_sigmas,
                                 sigma_min=sigma_min, sigma_max=sigma_max, timestep_spacing=timestep_spacing,
                                 timestep_type=timestep_type, steps_offset=steps_offset)
        if beta_schedule != "scaled_linear" or not use_karras_sigmas or prediction_type != "v_prediction" \
                or timestep_type != "continuous":
            raise NotImplementedError("only the SVD configuration (scaled_linear, Karras, continuous v-prediction)")
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        acp = torch.cumprod(1.0 - betas, dim=0)
        sig = ((1 - acp) / acp) ** 0.5
        self.sigmas = torch.cat([sig.flip(0), torch.zeros(1)])
        self.timesteps = None
        self.num_inference_steps = None
        self._step_index = None

    @classmethod
    def from_config(cls, config) -> "EulerDiscreteScheduler":
        """``config``: a dict or the path of a diffusers scheduler_config.json (e.g. <hub folder>/scheduler/).  Keys this
        class does not take (``_class_name``, ``_diffusers_version``, ``trained_betas``: null, ``clip_sample`` ...) are
        ignored; values it cannot honour raise in ``__init__``."""
        if isinstance(config, str):
            import json
            with open(config) as f:
                config = json.load(f)
        import inspect
        accepted = set(inspect.signature(cls.__init__).parameters) - {"self"}
        if config.get("trained_betas") is not None:
            raise NotImplementedError("trained_betas")
        return cls(**{k: v for k, v in config.items() if k in accepted})

    @property
    def step_index(self):
        return self._step_index

    @property
    def init_noise_sigma(self):
        m = self.sigmas.max()
        return m if self.config.timestep_spacing in ("linspace", "trailing") else (m ** 2 + 1) ** 0.5

    def set_timesteps(self, num_inference_steps: int, device=None):
        self.num_inference_steps = num_inference_steps
        ramp = np.linspace(0, 1, num_inference_steps)
        lo, hi = self.config.sigma_min ** (1 / 7.0), self.config.sigma_max ** (1 / 7.0)
        sig = torch.from_numpy((hi + ramp * (lo - hi)) ** 7.0).to(torch.float32)
        self.timesteps = (0.25 * sig.log()).to(device=device)
        self.sigmas = torch.cat([sig, torch.zeros(1)]).to(device=device)
        self._step_index = None

    def _init_step_index(self, timestep):
        idx = (self.timesteps == torch.as_tensor(timestep).to(self.timesteps.device)).nonzero()
        self._step_index = int((idx[1] if len(idx) > 1 else idx[0]).item())

    def scale_model_input(self, sample: torch.Tensor, timestep) -> torch.Tensor:
        if self._step_index is None:
            self._init_step_index(timestep)
        sigma = self.sigmas[self._step_index]
        return sample / ((sigma ** 2 + 1) ** 0.5)

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, return_dict: bool = True, **_):
        if self._step_index is None:
            self._init_step_index(timestep)
        sample = sample.to(torch.float32)
        sigma = self.sigmas[self._step_index]
        x0 = model_output * (-sigma / (sigma ** 2 + 1) ** 0.5) + sample / (sigma ** 2 + 1)
        prev = (sample + (sample - x0) / sigma * (self.sigmas[self._step_index + 1] - sigma)).to(model_output.dtype)
        self._step_index += 1
        if not return_dict:
            return (prev,)
        return EulerDiscreteSchedulerOutput(prev_sample=prev, pred_original_sample=x0)
