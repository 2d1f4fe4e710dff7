"""Oracle EulerDiscreteScheduler (diffusers==0.25.1 restatement) + the reference denoise loop body.

TEST INFRASTRUCTURE (see oracle/__init__.py).  The scheduler lives in diffusers
(call sites: svd/pipeline_stable_video_diffusion_controlnet.py:336,583-584,632,709);
restated from SURVEY.md A.9 with SVD's shipped scheduler_config.json values
(recalled, not present in /root/reference).  In-repo corroboration of the
v-prediction constants: train_code/train_csvd.py:766,818,902-904.
Known answers pinned in tests: SURVEY.md Appendix E sigma/timestep table.
"""
from __future__ import annotations

import numpy as np
import torch

SVD_SCHEDULER_CONFIG = dict(
    num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
    prediction_type="v_prediction", interpolation_type="linear", use_karras_sigmas=True,
    sigma_min=0.002, sigma_max=700.0, timestep_spacing="leading", timestep_type="continuous", steps_offset=1,
)


class EulerDiscreteScheduler:
    order = 1

    def __init__(self, **cfg):
        c = dict(SVD_SCHEDULER_CONFIG)
        c.update(cfg)
        self.config = c
        assert c["beta_schedule"] == "scaled_linear"
        betas = torch.linspace(c["beta_start"] ** 0.5, c["beta_end"] ** 0.5, c["num_train_timesteps"], dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        sig = ((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5
        self.sigmas = torch.cat([sig.flip(0), torch.zeros(1)])
        self.timesteps = None
        self._step_index = None

    @property
    def init_noise_sigma(self):
        m = self.sigmas.max()
        if self.config["timestep_spacing"] in ("linspace", "trailing"):
            return m
        return (m ** 2 + 1) ** 0.5

    def set_timesteps(self, num_inference_steps: int, device=None):
        c = self.config
        n = num_inference_steps
        assert c["timestep_spacing"] == "leading" and c["use_karras_sigmas"]
        # Karras ramp (rho = 7), float64 numpy then cast to fp32 -- A.9
        ramp = np.linspace(0, 1, n)
        lo, hi = c["sigma_min"] ** (1 / 7.0), c["sigma_max"] ** (1 / 7.0)
        sigmas = (hi + ramp * (lo - hi)) ** 7.0
        sigmas = torch.from_numpy(sigmas).to(dtype=torch.float32, device=device)
        assert c["timestep_type"] == "continuous" and c["prediction_type"] == "v_prediction"
        self.timesteps = torch.Tensor([0.25 * s.log() for s in sigmas]).to(device=device)
        self.sigmas = torch.cat([sigmas, torch.zeros(1, device=sigmas.device)])
        self._step_index = None

    def _init_step_index(self, timestep):
        idx = (self.timesteps == timestep).nonzero()
        self._step_index = (idx[1] if len(idx) > 1 else idx[0]).item()

    def scale_model_input(self, sample, timestep):
        if self._step_index is None:
            self._init_step_index(timestep)
        sigma = self.sigmas[self._step_index]
        return sample / ((sigma ** 2 + 1) ** 0.5)

    def step(self, model_output, timestep, sample):
        if self._step_index is None:
            self._init_step_index(timestep)
        sample = sample.to(torch.float32)
        sigma = self.sigmas[self._step_index]
        sigma_hat = sigma                      # s_churn = 0 -> gamma = 0
        pred_x0 = model_output * (-sigma / (sigma ** 2 + 1) ** 0.5) + (sample / (sigma ** 2 + 1))
        derivative = (sample - pred_x0) / sigma_hat
        dt = self.sigmas[self._step_index + 1] - sigma_hat
        prev = (sample + derivative * dt).to(model_output.dtype)
        self._step_index += 1
        return prev


def controlnet_keep(num_steps, control_guidance_start=0.0, control_guidance_end=1.0):
    """svd/pipeline_stable_video_diffusion_controlnet.py:611-617 for a single ControlNetModel: step i keeps the
    ControlNet (1.0) unless i/n < start or (i+1)/n > end (0.0)."""
    s = control_guidance_start[0] if isinstance(control_guidance_start, (list, tuple)) else control_guidance_start
    e = control_guidance_end[0] if isinstance(control_guidance_end, (list, tuple)) else control_guidance_end
    return [1.0 - float(i / num_steps < s or (i + 1) / num_steps > e) for i in range(num_steps)]


def denoise_loop(unet, controlnet, scheduler, latents, image_latents, encoder_hidden_states, added_time_ids,
                 controlnet_cond, guidance_scale, num_inference_steps=25, conditioning_scale=1.0, return_all=False,
                 control_guidance_start=0.0, control_guidance_end=1.0, use_instructpix2pix=False, image_guidance_scale=7.5):
    """Loop body of svd/pipeline_stable_video_diffusion_controlnet.py:624-720 (VL twin
    svd/pipeline_stable_video_diffusion.py:528-562 when ``controlnet is None``).

    latents [1,F,4,h,w] (already * init_noise_sigma), image_latents [2,F,4,h,w] (uncond zeros first),
    encoder_hidden_states [2,S,1024], added_time_ids [2,3], controlnet_cond [F,4,h,w] (pre-encoded gesture
    latents, loop-invariant -- quirk Q6/Q12), guidance_scale [1,F,1,1,1].
    ``use_instructpix2pix`` (:627-628,656-657,698-702): the constants carry a batch of 3 in the reference's order
    (context: cond, 0, 0 :182-184; image latents: cond, cond, 0 :208-210) and the guidance combines three predictions.
    """
    nb = 3 if use_instructpix2pix else 2
    scheduler.set_timesteps(num_inference_steps)
    keep = controlnet_keep(len(scheduler.timesteps), control_guidance_start, control_guidance_end)
    trace = []
    for i, t in enumerate(scheduler.timesteps):
        x = torch.cat([latents] * nb)
        x = scheduler.scale_model_input(x, t)
        x = torch.cat([x, image_latents], dim=2)
        down = mid = None
        if controlnet is not None:
            cc = torch.cat([controlnet_cond] * nb)
            down, mid = controlnet(x, t, encoder_hidden_states, added_time_ids, controlnet_cond=cc,
                                   conditioning_scale=conditioning_scale * keep[i], guess_mode=False)   # :639-645
        eps = unet(x, t, encoder_hidden_states, added_time_ids,
                   down_block_additional_residuals=down, mid_block_additional_residual=mid)
        if use_instructpix2pix:
            e1, c, u = eps.chunk(3)                 # "1st_frame", cond, uncond (:699)
            eps = u + guidance_scale * (c - u) + image_guidance_scale * (c - e1)
        else:
            u, c = eps.chunk(2)
            eps = u + guidance_scale * (c - u)
        latents = scheduler.step(eps, t, latents)
        if return_all:
            trace.append(latents.clone())
    return (latents, trace) if return_all else latents
