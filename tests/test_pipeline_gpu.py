"""GPU: the drop-in pipelines (VGL and VL ``__call__`` surface) with stub VAE / CLIP: end-to-end plumbing,
``output_type``, ``latents=``, callbacks, and agreement with the oracle loop fed the same request constants."""
import numpy as np
import pytest
import torch

from tests.parity_common import build_pair, err_stats
from tests.stubs import StubCLIPVision, StubTextEncoder, StubVAE

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def parts():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    p_unet, p_cn, o_unet, o_cn = build_pair("tiny_vgl", torch.float16, "cuda:0", True)
    vae, clip, txt = StubVAE().cuda().half(), StubCLIPVision().cuda().half(), StubTextEncoder().cuda().half()
    return p_unet, p_cn, o_unet, o_cn, vae, clip, txt


def _request():
    g = torch.Generator().manual_seed(11)
    image = torch.rand(1, 3, 64, 128, generator=g)                     # [0,1] tensor image
    cond = torch.rand(4, 3, 64, 128, generator=g).numpy().astype(np.float32)
    ids = torch.randint(0, 100, (1, 8), generator=g)
    return image, cond, ids


@torch.no_grad()
def test_vgl_pipeline_latent_output_matches_oracle_loop(parts):
    from oracle.scheduler import EulerDiscreteScheduler as OSched, denoise_loop
    from this_and_that_vdm_amd.svd import EulerDiscreteScheduler, StableVideoDiffusionControlNetPipeline
    p_unet, p_cn, o_unet, o_cn, vae, clip, txt = parts
    pipe = StableVideoDiffusionControlNetPipeline.from_pretrained(None, vae=vae, image_encoder=clip, unet=p_unet,
                                                                  scheduler=EulerDiscreteScheduler())
    pipe.set_progress_bar_config(disable=True)
    image, cond, ids = _request()
    lat0 = torch.randn(1, 4, 4, 8, 16, generator=torch.Generator().manual_seed(5))
    seen = []
    out = pipe(image.cuda(), cond, p_cn, prompt=ids.cuda(), use_text=True, text_encoder=txt, height=64, width=128, num_frames=4,
               num_inference_steps=3, fps=7, motion_bucket_id=200, noise_aug_strength=0.0, latents=lat0.clone(),
               output_type="latent", guess_mode=False, generator=torch.Generator().manual_seed(1),
               callback_on_step_end=lambda p, i, t, kw: seen.append((i, kw["latents"].shape)) or {})
    lat = out.frames
    assert lat.shape == (1, 4, 4, 8, 16) and len(seen) == 3 and pipe.num_timesteps == 3
    # same request constants through the oracle loop (constants come from the pipeline's own encoders)
    ehs = pipe.encode_clip(image.cuda(), ids.cuda(), True, txt, "cuda", 1, True).float().cpu()
    assert ehs.shape == (2, 5, 64) and float(ehs[0].abs().max()) == 0.0
    img = pipe.image_processor.preprocess(image, 64, 128)
    il = pipe._encode_vae_image(img.cuda().half(), "cuda", 1, True).float().cpu().unsqueeze(1).repeat(1, 4, 1, 1, 1)
    ges = vae.encode(torch.from_numpy(cond).cuda().half()).latent_dist.mode().float().cpu()
    sched = OSched()
    sched.set_timesteps(3)
    ref = denoise_loop(o_unet, o_cn, sched, lat0 * sched.init_noise_sigma, il, ehs, torch.tensor([[6.0, 200.0, 0.0]] * 2), ges,
                       torch.linspace(1, 3, 4).view(1, 4, 1, 1, 1), num_inference_steps=3)
    s = err_stats(lat, ref)
    print("pipeline vs oracle loop:", s)
    assert s["rel_l2"] <= 1e-2, s
    # control-guidance window (reference :611-617): 3 steps, end = 0.5 -> the ControlNet acts on step 0 only
    win = pipe(image.cuda(), cond, p_cn, prompt=ids.cuda(), use_text=True, text_encoder=txt, height=64, width=128, num_frames=4,
               num_inference_steps=3, fps=7, motion_bucket_id=200, noise_aug_strength=0.0, latents=lat0.clone(),
               output_type="latent", guess_mode=False, generator=torch.Generator().manual_seed(1),
               control_guidance_start=0.0, control_guidance_end=0.5).frames
    ref_w = denoise_loop(o_unet, o_cn, sched, lat0 * sched.init_noise_sigma, il, ehs, torch.tensor([[6.0, 200.0, 0.0]] * 2), ges,
                         torch.linspace(1, 3, 4).view(1, 4, 1, 1, 1), num_inference_steps=3, control_guidance_end=0.5)
    s = err_stats(win, ref_w)
    print("windowed pipeline vs oracle loop:", s)
    assert s["rel_l2"] <= 1e-2 and not torch.equal(win, lat), s


@torch.no_grad()
def test_vgl_pipeline_instructpix2pix_matches_oracle_loop(parts):
    """use_instructpix2pix=True through the drop-in __call__: constants in the reference's 3-way order, oracle loop beside."""
    from oracle.scheduler import EulerDiscreteScheduler as OSched, denoise_loop
    from this_and_that_vdm_amd.svd import EulerDiscreteScheduler, StableVideoDiffusionControlNetPipeline
    p_unet, p_cn, o_unet, o_cn, vae, clip, txt = parts
    pipe = StableVideoDiffusionControlNetPipeline.from_pretrained(None, vae=vae, image_encoder=clip, unet=p_unet,
                                                                  scheduler=EulerDiscreteScheduler())
    pipe.set_progress_bar_config(disable=True)
    image, cond, ids = _request()
    lat0 = torch.randn(1, 4, 4, 8, 16, generator=torch.Generator().manual_seed(5))
    lat = pipe(image.cuda(), cond, p_cn, prompt=ids.cuda(), use_text=True, text_encoder=txt, height=64, width=128, num_frames=4,
               num_inference_steps=3, fps=7, motion_bucket_id=200, noise_aug_strength=0.0, latents=lat0.clone(),
               output_type="latent", guess_mode=False, use_instructpix2pix=True, image_guidance_scale=1.5).frames
    ehs = pipe.encode_clip(image.cuda(), ids.cuda(), True, txt, "cuda", 1, True, True).float().cpu()
    assert ehs.shape == (3, 5, 64) and float(ehs[1:].abs().max()) == 0.0 and float(ehs[0].abs().max()) > 0.0
    img = pipe.image_processor.preprocess(image, 64, 128)
    il = pipe._encode_vae_image(img.cuda().half(), "cuda", 1, True, True).float().cpu()
    assert il.shape[0] == 3 and torch.equal(il[0], il[1]) and float(il[2].abs().max()) == 0.0
    il = il.unsqueeze(1).repeat(1, 4, 1, 1, 1)
    ges = vae.encode(torch.from_numpy(cond).cuda().half()).latent_dist.mode().float().cpu()
    sched = OSched()
    sched.set_timesteps(3)
    ref = denoise_loop(o_unet, o_cn, sched, lat0 * sched.init_noise_sigma, il, ehs, torch.tensor([[6.0, 200.0, 0.0]] * 3), ges,
                       torch.linspace(1, 3, 4).view(1, 4, 1, 1, 1), num_inference_steps=3, use_instructpix2pix=True,
                       image_guidance_scale=1.5)
    s = err_stats(lat, ref)
    print("instructpix2pix pipeline vs oracle loop:", s)
    assert s["rel_l2"] <= 1e-2, s


@torch.no_grad()
def test_vl_pipeline_decodes_frames(parts):
    from this_and_that_vdm_amd.svd import StableVideoDiffusionPipeline
    p_unet, _, _, _, vae, clip, _ = parts
    pipe = StableVideoDiffusionPipeline.from_pretrained(None, vae=vae, image_encoder=clip, unet=p_unet)
    pipe.set_progress_bar_config(disable=True)
    image, _, _ = _request()
    # VL without text: the context is the single CLIP image token (S = 1)
    frames = pipe(image.cuda(), height=64, width=128, num_frames=4, num_inference_steps=2, output_type="np",
                  generator=torch.Generator().manual_seed(2)).frames
    assert frames.shape == (1, 4, 64, 128, 3) and np.isfinite(frames).all()
    pil = pipe(image.cuda(), height=64, width=128, num_frames=4, num_inference_steps=2, output_type="pil",
               generator=torch.Generator().manual_seed(2)).frames
    assert len(pil) == 1 and len(pil[0]) == 4 and pil[0][0].size == (128, 64)


def test_argument_errors(parts):
    from this_and_that_vdm_amd.svd import StableVideoDiffusionControlNetPipeline
    p_unet, p_cn, _, _, vae, clip, _ = parts
    pipe = StableVideoDiffusionControlNetPipeline.from_pretrained(None, vae=vae, image_encoder=clip, unet=p_unet)
    image, cond, _ = _request()
    with pytest.raises(ValueError, match="divisible by 8"):
        pipe(image.cuda(), cond, p_cn, height=60, width=128, num_frames=4, guess_mode=False)
    with pytest.raises(NotImplementedError, match="guess_mode"):
        pipe(image.cuda(), cond, p_cn, height=64, width=128, num_frames=4, guess_mode=True)
    with pytest.raises(ValueError, match="window"):
        pipe(image.cuda(), cond, p_cn, height=64, width=128, num_frames=4, guess_mode=False, control_guidance_start=0.8,
             control_guidance_end=0.2)


@torch.no_grad()
def test_hub_folder_round_trip_runs_like_inference_py(parts, tmp_path):
    """test_code/inference.py:322-381,171-180 on a local hub folder: unet/ and controlnet/ through from_pretrained(subfolder=...),
    the pipeline through from_pretrained(folder, vae=, image_encoder=, unet=) picking feature_extractor/ and scheduler/ up from
    the folder.  The reloaded models must reproduce the in-memory models BIT FOR BIT (checkpoint I/O is lossless), also from
    a sharded checkpoint."""
    import json
    import os
    from safetensors.torch import save_file
    from this_and_that_vdm_amd.svd import (ControlNetModel, StableVideoDiffusionControlNetPipeline,
                                           UNetSpatioTemporalConditionModel)
    p_unet, p_cn, _, _, vae, clip, txt = parts
    root = str(tmp_path)
    p_unet.save_pretrained(os.path.join(root, "unet"))
    p_cn.save_pretrained(os.path.join(root, "controlnet"))
    os.makedirs(os.path.join(root, "feature_extractor"))
    os.makedirs(os.path.join(root, "scheduler"))
    json.dump({"image_mean": [0.48145466, 0.4578275, 0.40821073], "image_std": [0.26862954, 0.26130258, 0.27577711]},
              open(os.path.join(root, "feature_extractor", "preprocessor_config.json"), "w"))
    json.dump({"_class_name": "EulerDiscreteScheduler", "beta_start": 0.00085, "beta_end": 0.012, "beta_schedule": "scaled_linear",
               "num_train_timesteps": 1000, "prediction_type": "v_prediction", "interpolation_type": "linear",
               "use_karras_sigmas": True, "sigma_min": 0.002, "sigma_max": 700.0, "timestep_spacing": "leading",
               "timestep_type": "continuous", "steps_offset": 1}, open(os.path.join(root, "scheduler", "scheduler_config.json"), "w"))
    # a sharded copy of the unet next to it
    sd = {k: v.detach().cpu().contiguous() for k, v in p_unet.state_dict().items()}
    shard_dir = os.path.join(root, "unet_sharded")
    os.makedirs(shard_dir)
    json.dump({k: (list(v) if isinstance(v, tuple) else v) for k, v in p_unet.config.items()}, open(os.path.join(shard_dir, "config.json"), "w"))
    keys = sorted(sd)
    parts_ = {"diffusion_pytorch_model-00001-of-00002.safetensors": keys[: len(keys) // 2],
              "diffusion_pytorch_model-00002-of-00002.safetensors": keys[len(keys) // 2:]}
    for name, ks in parts_.items():
        save_file({k: sd[k] for k in ks}, os.path.join(shard_dir, name))
    json.dump({"weight_map": {k: n for n, ks in parts_.items() for k in ks}},
              open(os.path.join(shard_dir, "diffusion_pytorch_model.safetensors.index.json"), "w"))

    unet2 = UNetSpatioTemporalConditionModel.from_pretrained(root, subfolder="unet", low_cpu_mem_usage=True).to("cuda", torch.float16)
    unet3 = UNetSpatioTemporalConditionModel.from_pretrained(root, subfolder="unet_sharded", torch_dtype=torch.float16).to("cuda")
    cn2 = ControlNetModel.from_pretrained(root, subfolder="controlnet", low_cpu_mem_usage=True).to("cuda", torch.float16)
    image, cond, ids = _request()
    lat0 = torch.randn(1, 4, 4, 8, 16, generator=torch.Generator().manual_seed(5))
    call = dict(prompt=ids.cuda(), use_text=True, text_encoder=txt, height=64, width=128, num_frames=4, num_inference_steps=2, fps=7,
                motion_bucket_id=200, noise_aug_strength=0.0, output_type="latent", guess_mode=False)
    outs = []
    for unet, cn in ((p_unet, p_cn), (unet2, cn2), (unet3, cn2)):
        pipe = StableVideoDiffusionControlNetPipeline.from_pretrained(root, vae=vae, image_encoder=clip, unet=unet)
        pipe.set_progress_bar_config(disable=True)
        assert pipe.feature_extractor is not None and pipe.scheduler.config.sigma_max == 700.0
        outs.append(pipe(image.cuda(), cond, cn, latents=lat0.clone(), **call).frames.clone())
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0], outs[1]), "models reloaded from the folder must reproduce the in-memory models bit for bit"
    assert torch.equal(outs[0], outs[2]), "... also from a sharded checkpoint"


@torch.no_grad()
def test_vgl_pipeline_call_produces_frames_through_the_native_vae():
    """`pipe(image, condition_img, controlnet, ..., output_type="np")` END TO END with the native temporal VAE as the pipeline's
    one `vae=` object (test_code/inference.py:169-176,241-257): its encode() is the caller's stock encoder (a stub here), the
    denoise loop is the fused HIP loop, decode_latents (reference :257-283) runs the native decoder in chunks.  In the
    reference-precision mode (TT_F32) the frames must agree with the oracle loop + oracle/vae.py on the same request constants
    to the north-star tolerance on every pixel."""
    from oracle import vae as ov
    from oracle.scheduler import EulerDiscreteScheduler as OSched, denoise_loop
    from tests.parity_common import assert_north_star
    from this_and_that_vdm_amd.svd import EulerDiscreteScheduler, StableVideoDiffusionControlNetPipeline
    from this_and_that_vdm_amd.svd.autoencoder_kl_temporal_decoder import AutoencoderKLTemporalDecoder
    from this_and_that_vdm_amd.svd.pipeline_utils import tensor2vid
    from this_and_that_vdm_amd.utils.synthetic import fill_parameters_
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    dt = torch.float32
    p_unet, p_cn, o_unet, o_cn = build_pair("tiny_vgl", dt, "cuda:0", True)
    cfg = dict(block_out_channels=(32, 64, 64, 64), layers_per_block=2)
    o_vae = ov.AutoencoderKLTemporalDecoder(**cfg).eval()
    fill_parameters_(o_vae, "vae.", round_to=dt)
    stock, clip, txt = StubVAE().cuda(), StubCLIPVision().cuda(), StubTextEncoder().cuda()
    vae = AutoencoderKLTemporalDecoder(**cfg, encoder=stock).eval()
    vae.load_state_dict(o_vae.state_dict())
    vae = vae.to("cuda:0")
    vae.compute_dtype = torch.float32
    pipe = StableVideoDiffusionControlNetPipeline.from_pretrained(None, vae=vae, image_encoder=clip, unet=p_unet,
                                                                  scheduler=EulerDiscreteScheduler())
    pipe.set_progress_bar_config(disable=True)
    image, cond, ids = _request()
    lat0 = torch.randn(1, 4, 4, 8, 16, generator=torch.Generator().manual_seed(5))
    call = dict(prompt=ids.cuda(), use_text=True, text_encoder=txt, height=64, width=128, num_frames=4, num_inference_steps=3, fps=7,
                motion_bucket_id=200, noise_aug_strength=0.0, guess_mode=False, decode_chunk_size=3)
    frames = pipe(image.cuda(), cond, p_cn, latents=lat0.clone(), output_type="np", **call).frames
    assert frames.shape == (1, 4, 64, 128, 3) and np.isfinite(frames).all()
    # the oracle on the same request constants: loop -> / scaling_factor -> chunked decode -> the same post-processing
    ehs = pipe.encode_clip(image.cuda(), ids.cuda(), True, txt, "cuda", 1, True).float().cpu()
    img = pipe.image_processor.preprocess(image, 64, 128)
    il = pipe._encode_vae_image(img.cuda(), "cuda", 1, True).float().cpu().unsqueeze(1).repeat(1, 4, 1, 1, 1)
    ges = stock.encode(torch.from_numpy(cond).cuda().half().float()).latent_dist.mode().float().cpu()   # quirk Q7: fp16 gesture frames
    sched = OSched()
    sched.set_timesteps(3)
    lat = denoise_loop(o_unet, o_cn, sched, lat0 * sched.init_noise_sigma, il, ehs, torch.tensor([[6.0, 200.0, 0.0]] * 2), ges,
                       torch.linspace(1, 3, 4).view(1, 4, 1, 1, 1), num_inference_steps=3)
    z = lat.flatten(0, 1) / o_vae.scaling_factor
    ref = torch.cat([o_vae.decode(z[:3], num_frames=3), o_vae.decode(z[3:], num_frames=1)], 0)
    ref = ref.reshape(1, 4, *ref.shape[1:]).permute(0, 2, 1, 3, 4)
    ref_frames = tensor2vid(ref, pipe.image_processor, output_type="np")
    # latents first (the loop), then the frames (loop + native decoder + post-processing, values in [0, 1])
    got_lat = pipe(image.cuda(), cond, p_cn, latents=lat0.clone(), output_type="latent", **call).frames
    assert_north_star(got_lat, lat, "pipeline latents (TT_F32) vs oracle loop")
    st = err_stats(torch.from_numpy(frames), torch.from_numpy(ref_frames))
    print("pipeline frames through the native VAE vs oracle loop + oracle/vae.py:", st)
    assert_north_star(torch.from_numpy(frames), torch.from_numpy(ref_frames), "pipeline frames through the native VAE (TT_F32)")
    # and the same call in fp16 storage (the reference's inference dtype) runs end to end: finite frames close to the fp32 ones
    vae16 = AutoencoderKLTemporalDecoder(**cfg, encoder=StubVAE().cuda().half()).eval()
    vae16.load_state_dict(o_vae.state_dict())
    vae16 = vae16.to("cuda:0", torch.float16)
    p16, c16, _, _ = build_pair("tiny_vgl", torch.float16, "cuda:0", True)
    pipe16 = StableVideoDiffusionControlNetPipeline.from_pretrained(None, vae=vae16, image_encoder=StubCLIPVision().cuda().half(), unet=p16,
                                                                    scheduler=EulerDiscreteScheduler())
    pipe16.set_progress_bar_config(disable=True)
    call16 = dict(call, text_encoder=StubTextEncoder().cuda().half())
    f16 = pipe16(image.cuda(), cond, c16, latents=lat0.clone(), output_type="np", **call16).frames
    assert f16.shape == frames.shape and np.isfinite(f16).all()
    assert vae16.dtype == torch.float16                      # force_upcast round trip (config.force_upcast = True) restored the dtype
    print("fp16 frames vs TT_F32 frames: max abs", float(np.abs(f16 - frames).max()))
    assert float(np.abs(f16 - frames).max()) <= 0.05
