"""CPU: diffusers-folder save/load round trip, config surface and from_unet semantics of the drop-in classes
(reference call sites: test_code/inference.py:331-336,373-378; temporal_controlnet.py:311-339)."""
import json
import os

import pytest
import torch

from this_and_that_vdm_amd.svd import ControlNetModel, UNetSpatioTemporalConditionModel
from this_and_that_vdm_amd.utils.synthetic import fill_parameters_

KW = dict(block_out_channels=(64, 64, 64, 64), num_attention_heads=(1, 1, 1, 1), cross_attention_dim=32, num_frames=3)


def test_save_load_roundtrip(tmp_path):
    m = UNetSpatioTemporalConditionModel(**KW)
    fill_parameters_(m, "unet.")
    m.save_pretrained(os.path.join(tmp_path, "unet"))
    cfg = json.load(open(os.path.join(tmp_path, "unet", "config.json")))
    assert cfg["_class_name"] == "UNetSpatioTemporalConditionModel" and cfg["num_frames"] == 3
    m2 = UNetSpatioTemporalConditionModel.from_pretrained(str(tmp_path), subfolder="unet", low_cpu_mem_usage=True)
    assert m2.config.block_out_channels == (64, 64, 64, 64) and m2.config.in_channels == 8
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    assert m2.dtype == torch.float32 and m2.add_embedding.linear_1.in_features == 768
    with pytest.raises(OSError):
        UNetSpatioTemporalConditionModel.from_pretrained(str(tmp_path), subfolder="missing")


def test_from_unet_copies_encoder_but_not_add_embedding():
    kw = dict(KW)
    # ControlNetModel.from_unet builds with ITS OWN full-size defaults (quirk Q2), too slow for a CPU unit test:
    # exercise the same load_state_dict calls on matching tiny shapes
    u = UNetSpatioTemporalConditionModel(**kw)
    fill_parameters_(u, "unet.")
    kw.pop("num_frames")
    c = ControlNetModel(**kw)
    c.time_embedding.load_state_dict(u.time_embedding.state_dict())
    c.down_blocks.load_state_dict(u.down_blocks.state_dict())
    c.mid_block.load_state_dict(u.mid_block.state_dict())
    assert torch.equal(c.down_blocks[0].resnets[0].spatial_res_block.conv1.weight, u.down_blocks[0].resnets[0].spatial_res_block.conv1.weight)
    # zero-initialised pieces (temporal_controlnet.py:203-205,254-297)
    assert float(c.conv_in_concat.weight.abs().max()) == 0.0
    assert all(float(z.weight.abs().max()) == 0.0 for z in c.controlnet_down_blocks) and len(c.controlnet_down_blocks) == 12
    assert c.config.num_attention_heads == (1, 1, 1, 1)


def test_constructor_argument_checks():
    with pytest.raises(ValueError, match="same number of `block_out_channels`"):
        UNetSpatioTemporalConditionModel(block_out_channels=(64, 64))
    with pytest.raises(ValueError, match="does not exist"):
        UNetSpatioTemporalConditionModel(down_block_types=("Nope",) * 4, **KW)
    with pytest.raises(NotImplementedError, match="head_dim"):
        UNetSpatioTemporalConditionModel(block_out_channels=(32, 32, 32, 32), num_attention_heads=(1, 1, 1, 1))


def test_cpu_forward_fails_loudly():
    m = UNetSpatioTemporalConditionModel(**KW)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 3, 8, 8, 8), 1.0, torch.zeros(1, 2, 32), torch.zeros(1, 3))


def test_from_unet_is_callable_on_meta_device():
    """The real classmethod (temporal_controlnet.py:311-339): ControlNet built with ITS OWN defaults (heads (5,10,20,20),
    quirk Q2) and fed the UNet's time/down/mid weights by name -- shapes are head-independent, so the default-ctor UNet
    (heads (5,10,10,20)) loads.  Meta tensors: no 2.2 B-parameter allocation."""
    with torch.device("meta"):
        u = UNetSpatioTemporalConditionModel()
        c = ControlNetModel.from_unet(u)
    assert isinstance(c, ControlNetModel) and c.config.num_attention_heads == (5, 10, 20, 20)
    assert u.config.num_attention_heads == (5, 10, 10, 20)
    assert sum(p.numel() for p in c.parameters()) == 680_946_577
    assert c.conv_in_concat.weight.shape == (320, 12, 3, 3) and len(c.controlnet_down_blocks) == 12
    su, sc = u.down_blocks.state_dict(), c.down_blocks.state_dict()
    assert su.keys() == sc.keys() and all(su[k].shape == sc[k].shape for k in su)


def test_sharded_and_variant_checkpoints_load(tmp_path):
    """diffusers writes big models as shards + diffusion_pytorch_model.safetensors.index.json, and fp16 variants as
    diffusion_pytorch_model.fp16.safetensors: both must load to the same state dict."""
    from safetensors.torch import save_file
    m = UNetSpatioTemporalConditionModel(**KW)
    fill_parameters_(m, "unet.")
    sd = {k: v.contiguous() for k, v in m.state_dict().items()}
    folder = os.path.join(tmp_path, "unet")
    m.save_pretrained(folder)
    os.remove(os.path.join(folder, "diffusion_pytorch_model.safetensors"))
    keys = sorted(sd)
    half = len(keys) // 2
    shards = {"diffusion_pytorch_model-00001-of-00002.safetensors": keys[:half],
              "diffusion_pytorch_model-00002-of-00002.safetensors": keys[half:]}
    for name, ks in shards.items():
        save_file({k: sd[k] for k in ks}, os.path.join(folder, name), metadata={"format": "pt"})
    json.dump({"metadata": {}, "weight_map": {k: n for n, ks in shards.items() for k in ks}},
              open(os.path.join(folder, "diffusion_pytorch_model.safetensors.index.json"), "w"))
    m2 = UNetSpatioTemporalConditionModel.from_pretrained(str(tmp_path), subfolder="unet")
    assert all(torch.equal(sd[k], v) for k, v in m2.state_dict().items())
    # a shard named by the index but absent on disk is an error, not a partial load
    os.remove(os.path.join(folder, "diffusion_pytorch_model-00002-of-00002.safetensors"))
    with pytest.raises(OSError, match="missing"):
        UNetSpatioTemporalConditionModel.from_pretrained(str(tmp_path), subfolder="unet")
    # variant="fp16"
    save_file({k: v.half() for k, v in sd.items()}, os.path.join(folder, "diffusion_pytorch_model.fp16.safetensors"))
    m3 = UNetSpatioTemporalConditionModel.from_pretrained(str(tmp_path), subfolder="unet", variant="fp16", torch_dtype=torch.float16)
    assert m3.dtype == torch.float16 and all(torch.equal(sd[k].half(), v) for k, v in m3.state_dict().items())


def test_pipeline_from_pretrained_reads_the_local_hub_folder(tmp_path):
    """test_code/inference.py:171-178 passes vae / image_encoder / unet and lets from_pretrained find feature_extractor/ and
    scheduler/ in the folder.  The CLIP normalisation must be the reference formula (:134-156): (x - mean) / std on the
    antialias-resized [0,1] image -- and it must never be skipped silently."""
    from tests.stubs import StubCLIPVision, StubVAE
    from this_and_that_vdm_amd.svd import StableVideoDiffusionControlNetPipeline
    from this_and_that_vdm_amd.svd.pipeline_utils import CLIPFeatureExtractor, resize_with_antialiasing
    root = str(tmp_path)
    u = UNetSpatioTemporalConditionModel(**KW)
    u.save_pretrained(os.path.join(root, "unet"))
    os.makedirs(os.path.join(root, "feature_extractor"))
    os.makedirs(os.path.join(root, "scheduler"))
    mean, std = [0.5, 0.4, 0.3], [0.2, 0.25, 0.3]          # deliberately NOT the CLIP defaults: proves the file is read
    json.dump({"image_processor_type": "CLIPImageProcessor", "do_normalize": True, "image_mean": mean, "image_std": std,
               "size": {"shortest_edge": 224}}, open(os.path.join(root, "feature_extractor", "preprocessor_config.json"), "w"))
    json.dump({"_class_name": "EulerDiscreteScheduler", "_diffusers_version": "0.24.0", "beta_start": 0.00085, "beta_end": 0.012,
               "beta_schedule": "scaled_linear", "num_train_timesteps": 1000, "prediction_type": "v_prediction",
               "interpolation_type": "linear", "use_karras_sigmas": True, "sigma_min": 0.002, "sigma_max": 650.0,
               "timestep_spacing": "leading", "timestep_type": "continuous", "steps_offset": 1, "trained_betas": None,
               "clip_sample": False, "set_alpha_to_one": False, "skip_prk_steps": True},
              open(os.path.join(root, "scheduler", "scheduler_config.json"), "w"))
    seen = {}

    class Clip(StubCLIPVision):
        def forward(self, image):
            seen["pixel_values"] = image.detach().clone()
            return super().forward(image)

    pipe = StableVideoDiffusionControlNetPipeline.from_pretrained(root, vae=StubVAE(), image_encoder=Clip(), unet=u)
    assert pipe.feature_extractor.image_mean == mean and pipe.scheduler.config.sigma_max == 650.0
    import numpy as np
    import PIL.Image
    arr = (np.random.default_rng(0).random((48, 80, 3)) * 255).astype("uint8")
    ehs = pipe.encode_clip(PIL.Image.fromarray(arr), None, False, None, "cpu", 1, True)
    x = torch.from_numpy(arr.astype("float32") / 255.0).permute(2, 0, 1)[None]
    want = (resize_with_antialiasing(x * 2.0 - 1.0, (224, 224)) + 1.0) / 2.0
    want = (want - torch.tensor(mean).view(1, 3, 1, 1)) / torch.tensor(std).view(1, 3, 1, 1)
    torch.testing.assert_close(seen["pixel_values"], want, rtol=1e-6, atol=1e-6)
    assert ehs.shape[0] == 2 and float(ehs[0].abs().max()) == 0.0
    # unet can come from the folder too
    pipe2 = StableVideoDiffusionControlNetPipeline.from_pretrained(root, vae=StubVAE(), image_encoder=Clip())
    assert pipe2.unet.config.block_out_channels == (64, 64, 64, 64)
    # no folder: CLIP's published constants, never "no normalisation"
    pipe3 = StableVideoDiffusionControlNetPipeline.from_pretrained(None, vae=StubVAE(), image_encoder=Clip(), unet=u)
    assert pipe3.feature_extractor.image_mean == CLIPFeatureExtractor().image_mean
    pipe3.feature_extractor = None
    with pytest.raises(RuntimeError, match="feature_extractor"):
        pipe3.encode_clip(PIL.Image.fromarray(arr), None, False, None, "cpu", 1, True)


def test_clip_feature_extractor_matches_transformers():
    """Same numbers as transformers.CLIPImageProcessor for the call the reference makes (:145-152)."""
    transformers = pytest.importorskip("transformers")
    from this_and_that_vdm_amd.svd.pipeline_utils import CLIPFeatureExtractor
    x = torch.rand(2, 3, 224, 224, generator=torch.Generator().manual_seed(0))
    ours = CLIPFeatureExtractor()(images=x, do_normalize=True, do_center_crop=False, do_resize=False, do_rescale=False,
                                  return_tensors="pt").pixel_values
    try:
        theirs = transformers.CLIPImageProcessor()(images=x, do_normalize=True, do_center_crop=False, do_resize=False,
                                                   do_rescale=False, return_tensors="pt").pixel_values
    except Exception as e:          # an image-processor backend (PIL/torchvision) this image lacks
        pytest.skip(f"transformers CLIPImageProcessor unusable here: {e}")
    torch.testing.assert_close(ours, theirs, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_folded_layernorm_weights_keep_zero_row_sums_after_rounding(dtype):
    """packing.fold_layernorm + zero_sum_round: Linear(LN(x)) == rstd * (x W''^T) + b' for the ROUNDED W'' and rows of x
    with a large mean -- the rounded rows must still sum to ~0 or the row mean of x leaks into the output."""
    from this_and_that_vdm_amd.packing import fold_layernorm, zero_sum_round
    g = torch.Generator().manual_seed(0)
    k, n = 320, 512
    w, b = torch.randn(n, k, generator=g) * k ** -0.5, torch.randn(n, generator=g)
    gamma, beta = torch.rand(k, generator=g) + 0.5, torch.randn(k, generator=g)
    wf, bf = fold_layernorm(w, b, gamma, beta)
    plain, zs = wf.to(dtype), zero_sum_round(wf, dtype)
    assert zs.dtype == dtype
    r_plain, r_zs = plain.double().sum(1).abs().max(), zs.double().sum(1).abs().max()
    assert r_zs <= 1e-5 and r_zs < r_plain / 100, (float(r_plain), float(r_zs))
    # corrections go to elements whose own rounding error has the sign of the correction first: those land on their OTHER
    # rounding neighbour (< 1 ulp from the true value, against <= 0.5 ulp for plain rounding)
    mant, emin = (7, -126) if dtype == torch.bfloat16 else (10, -14)
    _, e = torch.frexp(plain.double().abs())
    ulp = 2.0 ** (torch.clamp(e - 1, min=emin) - mant).double()          # spacing in the binade of the rounded value
    dev_ulps = (zs.double() - wf.double()).abs() / ulp
    # (the few elements taken by the second descent -- rows whose fine binades hold too few elements rounded the right way; ~0.1 % -- may
    # sit up to 1.5 ulp off; everything else is within 1 ulp)
    assert dev_ulps.max() <= 1.5 + 1e-9 and (dev_ulps > 1.0 + 1e-9).double().mean() <= 2e-3, (float(dev_ulps.max()), float((dev_ulps > 1.0).double().mean()))
    x = torch.randn(64, k, generator=g) + 25.0                   # row mean = 25 sigma
    ref = torch.nn.functional.linear(torch.nn.functional.layer_norm(x, (k,), gamma, beta, 1e-5), w, b)
    rstd = 1.0 / torch.sqrt(x.var(1, unbiased=False, keepdim=True) + 1e-5)
    err = lambda wq: float(((rstd * (x.double() @ wq.double().T) + bf) - ref).abs().max())
    assert err(zs) < err(plain) / 10 and err(zs) < 0.05, (err(zs), err(plain))


def test_query_row_permutation_matches_the_mfma_operand_slots():
    """packing.permute_q_rows (the fused query projection of tt_attention, include/ttvdm.h TtAttnArgs.qx): the projection's MFMA
    accumulators are used as the Q^T operand of the score product.  In the accumulator layout lane half `hi` of fragment j holds
    output columns n = 32 j + 8 g + 4 hi + e; registers g = 2t, 2t+1 are the 8 contraction slots s = 4 (g - 2t) + e of key step
    ks = 2j + t, and slot s of half hi must be head dimension 16 ks + 8 hi + s (what the K fragment read supplies).  So row n of
    the permuted weight has to be original row d(n) -- checked for every (j, t, hi, s) of a 64-wide head, plus: involution, a
    permutation inside every 16-row group, applied identically to weight rows and bias entries."""
    import torch
    from this_and_that_vdm_amd.packing import permute_q_rows
    heads, d = 3, 64
    ident = torch.arange(heads * d)
    perm = permute_q_rows(ident)                      # perm[n] = d(n)
    assert torch.equal(permute_q_rows(perm), ident)
    assert torch.equal(perm.view(-1, 16).sort(1).values, ident.view(-1, 16))
    for h in range(heads):
        for j in range(2):
            for t in range(2):
                for hi in range(2):
                    for s_ in range(8):
                        g, e = 2 * t + s_ // 4, s_ % 4
                        n = h * d + 32 * j + 8 * g + 4 * hi + e
                        assert int(perm[n]) == h * d + 16 * (2 * j + t) + 8 * hi + s_, (h, j, t, hi, s_)
    w = torch.randn(heads * d, 40)
    b = torch.randn(heads * d)
    assert torch.equal(permute_q_rows(w), w[perm]) and torch.equal(permute_q_rows(b), b[perm])
