"""GPU: the fused, hipGraph-replayed denoise loop vs the oracle's restatement of the reference loop body
(svd/pipeline_stable_video_diffusion_controlnet.py:624-720) -- ControlNet + UNet + per-frame CFG + Euler, 4 steps.
Also: graph replay == eager launches bit-for-bit, and the fused residual add == the public two-call API.
Tolerance: latents are O(700) at the first steps; relative L2 of the final latents <= 1e-2 (fp16 storage)."""
import pytest
import torch

from tests.parity_common import build_pair, err_stats

pytestmark = pytest.mark.gpu


def _inputs():
    from this_and_that_vdm_amd.utils.synthetic import synthetic_inputs
    return synthetic_inputs(2, 4, 8, 16, ctx_tokens=5, ctx_dim=64, seed=3)


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
@pytest.mark.parametrize("with_cn", [True, False])
@torch.no_grad()
def test_fused_loop_matches_oracle_loop(with_cn, dtype):
    """fp16 storage: relative L2 <= 3e-3; TT_F32 (reference-precision mode): every element of the final latents inside the
    north-star tolerance rtol 1e-3 / atol 1e-4."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle.scheduler import EulerDiscreteScheduler as OSched, denoise_loop
    from tests.parity_common import assert_north_star
    from this_and_that_vdm_amd.svd.denoise import DenoiseLoop
    from this_and_that_vdm_amd.svd.scheduling_euler_discrete import EulerDiscreteScheduler
    p_unet, p_cn, o_unet, o_cn = build_pair("tiny_vgl", dtype, "cuda:0", True)
    if not with_cn:
        p_cn = o_cn = None
    inp = _inputs()
    steps = 4
    ref = denoise_loop(o_unet, o_cn, OSched(), inp["latents"], inp["image_latents"], inp["encoder_hidden_states"],
                       inp["added_time_ids"], inp["gesture_latents"], inp["guidance_scale"], num_inference_steps=steps,
                       conditioning_scale=1.0)
    sched = EulerDiscreteScheduler()
    sched.set_timesteps(steps)
    kw = dict(latents=inp["latents"], image_latents=inp["image_latents"], encoder_hidden_states=inp["encoder_hidden_states"],
              added_time_ids=inp["added_time_ids"], guidance_scale=inp["guidance_scale"], sigmas=sched.sigmas,
              timesteps=sched.timesteps, controlnet_cond=inp["gesture_latents"] if with_cn else None)
    outs = {}
    for graph in (True, False):
        loop = DenoiseLoop(p_unet, p_cn, use_graph=graph).begin(**kw)
        outs[graph] = loop.run().clone()
        torch.cuda.synchronize()
    assert torch.equal(outs[True], outs[False]), "graph replay must equal eager launches"
    # CFG halves as separate concurrent branches: same arithmetic per element, must agree exactly as well
    split = DenoiseLoop(p_unet, p_cn, use_graph=True, split_cfg=True).begin(**kw).run().clone()
    torch.cuda.synchronize()
    assert torch.equal(split, outs[True]), "split-CFG branches must reproduce the joint launch"
    s = err_stats(outs[True], ref)
    print(f"fused loop vs oracle loop ({dtype}):", s)
    if dtype == torch.float32:
        assert_north_star(outs[True].reshape(ref.shape), ref, "final latents of the 4-step fused loop (TT_F32)")
    else:
        assert s["rel_l2"] <= 3e-3 and s["cos"] >= 0.99999, s
    # second request on the same loop object re-uses the captured graph
    loop = DenoiseLoop(p_unet, p_cn, use_graph=True).begin(**kw)
    a = loop.run().clone()
    loop.begin(**kw)
    b = loop.run().clone()
    assert torch.equal(a, b)


@torch.no_grad()
def test_fused_residual_add_equals_public_api():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from this_and_that_vdm_amd.svd.denoise import DenoiseLoop
    from this_and_that_vdm_amd.svd.scheduling_euler_discrete import EulerDiscreteScheduler
    dtype = torch.float16
    p_unet, p_cn, _, _ = build_pair("tiny_vgl", dtype, "cuda:0", True)
    inp = {k: v.cuda() for k, v in _inputs().items()}
    sched = EulerDiscreteScheduler()
    sched.set_timesteps(3)
    sched.sigmas, sched.timesteps = sched.sigmas.cuda(), sched.timesteps.cuda()
    loop = DenoiseLoop(p_unet, p_cn, use_graph=False).begin(
        latents=inp["latents"], image_latents=inp["image_latents"], encoder_hidden_states=inp["encoder_hidden_states"],
        added_time_ids=inp["added_time_ids"], guidance_scale=inp["guidance_scale"], sigmas=sched.sigmas,
        timesteps=sched.timesteps, controlnet_cond=inp["gesture_latents"])
    loop.step()
    fused = loop.result().clone()
    # the same step through the reference-shaped public API (two forward() calls + scheduler)
    t = sched.timesteps[0]
    lat = inp["latents"]
    x = torch.cat([sched.scale_model_input(torch.cat([lat] * 2), t), inp["image_latents"]], dim=2)
    cc = torch.cat([inp["gesture_latents"]] * 2)
    down, mid = p_cn(x, t, inp["encoder_hidden_states"], inp["added_time_ids"], controlnet_cond=cc, return_dict=False)
    eps = p_unet(x, t, inp["encoder_hidden_states"], inp["added_time_ids"], down_block_additional_residuals=down,
                 mid_block_additional_residual=mid, return_dict=False)[0]
    u, c = eps.chunk(2)
    api = sched.step(u + inp["guidance_scale"] * (c - u), t, lat).prev_sample
    s = err_stats(fused, api)
    print("fused step vs public-API step:", s)
    assert s["rel_l2"] <= 2e-3, s


@torch.no_grad()
def test_control_guidance_window_matches_oracle_loop():
    """controlnet_keep (reference :611-617,639-645): steps outside [start, end] run with the residuals scaled by 0.
    The product replays a UNet-only graph for those steps; the oracle multiplies the scale like the reference."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle.scheduler import EulerDiscreteScheduler as OSched, controlnet_keep, denoise_loop
    from this_and_that_vdm_amd.svd.denoise import DenoiseLoop
    from this_and_that_vdm_amd.svd.scheduling_euler_discrete import EulerDiscreteScheduler
    p_unet, p_cn, o_unet, o_cn = build_pair("tiny_vgl", torch.float16, "cuda:0", True)
    inp = _inputs()
    steps = 4
    keep = controlnet_keep(steps, 0.25, 0.75)
    assert keep == [0.0, 1.0, 1.0, 0.0]
    ref = denoise_loop(o_unet, o_cn, OSched(), inp["latents"], inp["image_latents"], inp["encoder_hidden_states"],
                       inp["added_time_ids"], inp["gesture_latents"], inp["guidance_scale"], num_inference_steps=steps,
                       control_guidance_start=0.25, control_guidance_end=0.75)
    sched = EulerDiscreteScheduler()
    sched.set_timesteps(steps)
    kw = dict(latents=inp["latents"], image_latents=inp["image_latents"], encoder_hidden_states=inp["encoder_hidden_states"],
              added_time_ids=inp["added_time_ids"], guidance_scale=inp["guidance_scale"], sigmas=sched.sigmas,
              timesteps=sched.timesteps, controlnet_cond=inp["gesture_latents"])
    outs = {}
    for graph in (True, False):
        outs[graph] = DenoiseLoop(p_unet, p_cn, use_graph=graph).begin(**kw, controlnet_keep=keep).run().clone()
        torch.cuda.synchronize()
    assert torch.equal(outs[True], outs[False]), "two-graph replay must equal eager launches"
    s = err_stats(outs[True], ref)
    print("windowed loop vs oracle:", s)
    assert s["rel_l2"] <= 1e-2 and s["cos"] >= 0.9999, s
    full = DenoiseLoop(p_unet, p_cn, use_graph=True).begin(**kw).run().clone()
    assert not torch.equal(full, outs[True])
    # keep == 0 everywhere is the VL loop, bit for bit
    none = DenoiseLoop(p_unet, p_cn, use_graph=True).begin(**kw, controlnet_keep=[0.0] * steps).run().clone()
    kw_vl = dict(kw, controlnet_cond=None)
    vl = DenoiseLoop(p_unet, None, use_graph=True).begin(**kw_vl).run().clone()
    assert torch.equal(none, vl)
    with pytest.raises(ValueError):
        DenoiseLoop(p_unet, p_cn).begin(**kw, controlnet_keep=[1.0])


@torch.no_grad()
def test_instructpix2pix_loop_matches_oracle_loop():
    """use_instructpix2pix (reference :182-184,208-210,627-628,698-702): CFG batch of 3 = (context+image, image only,
    nothing); eps = uncond + g_f (cond - uncond) + image_guidance_scale (cond - first)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle.scheduler import EulerDiscreteScheduler as OSched, denoise_loop
    from this_and_that_vdm_amd.svd.denoise import DenoiseLoop
    from this_and_that_vdm_amd.svd.scheduling_euler_discrete import EulerDiscreteScheduler
    p_unet, p_cn, o_unet, o_cn = build_pair("tiny_vgl", torch.float16, "cuda:0", True)
    inp = _inputs()
    img, ctx = inp["image_latents"][1:], inp["encoder_hidden_states"][1:]
    il3 = torch.cat([img, img, torch.zeros_like(img)])
    ehs3 = torch.cat([ctx, torch.zeros_like(ctx), torch.zeros_like(ctx)])
    ids3 = inp["added_time_ids"][:1].repeat(3, 1)
    steps, igs = 3, 1.5
    ref = denoise_loop(o_unet, o_cn, OSched(), inp["latents"], il3, ehs3, ids3, inp["gesture_latents"], inp["guidance_scale"],
                       num_inference_steps=steps, use_instructpix2pix=True, image_guidance_scale=igs)
    sched = EulerDiscreteScheduler()
    sched.set_timesteps(steps)
    kw = dict(latents=inp["latents"], image_latents=il3, encoder_hidden_states=ehs3, added_time_ids=ids3,
              guidance_scale=inp["guidance_scale"], sigmas=sched.sigmas, timesteps=sched.timesteps,
              controlnet_cond=inp["gesture_latents"], image_guidance_scale=igs)
    outs = {}
    for graph in (True, False):
        outs[graph] = DenoiseLoop(p_unet, p_cn, use_graph=graph).begin(**kw).run().clone()
        torch.cuda.synchronize()
    assert torch.equal(outs[True], outs[False])
    s = err_stats(outs[True], ref)
    print("instructpix2pix loop vs oracle:", s)
    assert s["rel_l2"] <= 1e-2 and s["cos"] >= 0.9999, s
    split = DenoiseLoop(p_unet, p_cn, use_graph=True, split_cfg=True).begin(**kw).run().clone()
    assert torch.equal(split, outs[True]), "three concurrent CFG branches must reproduce the joint launch"
    other = DenoiseLoop(p_unet, p_cn, use_graph=True).begin(**dict(kw, image_guidance_scale=0.0)).run().clone()
    assert not torch.equal(other, outs[True])          # the image scale is live (and re-captures the graph)
    with pytest.raises(ValueError):
        DenoiseLoop(p_unet, p_cn).begin(**dict(kw, image_guidance_scale=None))


@torch.no_grad()
def test_guess_mode_without_cfg_uses_logspace_scales():
    """guess_mode without CFG (batch 1): residual i is scaled by logspace(-1, 0, 13)[i] * conditioning_scale
    (svd/temporal_controlnet.py:626-630).  One fused step vs the oracle's two forward() calls + scheduler."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle.scheduler import EulerDiscreteScheduler as OSched
    from this_and_that_vdm_amd.svd.denoise import DenoiseLoop
    from this_and_that_vdm_amd.svd.scheduling_euler_discrete import EulerDiscreteScheduler
    p_unet, p_cn, o_unet, o_cn = build_pair("tiny_vgl", torch.float16, "cuda:0", True)
    inp = _inputs()
    il, ehs, ids = inp["image_latents"][1:], inp["encoder_hidden_states"][1:], inp["added_time_ids"][:1]
    sched = EulerDiscreteScheduler()
    sched.set_timesteps(3)
    kw = dict(latents=inp["latents"], image_latents=il, encoder_hidden_states=ehs, added_time_ids=ids, guidance_scale=None,
              sigmas=sched.sigmas, timesteps=sched.timesteps, controlnet_cond=inp["gesture_latents"], conditioning_scale=0.7)
    loop = DenoiseLoop(p_unet, p_cn, use_graph=True).begin(**kw, guess_mode=True)
    loop.step()
    got = loop.result().clone()
    plain = DenoiseLoop(p_unet, p_cn, use_graph=True).begin(**kw)
    plain.step()
    assert not torch.equal(got, plain.result())
    osch = OSched()
    osch.set_timesteps(3)
    t = osch.timesteps[0]
    x = torch.cat([osch.scale_model_input(inp["latents"], t), il], dim=2)
    down, mid = o_cn(x, t, ehs, ids, controlnet_cond=inp["gesture_latents"], conditioning_scale=0.7, guess_mode=True)
    eps = o_unet(x, t, ehs, ids, down_block_additional_residuals=down, mid_block_additional_residual=mid)
    ref = osch.step(eps, t, inp["latents"])
    s = err_stats(got, ref)
    print("guess-mode step vs oracle:", s)
    assert s["rel_l2"] <= 1e-2, s
    with pytest.raises(NotImplementedError):
        DenoiseLoop(p_unet, p_cn).begin(**dict(kw, image_latents=inp["image_latents"], encoder_hidden_states=inp["encoder_hidden_states"],
                                               added_time_ids=inp["added_time_ids"], guidance_scale=inp["guidance_scale"]),
                                        guess_mode=True)


@torch.no_grad()
def test_captured_graphs_survive_scratch_growth_and_weight_reloads():
    """Captured hipGraphs hold raw pointers.  (1) VL request -> VGL request -> a larger request whose split-K GEMMs outgrow
    the scratch buffer -> the first VL loop again: its graph must still run on valid memory and reproduce its first result
    bit for bit.  (2) load_state_dict between two requests on the same loop object: the packed weights are rebuilt, so the
    loop must re-capture (pack generation in its key) and match an eager run on the new weights."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from this_and_that_vdm_amd import ops
    from this_and_that_vdm_amd.svd.denoise import DenoiseLoop
    from this_and_that_vdm_amd.svd.scheduling_euler_discrete import EulerDiscreteScheduler
    from this_and_that_vdm_amd.utils.synthetic import fill_parameters_, synthetic_inputs
    p_unet, p_cn, _, _ = build_pair("tiny_vgl", torch.float16, "cuda:0", True)
    sched = EulerDiscreteScheduler()
    sched.set_timesteps(3)

    def kw(h, w, with_cn, seed=3):
        inp = synthetic_inputs(2, 4, h, w, ctx_tokens=5, ctx_dim=64, seed=seed)
        return dict(latents=inp["latents"], image_latents=inp["image_latents"], encoder_hidden_states=inp["encoder_hidden_states"],
                    added_time_ids=inp["added_time_ids"], guidance_scale=inp["guidance_scale"], sigmas=sched.sigmas,
                    timesteps=sched.timesteps, controlnet_cond=inp["gesture_latents"] if with_cn else None)

    floor, ops.WS_FLOOR_BYTES = ops.WS_FLOOR_BYTES, 1 << 12        # tiny scratch floor: the larger request MUST outgrow it
    ops._WS_RETIRED.extend(ops._WS.values())                       # start from no scratch at all (earlier tests left 64 MiB
    ops._WS.clear()                                                # buffers; their graphs may live on, so retire, not free)
    try:
        retired0 = len(ops._WS_RETIRED)
        vl = DenoiseLoop(p_unet, None, use_graph=True)
        first = vl.begin(**kw(8, 16, False)).run().clone()
        vgl = DenoiseLoop(p_unet, p_cn, use_graph=True)
        vgl_first = vgl.begin(**kw(8, 16, True)).run().clone()
        big = DenoiseLoop(p_unet, p_cn, use_graph=True)
        big.begin(**kw(24, 40, True)).run()
        torch.cuda.synchronize()
        grew = len(ops._WS_RETIRED) > retired0
        print("split-K scratch outgrown by the larger request:", grew, "retired buffers:", len(ops._WS_RETIRED) - retired0)
        assert grew, "the larger request was meant to outgrow the split-K scratch (tiny floor)"
        again = vl.begin(**kw(8, 16, False)).run().clone()
        assert torch.equal(again, first), "VL graph after scratch growth must reproduce its first result"
        assert torch.equal(vgl.begin(**kw(8, 16, True)).run(), vgl_first)
    finally:
        ops.WS_FLOOR_BYTES = floor
    # (2) new weights under a live loop object
    gen0 = p_unet._pack_gen
    before = vl.begin(**kw(8, 16, False)).run().clone()
    sd = {k: v.clone() for k, v in p_unet.state_dict().items()}
    fill_parameters_(p_unet, "other-unet.", round_to=torch.float16)
    after = vl.begin(**kw(8, 16, False)).run().clone()
    assert p_unet._pack_gen > gen0 and not torch.equal(after, before), "the reloaded weights must take effect"
    eager = DenoiseLoop(p_unet, None, use_graph=False).begin(**kw(8, 16, False)).run().clone()
    assert torch.equal(after, eager), "re-captured graph must equal eager launches on the new weights"
    p_unet.load_state_dict(sd)
    assert torch.equal(vl.begin(**kw(8, 16, False)).run(), before), "restoring the weights restores the result"


@torch.no_grad()
def test_film_table_rows_are_the_per_step_film_rows_bit_for_bit():
    """DenoiserBase.film_table (round 4): the time embedding + FiLM rows of all steps of a request evaluated at once must equal,
    bit for bit, what the per-step path computes inside the graph (_embed + _step_context) for every step and both models --
    including a CFG batch of 3 (10 steps per launch) and more rows than one launch holds (25 steps x 2 = 50 rows)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from tests.parity_common import build_pair
    from this_and_that_vdm_amd.svd.scheduling_euler_discrete import EulerDiscreteScheduler
    p_unet, p_cn, _, _ = build_pair("tiny_vgl", torch.float16, "cuda:0", True)
    sched = EulerDiscreteScheduler()
    sched.set_timesteps(25)
    for model in (p_unet, p_cn):
        model.prepare()
        for batch in (1, 2, 3):
            ids = torch.tensor([[6.0, 127.0, 0.02]] * batch, device="cuda")
            tab = model.film_table(sched.timesteps, ids, batch)
            assert tab.shape[:2] == (25, batch) and tab.dtype == torch.float32
            for i in (0, 7, 24):
                t = sched.timesteps[i:i + 1].to("cuda", torch.float32)
                ctx = model._step_context(model._embed(t, ids, batch, torch.device("cuda:0")), (None, None, 0, 0, 0))
                assert torch.equal(tab[i], ctx.film), (type(model).__name__, batch, i)
