#!/usr/bin/env python3
"""Generates tests/golden/full_size_autocast_yardstick.json: how far the reference's OWN 16-bit inference mode (16-bit parameters
under torch.autocast, test_code/inference.py:246) sits from its fp32 result at BASELINE's full size, measured on the oracle
(CPU) with exactly the weights / inputs / step of tests/test_full_size_gpu.py: the network contribution to the latents after
step 1 of the VGL loop.  The GPU test asserts that the HIP 16-bit modes are not further from the fp32 oracle than 1.25x these
numbers.  Runs on the CPU only (about 5 minutes on 8 cores):   python tests/golden/make_autocast_yardstick.py"""
import copy
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from oracle import models as om                                    # noqa: E402
from oracle.scheduler import EulerDiscreteScheduler as OSched      # noqa: E402
from tests.parity_common import err_stats                          # noqa: E402
from this_and_that_vdm_amd.utils.synthetic import fill_parameters_, synthetic_inputs   # noqa: E402

FRAMES, H, W, CTX_TOKENS, CTX_DIM, HEADS = 14, 32, 56, 78, 1024, (5, 10, 20, 20)       # == tests/test_full_size_gpu.py


def step1(unet, cn, inp, x, t, osched, dtype=None):
    lo = (lambda v: v.to(dtype)) if dtype is not None else (lambda v: v)
    ctxm = torch.autocast("cpu", dtype=dtype) if dtype is not None else torch.autocast("cpu", enabled=False)
    with ctxm:
        down, mid = cn(lo(x), t, lo(inp["encoder_hidden_states"]), lo(inp["added_time_ids"]),
                       controlnet_cond=lo(torch.cat([inp["gesture_latents"]] * 2)))
        eps = unet(lo(x), t, lo(inp["encoder_hidden_states"]), lo(inp["added_time_ids"]), down_block_additional_residuals=down,
                   mid_block_additional_residual=mid)
    u, c = eps.float().chunk(2)
    out = osched.step(u + inp["guidance_scale"] * (c - u), t, inp["latents"])
    return out[0] if isinstance(out, (tuple, list)) else getattr(out, "prev_sample", out)


@torch.no_grad()
def main():
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    with torch.device("meta"):
        o_unet = om.UNetSpatioTemporalConditionModel(num_attention_heads=HEADS, num_frames=FRAMES)
        o_cn = om.ControlNetModel()
    o_unet, o_cn = o_unet.to_empty(device="cpu").eval(), o_cn.to_empty(device="cpu").eval()
    fill_parameters_(o_unet, "unet.", round_to=torch.bfloat16)
    fill_parameters_(o_cn, "controlnet.", round_to=torch.bfloat16)
    inp = synthetic_inputs(2, FRAMES, H, W, CTX_TOKENS, CTX_DIM, seed=0)
    osched = OSched()
    osched.set_timesteps(25)
    t = osched.timesteps[0]
    x = torch.cat([osched.scale_model_input(torch.cat([inp["latents"]] * 2), t), inp["image_latents"]], dim=2)
    ref = step1(o_unet, o_cn, inp, x, t, osched)
    sample = inp["latents"].double()
    share = float(osched.sigmas[1]) / float(osched.sigmas[0])
    contrib = lambda z: (z.double().reshape(sample.shape) - sample * share).float()
    doc = {"_what": "network contribution to the latents after step 1 of the full-size VGL loop: oracle with 16-bit parameters under "
                    "torch.autocast('cpu') vs the fp32 oracle (tests/golden/make_autocast_yardstick.py)",
           "_torch": torch.__version__}
    for name, dtype in (("float16", torch.float16), ("bfloat16", torch.bfloat16)):
        o_unet16, o_cn16 = copy.deepcopy(o_unet).to(dtype), copy.deepcopy(o_cn).to(dtype)
        osched2 = OSched()
        osched2.set_timesteps(25)
        got = step1(o_unet16, o_cn16, inp, x, t, osched2, dtype)
        doc[name] = err_stats(contrib(got), contrib(ref))
        print(name, doc[name], flush=True)
        del o_unet16, o_cn16
    json.dump(doc, open(os.path.join(REPO, "tests", "golden", "full_size_autocast_yardstick.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
