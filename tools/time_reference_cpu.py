#!/usr/bin/env python3
"""BASELINE.md section 3's one-off cross-check (BUILD CONTAINER ONLY -- needs /root/reference, never runs on the GPU box):
the reference's own model files (imported unmodified through the in-memory diffusers stand-in of tests/golden/make_golden.py)
timed at FULL size on the same inputs and the same weights as the oracle, on the same cores.  Shows that `cpu_baseline`
(the oracle) costs what the reference costs, and pins the oracle's full-size output against the reference's composition.

    python tools/time_reference_cpu.py [--out profiles/r2_reference_vs_oracle_cpu.json]

One VGL network evaluation = ControlNet + UNet forward on the CFG batch of 2 x 14 frames at 32x56 latents, fp32 eager.
"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests", "golden"))


def build(ctor, pat):
    with torch.device("meta"):
        m = ctor()
    m = m.to_empty(device="cpu").eval()
    for name, p in m.named_parameters():
        n = p.numel()
        p.data.view(-1).copy_(pat.repeat((n + pat.numel() - 1) // pat.numel())[:n])
        if p.dim() == 1 and name.endswith("weight"):
            p.data.fill_(1.0)
    return m


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(REPO, "profiles", "r2_reference_vs_oracle_cpu.json"))
    ap.add_argument("--threads", type=int, default=os.cpu_count())
    args = ap.parse_args()
    import make_golden
    from oracle import models as om
    from this_and_that_vdm_amd.utils.synthetic import synthetic_inputs
    assert os.path.isdir(make_golden.REF), "needs /root/reference (build container)"
    make_golden.install_diffusers_standin()
    cwd = os.getcwd()
    os.chdir(make_golden.REF)
    sys.path.insert(0, make_golden.REF)
    from svd.unet_spatio_temporal_condition import UNetSpatioTemporalConditionModel as RefUNet
    from svd.temporal_controlnet import ControlNetModel as RefCN
    os.chdir(cwd)
    torch.set_grad_enabled(False)
    torch.set_num_threads(args.threads)
    torch.set_flush_denormal(True)
    pat = torch.randn(1 << 20, generator=torch.Generator().manual_seed(0)) * 0.02
    inp = synthetic_inputs(2, 14, 32, 56, 78, 1024, seed=0)
    x = torch.cat([torch.cat([inp["latents"]] * 2) / (700.0 ** 2 + 1) ** 0.5, inp["image_latents"]], dim=2)
    t = torch.tensor(1.6377)
    ehs, ati, cond = inp["encoder_hidden_states"], inp["added_time_ids"], torch.cat([inp["gesture_latents"]] * 2)
    res = {"threads": args.threads, "cpu_count": os.cpu_count(), "torch": torch.__version__,
           "workload": "ControlNet + UNet forward, CFG batch 2 x 14 frames, 32x56 latents, fp32 eager, same weights and inputs"}

    def run(unet_ctor, cn_ctor, tag, ref):
        unet = build(unet_ctor, pat)
        cn = build(cn_ctor, pat)
        sd_u = {k: v.shape for k, v in unet.state_dict().items()}
        t0 = time.perf_counter()
        if ref:
            down, mid = cn(x, t, ehs, ati, controlnet_cond=cond, conditioning_scale=1.0, return_dict=False)
            out = unet(x, t, ehs, ati, down_block_additional_residuals=down, mid_block_additional_residual=mid, return_dict=False)[0]
        else:
            down, mid = cn(x, t, ehs, ati, controlnet_cond=cond)
            out = unet(x, t, ehs, ati, down_block_additional_residuals=down, mid_block_additional_residual=mid)
        dt = time.perf_counter() - t0
        res[tag + "_seconds"] = round(dt, 2)
        print(tag, f"{dt:.1f} s", flush=True)
        del unet, cn
        return out, sd_u

    out_ref, sd_ref = run(lambda: RefUNet(num_attention_heads=(5, 10, 20, 20), num_frames=14), lambda: RefCN(), "reference", True)
    out_orc, sd_orc = run(lambda: om.UNetSpatioTemporalConditionModel(num_attention_heads=(5, 10, 20, 20), num_frames=14),
                          lambda: om.ControlNetModel(), "oracle", False)
    assert sd_ref == sd_orc, "state-dict keys/shapes differ between the reference and the oracle"
    # the first evaluation in a process also pays for the allocator / thread-pool warm-up: time the reference a second time
    out_ref2, _ = run(lambda: RefUNet(num_attention_heads=(5, 10, 20, 20), num_frames=14), lambda: RefCN(), "reference_second_run", True)
    assert torch.equal(out_ref, out_ref2)
    del out_ref2
    d = (out_ref - out_orc).abs()
    res["max_abs_diff"] = float(d.max())
    res["rel_l2"] = float(d.norm() / out_ref.norm())
    res["ratio_oracle_over_reference"] = round(res["oracle_seconds"] / res["reference_second_run_seconds"], 3)
    res["note"] = ("reference = /root/reference svd/{unet_spatio_temporal_condition,temporal_controlnet}.py + diffusion_arch/* executed "
                   "unmodified, leaf modules from oracle/leaves.py through the in-memory diffusers stand-in (diffusers 0.25.1 is not "
                   "installed); oracle = oracle/models.py.  Single evaluations in one process, in the order reference, oracle, reference; the ratio uses the reference's second run (the first also pays the process warm-up).")
    print(json.dumps(res, indent=1))
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
