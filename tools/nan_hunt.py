#!/usr/bin/env python3
"""debug: run one eager full-size VGL step and report the first libttvdm op whose output is not finite."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from this_and_that_vdm_amd import ops
import bench

dtype = torch.bfloat16 if len(sys.argv) < 2 or sys.argv[1] == "bf16" else torch.float16
found = []

def wrap(name):
    fn = getattr(ops, name)
    def inner(*a, **k):
        out = fn(*a, **k)
        outs = out if isinstance(out, (tuple, list)) else (out,)
        torch.cuda.synchronize()
        for o in outs:
            if torch.is_tensor(o) and o.is_floating_point() and not bool(torch.isfinite(o).all()):
                ins = [(tuple(t.shape), str(t.dtype), bool(torch.isfinite(t).all())) for t in a if torch.is_tensor(t)]
                kw = {kk: (tuple(v.shape), bool(torch.isfinite(v).all())) if torch.is_tensor(v) else v for kk, v in k.items()}
                nbad = int((~torch.isfinite(o)).sum())
                found.append((name, tuple(o.shape), nbad, ins, kw))
                print("NON-FINITE after", name, tuple(o.shape), "bad elements", nbad, "inputs", ins, "kwargs", kw, flush=True)
                if len(found) >= 3:
                    raise SystemExit(1)
        return out
    setattr(ops, name, inner)

mode = sys.argv[2] if len(sys.argv) > 2 else "serial"
if mode == "serial":
    for n in ("gemm", "attention", "temporal_attention", "groupnorm_stats", "groupnorm_apply", "add_rowvec", "small_linear", "prep_model_input"):
        wrap(n)
elif mode.startswith("check:"):              # overlap on, but synchronise + check after the named ops only
    for n in mode[6:].split(","):
        wrap(n)
unet, cn, _, _ = bench.build_models("vgl", dtype, torch.device("cuda", 0), 0, 1)
loop, args = bench.make_loop(unet, cn, "lo", torch.device("cuda", 0), seed=0)
loop.use_graph = mode == "graph"
loop.overlap_branches = mode != "serial"
for _ in range(3):
    loop.step()
torch.cuda.synchronize()
print("step done; finite latents:", bool(torch.isfinite(loop.result()).all()), "offenders:", len(found))
