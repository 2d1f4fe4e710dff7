#!/usr/bin/env python3
"""How much idle capacity does one GEMM launch leave?  Same GEMM on 1 / 2 / 3 concurrent streams (GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from this_and_that_vdm_amd import ops
dt, dev = torch.bfloat16, "cuda"
def run(m, n, k, nstreams, iters=10):
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    bufs = [(torch.randn(m, k, device=dev, dtype=dt), torch.randn(n, k, device=dev, dtype=dt), torch.empty(m, n, device=dev, dtype=dt)) for _ in range(nstreams)]
    for s, (a, w, o) in zip(streams, bufs):
        with torch.cuda.stream(s): ops.gemm(a, w, out=o)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for s in streams: s.wait_event(e0)
    for _ in range(iters):
        for s, (a, w, o) in zip(streams, bufs):
            with torch.cuda.stream(s): ops.gemm(a, w, out=o)
    for s in streams: torch.cuda.current_stream().wait_stream(s)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / iters
    return 2.0 * m * n * k * nstreams / t / 1e12
for shape in [(50176, 2560, 320), (12544, 5120, 640), (50176, 320, 320), (12544, 640, 640), (3136, 1280, 1280), (3136, 1280, 5120), (8192, 8192, 8192)]:
    print(shape, " ".join(f"{ns} streams: {run(*shape, ns):6.0f} TF" for ns in (1, 2, 3)))
