#!/usr/bin/env python3
"""Launch time of tt_gemm against K at fixed (M, N): intercept = prologue + epilogue + launch, slope = main loop."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from this_and_that_vdm_amd import ops, _lib

def main():
    lib = _lib.load()
    dt, dev = torch.bfloat16, "cuda"
    cases = [(50176, 320, 10), (50176, 320, 18), (12544, 640, 18), (3136, 1280, 18), (50176, 2560, 3), (50176, 2560, 18)]
    if len(sys.argv) > 3:
        cases = [(int(sys.argv[1]), int(sys.argv[2]), int(c)) for c in sys.argv[3:]]
    for m, n, cfg in cases:
        lib.tt_gemm_set_tile_override(cfg)
        line = []
        for k in (64, 128, 320, 640, 1280, 2560, 5120):
            nb = max(2, min(16, (1 << 30) // ((m * k + m * n) * 2) + 1))
            A = [torch.randn(m, k, device=dev, dtype=dt) for _ in range(nb)]
            O = [torch.empty(m, n, device=dev, dtype=dt) for _ in range(nb)]
            w = torch.randn(n, k, device=dev, dtype=dt)
            for i in range(nb): ops.gemm(A[i], w, out=O[i])
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(3 * nb): ops.gemm(A[i % nb], w, out=O[i % nb])
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / (3 * nb) * 1e3
            line.append(f"K={k}: {us:6.1f}us {2*m*n*k/us/1e6:5.0f}TF")
            del A, O
        print(f"M={m} N={n} cfg{cfg}: " + " | ".join(line))
    lib.tt_gemm_set_tile_override(-1)

if __name__ == "__main__":
    main()
