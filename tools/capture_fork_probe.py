#!/usr/bin/env python3
"""Which fork/join patterns does hipGraph stream capture (torch.cuda.graph) accept?  (a) one side stream re-forked N times,
(b) a fresh pre-created side stream per fork, (c) a side stream forked from a forked branch."""
import sys
import torch

def run(mode, n):
    x = torch.zeros(1 << 20, device="cuda")
    pool = [torch.cuda.Stream() for _ in range(n + 2)]
    keep = []
    g = torch.cuda.CUDAGraph()
    s0 = torch.cuda.Stream()
    s0.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s0):
        x.add_(1)
    torch.cuda.current_stream().wait_stream(s0)
    with torch.cuda.graph(g):
        main = torch.cuda.current_stream()
        branch = main
        if mode == "c":
            branch = pool[n + 1]
            e = torch.cuda.Event(); keep.append(e); e.record(main); branch.wait_event(e)
        for i in range(n):
            side = pool[0] if mode in ("a", "c") else pool[i]
            with torch.cuda.stream(branch):
                e = torch.cuda.Event(); keep.append(e); e.record(branch); side.wait_event(e)
                with torch.cuda.stream(side):
                    x[: 1 << 19].add_(1)
                x[1 << 19:].add_(1)
                e2 = torch.cuda.Event(); keep.append(e2); e2.record(side); branch.wait_event(e2)
                x.mul_(1.0)
        if mode == "c":
            e = torch.cuda.Event(); keep.append(e); e.record(branch); main.wait_event(e)
    g.replay()
    torch.cuda.synchronize()
    print(mode, n, "ok", float(x[0]), float(x[-1]))

run(sys.argv[1], int(sys.argv[2]))
