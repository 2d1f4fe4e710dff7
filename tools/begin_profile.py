import sys, os, time, cProfile, pstats
sys.path.insert(0, os.getcwd())
import torch, bench
dev = torch.device("cuda", 0)
unet, cn, _, _ = bench.build_models("vgl", torch.bfloat16, dev, 0, 1)
loop, args = bench.make_loop(unet, cn, "lo", dev, 0)
bench.advance(loop, args, 3)
torch.cuda.synchronize()
for i in range(3):
    t0 = time.perf_counter(); loop.begin(**args); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"begin host {1e3*(t1-t0):.2f} ms, + drain {1e3*(t2-t1):.2f} ms")
pr = cProfile.Profile(); pr.enable(); loop.begin(**args); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
