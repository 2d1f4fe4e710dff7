#!/usr/bin/env python3
"""Per-block timeline of one tt_gemm launch (prologue / first tile / main loop / epilogue) from the wall-clock stamps of
the TT_GEMM_TIMELINE debug build.  make -C this_and_that_vdm_amd/csrc timeline; python tools/gemm_timeline.py M N K cfg [r] [g]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from this_and_that_vdm_amd import ops, _lib

def main():
    m, n, k, cfg = [int(x) for x in sys.argv[1:5]]
    res = 'r' in sys.argv[5:]
    gg = 'g' in sys.argv[5:]
    _lib.LIB_PATH = os.path.join(_lib.CSRC, "libttvdm_tl.so")       # `make -C this_and_that_vdm_amd/csrc timeline`
    lib = _lib.load()
    raw = C.CDLL(_lib.LIB_PATH)
    dt, dev = torch.bfloat16, "cuda"
    nb = 6
    A = [torch.randn(m, k, device=dev, dtype=dt) for _ in range(nb)]
    O = [torch.empty(m, n // 2 if 'g' in sys.argv[5:] else n, device=dev, dtype=dt) for _ in range(nb)]
    R = [torch.randn(m, n, device=dev, dtype=dt) for _ in range(nb)]
    w = torch.randn(n, k, device=dev, dtype=dt)
    lib.tt_gemm_set_tile_override(cfg)
    for i in range(nb): ops.gemm(A[i], w, out=O[i], residual=R[i] if res else None, geglu=gg)
    torch.cuda.synchronize()
    ops.gemm(A[0], w, out=O[0], residual=R[0] if res else None, geglu=gg)
    torch.cuda.synchronize()
    buf = np.zeros(8 * 8192, dtype=np.int64)
    raw.tt_debug_timeline(buf.ctypes.data_as(C.c_void_p), C.c_int(8 * 8192))
    t = buf.reshape(8192, 8)
    nblk = int((t[:, 0] > 0).sum())
    extra = t[:nblk, 6:8].astype(np.float64)       # inside the prologue: 6 = addresses done (before the first DMA / the residual batch), 7 = first tile requested
    t = t[:nblk, :6].astype(np.float64)
    t0 = t[:, 0].min()
    extra = (extra - t0) * 0.01
    t = (t - t0) * 0.01      # 100 MHz -> us
    names = ["start", "prologue done", "tile0 landed", "mainloop done", "ring free", "stores done"]
    print(f"M={m} N={n} K={k} cfg{cfg} res={res}: {nblk} blocks; kernel span {t[:, 5].max():.2f} us")
    order = np.argsort(t[:, 0])
    for q in (0, nblk // 4, nblk // 2, 3 * nblk // 4, nblk - 1):
        b = order[q]
        print(f"  block#{q:4d} (by start): " + " | ".join(f"{names[i]} {t[b, i]:6.2f}" for i in range(6)))
    d = np.diff(t, axis=1)
    print("  mean phase us: " + " | ".join(f"{names[i+1]} +{d[:, i].mean():.2f}" for i in range(5)))
    if (extra[:, 0] > 0).all():
        print(f"  inside the prologue (mean us after start): arguments + tile mapping + address set-up {np.mean(extra[:, 0] - t[:, 0]):.2f}" +
              (f" | first tile requested {np.mean(extra[:, 1] - t[:, 0]):.2f}" if (extra[:, 1] > 0).all() else "") +
              f" | residual batch requested {np.mean(t[:, 1] - t[:, 0]):.2f}")
    print(f"  start times: min {t[:,0].min():.2f} median {np.median(t[:,0]):.2f} max {t[:,0].max():.2f}")

if __name__ == "__main__":
    main()
