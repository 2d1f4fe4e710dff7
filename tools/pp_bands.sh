#!/bin/bash
# gemm_pp's XCD column bands (TT_PP_XCD_BANDS): kernel time (graph replays) and L2 <-> fabric bytes per launch (rocprofv3 --pmc
# FETCH_SIZE / WRITE_SIZE, separate passes) of the three LayerNorm + GEGLU projections, bands forced to 1 against the host's choice.
# usage (GPU box, repo root): tools/pp_bands.sh <outdir>
set -u
out=${1:-gpurun_out/pp_bands}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cat > $out/run.py <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from this_and_that_vdm_amd import ops
dt, dev = torch.bfloat16, "cuda"
r = lambda *s: torch.randn(*s, device=dev, dtype=dt)
time_it = len(sys.argv) > 1
for m, c in [(50176, 320), (12544, 640), (3136, 1280)]:
    x = r(m, c); w1 = r(8 * c, c) * c ** -0.5; h = torch.empty(m, 4 * c, device=dev, dtype=dt); b1 = torch.randn(8 * c, device=dev)
    f = lambda: ops.gemm(x, w1, bias=b1, ln_fold=1, ln_eps=1e-5, geglu=True, out=h)
    for _ in range(4): f()
    torch.cuda.synchronize()
    if time_it:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(10): f()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): g.replay()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        print(f"M={m} C={c}: {us:7.1f} us  {2.0*m*8*c*c/us/1e6:7.1f} TFLOP/s")
PY
for bands in ${PP_BANDS_LIST:-1 0}; do
  echo "== TT_PP_XCD_BANDS=$bands (0 = host's choice)"
  TT_PP_XCD_BANDS=$bands python $out/run.py time
  for c in FETCH_SIZE WRITE_SIZE; do
    TT_PP_XCD_BANDS=$bands timeout 300 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "gemm_pp_kernel" --output-format csv -d $out/raw_${bands}_$c -o p -- python $out/run.py > $out/run_${bands}_$c.log 2>&1
  done
  python - "$out" "$bands" <<'PY'
import csv, glob, sys, os
out, bands = sys.argv[1:3]
vals = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = []
    for f in glob.glob(os.path.join(out, f"raw_{bands}_{c}", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c: rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    rows.sort()
    vals[c] = [v for _, v in rows]
n = len(vals["FETCH_SIZE"]) // 3
for i, (m, cc) in enumerate([(50176, 320), (12544, 640), (3136, 1280)]):
    # the 3136-row problem launches the persistent kernel on 3072 rows (+ a tiled launch for the rest, filtered out)
    f = sum(vals["FETCH_SIZE"][i * n:(i + 1) * n]) / n; w = sum(vals["WRITE_SIZE"][i * n:(i + 1) * n]) / n
    mm = m // 256 * 256
    alg = (mm * cc + 8 * cc * cc + mm * 4 * cc) * 2
    print(f"  M={m} C={cc}: fetched {2*f*1024/1e6:7.1f} MB  written {w*1024/1e6:7.1f} MB  total {(2*f+w)*1024/1e6:7.1f} MB  algorithmic {alg/1e6:6.1f} MB  ratio {(2*f+w)*1024/alg:4.2f}")
PY
done
rm -rf $out/raw_*
