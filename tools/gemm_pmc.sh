#!/bin/bash
# SQ counters of the GEMM kernels on the model's biggest shapes (one rocprofv3 --pmc pass per counter group; no tracing domains
# besides --kernel-trace).  usage (GPU box, repo root): tools/gemm_pmc.sh <outdir>
set -u
out=${1:-gpurun_out/gemm_pmc}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cat > $out/run.py <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from this_and_that_vdm_amd import ops, packing
dt, dev = torch.bfloat16, "cuda"
r = lambda *s: torch.randn(*s, device=dev, dtype=dt)
# the three GEGLU projections with the LayerNorm fold (gemm_pp), FF2 with its residual at the three finest levels (gemm_w320 / tiled),
# a 3x3 conv and a temporal conv of the finest level (gemm_w320 implicit-GEMM producers), one library-sized square problem
for m, c in [(50176, 320), (12544, 640), (3136, 1280)]:
    x = r(m, c); w1 = r(8 * c, c) * c ** -0.5; h = torch.empty(m, 4 * c, device=dev, dtype=dt)
    w2 = r(c, 4 * c) * (4 * c) ** -0.5; b1 = torch.randn(8 * c, device=dev); b2 = torch.randn(c, device=dev)
    for _ in range(4):
        ops.gemm(x, w1, bias=b1, ln_fold=1, ln_eps=1e-5, geglu=True, out=h)
        ops.gemm(h, w2, bias=b2, residual=x, out=x)
x, wc = r(50176, 320), r(320, 9 * 320) * (9 * 320) ** -0.5
wt = r(320, 960) * 960 ** -0.5
o = torch.empty(50176, 320, device=dev, dtype=dt)
for _ in range(4):
    ops.gemm(x, wc, mode=1, conv=(28, 32, 56, 32, 56, 1, 0), bias=torch.randn(320, device=dev), out=o)
    ops.gemm(x, wt, mode=2, tconv=(14, 1792), bias=torch.randn(320, device=dev), residual=o, blend=o, alpha=0.3, out=o)
a = r(8192, 8192); w = r(8192, 8192) * 8192 ** -0.5; out = torch.empty(8192, 8192, device=dev, dtype=dt)
for _ in range(4):
    ops.gemm(a, w, out=out)
torch.cuda.synchronize()
PY
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --kernel-include-regex "gemm_kernel|gemm_pp_kernel|gemm_w320" --output-format csv -d $out/raw_$tag -o p -- python $out/run.py > $out/run_$tag.log 2>&1
done
python - "$out" <<'PY'
import csv, glob, sys, collections, os, re
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "raw_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(gemm(?:_pp|_w320h?)?_kernel<[^>]*>)", r["Kernel_Name"])
        gs = int(r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X", "0"))
        wg = int(r.get("Workgroup_Size", r.get("Workgroup_Size_X", "256")) or 256)
        acc[(m.group(1) if m else r["Kernel_Name"][-40:], f"{gs // max(wg,1)} blocks")][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(os.path.join(out, "gemm_sq_counters.txt"), "w") as w:
    for key, c in acc.items():
        m = {k: sum(v) / len(v) for k, v in c.items()}
        line = f"{key}: " + "  ".join(f"{k}={v:.4g}" for k, v in sorted(m.items()))
        if "SQ_INSTS_MFMA" in m and "SQ_INSTS_VALU" in m: line += f"  | valu_per_mfma={m['SQ_INSTS_VALU'] / max(m['SQ_INSTS_MFMA'], 1):.2f}"
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "SQ_BUSY_CYCLES" in m: line += f"  mfma_busy/busy={m['SQ_VALU_MFMA_BUSY_CYCLES'] / max(m['SQ_BUSY_CYCLES'], 1):.3f}"
        if "SQ_ACTIVE_INST_VALU" in m and "SQ_WAVE_CYCLES" in m:
            line += f"  wait_inst/wave={m.get('SQ_WAIT_INST_ANY', 0) / max(m['SQ_WAVE_CYCLES'], 1):.3f}  wait_any/wave={m.get('SQ_WAIT_ANY', 0) / max(m['SQ_WAVE_CYCLES'], 1):.3f}"
        if "SQ_LDS_BANK_CONFLICT" in m and "SQ_ACTIVE_INST_LDS" in m: line += f"  lds_conflict/lds_active={m['SQ_LDS_BANK_CONFLICT'] / max(m['SQ_ACTIVE_INST_LDS'], 1):.4f}"
        print(line); w.write(line + "\n")
PY
rm -rf $out/raw_*
