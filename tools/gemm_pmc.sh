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
for m, n, k in [(50176, 2560, 320), (12544, 5120, 640), (3136, 10240, 1280), (50176, 320, 1280), (12544, 640, 2560), (8192, 8192, 8192)]:
    a = torch.randn(m, k, device=dev, dtype=dt); w = torch.randn(n, k, device=dev, dtype=dt) * k ** -0.5
    out = torch.empty(m, n, device=dev, dtype=dt)
    for _ in range(4):
        ops.gemm(a, w, out=out)
    torch.cuda.synchronize()
PY
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --kernel-include-regex "gemm_kernel" --output-format csv -d $out/raw_$tag -o p -- python $out/run.py > $out/run_$tag.log 2>&1
done
python - "$out" <<'PY'
import csv, glob, sys, collections, os, re
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "raw_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(gemm_kernel<[^>]*>)", r["Kernel_Name"])
        gs = int(r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X", "0"))
        wg = int(r.get("Workgroup_Size", r.get("Workgroup_Size_X", "256")) or 256)
        acc[(m.group(1) if m else r["Kernel_Name"][-40:], f"{gs // max(wg,1)} blocks")][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(os.path.join(out, "gemm_sq_counters.txt"), "w") as w:
    for key, c in acc.items():
        m = {k: sum(v) / len(v) for k, v in c.items()}
        line = f"{key}: " + "  ".join(f"{k}={v:.4g}" for k, v in sorted(m.items()))
        if "SQ_INSTS_MFMA" in m and "SQ_INSTS_VALU" in m: line += f"  | valu_per_mfma={m['SQ_INSTS_VALU'] / max(m['SQ_INSTS_MFMA'], 1):.2f}"
        if "SQ_ACTIVE_INST_VALU" in m and "SQ_WAVE_CYCLES" in m:
            line += f"  wait_inst/wave={m.get('SQ_WAIT_INST_ANY', 0) / max(m['SQ_WAVE_CYCLES'], 1):.3f}  wait_any/wave={m.get('SQ_WAIT_ANY', 0) / max(m['SQ_WAVE_CYCLES'], 1):.3f}"
        if "SQ_LDS_BANK_CONFLICT" in m and "SQ_ACTIVE_INST_LDS" in m: line += f"  lds_conflict/lds_active={m['SQ_LDS_BANK_CONFLICT'] / max(m['SQ_ACTIVE_INST_LDS'], 1):.4f}"
        print(line); w.write(line + "\n")
PY
rm -rf $out/raw_*
