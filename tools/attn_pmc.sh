#!/bin/bash
# SQ counters of the spatial self-attention kernel at L = 7168 / 1792, d = 64 (one rocprofv3 --pmc pass; no tracing domains
# besides --kernel-trace).  usage (GPU box, repo root): tools/attn_pmc.sh <outdir>
set -u
out=${1:-gpurun_out/attn_pmc}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY \
  --kernel-trace --kernel-include-regex "attn_kernel|attn_pipe_kernel|attn8_kernel" --output-format csv -d $out/raw -o p -- python tools/attn_bench.py > $out/run.log 2>&1
python - "$out" <<'PY'
import csv, glob, sys, collections, os
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "raw", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        import re
        m = re.search(r"(attn8?_kernel<[^>]*>)", r["Kernel_Name"])
        gs = int(r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X", "0"))
        key = (m.group(1) if m else r["Kernel_Name"][-40:], f"{gs // 256} blocks of 256 threads")
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(os.path.join(out, "attn_sq_counters.txt"), "w") as w:
    for key, c in acc.items():
        m = {k: sum(v) / len(v) for k, v in c.items()}
        line = f"{key}: " + "  ".join(f"{k}={v:.4g}" for k, v in sorted(m.items()))
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "SQ_BUSY_CU_CYCLES" in m:
            line += f"  | matrix pipe busy (SQ_VALU_MFMA_BUSY_CYCLES / 4 SIMDs / SQ_BUSY_CU_CYCLES)={m['SQ_VALU_MFMA_BUSY_CYCLES'] / 4 / max(m['SQ_BUSY_CU_CYCLES'], 1):.3f}  valu_per_mfma={m['SQ_INSTS_VALU'] / max(m['SQ_INSTS_MFMA'], 1):.1f}"
        if "SQ_ACTIVE_INST_VALU" in m and "SQ_WAVE_CYCLES" in m:
            line += f"  valu_active/wave_cycles={m['SQ_ACTIVE_INST_VALU'] / max(m['SQ_WAVE_CYCLES'], 1):.3f}  wait_inst/wave={m.get('SQ_WAIT_INST_ANY', 0) / max(m['SQ_WAVE_CYCLES'], 1):.3f}  wait_any/wave={m.get('SQ_WAIT_ANY', 0) / max(m['SQ_WAVE_CYCLES'], 1):.3f}"
        print(line); w.write(line + "\n")
PY
rm -rf $out/raw
