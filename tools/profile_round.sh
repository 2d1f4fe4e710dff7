#!/bin/bash
# One GPU-box pass that refreshes the committed measurement artefacts of a round:
#   profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of `bench.py` (graph replays)
#   profiles/<tag>_kernel_by_grid.txt per (kernel, grid) durations of the timed steps (tools/trace_agg.py)
#   profiles/r1_hbm_traffic.json      HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE passes (tools/pmc_traffic.py)
# usage (on the GPU box, from the repo root): tools/profile_round.sh r1_vgl_lo_bf16
set -u
tag=${1:-r1_vgl_lo_bf16}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag; rm -rf $out; mkdir -p $out gpurun_out/profiles
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- \
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-profile > $out/stats.log 2>&1
cp "$(find $out/stats -name '*kernel_stats.csv' | head -1)" gpurun_out/profiles/${tag}_kernel_stats.csv
python tools/trace_agg.py "$(find $out/stats -name '*kernel_trace.csv' | head -1)" --steps 5 > gpurun_out/profiles/${tag}_kernel_by_grid.txt
tail -1 $out/stats.log | cut -c1-160
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "gemm_kernel|attn_kernel" --output-format csv -d $out/$c -o p -- \
    python tools/pmc_step.py vgl lo > $out/$c.log 2>&1 || echo "pmc pass $c failed"
done
cp profiles/r1_hbm_traffic.json gpurun_out/profiles/r1_hbm_traffic.json 2>/dev/null
python tools/pmc_traffic.py $out/FETCH_SIZE $out/WRITE_SIZE vgl_lo gpurun_out/profiles/r1_hbm_traffic.json gpurun_out/profiles/r1_vgl_lo_pmc_fetch_write_raw.json | head -20
rm -rf $out
