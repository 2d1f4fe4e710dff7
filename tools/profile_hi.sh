#!/bin/bash
# GPU-box pass for the 512x896 workload (BASELINE config 5's size, bf16): bench line + rocprofv3 kernel stats + per-grid split.
# usage (from the repo root on the GPU box): tools/profile_hi.sh r1_vgl_hi_bf16
set -u
tag=${1:-r1_vgl_hi_bf16}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag; rm -rf $out; mkdir -p $out gpurun_out/profiles
python bench.py --res hi --no-cpu-baseline > gpurun_out/bench_${tag}.json 2> $out/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- \
  python bench.py --res hi --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-profile > $out/stats.log 2>&1
cp "$(find $out/stats -name '*kernel_stats.csv' | head -1)" gpurun_out/profiles/${tag}_kernel_stats.csv
python tools/trace_agg.py "$(find $out/stats -name '*kernel_trace.csv' | head -1)" --steps 5 > gpurun_out/profiles/${tag}_kernel_by_grid.txt
tail -1 $out/stats.log | cut -c1-200
cut -c1-400 gpurun_out/bench_${tag}.json
rm -rf $out
