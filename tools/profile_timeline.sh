#!/bin/bash
# rocprofv3 kernel trace of the headline bench -> gpurun_out/timeline_<tag>.txt (tools/trace_timeline.py) + the per-grid summary
tag=${1:-r4}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_tl_$tag; rm -rf $out; mkdir -p $out
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out/stats -o s -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-profile "$@" > $out/stats.log 2>&1
tr=$(find $out/stats -name '*kernel_trace.csv' | head -1)
python tools/trace_timeline.py "$tr" > gpurun_out/timeline_$tag.txt
python tools/trace_agg.py "$tr" --steps 3 > gpurun_out/bygrid_$tag.txt
tail -1 $out/stats.log | cut -c1-120
rm -rf $out
