#!/bin/bash
# One GPU-box pass that refreshes the committed measurement artefacts of a round (run from the repo root on the GPU box):
#   profiles/<tag>_kernel_stats.csv    rocprofv3 --kernel-trace --stats of `bench.py` (graph replays)
#   profiles/<tag>_kernel_by_grid.txt  per (kernel, grid) durations of the timed steps (tools/trace_agg.py)
#   profiles/r6_hbm_traffic.json       HBM-side bytes per launch from the FETCH_SIZE / WRITE_SIZE passes (tools/pmc_traffic.py),
#                                      stamped with the commit it was measured at (bench.py copies it into `roofline.traffic`)
# usage: tools/profile_round.sh <tag> [lo|hi] [extra bench.py flags]     e.g.  tools/profile_round.sh r4_vgl_lo_bf16 lo
set -u
tag=${1:-r6_vgl_lo_bf16}; res=${2:-lo}; shift 2 2>/dev/null
extra="$*"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag; rm -rf $out; mkdir -p $out gpurun_out/profiles
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- \
  python bench.py --steps 5 --warmup 2 --res $res --no-cpu-baseline --no-kernel-profile $extra > $out/stats.log 2>&1
cp "$(find $out/stats -name '*kernel_stats.csv' | head -1)" gpurun_out/profiles/${tag}_kernel_stats.csv
python tools/trace_agg.py "$(find $out/stats -name '*kernel_trace.csv' | head -1)" --steps 5 > gpurun_out/profiles/${tag}_kernel_by_grid.txt
tail -1 $out/stats.log | cut -c1-160
if [ "$res" = "lo" ] && [ -z "$extra" ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 500 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "gemm_kernel|gemm_pp_kernel|gemm_w320_kernel|gemm_w320h_kernel|attn_kernel|attn_pipe_kernel|attn8_kernel" --output-format csv -d $out/$c -o p -- \
      python tools/pmc_step.py vgl lo > $out/$c.log 2>&1 || echo "pmc pass $c failed"
  done
  python tools/pmc_traffic.py $out/FETCH_SIZE $out/WRITE_SIZE vgl_lo gpurun_out/profiles/r6_hbm_traffic.json gpurun_out/profiles/r6_vgl_lo_pmc_fetch_write_raw.json | head -24
fi
rm -rf $out
