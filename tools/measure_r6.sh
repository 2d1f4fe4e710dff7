#!/bin/bash
# One GPU-box pass that produces the round-6 measurement set under gpurun_out/profiles/ (copied into profiles/ afterwards).
# Order matters: the split16 line is written FIRST (also into profiles/ of the box's copy) so that the headline line's `tolerance_mode`
# quotes a measurement of the same call and the same kernel sources; profiles/r6_hbm_traffic.json must already belong to the current
# kernel sources (tools/profile_round.sh; bench.py checks the stamp).
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/profiles; P=gpurun_out/profiles
b() { python bench.py "$@" 2>/dev/null | tail -1; }
b --dtype split16 --no-cpu-baseline --steps 10 --warmup 3 > $P/r6_bench_lo_split16.json; cp $P/r6_bench_lo_split16.json profiles/r6_bench_lo_split16.json
b --dtype f32 --no-cpu-baseline --steps 5 --warmup 2 > $P/r6_bench_lo_f32.json
b > $P/r6_bench_lo.json
b --mode vl --no-cpu-baseline > $P/r6_bench_vl.json
b --dtype fp16 --no-cpu-baseline > $P/r6_bench_lo_fp16.json
b --res ref --no-cpu-baseline > $P/r6_bench_ref.json
b --res hi --no-cpu-baseline --steps 10 --warmup 3 > $P/r6_bench_hi_bf16.json
b --res hi --attn fp8 --no-cpu-baseline --steps 10 --warmup 3 > $P/r6_bench_hi_fp8.json
b --block l0hi --steps 20 --warmup 3 > $P/r6_block_l0hi_bf16.json
b --block l0hi --attn fp8 --steps 20 --warmup 3 > $P/r6_block_l0hi_fp8.json
for f in lo_split16 lo_f32 lo vl lo_fp16 ref hi_bf16 hi_fp8; do python -c "import json,sys; d=json.load(open('$P/r6_bench_$f.json')); print('$f', round(d['ms_per_step'],3), 'ms', [round(v,3) for v in d['config']['ms_per_step_windows']], round(d['config']['step_mfma_frac_of_peak'],4), d['roofline']['kernel'] if d['roofline'] else '', round(d['roofline']['frac'],4) if d['roofline'] else '', d['roofline'].get('traffic') if d['roofline'] else '')"; done
for f in bf16 fp8; do python -c "import json; d=json.load(open('$P/r6_block_l0hi_$f.json')); print('block $f', round(d['ms_per_step'],3), 'ms frac', round(d['roofline']['frac'],4))"; done
python tools/gn_census.py 2>/dev/null > $P/r6_gn_census.txt; tail -1 $P/r6_gn_census.txt
