#!/bin/bash
# Round-3 measurement pass on the GPU box (from the repo root): everything lands in gpurun_out/profiles/ and is copied to profiles/.
#   tools/profile_r3.sh [quick]      quick: skip the parity log and the probes
set -u
mkdir -p gpurun_out/profiles
P=gpurun_out/profiles
tools/profile_round.sh r3_vgl_lo_bf16 lo
cp gpurun_out/profiles/r3_hbm_traffic.json profiles/      # bench.py reads the traffic of the dominant kernel from the tracked copy
b() { python bench.py "$@" 2>/dev/null | tail -1; }
b > $P/r3_bench_lo.json
b --mode vl --no-cpu-baseline > $P/r3_bench_vl.json
b --res hi --no-cpu-baseline > $P/r3_bench_hi_bf16.json
b --res hi --attn fp8 --no-cpu-baseline > $P/r3_bench_hi_fp8.json
b --dtype fp16 --no-cpu-baseline > $P/r3_bench_lo_fp16.json
b --dtype f32 --steps 4 --warmup 1 --no-cpu-baseline > $P/r3_bench_lo_f32.json
b --block l0hi --steps 20 --warmup 3 > $P/r3_block_l0hi_bf16.json
b --block l0hi --attn fp8 --steps 20 --warmup 3 > $P/r3_block_l0hi_fp8.json
python tools/decode_bench.py 2>/dev/null | tail -1 > $P/r3_decode_bench.txt
python tools/decode_bench.py --dtype fp16 2>/dev/null | tail -1 >> $P/r3_decode_bench.txt
for f in $P/r3_bench_*.json $P/r3_block_*.json; do echo "$f: $(cut -c1-230 $f)"; done
cat $P/r3_decode_bench.txt
if [ "${1:-}" != "quick" ]; then
  # (the probes are built in the build container: hipcc --offload-arch=gfx950 -O3 tools/<name>.hip -o tools/<name>.bin; *.bin is git-ignored and travels with gpurun)
  tools/gemm_pp_probe.bin > $P/r3_gemm_pp_probe.txt 2>&1
  tools/dma_shape_test.bin > $P/r3_dma_shape_test.txt 2>&1
  tools/mfma_f8f6f4_probe.bin > $P/r3_mfma_f8f6f4_probe.txt 2>&1
  python tools/gemm_bench.py -1 9 --lib 2>/dev/null > $P/r3_gemm_vs_lib.txt
  python -m pytest tests/test_model_gpu.py tests/test_full_size_gpu.py tests/test_vae_decoder_gpu.py -q -s -k "16bit or two_steps or config5 or f32_mode or decoder_matches or zero_context" 2>&1 | grep -v "^$" | grep -v amdgpu.ids | cut -c1-400 > $P/r3_gpu_parity_log.txt
  tail -3 $P/r3_gpu_parity_log.txt
fi
