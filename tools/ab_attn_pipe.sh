# interleaved A/B of the software-pipelined self-attention (TT_ATTN_PIPE=0: attn_kernel everywhere) inside ONE gpurun call
set -e
python -m pytest tests/test_ops_gpu.py -x -q -k "attention" 2>&1 | tail -2
for i in 1 2; do
  for v in 1 0; do
    TT_ATTN_PIPE=$v python bench.py --no-cpu-baseline --no-kernel-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lo pipe=$v', d['ms_per_step'], d['config'].get('ms_per_step_windows'))"
  done
done
for v in 1 0; do
  TT_ATTN_PIPE=$v python bench.py --block l0hi --no-cpu-baseline --no-kernel-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('block pipe=$v', d['ms_per_step'], d.get('value'))"
  TT_ATTN_PIPE=$v python bench.py --res hi --steps 10 --no-cpu-baseline --no-kernel-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('hi pipe=$v', d['ms_per_step'])"
done
