#!/bin/bash
# A/B of environment switches on the headline bench inside ONE gpurun call (box-to-box spread is +-3 %):
#   tools/ab_env.sh "NAME=VALUE ..." "NAME=VALUE ..." [...]     -> ms/step of each variant, two interleaved rounds
run() { env $1 python bench.py --no-cpu-baseline --no-kernel-profile --steps 25 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.3f ms' % d['ms_per_step'])"; }
for rep in 1 2; do for v in "$@"; do echo "[$v] $(run "$v")"; done; done
