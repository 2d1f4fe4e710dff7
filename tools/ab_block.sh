run() { env $1 python bench.py --block l0hi --steps 20 --warmup 3 $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.3f ms' % d['ms_per_step'])"; }
for rep in 1 2; do for v in "TT_ATTN_QPROJ=0" "TT_ATTN_QPROJ=1"; do echo "[block bf16 $v] $(run $v)"; done; done
for v in "TT_ATTN_QPROJ=0" "TT_ATTN_QPROJ=1"; do echo "[block fp8 $v] $(run $v '--attn fp8')"; done
runhi() { env $1 python bench.py --res hi --attn fp8 --no-cpu-baseline --no-kernel-profile --steps 8 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.3f ms' % d['ms_per_step'])"; }
for v in "TT_ATTN_QPROJ=0" "TT_ATTN_QPROJ=1"; do echo "[hi fp8 $v] $(runhi $v)"; done
