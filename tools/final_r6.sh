cd "$GRAFT_REPO_ROOT"
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
bash tools/profile_round.sh r6_vgl_lo_bf16 lo 2>&1 | tail -8
cp gpurun_out/profiles/r6_hbm_traffic.json profiles/r6_hbm_traffic.json
bash tools/measure_r6.sh 2>&1 | tail -14
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
