cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_fullsize_golden_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu -k "merged or production_mode" -s 2>&1 | grep "parity\|passed\|failed\|Error" | tail -12
for s in 1 2 3; do DPPR_SOAK_MERGE=1 timeout 600 python tools/soak.py $s 150 16 2>&1 | tail -1; done
DPPR_SOAK_MERGE=1 timeout 600 python tools/soak.py 4 150 16 one-sweep 2>&1 | tail -1
DPPR_SOAK_MERGE=1 DPPR_SOAK_TUNE="persistent=0,binned=2" timeout 600 python tools/soak.py 5 150 16 one-sweep 2>&1 | tail -1
DPPR_SOAK_MERGE=1 DPPR_SOAK_TUNE="pull_min_frontier=2000" timeout 600 python tools/soak.py 6 150 16 2>&1 | tail -1
BENCH_EXTRA="--sources 1" bash tools/r03/sweep_bin.sh friendster binned=1
