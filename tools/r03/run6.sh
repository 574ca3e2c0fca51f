cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "binned" 2>&1 | tail -3
bash tools/r03/stamps_bin.sh friendster 1 2>&1 | grep -v "longest"
bash tools/r03/sweep_bin.sh twitter binned=1 binned=1,128,48,196608
BENCH_EXTRA="--sources 1" bash tools/r03/sweep_bin.sh friendster binned=1 binned=1,128,48,196608 binned=1,288,112,393216
