cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "binned or merged" 2>&1 | tail -5
BENCH_EXTRA="--sources 1" bash tools/r03/sweep_bin.sh twitter binned=1 binned=1,128,48,196608,0,65536
BENCH_EXTRA="--sources 1" bash tools/r03/sweep_bin.sh friendster binned=1
