cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_engine_gpu.py tests/test_fullsize_gpu.py tests/test_renumbering_gpu.py -x -q -m gpu -k "livejournal or full_size or renumber" 2>&1 | tail -4
BENCH_EXTRA="--sources 1 --pick top10 --steps 15" bash tools/r03/sweep_bin.sh livejournal binned=1 binned=0
BENCH_EXTRA="--sources 1" bash tools/r03/sweep_bin.sh twitter binned=1
