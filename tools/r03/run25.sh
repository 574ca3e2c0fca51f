cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "binned or merged" 2>&1 | tail -3
BENCH_EXTRA="--sources 1" bash tools/r03/sweep_bin.sh twitter binned=1
BENCH_EXTRA="--sources 1 --pick top10 --steps 15" bash tools/r03/sweep_bin.sh livejournal binned=1
BENCH_EXTRA="--sources 1" bash tools/r03/sweep_bin.sh friendster binned=1
bash tools/prof_timeline.sh twitter_1src --config twitter --sources 1 --steps 4 --warmup 2 | grep "k_bin"
