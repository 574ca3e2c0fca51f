cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
BENCH_EXTRA="--sources 1 --pick top10 --steps 15" bash tools/r03/sweep_bin.sh livejournal binned=0 binned=2 binned=2,128,48,32768 binned=2,64,24,32768,0,8192 binned=2,128,24,16384,0,8192,1048576 binned=2,32,16,16384,0,8192,262144
