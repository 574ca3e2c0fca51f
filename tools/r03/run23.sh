cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
BENCH_EXTRA="--sources 1 --pick top10 --steps 15" bash tools/r03/sweep_bin.sh livejournal binned=2,128,48,8192 binned=2,128,48,16384 binned=2,128,48,32768 binned=2,128,48,65536 binned=2,128,48,32768,0,16384 binned=2,128,48,32768,0,32768,1048576 binned=2,256,48,32768
