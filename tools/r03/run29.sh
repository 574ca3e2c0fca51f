cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python tools/soak.py 7 300 16 2>&1 | tail -1
DPPR_SOAK_TUNE="persistent=0,binned=2" timeout 900 python tools/soak.py 8 300 16 one-sweep 2>&1 | tail -1
DPPR_SOAK_TUNE="variant=2,pull_min_frontier=3000" timeout 900 python tools/soak.py 9 200 16 2>&1 | tail -1
DPPR_SOAK_TUNE="variant=3,group_at_slide=0" timeout 900 python tools/soak.py 10 200 16 2>&1 | tail -1
DPPR_SOAK_MERGE=1 DPPR_SOAK_TUNE="persistent=0,binned=2,pull_min_frontier=500" timeout 900 python tools/soak.py 11 300 17 one-sweep 2>&1 | tail -1
