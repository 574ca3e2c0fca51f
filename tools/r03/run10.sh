cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
for k in livejournal twitter friendster; do timeout 1200 python tools/slide_costs.py $k 2>&1 | tail -5; done
