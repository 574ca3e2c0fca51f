cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
run() { cfg=$1; shift; timeout 900 python bench.py --config $cfg --sources 1 --steps 5 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg $*', d['ms_per_step'], 'iters', d['iterations_per_step'], 'pull', d['pull_iterations_per_step'], 'merged', (d.get('merged_loop') or {}).get('ms_per_step'))"; }
run friendster
run friendster --tune pull_min_frontier=3760000
run friendster --tune pull_min_frontier=5640000
run friendster --tune pull_min_frontier=9000000
run twitter --tune pull_min_frontier=1530000
run twitter --tune pull_min_frontier=2300000
