cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
for cfg in twitter friendster; do for S in 2 3 4; do timeout 900 python bench.py --config $cfg --sources $S --steps 4 --warmup 2 --no-cpu-baseline --no-merged 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg group of $S', d['ms_per_step'], 'per source', round(d['ms_per_step']/$S,1))"; done; done
