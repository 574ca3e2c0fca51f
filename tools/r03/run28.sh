cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
for rep in 1 2; do for v in 1 0; do DPPR_SYNC_SPIN=$v timeout 600 python bench.py --steps 20 --warmup 5 --no-extra --no-merged --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lj10 spin=$v', d['ms_per_step'], d['event_ms_per_step'])"; done; done
for v in 1 0; do DPPR_SYNC_SPIN=$v timeout 600 python bench.py --config twitter --sources 1 --steps 5 --warmup 2 --no-merged --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tw1 spin=$v', d['ms_per_step'])"; done
