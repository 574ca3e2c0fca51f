cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
for i in 1 2 3; do timeout 600 python -m pytest tests/test_engine_gpu.py tests/test_cli.py tests/test_fullsize_gpu.py -x -q -m gpu -k "merged or merge or (production_mode and youtube)" 2>&1 | tail -1; done
for t in "--tune merge_phases=4" ""; do timeout 600 python bench.py --config youtube --steps 40 --warmup 5 --no-cpu-baseline $t 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('youtube $t', d['ms_per_step'], d['iterations_per_step'], (d.get('merged_loop') or {}).get('ms_per_step'), d['parity']['ok'])"; done
DPPR_SOAK_MERGE=1 timeout 600 python tools/soak.py 21 200 16 2>&1 | tail -1
DPPR_SOAK_MERGE=1 timeout 600 python tools/soak.py 22 200 16 2>&1 | tail -1
