cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_engine_gpu.py tests/test_cli.py tests/test_renumbering_gpu.py -x -q -m gpu 2>&1 | tail -3
for cfg in friendster twitter; do timeout 900 python bench.py --config $cfg --sources 1 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg 1src', d['ms_per_step'], d['roofline']['frac'], d['parity']['ok'], (d.get('merged_loop') or {}).get('ms_per_step'))"; done
bash tools/prof_timeline.sh friendster_1src --config friendster --sources 1 --steps 4 --warmup 2 --no-merged | grep "k_push\|median span"
