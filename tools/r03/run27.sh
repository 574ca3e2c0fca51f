cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_engine_gpu.py tests/test_cli.py tests/test_renumbering_gpu.py -x -q -m gpu 2>&1 | tail -4
for t in "" "--tune group_at_slide=0"; do timeout 600 python bench.py --config youtube --steps 40 --warmup 5 --no-cpu-baseline $t 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('youtube $t', d['ms_per_step'], (d.get('merged_loop') or {}).get('ms_per_step'))"; done
bash tools/prof_timeline.sh youtube_1src --config youtube --steps 40 --warmup 5 --no-merged | head -12
