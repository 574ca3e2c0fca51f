cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_bench_gpu.py tests/test_cli.py -x -q -m gpu 2>&1 | tail -6
S=$(date +%s); timeout 1500 python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err; echo "default bench rc=$? wall=$(( $(date +%s) - S )) s"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_bench_default.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], json.dumps(d['merged_loop']), json.dumps(d['configs1_single_source']))
PY
