cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
S=$(date +%s); timeout 1500 python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err; echo "default bench rc=$? wall=$(( $(date +%s) - S )) s"
tail -2 gpurun_out/r03_bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_bench_default.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], json.dumps(d['parity']), json.dumps({k:d['roofline'][k] for k in ('frac','frac_group_adjusted','frac_traffic','avg_launch_us')}), json.dumps(d['cpu_baseline'])[:600], d['config']['stream_file'].get('origin'))
PY
