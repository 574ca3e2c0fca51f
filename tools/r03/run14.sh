cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
run() { tag=$1; shift; timeout 900 python bench.py "$@" > gpurun_out/mp_$tag.json 2> gpurun_out/mp_$tag.err; python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/mp_$tag.json').read().strip().splitlines()[-1])
    print('$tag', 'ms/step', d['ms_per_step'], 'value', d['value'], 'iters/step', d['iterations_per_step'], 'frac', d['roofline']['frac'], 'parity', d['parity']['ok'], d['parity']['max_abs_residual'], d['parity'].get('max_abs_dp_vs_cpu_t1'))
except Exception as ex:
    print('$tag FAILED', ex); print(open('gpurun_out/mp_$tag.err').read()[-500:])
PY
}
run lj10_two --steps 20 --warmup 5 --no-extra
run lj10_merged --steps 20 --warmup 5 --no-extra --tune merge_phases=4
run yt1_two --config youtube --steps 40 --warmup 5
run yt1_merged --config youtube --steps 40 --warmup 5 --tune merge_phases=4
run tw1_two --config twitter --sources 1 --steps 5 --warmup 2 --no-cpu-baseline
run tw1_merged --config twitter --sources 1 --steps 5 --warmup 2 --no-cpu-baseline --tune merge_phases=4
run tw8_merged --config twitter --steps 5 --warmup 2 --no-cpu-baseline --tune merge_phases=4
