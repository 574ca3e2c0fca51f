cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/ab
b() { tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-merged "$@" > gpurun_out/ab/$tag.json 2> gpurun_out/ab/$tag.err; python - <<PY
import json
try:
    d=json.loads([l for l in open('gpurun_out/ab/$tag.json') if l.startswith('{')][-1]); r=d['roofline']
    print('$tag', 'ms/step', d['ms_per_step'], 'iters', d['iterations_per_step'], 'launch_us', r['avg_launch_us'], 'frac', r['frac'], 'parity', d['parity']['ok'])
except Exception as ex:
    print('$tag FAILED', ex); print(open('gpurun_out/ab/$tag.err').read()[-600:])
PY
}
for cfg in youtube dblp; do for pb in 1024 512 256 1024; do b ${cfg}_pb$pb --config $cfg --steps 60 --warmup 10 --tune pull_block=$pb; done; done
