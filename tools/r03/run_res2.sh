cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/res
b() { tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline "$@" > gpurun_out/res/$tag.json 2> gpurun_out/res/$tag.err; python - <<PY
import json
try:
    d=json.loads([l for l in open('gpurun_out/res/$tag.json') if l.startswith('{')][-1]); r=d['roofline']; m=d.get('merged_loop') or {}
    print('$tag', 'ms/step', d['ms_per_step'], 'iters', d['iterations_per_step'], 'launch_us', r['avg_launch_us'], 'frac', r['frac'], 'parity', d['parity']['ok'], 'merged', m.get('ms_per_step'))
except Exception as ex:
    print('$tag FAILED', ex); print(open('gpurun_out/res/$tag.err').read()[-600:])
PY
}
for rep in 1 2 3; do for m in 0 1; do b yt_slots${m}_$rep --config youtube --steps 60 --warmup 10 --tune resident_slots=$m; done; done
for rep in 1 2; do for m in 0 1; do b dblp_slots${m}_$rep --config dblp --steps 60 --warmup 10 --tune resident_slots=$m; done; done
