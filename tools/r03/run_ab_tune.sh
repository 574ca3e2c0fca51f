# A/B of two engine tunings through bench.py on one box: tools/r03/run_ab_tune.sh "<tune A>" "<tune B>"   (e.g. resident_update=0 resident_update=1)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/ab
b() { tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline "$@" > gpurun_out/ab/$tag.json 2> gpurun_out/ab/$tag.err; python - <<PY
import json
try:
    d=json.loads([l for l in open('gpurun_out/ab/$tag.json') if l.startswith('{')][-1]); r=d['roofline']; m=d.get('merged_loop') or {}
    print('$tag', 'ms/step', d['ms_per_step'], 'iters', d['iterations_per_step'], 'launch_us', r['avg_launch_us'], 'frac', r['frac'], 'parity', d['parity']['ok'], d['parity'].get('max_abs_dp_vs_cpu_t1'), 'merged', m.get('ms_per_step'))
except Exception as ex:
    print('$tag FAILED', ex); print(open('gpurun_out/ab/$tag.err').read()[-600:])
PY
}
for rep in 1 2 3; do
  b yt_A_$rep --config youtube --steps 60 --warmup 10 --tune $1
  b yt_B_$rep --config youtube --steps 60 --warmup 10 --tune $2
done
for rep in 1 2; do
b dblp_A_$rep --config dblp --steps 60 --warmup 10 --tune $1
b dblp_B_$rep --config dblp --steps 60 --warmup 10 --tune $2
done
