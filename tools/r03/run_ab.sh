# A/B of two library builds on one box: dynamicppr_amd/libdppr_hip_prev.so (DPPR_LIB) against the in-tree build
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/ab
b() { tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline "$@" > gpurun_out/ab/$tag.json 2> gpurun_out/ab/$tag.err; python - <<PY
import json
try:
    d=json.loads([l for l in open('gpurun_out/ab/$tag.json') if l.startswith('{')][-1]); r=d['roofline']; m=d.get('merged_loop') or {}
    print('$tag', 'ms/step', d['ms_per_step'], 'iters', d['iterations_per_step'], 'launch_us', r['avg_launch_us'], 'frac', r['frac'], 'parity', d['parity']['ok'], 'merged', m.get('ms_per_step'))
except Exception as ex:
    print('$tag FAILED', ex); print(open('gpurun_out/ab/$tag.err').read()[-600:])
PY
}
for rep in 1 2 3; do
  DPPR_LIB=$PWD/dynamicppr_amd/libdppr_hip_prev.so b yt_prev_$rep --config youtube --steps 60 --warmup 10
  b yt_new_$rep --config youtube --steps 60 --warmup 10
done
DPPR_LIB=$PWD/dynamicppr_amd/libdppr_hip_prev.so b dblp_prev --config dblp --steps 60 --warmup 10
b dblp_new --config dblp --steps 60 --warmup 10
