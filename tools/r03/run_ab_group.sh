# A/B of two library builds on the group sweeps: dynamicppr_amd/libdppr_hip_prev.so (DPPR_LIB) against the in-tree build
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/ab
b() { tag=$1; shift; timeout 900 python bench.py --no-cpu-baseline --no-merged --no-extra "$@" > gpurun_out/ab/$tag.json 2> gpurun_out/ab/$tag.err; python - <<PY
import json
try:
    d=json.loads([l for l in open('gpurun_out/ab/$tag.json') if l.startswith('{')][-1]); r=d['roofline']
    print('$tag', 'ms/step', d['ms_per_step'], 'iters', d['iterations_per_step'], 'launch_us', r['avg_launch_us'], 'frac', r['frac'], 'parity', d['parity']['ok'])
except Exception as ex:
    print('$tag FAILED', ex); print(open('gpurun_out/ab/$tag.err').read()[-600:])
PY
}
for rep in 1 2; do
  DPPR_LIB=$PWD/dynamicppr_amd/libdppr_hip_prev.so b lj_prev_$rep --steps 20 --warmup 5
  b lj_new_$rep --steps 20 --warmup 5
done
if [ -n "$TW" ]; then
DPPR_LIB=$PWD/dynamicppr_amd/libdppr_hip_prev.so b tw_prev --config twitter --steps 5 --warmup 2
b tw_new --config twitter --steps 5 --warmup 2
fi
