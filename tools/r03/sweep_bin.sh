#!/bin/bash
# A/B of the binned sweep's block shapes on a stand-in: tools/r03/sweep_bin.sh <config> "<tune1>" "<tune2>" ...
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
CFG=$1; shift
for t in "$@"; do
  tag=$(echo "$t" | tr ',=' '__')
  timeout 900 python bench.py --config $CFG --steps 5 --warmup 2 --no-cpu-baseline $BENCH_EXTRA --tune "$t" > gpurun_out/sw_${CFG}_$tag.json 2> gpurun_out/sw_${CFG}_$tag.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/sw_${CFG}_$tag.json').read().strip().splitlines()[-1])
    print('$CFG $t', 'ms/step', d['ms_per_step'], 'pull/step', d['pull_iterations_per_step'], 'launch_us', d['roofline']['avg_launch_us'], 'frac', d['roofline']['frac'], 'parity', d['parity']['ok'])
except Exception as ex:
    print('$CFG $t FAILED', ex); print(open('gpurun_out/sw_${CFG}_$tag.err').read()[-600:])
PY
done
