# soak of the resident sweep's data-flow protocol: medium-size randomised streams against the oracle, and a long configs[1] run
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/soak
for seed in 1 2 3 4 5 6; do
  timeout 900 python tools/soak.py $seed 250 16 > gpurun_out/soak/soak_$seed.log 2>&1; echo "soak seed $seed rc=$? $(tail -1 gpurun_out/soak/soak_$seed.log | cut -c1-200)"
done
for seed in 7 8; do
  DPPR_SOAK_MERGE=1 timeout 900 python tools/soak.py $seed 250 16 > gpurun_out/soak/soak_m$seed.log 2>&1; echo "soak merged seed $seed rc=$? $(tail -1 gpurun_out/soak/soak_m$seed.log | cut -c1-200)"
done
timeout 1200 python bench.py --config youtube --steps 850 --warmup 5 --no-cpu-baseline > gpurun_out/soak/yt_long.json 2> gpurun_out/soak/yt_long.err; echo "long youtube rc=$?"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/soak/yt_long.json') if l.startswith('{')][-1])
print('youtube 850 pre-staged batches: ms/step', d['ms_per_step'], 'parity', d['parity'], 'merged', (d.get('merged_loop') or {}).get('ms_per_step'))
PY
