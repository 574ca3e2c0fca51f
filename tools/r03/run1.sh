cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "binned" 2>&1 | tail -15
for t in "" "--tune binned=0"; do
  tag=$( [ -z "$t" ] && echo bin || echo nobin )
  timeout 600 python bench.py --config twitter --steps 5 --warmup 2 --no-cpu-baseline $t > gpurun_out/r03_tw_$tag.json 2> gpurun_out/r03_tw_$tag.err
  echo "== twitter $tag rc=$?"; tail -c 1500 gpurun_out/r03_tw_$tag.json | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['pull_iterations_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['parity'])" 2>&1 | tail -3
  tail -3 gpurun_out/r03_tw_$tag.err
done
