#!/bin/bash
# round 6, after the capture: (1) the three binned-sweep workloads' kernel-trace half again (bench.py prices the binned sweep with the sweep
# formula since; the library -- and with it the counter summaries -- is unchanged), (2) the unprofiled driver-form lines on the final build
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
CAP=$ROOT/gpurun_out/r06cap2; mkdir -p $CAP
cd $ROOT
retime() { # tag, bench args
  TAG=$1; shift
  bash $ROOT/tools/prof_timeline.sh r06_$TAG --no-merged --no-extra --no-extra-passes "$@" > $CAP/timeline_$TAG.log 2>&1 || true
  T=$ROOT/gpurun_out/timeline_r06_$TAG
  cp $T/kernel_stats.csv $CAP/r06_kernel_stats_$TAG.csv; cp $T/timeline.json $CAP/r06_batch_timeline_$TAG.json
  cp $T/timeline.txt $CAP/r06_batch_timeline_$TAG.txt; cp $T/bench.json $CAP/r06_bench_${TAG}_1gpu.json; rm -rf $T
}
retime livejournal_1src --config livejournal --sources 1 --steps 20 --warmup 5
retime twitter_1src --config twitter --sources 1 --steps 8 --warmup 2
retime friendster_1src --config friendster --sources 1 --steps 6 --warmup 2
python bench.py --steps 20 --warmup 5 > $CAP/r06_bench_default_driver_form.json 2> $CAP/default.err
python bench.py --steps 20 --warmup 5 --prestage --no-extra --no-cpu-baseline > $CAP/r06_bench_default_prestage.json 2> $CAP/prestage.err
python bench.py --config friendster --steps 20 --warmup 5 --no-cpu-baseline > $CAP/r06_bench_friendster_group_steps20.json 2> $CAP/fr20.err
python bench.py --gpus 2 --config twitter --steps 20 --warmup 5 --no-cpu-baseline > $CAP/r06_bench_twitter_2ranks_steps20.json 2> $CAP/tw2.err
python bench.py --gpus 2 --steps 8 --warmup 2 --no-cpu-baseline --strong-steps 3 > $CAP/r06_bench_default_2ranks.json 2> $CAP/def2.err
for f in $CAP/r06_bench_*.json; do python3 - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().splitlines()[-1]); r=d['roofline']
    print(sys.argv[1].split('/')[-1], 'ms', d['ms_per_step'], 'frac', r['frac'], 'traffic_frac', r.get('frac_traffic'), 'parity', d['parity']['ok'], d['parity'].get('sources_compared'), d['parity'].get('max_abs_dp'), 'hbm', d['config']['hbm']['hbm_peak_bytes']/1e9, 'strong', (d.get('configs3_strong') or {}).get('ms_per_step'), (d.get('configs3_strong') or {}).get('per_rank_ms_per_step'))
except Exception as ex: print(sys.argv[1], 'NO LINE', ex)
PY
done
