#!/bin/bash
# the unprofiled driver-form lines on the final build (after tools/r06/capture_all.sh)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
CAP=$ROOT/gpurun_out/r06cap2; mkdir -p $CAP
cd $ROOT
python bench.py --steps 20 --warmup 5 > $CAP/r06_bench_default_driver_form.json 2> $CAP/default.err
python bench.py --steps 20 --warmup 5 --prestage --no-extra --no-cpu-baseline > $CAP/r06_bench_default_prestage.json 2> $CAP/prestage.err
python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline > $CAP/ring_again.json 2> $CAP/ring_again.err
python bench.py --config friendster --steps 20 --warmup 5 --no-cpu-baseline > $CAP/r06_bench_friendster_group_steps20.json 2> $CAP/fr20.err
python bench.py --gpus 2 --config twitter --steps 20 --warmup 5 --no-cpu-baseline > $CAP/r06_bench_twitter_2ranks_steps20.json 2> $CAP/tw2.err
python bench.py --gpus 2 --steps 8 --warmup 2 --no-cpu-baseline --strong-steps 3 > $CAP/r06_bench_default_2ranks.json 2> $CAP/def2.err
for f in $CAP/*.json; do python3 - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().splitlines()[-1]); r=d['roofline']
    print(sys.argv[1].split('/')[-1], 'ms', d['ms_per_step'], 'frac', r['frac'], 'traffic_frac', r.get('frac_traffic'), 'parity', d['parity']['ok'], d['parity'].get('sources_compared'), d['parity'].get('max_abs_dp'), 'hbm', d['config']['hbm']['hbm_peak_bytes']/1e9, 'strong', (d.get('configs3_strong') or {}).get('ms_per_step'), (d.get('configs3_strong') or {}).get('per_rank_ms_per_step'))
except Exception as ex: print(sys.argv[1], 'NO LINE', ex)
PY
done
