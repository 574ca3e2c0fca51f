for cfg in "1 1" "0 1" "1 0" "0 0"; do set -- $cfg; export DPPR_SWEEP_BITS=$1 DPPR_HOT_BLOCKS=$2
 for args in "--config livejournal --sources 1 --pick top10 --steps 10 --warmup 3" "--config twitter --sources 1 --steps 4 --warmup 2"; do
  python bench.py --no-cpu-baseline $args 2>/dev/null | python -c "
import json,sys,os; d=json.loads(sys.stdin.read()); print('bits',os.environ['DPPR_SWEEP_BITS'],'blocks',os.environ['DPPR_HOT_BLOCKS'], d['config']['workload'][:12], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['parity']['ok'])"; done; done
