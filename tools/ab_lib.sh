#!/bin/bash
# A/B of two builds of the library on the same box: tools/ab_lib.sh <other.so> -- runs the given bench
# configs with the default library and with DPPR_LIB=<other.so>
OTHER=$1; shift
for lib in "" "$OTHER"; do
  for args in "--config livejournal --sources 1 --pick top10 --steps 10 --warmup 3" "--config twitter --steps 4 --warmup 2" "--config livejournal --steps 10 --warmup 3"; do
    DPPR_LIB=$lib python bench.py --no-cpu-baseline $args 2>/dev/null | python -c "
import json,sys,os; d=json.loads(sys.stdin.read()); print('lib=', os.environ.get('DPPR_LIB') or 'default', d['config']['workload'][:12], len(d['config']['sources']), 'src', d['ms_per_step'], d['roofline']['avg_launch_us'], d['parity']['ok'])"
  done
done
