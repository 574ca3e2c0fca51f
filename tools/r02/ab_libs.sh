#!/bin/bash
# A/B of several builds of the library on one box: tools/r02/ab_libs.sh "<bench args>" lib1.so lib2.so ... ("" = default build)
ARGS=$1; shift
for lib in "" "$@"; do
  export DPPR_LIB=$lib
  python bench.py --no-cpu-baseline $ARGS 2>/dev/null | python -c "
import json,sys,os; d=json.loads(sys.stdin.read()); print('lib=', os.path.basename(os.environ.get('DPPR_LIB') or 'default'), d['config']['workload'][:12], len(d['config']['sources']), 'src', d['ms_per_step'], d['roofline']['avg_launch_us'], d['parity']['ok'])"
done
