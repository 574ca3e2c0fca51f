#!/bin/bash
# three sources on a big window: in series (default since round 6) against one source group (--force-group)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
python -m pytest tests/test_bench_gpu.py -x -q 2>&1 | tail -3
Q="--steps 4 --warmup 2 --no-cpu-baseline --no-extra --no-merged --no-extra-passes --no-ceilings"
show() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().splitlines()[-1]); print('$1', d['ms_per_step'], d['parity']['ok'], d['roofline']['kernel'].split(' (')[0])"; }
python3 bench.py --config friendster --sources 3 --pick top1000 $Q 2>/dev/null | show "friendster S=3 series"
python3 bench.py --config twitter --sources 3 --pick top10 $Q 2>/dev/null | show "twitter S=3 series"
python3 bench.py --config twitter --sources 3 --pick top10 --force-group $Q 2>/dev/null | show "twitter S=3 group"
