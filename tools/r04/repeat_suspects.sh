#!/bin/bash
# the two suspects of round 3's one unexplained 300-second guard, repeated. Every ./pagerank child runs under
# its watchdog (DPPR_WATCHDOG_S, tests/test_cli.py), the thread test dumps both engines on a 90-second limit: a hang leaves a
# post-mortem in the log. usage: tools/r04/repeat_suspects.sh <repetitions> [tag]
N=${1:-100}; TAG=${2:-a}
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/suspects_$TAG.log
: > $OUT
ok=0
for i in $(seq 1 $N); do
  if timeout 600 python3 -m pytest tests/test_engine_gpu.py tests/test_cli.py -m gpu -x -q \
       -k "two_engines_on_one_device or device_threads_share_the_device or debug_dump or (cli_end_to_end and split)" > /tmp/rep.log 2>&1; then
    ok=$((ok+1))
  else
    echo "=== repetition $i FAILED" >> $OUT; cat /tmp/rep.log >> $OUT
  fi
done
echo "repetitions $N clean $ok box $(hostname) $(date -u +%FT%TZ)" | tee -a $OUT
tail -3 /tmp/rep.log | tee -a $OUT
