#!/bin/bash
# The mechanisms most likely to hang or race, repeated (round 5: with the graph build running beside the solve): the builder /
# solver pair of dppr_slide_concurrent on a churning stream (single source, source group, push-only), ./pagerank's overlapped
# loop end to end incl. -g N device threads on one device, two engines from two threads, the post-mortem dump. Every ./pagerank
# child runs under its watchdog (DPPR_WATCHDOG_S, exit code 125 with a dump). usage: tools/r05/repeat_suspects.sh <repetitions> [tag]
N=${1:-50}; TAG=${2:-a}
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05_suspects_$TAG.log
: > $OUT
ok=0
t0=$(date +%s)
for i in $(seq 1 $N); do
  if timeout 600 python3 -m pytest tests/test_overlap_gpu.py tests/test_engine_gpu.py tests/test_cli.py -m gpu -x -q \
       -k "next_graph_is_built or two_engines_on_one_device or device_threads_share_the_device or debug_dump or cli_end_to_end or churning" > /tmp/rep.log 2>&1; then
    ok=$((ok+1))
  else
    echo "=== repetition $i FAILED" >> $OUT; cat /tmp/rep.log >> $OUT
  fi
done
echo "repetitions $N clean $ok in $(( $(date +%s) - t0 )) s, box $(hostname) $(date -u +%FT%TZ)" | tee -a $OUT
tail -3 /tmp/rep.log | tee -a $OUT
