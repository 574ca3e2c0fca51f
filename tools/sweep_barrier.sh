#!/bin/bash
# Diagnostic: grid-barrier cost of the resident sweep for replica counts / poll pacing.
cd $GRAFT_REPO_ROOT
for CFG in "4 1" "8 1" "16 1" "32 1" "16 4" "16 16" "8 8" "32 8"; do
set -- $CFG
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -ffp-contract=off -munsafe-fp-atomics \
   -DDPPR_BAR_REPS=$1 -DDPPR_BAR_SLEEP=$2 -o /tmp/libdppr_sweep.so dynamicppr_amd/csrc/dppr_engine.hip
echo -n "reps=$1 sleep=$2  "
DPPR_LIB=/tmp/libdppr_sweep.so python3 bench.py --no-cpu-baseline --steps 40 --warmup 5 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['avg_launch_us'])"
done
