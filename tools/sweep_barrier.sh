#!/bin/bash
# Diagnostic: pacing of the first gather round of the data-flow resident sweep (s_sleep units of 64 clocks).
cd $GRAFT_REPO_ROOT
for PACE in 0 8 16 24 32 48 64; do
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -ffp-contract=off -munsafe-fp-atomics \
   -DDPPR_FLOW_PACE=$PACE -o /tmp/libdppr_sweep.so dynamicppr_amd/csrc/dppr_engine.hip
echo -n "pace=$PACE  "
DPPR_LIB=/tmp/libdppr_sweep.so python3 bench.py --no-cpu-baseline --steps 40 --warmup 5 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['avg_launch_us'])"
done
