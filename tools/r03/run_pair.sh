cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
for cfg in twitter friendster; do timeout 900 python bench.py --config $cfg --sources 2 --steps 4 --warmup 2 --no-cpu-baseline 2>gpurun_out/pair_$cfg.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg two sources', d['ms_per_step'], d['value'], d['roofline']['kernel'][:30], d['roofline']['frac'], d['parity']['ok'], d['parity']['sources_checked'], (d.get('merged_loop') or {}).get('ms_per_step'), d['config']['workload'][-90:])"; tail -2 gpurun_out/pair_$cfg.err; done
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --config twitter --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('twitter 2 ranks on one GPU', d['ms_per_step'], d['config']['parallelism'], d['parity']['ok'])"
timeout 600 python -m pytest tests/test_bench_gpu.py tests/test_cli.py -x -q -m gpu 2>&1 | tail -2
