cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 6 --warmup 2 > gpurun_out/r03_bench_2ranks_1gpu.json 2> gpurun_out/r03_bench_2ranks_1gpu.err; echo "rc=$?"
tail -3 gpurun_out/r03_bench_2ranks_1gpu.err | cut -c1-300
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r03_bench_2ranks_1gpu.json') if l.startswith('{')][-1])
print(d['n_gpus'], d['scaling'], d['ms_per_step'], d['value'], d['config']['parallelism'], d['parity']['ok'], d.get('merged_loop'))
PY
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --config twitter --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03_bench_tw_2ranks_1gpu.json 2> gpurun_out/r03_bench_tw_2ranks_1gpu.err; echo "rc=$?"
tail -2 gpurun_out/r03_bench_tw_2ranks_1gpu.err | cut -c1-300
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r03_bench_tw_2ranks_1gpu.json') if l.startswith('{')][-1])
print(d['n_gpus'], d['scaling'], d['ms_per_step'], d['value'], d['config']['parallelism'], d['config']['sources'], d['parity']['ok'], d.get('single_gpu_group_alternative'))
PY
