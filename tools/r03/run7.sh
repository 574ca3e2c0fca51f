cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_bench_gpu.py -x -q -m gpu 2>&1 | tail -5
/usr/bin/time -v -o gpurun_out/r03_bench_default.time timeout 1500 python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err
grep -E "Elapsed|Maximum resident" gpurun_out/r03_bench_default.time; tail -2 gpurun_out/r03_bench_default.err
bash tools/prof_pmc.sh r03_tw1_gather --config twitter --sources 1 --steps 3 --warmup 1 --tune binned=0 | grep "k_pull_iter\|k_bin"
bash tools/prof_pmc.sh r03_tw1_binned --config twitter --sources 1 --steps 3 --warmup 1 | grep "k_pull_iter\|k_bin"
bash tools/prof_pmc.sh r03_fr1_gather --config friendster --sources 1 --steps 3 --warmup 1 --tune binned=0 | grep "k_pull_iter\|k_bin"
bash tools/prof_pmc.sh r03_fr1_binned --config friendster --sources 1 --steps 3 --warmup 1 | grep "k_pull_iter\|k_bin"
