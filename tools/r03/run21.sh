cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "incremental_batch_update or seed_lists or merged or eager_schedule" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_fullsize_golden_gpu.py -x -q -m gpu -k "twitter_single" 2>&1 | tail -3
bash tools/prof_timeline.sh twitter_1src --config twitter --sources 1 --steps 4 --warmup 2 | grep "k_su\|median span\|k_bin"
