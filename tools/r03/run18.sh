cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_engine_gpu.py tests/test_cli.py -x -q -m gpu 2>&1 | tail -8
