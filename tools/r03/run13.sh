cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "merged" 2>&1 | tail -15
