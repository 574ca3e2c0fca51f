cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_fullsize_golden_gpu.py -x -q -m gpu -k "friendster" -s 2>&1 | grep "parity\|passed\|failed\|Error\|assert" | tail -8
timeout 600 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "binned or two_engines" 2>&1 | tail -2
