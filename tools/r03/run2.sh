cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "binned" 2>&1 | tail -5
bash tools/r03/stamps_bin.sh twitter 1,128,48,98304
bash tools/r03/sweep_bin.sh twitter binned=1,128,48,98304 binned=1,128,48,49152 binned=1,128,48,196608 binned=1,256,48,98304 binned=1,128,48,98304,0,16384 binned=1,128,48,98304,0,65536
