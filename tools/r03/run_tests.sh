cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
for i in 1 2 3 4 5 6 7 8; do timeout 600 python -m pytest tests/test_engine_gpu.py tests/test_cli.py -x -q -m gpu -k "merged or merge or end_to_end" 2>&1 | tail -1; done
S=$(date +%s); timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -4; echo "wall $(( $(date +%s) - S )) s"
