cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
S=$(date +%s); timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -8; echo "wall $(( $(date +%s) - S )) s"
