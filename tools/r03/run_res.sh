# resident sweep changes: the tests that reach k_pull_resident, then the A/B of dynamicppr_amd/libdppr_hip_prev.so against the in-tree build
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/res
timeout 1500 python -m pytest tests/test_engine_gpu.py tests/test_cli.py -x -q -k "resident or rollcall or merged or pull or update or default or cli" > gpurun_out/res/tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/res/tests.log
[ -f dynamicppr_amd/libdppr_hip_prev.so ] && bash tools/r03/run_ab.sh
