# resident sweep changes: tests that reach k_pull_resident, then the A/B against dynamicppr_amd/libdppr_hip_prev.so
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/res
timeout 1500 python -m pytest tests/test_engine_gpu.py tests/test_cli.py -x -q -k "resident or rollcall or merged or pull or update or default or cli" > gpurun_out/res/tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/res/tests.log
bash tools/r03/run_ab.sh
