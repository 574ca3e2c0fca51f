cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
bash tools/prof_timeline.sh livejournal_group10 --steps 12 --warmup 3 --no-extra | tail -30
bash tools/prof_timeline.sh twitter_1src --config twitter --sources 1 --steps 4 --warmup 2 | tail -16
