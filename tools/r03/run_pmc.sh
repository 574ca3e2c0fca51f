cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
bash tools/r02/prof_pmc.sh r03_livejournal_group10 --steps 6 --warmup 2 --no-extra --no-merged | grep "k_gsweep\|k_gpush_expand"
bash tools/r02/prof_pmc.sh r03_twitter_group8 --config twitter --steps 3 --warmup 1 --no-merged | grep "k_gsweep"
bash tools/r02/prof_pmc.sh r03_lj1_binned --config livejournal --sources 1 --pick top10 --steps 6 --warmup 2 --no-merged | grep "k_bin_scatter\|k_bin_reduce"
bash tools/r02/prof_pmc.sh r03_youtube_1src --config youtube --steps 20 --warmup 3 --no-merged | grep "k_pull_resident"
