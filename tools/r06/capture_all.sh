#!/bin/bash
# All eight round-6 profile sets on ONE box with ONE build (tools/r06/capture.sh each): counter traffic, kernel stats, batch timeline
# and the bench line of the profiled run. usage: GIT_COMMIT=<sha> tools/r06/capture_all.sh   -> gpurun_out/r06cap/
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cap() { bash $ROOT/tools/r06/capture.sh "$@" 2>&1 | tail -1; }
cap livejournal_group10 livejournal_group10
cap youtube_1src youtube_1src --config youtube --steps 40 --warmup 5
cap dblp_1src dblp_1src --config dblp --steps 40 --warmup 5
cap livejournal_1src livejournal_1src --config livejournal --sources 1 --steps 20 --warmup 5
cap twitter_1src twitter_1src --config twitter --sources 1 --steps 8 --warmup 2
cap twitter_group twitter_group8 --config twitter --steps 6 --warmup 2
cap friendster_1src friendster_1src --config friendster --sources 1 --steps 6 --warmup 2
cap friendster_group friendster_group10 --config friendster --steps 4 --warmup 2
