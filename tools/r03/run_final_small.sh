# refresh of the committed bench lines / timeline / counters of the resident-path configurations (configs[0], [1]) with the final build
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/final
b() { tag=$1; shift; timeout 900 python bench.py "$@" > gpurun_out/final/r03_bench_$tag.json 2> gpurun_out/final/r03_bench_$tag.err; echo "$tag rc=$?"; }
b youtube_1src_1gpu --config youtube --steps 40 --warmup 5
b dblp_1src_1gpu --config dblp --steps 40 --warmup 5
bash tools/prof_timeline.sh youtube_1src --config youtube --steps 40 --warmup 5 --no-merged > /dev/null 2>&1
bash tools/r02/prof_pmc.sh r03_youtube_1src --config youtube --steps 20 --warmup 3 --no-merged | grep "k_pull_resident"
python - <<'PY'
import json
for t in ('youtube_1src_1gpu','dblp_1src_1gpu'):
    d=json.loads([l for l in open(f'gpurun_out/final/r03_bench_{t}.json') if l.startswith('{')][-1]); r=d['roofline']; m=d.get('merged_loop') or {}
    print(t, d['ms_per_step'], r['frac'], r['frac_traffic'], r['avg_launch_us'], d['parity']['ok'], d['parity'].get('max_abs_dp_vs_cpu_t1'), 'merged', m.get('ms_per_step'), (m.get('parity') or {}).get('max_abs_dp_vs_cpu_t1'))
PY
head -8 gpurun_out/timeline_youtube_1src/timeline.txt | cut -c1-160
