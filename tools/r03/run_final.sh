cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/final
b() { tag=$1; shift; S=$(date +%s); timeout 1500 python bench.py "$@" > gpurun_out/final/r03_bench_$tag.json 2> gpurun_out/final/r03_bench_$tag.err; echo "$tag rc=$? wall=$(( $(date +%s) - S ))s"; python - <<PY
import json
try:
    d=json.loads([l for l in open('gpurun_out/final/r03_bench_$tag.json') if l.startswith('{')][-1])
    r=d['roofline']; m=d.get('merged_loop') or {}
    print('   ms/step', d['ms_per_step'], 'value', d['value'], 'iters', d['iterations_per_step'], 'kernel', r['kernel'][:40], 'launch_us', r['avg_launch_us'], 'frac', r['frac'], 'adj', r['frac_group_adjusted'], 'traffic', r['frac_traffic'], 'parity', d['parity']['ok'], d['parity'].get('max_abs_dp_vs_cpu_t1'), '| merged', m.get('ms_per_step'), (m.get('parity') or {}).get('max_abs_dp_vs_cpu_t1'))
except Exception as ex:
    print('   FAILED', ex); print(open('gpurun_out/final/r03_bench_$tag.err').read()[-400:])
PY
}
b livejournal_group10_1gpu
b youtube_1src_1gpu --config youtube --steps 40 --warmup 5
b dblp_1src_1gpu --config dblp --steps 40 --warmup 5
b livejournal_1src_1gpu --config livejournal --sources 1 --pick top10 --steps 20 --warmup 5
b twitter_group_1gpu --config twitter --steps 8 --warmup 2 --no-cpu-baseline
b twitter_1src_1gpu --config twitter --sources 1 --steps 8 --warmup 2 --no-cpu-baseline
b friendster_group_1gpu --config friendster --steps 6 --warmup 2 --no-cpu-baseline
b friendster_1src_1gpu --config friendster --sources 1 --steps 6 --warmup 2 --no-cpu-baseline
bash tools/prof_timeline.sh livejournal_group10 --steps 12 --warmup 3 --no-extra --no-merged > /dev/null 2>&1
bash tools/prof_timeline.sh twitter_1src --config twitter --sources 1 --steps 4 --warmup 2 --no-merged > /dev/null 2>&1
bash tools/prof_timeline.sh youtube_1src --config youtube --steps 40 --warmup 5 --no-merged > /dev/null 2>&1
bash tools/r02/prof_pmc.sh r03_youtube_1src --config youtube --steps 20 --warmup 3 --no-merged | grep "k_pull_resident"
ls gpurun_out/timeline_livejournal_group10 gpurun_out/timeline_twitter_1src gpurun_out/timeline_youtube_1src
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
