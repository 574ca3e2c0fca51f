cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
for k in livejournal twitter; do timeout 1200 python tools/slide_costs.py $k 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['config'], d['variant'], 'set_batch', d['set_batch_ms'], 'slide', d['slide_ms'], 'renumbering slide', d.get('renumbering_slide_ms'), json.dumps(d.get('renumbering_slide_phases_ms') or d['slide_phases_ms']))
"; done
timeout 1200 python -m pytest tests/test_renumbering_gpu.py -x -q -m gpu 2>&1 | tail -3
