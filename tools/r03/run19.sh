cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
python - <<'PY'
import sys; sys.path.insert(0, '.')
from dynamicppr_amd import datagen
p = datagen.ensure_stand_in("youtube", "/tmp/dppr_data")
V, e1, e2 = datagen.read_bin(p)
W = int(len(e1) * 0.1)
print(p, int(datagen.top_sources(V, e1, e2, W, 0, 10)[3]))
PY
F=/tmp/dppr_data/com-youtube.ungraph.rmat20.s2.bin
S=$(python -c "
import sys; sys.path.insert(0,'.')
from dynamicppr_amd import datagen
V,e1,e2=datagen.read_bin('$F'); print(int(datagen.top_sources(V,e1,e2,int(len(e1)*0.1),0,10)[3]))")
python tools/sweep.py variant --data $F --directed 0 --source $S --log-dir /tmp/sweeplog | tee gpurun_out/r03_variant_ablation_youtube.jsonl
python tools/sweep.py variant --data $F --directed 0 --source $S --log-dir /tmp/sweeplog > gpurun_out/r03_variant_ablation_youtube_run2.jsonl
