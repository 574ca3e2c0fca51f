#!/usr/bin/env python3
"""Prints the tables of DESIGN.md section 6 / 9 from the committed round-5 artefacts (profiles/r05_bench_*_1gpu.json,
profiles/r05_sweep_*.jsonl), so that the document quotes the files and not a transcription of them."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
R04 = {"livejournal_group10": "12.11 / 57.0 M", "youtube_1src": "0.435", "dblp_1src": "0.294", "livejournal_1src": "4.74", "twitter_1src": "93.8",
       "twitter_group": "248.0", "friendster_1src": "133.7", "friendster_group": "551.0"}
NAMES = {"livejournal_group10": "configs[2] LiveJournal, 10 sources (headline)", "youtube_1src": "configs[1] youtube, 1 source", "dblp_1src": "configs[0] dblp, 1 source",
         "livejournal_1src": "LiveJournal, 1 source", "twitter_1src": "configs[3] twitter, 1 source (`--sources 1`)", "twitter_group": "configs[3] twitter, all 8 sources as one group",
         "friendster_1src": "configs[4] friendster, 1 source", "friendster_group": "configs[4] friendster, all 10 sources as one group"}
for tag in NAMES:
    d = json.loads(open(os.path.join(ROOT, "profiles", f"r05_bench_{tag}_1gpu.json")).read().strip().splitlines()[-1])
    r = d["roofline"]
    k = r["kernel"].split(" (")[0]
    print(f"| {NAMES[tag]} | {d['ms_per_step']:.4g} | {d['value'] / 1e6:.3g} M | `{k}` {r['frac']:.2f} / {r['frac_traffic']:.2f} ({r['traffic'] / 1e9:.3g} GB per launch, "
          f"{r['avg_launch_us']:.0f} µs) | {R04[tag]} |")
print()
for key in ("youtube", "livejournal"):
    rows = [json.loads(l) for l in open(os.path.join(ROOT, "profiles", f"r05_sweep_batch_ratio_{key}.jsonl"))]
    for feat in ("top10", "top1000", "top1000000"):
        vals = [next(x["ppr_latency_ms"] for x in rows if x["source_feature"] == feat and x["batch_ratio"] == r) for r in (0.01, 0.001, 0.0001)]
        print(f"| {key} | {feat} | " + " | ".join(f"{v:.3g}" for v in vals) + " |")
