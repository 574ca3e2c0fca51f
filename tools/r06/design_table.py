#!/usr/bin/env python3
"""Prints the tables of DESIGN.md section 6 / 9 from the committed round-6 artefacts (profiles/r06_bench_*_1gpu.json,
profiles/r06_sweep_*.jsonl), so that the document quotes the files and not a transcription of them."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
R05 = {"livejournal_group10": "12.2 / 56.5 M", "youtube_1src": "0.4737", "dblp_1src": "0.2933", "livejournal_1src": "5.016", "twitter_1src": "93.25",
       "twitter_group": "247.9", "friendster_1src": "124.5", "friendster_group": "541.8"}
NAMES = {"livejournal_group10": "configs[2] LiveJournal, 10 sources (headline)", "youtube_1src": "configs[1] youtube, 1 source", "dblp_1src": "configs[0] dblp, 1 source",
         "livejournal_1src": "LiveJournal, 1 source", "twitter_1src": "configs[3] twitter, 1 source (`--sources 1`)", "twitter_group": "configs[3] twitter, all 8 sources as one group",
         "friendster_1src": "configs[4] friendster, 1 source", "friendster_group": "configs[4] friendster, all 10 sources as one group"}
for tag in NAMES:
    d = json.loads(open(os.path.join(ROOT, "profiles", f"r06_bench_{tag}_1gpu.json")).read().strip().splitlines()[-1])
    r = d["roofline"]
    k = r["kernel"].split(" (")[0]
    print(f"| {NAMES[tag]} | {d['ms_per_step']:.4g} | {d['value'] / 1e6:.3g} M | `{k}` {r['frac']:.2f} / {r['frac_traffic']:.2f} ({r['traffic'] / 1e9:.3g} GB per launch, "
          f"{r['avg_launch_us']:.0f} µs) | {R05[tag]} |")
