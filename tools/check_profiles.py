#!/usr/bin/env python3
"""Keeps the committed profile artefacts honest with each other.

    check_profiles.py --cut <kernel_trace.csv> <kernel_stats.csv> <outdir>    (used by tools/prof_timeline.sh)
        writes kernel_stats.csv (short kernel names), timeline.json and timeline.txt from ONE profiler run
    check_profiles.py [profiles/]                                              (CPU test suite: tests/test_profiles.py)
        for every pair r*_kernel_stats_<tag>.csv / r*_batch_timeline_<tag>.json: the timeline's whole-run totals per
        kernel must equal the stats file's within 10 % (two files of one run), and every kernel's average launch inside
        the chosen batch must lie within the stats file's [min, max] for that kernel (a batch from another build, or a
        pathological one, does not). Exit status 1 on any disagreement.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    n = name.split("(")[0].replace("void ", "").strip()
    n = re.sub(r"rocprim::ROCPRIM_\d+_NS::detail::", "rocprim::", n)
    if n.startswith("rocprim::trampoline_kernel"):
        m = re.search(r"(radix_sort_onesweep\w*|partition_impl|merge_impl|radix_sort_block_sort\w*|radix_sort_merge\w*|lookback_scan\w*|scan\w*|histogram\w*)", name)
        n = "rocprim::" + (m.group(1) if m else "kernel")
    return n.replace("dppr::", "")[:60]


def cut(trace_csv, stats_csv, outdir):
    rows = list(csv.DictReader(open(trace_csv)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ev = [(short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
    whole = defaultdict(lambda: [0, 0, 10**18, 0])
    for n, s, e in ev:
        w = whole[n]
        w[0] += 1; w[1] += e - s; w[2] = min(w[2], e - s); w[3] = max(w[3], e - s)
    with open(os.path.join(outdir, "kernel_stats.csv"), "w") as f:
        f.write("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs\n")
        for n, w in sorted(whole.items(), key=lambda kv: -kv[1][1]):
            f.write(f'"{n}",{w[0]},{w[1]},{w[1] / w[0]:.1f},{w[2]},{w[3]}\n')
    # First kernel of every batch's timed region: the IncrementalBatchUpdate kernel (k_su_apply_fused, or k_su_terms of the
    # two-kernel form), or the k_su_keys just before it (counter clear; with dppr_set_batch_grouping(0) k_su_keys + the
    # device sort). The k_su_keys of a slide-time grouping belongs to the SLIDE and is not a mark.
    # With the update applied inside the resident launch (PLAN_UPDATE) a batch is k_su_keys (counter clear) -> k_pull_resident.
    # Round 5 (grouping and CopyOutDegree inside the timed region by default): a batch starts at k_su_group_rank (one launch: degrees,
    # grouping, counter clear) or, beyond 4 Ki records, at k_su_grp_hist -> k_su_grp_scatter -> k_su_grp_rank (round 6; rounds 1-5 and
    # DPPR_GROUPING_RADIX=1: k_copy_out_degree -> k_su_keys -> device sort).
    marks = []
    for i, (n, _, _) in enumerate(ev):
        if n == "k_su_keys" and i + 1 < len(ev) and ev[i + 1][0].startswith("k_pull_resident"):
            marks.append(i)
            continue
        if n not in ("k_su_apply_fused", "k_su_terms"):
            continue
        j = i - 1
        while j >= 0 and ev[j][0].startswith("rocprim::radix"):
            j -= 1
        if j >= 0 and ev[j][0] == "k_su_group_rank":
            marks.append(j)
            continue
        if j >= 2 and ev[j][0] == "k_su_grp_rank" and ev[j - 1][0] == "k_su_grp_scatter" and ev[j - 2][0] == "k_su_grp_hist":
            marks.append(j - 2)   # round 6: the hand-written grouping of a batch beyond 4 Ki records (its first launch also does CopyOutDegree)
            continue
        if j >= 0 and ev[j][0] == "k_su_keys":
            marks.append(j - 1 if j >= 1 and ev[j - 1][0] == "k_copy_out_degree" else j)
            continue
        marks.append(i)
    batches = [(marks[i], marks[i + 1]) for i in range(len(marks) - 1)]
    spans = []
    slide_kernels = ("k_make", "k_deg", "k_mark", "k_del_pos", "k_merge_tiles", "k_build", "k_assign", "k_tile", "k_gather_deg", "k_gtables", "k_bin_keys", "k_bin_fill", "k_bin_vertex",
                     "k_bin_quant", "k_bin_big", "k_bin_count", "k_bin_btables", "k_bin_atables", "k_bin_tb", "k_bin_swap", "k_bin_group", "k_in_degree", "k_number", "k_live", "k_remap", "k_permute", "k_rows", "k_res_")
    for lo, hi in batches:                                                       # a batch ends where the next slide's kernels begin
        seg = []
        for x in ev[lo:hi]:
            if x[0].startswith(slide_kernels):
                break
            seg.append(x)
        spans.append((seg[-1][2] - seg[0][1], lo, seg))
    spans.sort(key=lambda t: t[0])
    span, lo, seg = spans[len(spans) // 2]
    busy = sum(e - s for _, s, e in seg)
    per = defaultdict(lambda: [0, 0])
    for n, s, e in seg:
        per[n][0] += 1; per[n][1] += e - s
    gaps = sorted(((seg[i + 1][1] - seg[i][2]) / 1e3 for i in range(len(seg) - 1)))
    tl = {"batches_in_run": len(batches), "batch_spans_us": [round(s[0] / 1e3, 1) for s in sorted(spans, key=lambda t: t[1])],
          "chosen": "the batch of median span", "span_us": round(span / 1e3, 1), "busy_us": round(busy / 1e3, 1), "dispatches": len(seg),
          "median_gap_us": round(gaps[len(gaps) // 2], 2) if gaps else None,
          "kernels": {n: {"launches": c, "total_us": round(t / 1e3, 1), "avg_us": round(t / c / 1e3, 2)} for n, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1])},
          "iteration_kernel_durations_us": [round((e - s) / 1e3, 1) for n, s, e in seg if n.startswith(("k_gsweep", "k_pull_iter", "k_bin_reduce", "k_pull_resident"))],
          "whole_run": {n: {"launches": w[0], "total_us": round(w[1] / 1e3, 1)} for n, w in whole.items()}}
    json.dump(tl, open(os.path.join(outdir, "timeline.json"), "w"), indent=1)
    with open(os.path.join(outdir, "timeline.txt"), "w") as f:
        f.write(f"{len(batches)} batches in the run, spans us: {tl['batch_spans_us']}\n")
        f.write(f"batch of median span: {tl['dispatches']} dispatches, span {tl['span_us']} us, busy {tl['busy_us']} us, median gap {tl['median_gap_us']} us\n")
        for n, v in tl["kernels"].items():
            f.write(f"  {n:44s} x{v['launches']:4d}  total {v['total_us']:10.1f} us  avg {v['avg_us']:9.2f} us\n")
        f.write(f"iteration kernels in launch order (us): {tl['iteration_kernel_durations_us']}\n")


def check(directory):
    bad = 0
    pairs = 0
    for tl_path in sorted(glob.glob(os.path.join(directory, "r*_batch_timeline_*.json"))):
        tag = re.sub(r"^r\d+_batch_timeline_", "", os.path.basename(tl_path))[:-5]
        rnd = os.path.basename(tl_path).split("_")[0]
        st_path = os.path.join(directory, f"{rnd}_kernel_stats_{tag}.csv")
        if not os.path.exists(st_path):
            print(f"FAIL {os.path.basename(tl_path)}: no {os.path.basename(st_path)} next to it")
            bad += 1
            continue
        pairs += 1
        tl = json.load(open(tl_path))
        stats = {r["Name"]: r for r in csv.DictReader(open(st_path))}
        for n, w in tl["whole_run"].items():
            if n not in stats:
                print(f"FAIL {tag}: {n} is in the timeline's run but not in the stats file")
                bad += 1
                continue
            tot = float(stats[n]["TotalDurationNs"]) / 1e3
            if w["total_us"] > 50 and abs(tot - w["total_us"]) > 0.10 * max(tot, w["total_us"]):
                print(f"FAIL {tag}: {n}: {w['total_us']:.0f} us in the timeline's run, {tot:.0f} us in the stats file")
                bad += 1
        for n, v in tl["kernels"].items():
            if n in stats and not (float(stats[n]["MinNs"]) / 1e3 * 0.999 <= v["avg_us"] <= float(stats[n]["MaxNs"]) / 1e3 * 1.001):
                print(f"FAIL {tag}: {n}: average {v['avg_us']} us in the chosen batch, outside [{float(stats[n]['MinNs']) / 1e3:.1f}, {float(stats[n]['MaxNs']) / 1e3:.1f}] of the stats file")
                bad += 1
        if not tl.get("iteration_kernel_durations_us"):
            print(f"FAIL {tag}: the chosen batch holds no iteration kernel -- the cut is not a batch's timed region")
            bad += 1
        if tl["busy_us"] > tl["span_us"] * 1.005:   # (the profiler's timestamps of adjacent dispatches may overlap by a fraction of a microsecond each)
            print(f"FAIL {tag}: busy {tl['busy_us']} us exceeds the span {tl['span_us']} us")
            bad += 1
    # Round 4 on: a committed bench line's roofline must be reproducible from the kernel-stats file of the same command --
    # frac = algorithmic bytes per launch / (average launch time of the dominant kernel(s) in the CSV) / peak, within 10 %
    # (friendster's 0.32 came from another box's run than the committed CSV, which gives 0.27).
    lines = 0
    for b_path in sorted(glob.glob(os.path.join(directory, "r*_bench_*_1gpu.json"))):
        rnd = os.path.basename(b_path).split("_")[0]
        if int(rnd[1:]) < 4:
            continue
        tag = re.sub(r"^r\d+_bench_", "", os.path.basename(b_path))[:-len("_1gpu.json")]
        st_path = os.path.join(directory, f"{rnd}_kernel_stats_{tag}.csv")
        if not os.path.exists(st_path):
            continue   # (a line without a profile of its own makes no claim the profile could contradict)
        line = json.loads(open(b_path).read().strip().splitlines()[-1])
        roof = line.get("roofline") or {}
        heads = [h.strip().split(" ")[0] for h in roof.get("kernel", "").split("(")[0].split("+")]
        stats = list(csv.DictReader(open(st_path)))
        avg_ns = 0.0
        for h in heads:
            rows = [r for r in stats if r["Name"].startswith(h)]
            calls = sum(int(r["Calls"]) for r in rows)
            if not calls:
                print(f"FAIL {tag}: the line's dominant kernel {h} is not in {os.path.basename(st_path)}")
                bad += 1
                avg_ns = 0.0
                break
            avg_ns += sum(float(r["TotalDurationNs"]) for r in rows) / calls
        if avg_ns <= 0:
            continue
        # a "launch" of a two-kernel sweep (k_bin_scatter + k_bin_reduce) is the PAIR as the event bracket of bench.py sees it: the
        # dispatch gap between the two kernels is inside the bracket and in no kernel's duration -- taken from the timeline of the same run
        tl_path = os.path.join(directory, f"{rnd}_batch_timeline_{tag}.json")
        if len(heads) > 1 and os.path.exists(tl_path):
            avg_ns += (len(heads) - 1) * 1e3 * float(json.load(open(tl_path)).get("median_gap_us") or 0.0)
        lines += 1
        per = roof.get("iterations_per_launch", 1.0) if heads[0].startswith("k_pull_resident") else 1.0
        frac_csv = roof["algorithmic_bytes_per_launch"] / (avg_ns * 1e-9) / (roof["peak"] * 1e9)
        # (10 %; 15 % where the bracketed launch is shorter than 100 us: the hipEvent pair around it costs ~6 us that no kernel's duration holds)
        tol = 0.15 if avg_ns < 100e3 else 0.10
        if abs(frac_csv - roof["frac"]) > tol * max(frac_csv, roof["frac"]):
            print(f"FAIL {tag}: roofline.frac {roof['frac']:.3f} in {os.path.basename(b_path)}, {frac_csv:.3f} from {os.path.basename(st_path)} "
                  f"({roof['algorithmic_bytes_per_launch'] / 1e6:.1f} MB per launch / {avg_ns / 1e3:.1f} us)")
            bad += 1
        # Round 6 on: the algorithmic bytes are priced for the form that ran (a pull form performs no residual RMW
        # per edge), so that they do not exceed what the counters saw: frac <= 1.1 x frac_traffic wherever a counter profile exists
        if int(rnd[1:]) >= 6 and roof.get("frac_traffic") is not None and roof["frac"] > 1.1 * roof["frac_traffic"]:
            print(f"FAIL {tag}: roofline.frac {roof['frac']:.3f} exceeds 1.1 x frac_traffic {roof['frac_traffic']:.3f}: the byte model claims more than the counters saw")
            bad += 1
        del per
    # Round 5 on: the artefacts say which build of the library produced them (profiles/rNN_manifest.json, tools/r05/manifest.py).
    # A kernel change without a re-capture must not go unnoticed: the tree's build id (tools/build_id.py) has to be the manifest's,
    # every bench line and counter summary has to carry it, and every stamped kind of artefact has to be listed.
    stamped = 0
    manifests = sorted(glob.glob(os.path.join(directory, "r*_manifest.json")))
    for m_path in manifests:
        rnd = os.path.basename(m_path).split("_")[0]
        man = json.load(open(m_path))
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import build_id as _bid
        real_profiles = os.path.abspath(directory) == os.path.join(_bid.ROOT, "profiles")
        # (the NEWEST round's manifest is the one the tree has to match; an older round's artefacts stay what they were captured on --
        # they are still held to their own manifest below)
        if real_profiles and m_path == manifests[-1] and man.get("build_id") != _bid.tree_build_id():
            print(f"FAIL {os.path.basename(m_path)}: captured on build {man.get('build_id')}, the tree is build {_bid.tree_build_id()} -- "
                  f"kernel / engine sources changed after the capture: re-capture (tools/{rnd}/capture.sh) or revert")
            bad += 1
        listed = set(man.get("files", []))
        for kind in ("bench", "kernel_stats", "batch_timeline", "pmc_fabric"):
            for f_path in sorted(glob.glob(os.path.join(directory, f"{rnd}_{kind}_*"))):
                f = os.path.basename(f_path)
                stamped += 1
                if f not in listed:
                    print(f"FAIL {f}: not listed in {os.path.basename(m_path)}")
                    bad += 1
                    continue
                got = None
                if kind == "bench":
                    got = json.loads(open(f_path).read().strip().splitlines()[-1]).get("build_id")
                elif kind == "pmc_fabric":
                    got = (json.load(open(f_path)).get("_stamp") or {}).get("build_id")
                if kind in ("bench", "pmc_fabric") and got != man.get("build_id"):
                    print(f"FAIL {f}: build id {got}, the manifest says {man.get('build_id')}")
                    bad += 1
    print(f"{pairs} timeline / stats pair(s), {lines} bench line(s) and {stamped} stamped artefact(s) checked, {bad} disagreement(s)")
    return 1 if bad else 0


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--cut":
        cut(sys.argv[2], sys.argv[3], sys.argv[4])
    else:
        sys.exit(check(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")))
