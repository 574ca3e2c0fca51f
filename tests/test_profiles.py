"""The committed profile artefacts agree with each other: every batch timeline under profiles/ was cut from the same profiler
run as the kernel-stats file next to it (tools/prof_timeline.sh, tools/check_profiles.py)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_batch_timelines_agree_with_their_kernel_stats():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_profiles.py"), os.path.join(ROOT, "profiles")],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    assert "checked, 0 disagreement(s)" in r.stdout and not r.stdout.startswith("0 timeline")


def test_the_checker_catches_artefacts_of_another_build(tmp_path):
    """Round 5: profiles are stamped with the build id of the library that produced them. A counter summary or a bench line from
    another build than the manifest's, and an artefact the manifest does not list, each fail the check."""
    import json
    man = {"build_id": "aaaaaaaaaaaaaaaa", "git_commit": "0000000", "files": ["r09_pmc_fabric_x.json", "r09_bench_x_1gpu.json"]}
    json.dump(man, open(tmp_path / "r09_manifest.json", "w"))
    json.dump({"_stamp": {"build_id": "aaaaaaaaaaaaaaaa"}, "k_gsweep": {}}, open(tmp_path / "r09_pmc_fabric_x.json", "w"))
    open(tmp_path / "r09_bench_x_1gpu.json", "w").write(json.dumps({"build_id": "aaaaaaaaaaaaaaaa", "roofline": {}}) + "\n")
    run = lambda: subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_profiles.py"), str(tmp_path)], stdout=subprocess.PIPE, text=True)
    r = run()
    assert r.returncode == 0 and "2 stamped artefact(s) checked, 0 disagreement(s)" in r.stdout, r.stdout
    json.dump({"_stamp": {"build_id": "bbbbbbbbbbbbbbbb"}, "k_gsweep": {}}, open(tmp_path / "r09_pmc_fabric_x.json", "w"))
    r = run()
    assert r.returncode == 1 and "r09_pmc_fabric_x.json: build id bbbbbbbbbbbbbbbb" in r.stdout, r.stdout
    json.dump({"_stamp": {"build_id": "aaaaaaaaaaaaaaaa"}, "k_gsweep": {}}, open(tmp_path / "r09_pmc_fabric_x.json", "w"))
    open(tmp_path / "r09_kernel_stats_y.csv", "w").write("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs\n")
    r = run()
    assert r.returncode == 1 and "r09_kernel_stats_y.csv: not listed" in r.stdout, r.stdout


def test_the_checker_catches_a_timeline_from_another_run(tmp_path):
    import json
    import shutil
    src = os.path.join(ROOT, "profiles")
    for f in ("r03_batch_timeline_livejournal_group10.json", "r03_kernel_stats_livejournal_group10.csv"):
        shutil.copy(os.path.join(src, f), tmp_path / f)
    tl = json.load(open(tmp_path / "r03_batch_timeline_livejournal_group10.json"))
    k = next(n for n in tl["kernels"] if n.startswith("k_gpush_tiny"))
    tl["kernels"][k]["avg_us"] = 900.0           # what round 2's committed timeline showed against a 264 us maximum in the stats
    json.dump(tl, open(tmp_path / "r03_batch_timeline_livejournal_group10.json", "w"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_profiles.py"), str(tmp_path)], stdout=subprocess.PIPE, text=True)
    assert r.returncode == 1 and "k_gpush_tiny" in r.stdout


def test_the_cut_finds_a_batch_in_each_of_its_forms(tmp_path):
    """tools/check_profiles.py --cut on a synthetic kernel trace: the slide's kernels (incl. the k_su_keys + radix sort of the
    slide-time grouping) are never taken for a batch; a batch starts at its IncrementalBatchUpdate kernel, at the counter-clearing
    k_su_keys in front of it, or -- with the update inside the resident launch -- at the k_su_keys in front of k_pull_resident."""
    import json
    slide = ["k_make_keys_seg", "rocprim::radix_sort_onesweep", "k_build_csr", "k_su_keys", "rocprim::radix_sort_block_sort"]
    forms = {"group": ["k_su_terms", "k_su_apply", "k_gsweep<2, 512, false>", "k_gsweep<2, 512, false>", "k_gpush_scan"],
             "fused": ["k_su_keys", "k_su_apply_fused", "k_pull_resident<1024>", "__amd_rocclr_copyBuffer"],
             "in-launch": ["k_su_keys", "k_pull_resident<1024>", "__amd_rocclr_copyBuffer"],
             "grouping-in-update": ["k_su_keys", "rocprim::radix_sort_block_sort", "k_su_apply_fused", "k_pull_iter<1024, false>"],
             # round 5 (grouping + CopyOutDegree inside the timed region by default): one ranking launch, or degrees + keys + device sort
             "ranked-in-update": ["k_su_group_rank", "k_su_terms", "k_su_apply", "k_gsweep<2, 512, false>", "k_gsweep<2, 512, false>"],
             "sorted-in-update": ["k_copy_out_degree", "k_su_keys", "rocprim::radix_sort_block_sort", "k_su_terms", "k_su_apply", "k_pull_iter<1024, false>"]}
    for form, batch in forms.items():
        d = tmp_path / form.replace(" ", "_")
        d.mkdir()
        t, rows = 1000, []
        for _ in range(4):
            for n in slide + batch:
                dur = 200000 if n.startswith(("k_gsweep", "k_pull")) else 3000
                rows.append((n, t, t + dur))
                t += dur + 500
        with open(d / "trace.csv", "w") as f:
            f.write('"Kernel_Name","Start_Timestamp","End_Timestamp"\n')
            for n, s, e in rows:
                f.write(f'"{n}",{s},{e}\n')
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_profiles.py"), "--cut", str(d / "trace.csv"), "unused", str(d)],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0, (form, r.stdout)
        tl = json.load(open(d / "timeline.json"))
        assert tl["batches_in_run"] == 3, (form, tl["batches_in_run"])       # four marks delimit three complete batches
        assert len(tl["iteration_kernel_durations_us"]) == sum(n.startswith(("k_gsweep", "k_pull")) for n in batch), form
        assert not any(n.startswith(("k_make", "k_build", "rocprim::radix_sort_onesweep")) for n in tl["kernels"]), (form, tl["kernels"])


def test_the_gap_report_reads_a_kernel_trace(tmp_path):
    """tools/r04/gaps.py on a synthetic trace: batches from one IncrementalBatchUpdate kernel to the next (the slide's kernels at
    a batch's tail are not part of it), idle time = span - busy, every gap above 5 us listed with the kernels around it."""
    rows, t = [], 1000
    for b in range(5):
        for n, dur, gap in [("k_su_terms", 30000, 0), ("k_su_apply", 80000, 500), ("k_gsweep<2, 10, 512, false, 0>", 150000, 500),
                            ("__amd_rocclr_copyBuffer", 4000, 500), ("k_gsweep<2, 10, 512, false, 1>", 160000, 22000),
                            ("k_make_keys_seg", 9000, 40000), ("k_build_csr", 500000, 500)]:
            t += gap
            rows.append((n, t, t + dur))
            t += dur
    with open(tmp_path / "trace.csv", "w") as f:
        f.write('"Kernel_Name","Start_Timestamp","End_Timestamp"\n')
        for n, a, b in rows:
            f.write(f'"{n}",{a},{b}\n')
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "r04", "gaps.py"), str(tmp_path / "trace.csv")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    assert "4 batches; median span 447.5 us, 5 dispatches" in r.stdout, r.stdout
    assert "busy 424.0 us, idle 23.5 us" in r.stdout and "gap   22.0 us  after __amd_rocclr_copyBuffer" in r.stdout and "gaps > 5 us: 22.0 us" in r.stdout, r.stdout

