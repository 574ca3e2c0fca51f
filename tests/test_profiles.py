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
    assert "pair(s) checked, 0 disagreement(s)" in r.stdout and not r.stdout.startswith("0 timeline")


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
