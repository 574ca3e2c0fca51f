#!/usr/bin/env python3
"""profiles/r06_manifest.json: which build of the library (tools/build_id.py) and which commit the round-6 bench lines, kernel-stats
files, batch timelines and counter summaries under profiles/ were captured on. tools/check_profiles.py fails when the tree's build
id differs from the manifest's, when a bench line or a counter summary carries another build id than the manifest, or when an
r06 artefact of those kinds is not listed.

    python tools/r06/manifest.py <git commit>       (after copying gpurun_out/r06cap/* into profiles/)
"""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import build_id  # noqa: E402

KINDS = ("bench", "kernel_stats", "batch_timeline", "pmc_fabric")


def stamped_files(directory):
    out = []
    for k in KINDS:
        out += sorted(os.path.basename(p) for p in glob.glob(os.path.join(directory, f"r06_{k}_*")))
    return out


def main():
    prof = os.path.join(ROOT, "profiles")
    files = stamped_files(prof)
    ids = set()
    for f in files:
        if f.startswith("r06_bench_"):
            ids.add(json.loads(open(os.path.join(prof, f)).read().strip().splitlines()[-1]).get("build_id"))
        if f.startswith("r06_pmc_fabric_"):
            ids.add((json.load(open(os.path.join(prof, f))).get("_stamp") or {}).get("build_id"))
    if len(ids) != 1:
        sys.exit(f"the artefacts carry {len(ids)} different build ids: {sorted(map(str, ids))}")
    json.dump({"build_id": ids.pop(), "git_commit": sys.argv[1] if len(sys.argv) > 1 else None, "tree_build_id_when_written": build_id.tree_build_id(),
               "files": files}, open(os.path.join(prof, "r06_manifest.json"), "w"), indent=1)
    print("wrote profiles/r06_manifest.json:", len(files), "files")


if __name__ == "__main__":
    main()
