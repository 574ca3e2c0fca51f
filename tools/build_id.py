#!/usr/bin/env python3
"""Identity of the library sources: first 16 hex digits of the SHA-256 over csrc/*.hpp (name order), csrc/dppr_engine.hip and
include/dppr.h -- what csrc/Makefile compiles into libdppr_hip.so (dppr_build_id). Profiles carry the id of the build that
produced them (tools/r05/*.sh), bench.py and tools/check_profiles.py compare.

    python tools/build_id.py            -> the id of the tree
"""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_files(root=ROOT):
    csrc = os.path.join(root, "dynamicppr_amd", "csrc")
    return sorted(glob.glob(os.path.join(csrc, "*.hpp"))) + [os.path.join(csrc, "dppr_engine.hip"), os.path.join(root, "include", "dppr.h")]


def tree_build_id(root=ROOT):
    h = hashlib.sha256()
    for f in source_files(root):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(tree_build_id())
