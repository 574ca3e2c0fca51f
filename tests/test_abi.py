"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads,
and exports every symbol include/dppr.h declares. No compute call is made (no GPU here)."""
import ctypes
import os
import re

import pytest

from dynamicppr_amd import engine as eng

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "dppr.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dppr_[a-z_]+)\s*\(", text)))


def test_header_declares_expected_surface():
    syms = declared_symbols()
    for must in ("dppr_create", "dppr_load_window", "dppr_set_batch", "dppr_slide", "dppr_add_source",
                 "dppr_init_solve", "dppr_update", "dppr_incremental_batch_update", "dppr_execute_main_loop",
                 "dppr_read", "dppr_stats", "dppr_destroy", "dppr_strerror"):
        assert must in syms


def test_library_builds_and_exports_every_declared_symbol():
    path = eng.build()
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} declared in include/dppr.h but not exported"
    assert sorted(eng.EXPORTS) == declared_symbols()
    assert lib.dppr_abi_version() == 5   # 5: binned tables as runs + tiles (dppr_debug_bin_tables), map lock for reads beside a concurrent slide (round 6)
    # the library knows which sources it was built from, and says the same as the tree (profiles are stamped with it)
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import build_id
    lib.dppr_build_id.restype = ctypes.c_char_p
    assert lib.dppr_build_id().decode() == build_id.tree_build_id() == eng.build_id()


def test_strerror_and_argument_validation_without_gpu():
    L = eng.lib()
    assert L.dppr_strerror(0) == b"ok"
    assert b"no CPU fallback" in L.dppr_strerror(-4)
    h = ctypes.c_void_p()
    # invalid arguments are rejected before any device is touched
    assert L.dppr_create(ctypes.byref(h), 0, 0, 10, 1, 1, 1) == -1
    assert L.dppr_create(ctypes.byref(h), 0, 10, 10, 1, 1, 0) == -1


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="only meaningful without a GPU")
def test_fails_loudly_without_device():
    """The product path has no CPU fallback: creating an engine without a GPU raises."""
    with pytest.raises(eng.DpprError):
        eng.Engine(16, 4, 1, 1)


def test_product_never_imports_oracle():
    """oracle/ is test infrastructure: nothing under dynamicppr_amd/ may load or import it."""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "dynamicppr_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", ".cc")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                if "liboracle" in text or "dppr_oracle" in text or re.search(r"(from|import)\s+oracle", text):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad
