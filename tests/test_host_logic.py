"""Host logic of the engine kept in HIP-free headers. Vertex numbering (dynamicppr_amd/csrc/dppr_idspace.hpp: id maps, parked zone, composed
row moves of revived vertices, renumbering permutations) driven on the CPU by tests/native/idspace_test.cpp against
plain host arrays, built with the address and undefined-behaviour sanitizers. CPU only."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_idspace_driver(exe, extra):
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-Wall",
                           "-Werror", "-pthread"] + extra + ["-o", exe, os.path.join(ROOT, "tests", "native", "idspace_test.cpp")])
    return exe


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    return build_idspace_driver(str(tmp_path_factory.mktemp("native") / "idspace_test"), [])


@pytest.fixture(scope="module")
def threaded_driver(tmp_path_factory):
    # translate / renumber split their loops over host threads in pieces of >= 64 K ids; with a piece size of 3 the toy
    # arrays of the test take those paths too (a thread start per call: fewer operations)
    return build_idspace_driver(str(tmp_path_factory.mktemp("native") / "idspace_test_mt"), ["-DDPPR_PAR_MIN_PIECE=3"])


@pytest.mark.parametrize("seed,cap,ops", [(11, 40, 1500), (12, 7, 1000), (13, 500, 2000)])
def test_threaded_paths_of_translate_and_renumber(threaded_driver, seed, cap, ops):
    r = subprocess.run([threaded_driver, str(seed), str(cap), str(ops)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:]
    assert " 0 failures" in r.stdout


# (capacity, operations): tiny ranges keep the live and the parked zone touching nearly all the time
@pytest.mark.parametrize("seed,cap,ops", [(1, 40, 20000), (2, 40, 20000), (3, 7, 20000), (4, 3, 5000), (5, 500, 30000), (6, 97, 30000),
                                          (7, 2, 2000), (8, 1, 500)])
def test_random_sightings_revivals_flushes_and_renumberings(driver, seed, cap, ops):
    r = subprocess.run([driver, str(seed), str(cap), str(ops)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:]
    assert " 0 failures" in r.stdout


@pytest.fixture(scope="module")
def numbering_driver(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("native") / "numbering_test")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-Wall",
                           "-Werror", "-o", exe, os.path.join(ROOT, "tests", "native", "numbering_test.cpp")])
    return exe


# n: hash only (small windows), one counting pass on 16 / 21 key bits, with hot blocks of falling in-degree or two blocks
@pytest.mark.parametrize("seed,n,hot", [(1, 300000, 1), (2, 700000, 1), (3, 700000, 0), (4, 5000, 1), (5, 100000, 1), (6, 262145, 1),
                                        (7, 1, 1), (8, 65536, 1), (9, 524289, 0)])
def test_numbering_order_equals_its_plain_restatement(numbering_driver, seed, n, hot):
    r = subprocess.run([numbering_driver, str(seed), str(n), str(hot)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and ": 0 mismatches" in r.stdout, r.stdout[-2000:]


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_sweep_group_cuts(tmp_path, seed):
    """dppr_cut.hpp: cuts are well formed on random tile weights, a greedy group is closed by the tile that takes it to the
    target, and the min-max cut of a resident launch equals an exhaustive optimum on small inputs."""
    exe = str(tmp_path / "cut_test")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-Wall",
                           "-Werror", "-o", exe, os.path.join(ROOT, "tests", "native", "cut_test.cpp")])
    r = subprocess.run([exe, str(seed), "3000"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and " 0 failures" in r.stdout, r.stdout[-2000:]
