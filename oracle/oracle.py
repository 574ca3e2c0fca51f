"""ctypes binding of the CPU ORACLE (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg. Product code (dynamicppr_amd/) must never import
this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")


def build(force: bool = False) -> str:
    src = [os.path.join(_HERE, f) for f in ("dppr_oracle.c", "dppr_oracle.h")]
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in src)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "all"])
    return _LIB_PATH


class _Vec(C.Structure):
    _fields_ = [("d", C.POINTER(C.c_int)), ("head", C.c_int), ("n", C.c_int), ("cap", C.c_int)]


class _Graph(C.Structure):
    _fields_ = [
        ("V", C.c_int), ("directed", C.c_int), ("stream_len", C.c_int64),
        ("s1", C.POINTER(C.c_int)), ("s2", C.POINTER(C.c_int)),
        ("W", C.c_int), ("pos", C.c_int64), ("c", C.c_int), ("edge_count", C.c_int),
        ("deg", C.POINTER(C.c_int)), ("out", C.POINTER(_Vec)), ("in_", C.POINTER(_Vec)),
        ("b1", C.POINTER(C.c_int)), ("b2", C.POINTER(C.c_int)), ("bins", C.POINTER(C.c_uint8)),
        ("blen", C.c_int),
        ("n1", C.POINTER(C.c_int)), ("n2", C.POINTER(C.c_int)), ("nlen", C.c_int),
        ("out_change", C.POINTER(C.c_int)), ("in_change", C.POINTER(C.c_int)),
    ]


class _State(C.Structure):
    _fields_ = [
        ("V", C.c_int), ("source", C.c_int), ("eps", C.c_double),
        ("p", C.POINTER(C.c_double)), ("r", C.POINTER(C.c_double)), ("predeg", C.POINTER(C.c_int)),
        ("ft", C.POINTER(C.c_int)), ("ft2", C.POINTER(C.c_int)), ("ft_r", C.POINTER(C.c_double)),
        ("ft_count", C.c_int),
        ("edge_ind", C.POINTER(C.c_int)), ("edge_flag", C.POINTER(C.c_uint8)),
        ("vertex_offset", C.POINTER(C.c_int64)), ("edge_cap", C.c_int64),
        ("iteration_id", C.c_int),
        ("status", C.POINTER(C.c_int)), ("q", C.POINTER(C.c_int)),
        ("qhead", C.c_int64), ("qtail", C.c_int64), ("qcap", C.c_int64),
        ("stat_iters", C.c_int64), ("stat_F", C.c_int64), ("stat_E", C.c_int64), ("stat_N", C.c_int64),
        ("trace_on", C.c_int), ("trace_v", C.POINTER(C.c_int)),
        ("trace_len", C.c_int64), ("trace_cap", C.c_int64),
        ("trace_off", C.POINTER(C.c_int64)), ("trace_iters", C.c_int64), ("trace_off_cap", C.c_int64),
    ]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(build())
    GP, SP = C.POINTER(_Graph), C.POINTER(_State)
    ip, dp = C.POINTER(C.c_int), C.POINTER(C.c_double)
    L.orc_workload_config.argtypes = [C.c_int64, C.c_double, C.c_int, C.c_double, C.c_int64, C.c_int64,
                                      C.c_int64, ip, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                      C.POINTER(C.c_int64)]
    L.orc_workload_config.restype = None
    L.orc_graph_create.argtypes = [C.c_int, ip, ip, C.c_int64, C.c_int, C.c_int, C.c_int]
    L.orc_graph_create.restype = GP
    L.orc_graph_destroy.argtypes = [GP]
    L.orc_graph_stream_updates.argtypes = [GP]
    L.orc_graph_stream_updates.restype = C.c_int
    L.orc_graph_inc_construct.argtypes = [GP, C.c_int]
    L.orc_graph_scratch_construct.argtypes = [GP]
    L.orc_graph_flatten.argtypes = [GP, C.c_int, ip, ip]
    L.orc_state_create.argtypes = [C.c_int, C.c_int, C.c_double]
    L.orc_state_create.restype = SP
    L.orc_state_destroy.argtypes = [SP]
    L.orc_state_trace.argtypes = [SP, C.c_int]
    L.orc_state_reset_stats.argtypes = [SP]
    L.orc_is_legal_push.argtypes = [C.c_double, C.c_int, C.c_double]
    L.orc_is_legal_push.restype = C.c_int
    for name in ("orc_cilk_execute", "orc_cilk_inc_execute", "orc_fifo_execute", "orc_fifo_inc_execute",
                 "orc_sync_execute", "orc_sync_inc_execute", "orc_copy_revert_out_degree",
                 "orc_stream_update"):
        getattr(L, name).argtypes = [SP, GP]
        getattr(L, name).restype = None
    for name in ("orc_variant_execute", "orc_variant_inc_execute"):
        getattr(L, name).argtypes = [SP, GP, C.c_int]
        getattr(L, name).restype = None
    L.orc_cilk_init.argtypes = [SP]
    for name in ("orc_merged_main_loop", "orc_merged_inc_execute"):
        getattr(L, name).argtypes = [SP, GP, C.c_double]
        getattr(L, name).restype = None
    L.orc_cilk_inc_execute_mt.argtypes = [SP, GP, C.c_int]
    L.orc_cilk_inc_execute_mt.restype = None
    L.orc_cilk_main_loop_mt.argtypes = [SP, GP, C.c_int, C.c_int]
    L.orc_cilk_main_loop_mt.restype = None
    L.orc_max_threads.restype = C.c_int
    for name in ("orc_cilk_main_loop", "orc_sync_main_loop", "orc_dyn_push_init"):
        getattr(L, name).argtypes = [SP, GP, C.c_int]
        getattr(L, name).restype = None
    L.orc_inspect.argtypes = [SP, C.c_int, ip]
    L.orc_inspect.restype = C.c_int
    L.orc_pow_rev.argtypes = [GP, C.c_int, C.c_double, dp]
    L.orc_pow_rev.restype = C.c_int64
    L.orc_invariant_max_err.argtypes = [SP, GP]
    L.orc_invariant_max_err.restype = C.c_double
    L.orc_max_abs_residual.argtypes = [SP]
    L.orc_max_abs_residual.restype = C.c_double
    _lib = L
    return L


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def workload_config(stream_len, window_ratio=0.1, cfg_type=0, ratio=-1.0, batch_count=0, per_batch=0, total=0):
    """(W, per_batch, batch_count, total) as SlidingGraphVec.h:46-66 derives them."""
    W = C.c_int()
    pb, bc, tot = C.c_int64(), C.c_int64(), C.c_int64()
    lib().orc_workload_config(stream_len, window_ratio, cfg_type, ratio, batch_count, per_batch, total,
                              C.byref(W), C.byref(pb), C.byref(bc), C.byref(tot))
    return W.value, pb.value, bc.value, tot.value


class Graph:
    """Sliding-window host graph (restates SlidingGraphVec)."""

    def __init__(self, V, e1, e2, directed, W, c):
        e1 = np.ascontiguousarray(e1, dtype=np.int32)
        e2 = np.ascontiguousarray(e2, dtype=np.int32)
        self._g = lib().orc_graph_create(int(V), _ip(e1), _ip(e2), len(e1), int(directed), int(W), int(c))
        self.V, self.directed, self.W, self.c = int(V), int(directed), int(W), int(c)

    def __del__(self):
        if getattr(self, "_g", None):
            lib().orc_graph_destroy(self._g)
            self._g = None

    @property
    def edge_count(self):
        return self._g.contents.edge_count

    @property
    def pos(self):
        return self._g.contents.pos

    def stream_updates(self) -> bool:
        """True when the stream is over (the partial batch is dropped)."""
        return bool(lib().orc_graph_stream_updates(self._g))

    def inc_construct(self, mode=1):
        lib().orc_graph_inc_construct(self._g, mode)

    def scratch_construct(self):
        lib().orc_graph_scratch_construct(self._g)

    def batch(self):
        g = self._g.contents
        n = g.blen
        return (np.ctypeslib.as_array(g.b1, (n,)).copy(), np.ctypeslib.as_array(g.b2, (n,)).copy(),
                np.ctypeslib.as_array(g.bins, (n,)).copy())

    def new_stream(self):
        g = self._g.contents
        n = g.nlen
        return np.ctypeslib.as_array(g.n1, (n,)).copy(), np.ctypeslib.as_array(g.n2, (n,)).copy()

    def deg(self):
        return np.ctypeslib.as_array(self._g.contents.deg, (self.V,)).copy()

    def flatten(self, which):
        """CSR of the out (0) / in (1) adjacency in list order."""
        row = np.empty(self.V + 1, dtype=np.int32)
        col = np.empty(max(self.edge_count, 1), dtype=np.int32)
        lib().orc_graph_flatten(self._g, which, _ip(row), _ip(col))
        return row, col[:row[-1]]

    def window_edges(self):
        g = self._g.contents
        lo, hi = g.pos - g.W, g.pos
        s1 = np.ctypeslib.as_array(g.s1, (g.stream_len,))
        s2 = np.ctypeslib.as_array(g.s2, (g.stream_len,))
        return s1[lo:hi].copy(), s2[lo:hi].copy()


class State:
    """p/r state plus the three push schedules."""

    def __init__(self, V, source, eps):
        self._s = lib().orc_state_create(int(V), int(source), float(eps))
        self.V = int(V)

    def __del__(self):
        if getattr(self, "_s", None):
            lib().orc_state_destroy(self._s)
            self._s = None

    @property
    def p(self):
        return np.ctypeslib.as_array(self._s.contents.p, (self.V,))

    @property
    def r(self):
        return np.ctypeslib.as_array(self._s.contents.r, (self.V,))

    @property
    def predeg(self):
        return np.ctypeslib.as_array(self._s.contents.predeg, (self.V,))

    def frontier(self):
        s = self._s.contents
        return np.ctypeslib.as_array(s.ft, (max(s.ft_count, 1),))[:s.ft_count].copy()

    def stats(self):
        s = self._s.contents
        return dict(iters=s.stat_iters, F=s.stat_F, E=s.stat_E, N=s.stat_N)

    def reset_stats(self):
        lib().orc_state_reset_stats(self._s)

    def trace(self, on=True):
        lib().orc_state_trace(self._s, int(on))

    def traced_frontiers(self):
        s = self._s.contents
        n = s.trace_iters
        if n == 0:
            return []
        off = np.ctypeslib.as_array(s.trace_off, (n + 1,))
        v = np.ctypeslib.as_array(s.trace_v, (max(s.trace_len, 1),))
        return [v[off[i]:off[i + 1]].copy() for i in range(n)]

    # schedules
    def cilk_init(self): lib().orc_cilk_init(self._s)
    def cilk_execute(self, g): lib().orc_cilk_execute(self._s, g._g)
    def cilk_inc_execute(self, g): lib().orc_cilk_inc_execute(self._s, g._g)
    def cilk_inc_execute_mt(self, g, threads): lib().orc_cilk_inc_execute_mt(self._s, g._g, int(threads))
    def cilk_main_loop_mt(self, g, phase, threads): lib().orc_cilk_main_loop_mt(self._s, g._g, phase, int(threads))
    def fifo_execute(self, g): lib().orc_fifo_execute(self._s, g._g)
    def fifo_inc_execute(self, g): lib().orc_fifo_inc_execute(self._s, g._g)
    def sync_execute(self, g): lib().orc_sync_execute(self._s, g._g)
    def sync_inc_execute(self, g): lib().orc_sync_inc_execute(self._s, g._g)
    def merged_main_loop(self, g, eps): lib().orc_merged_main_loop(self._s, g._g, float(eps))
    def merged_inc_execute(self, g, eps): lib().orc_merged_inc_execute(self._s, g._g, float(eps))
    def sync_main_loop(self, g, phase): lib().orc_sync_main_loop(self._s, g._g, phase)
    def cilk_main_loop(self, g, phase): lib().orc_cilk_main_loop(self._s, g._g, phase)
    def variant_execute(self, g, variant): lib().orc_variant_execute(self._s, g._g, int(variant))
    def variant_inc_execute(self, g, variant): lib().orc_variant_inc_execute(self._s, g._g, int(variant))
    def copy_revert_out_degree(self, g): lib().orc_copy_revert_out_degree(self._s, g._g)
    def stream_update(self, g): lib().orc_stream_update(self._s, g._g)
    def dyn_push_init(self, g, phase): lib().orc_dyn_push_init(self._s, g._g, phase)

    def inspect(self, phase):
        out = np.empty(self.V, dtype=np.int32)
        n = lib().orc_inspect(self._s, phase, _ip(out))
        return out[:n].copy()

    def invariant_max_err(self, g):
        return lib().orc_invariant_max_err(self._s, g._g)

    def max_abs_residual(self):
        return lib().orc_max_abs_residual(self._s)


def pow_rev(g: Graph, source, alpha=0.15):
    out = np.empty(g.V, dtype=np.float64)
    iters = lib().orc_pow_rev(g._g, int(source), float(alpha), out.ctypes.data_as(C.POINTER(C.c_double)))
    return out, iters


def max_threads():
    return int(lib().orc_max_threads())


def is_legal_push(r, phase, eps):
    return bool(lib().orc_is_legal_push(float(r), int(phase), float(eps)))
