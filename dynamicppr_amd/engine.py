"""ctypes binding of libdppr_hip.so (the C ABI of include/dppr.h).

This is plumbing for tests and bench.py; the product's host side is the C++
program under dynamicppr_amd/host/ (``./pagerank``), which calls the same C ABI.
There is no CPU fallback: if the HIP library is missing or no device is present
every entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DPPR_LIB") or os.path.join(_HERE, "libdppr_hip.so")  # DPPR_LIB: diagnostic builds only

SCHEDULE_EAGER = 0
SCHEDULE_SYNC = 1


class DpprError(RuntimeError):
    pass


class Stats(C.Structure):
    _fields_ = [("iterations", C.c_int64), ("sum_F", C.c_int64), ("sum_E", C.c_int64), ("sum_N", C.c_int64),
                ("records", C.c_int64), ("inspected", C.c_int64), ("batches", C.c_int64),
                ("pull_iterations", C.c_int64),
                ("algorithmic_bytes", C.c_int64), ("gpu_ms", C.c_double), ("push_ms", C.c_double),
                ("push_launches", C.c_int64), ("persist_launches", C.c_int64), ("persist_aborts", C.c_int64),
                ("binned_sweeps", C.c_int64), ("sweep_F", C.c_int64), ("sweep_E", C.c_int64), ("sweep_ms", C.c_double),
                ("sweep_launches", C.c_int64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def build(force: bool = False) -> str:
    """Compile the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    csrc = os.path.join(_HERE, "csrc")
    srcs = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".hpp"))]
    srcs.append(os.path.join(os.path.dirname(_HERE), "include", "dppr.h"))
    stale = (not os.path.exists(LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", csrc, "-s", "all"])
    return LIB_PATH


_lib = None
EXPORTS = [
    "dppr_abi_version", "dppr_strerror", "dppr_last_error", "dppr_create", "dppr_destroy", "dppr_set_schedule", "dppr_set_profiling", "dppr_set_tuning", "dppr_set_persistent", "dppr_set_incremental_graph",
    "dppr_load_window", "dppr_set_batch", "dppr_slide", "dppr_add_source", "dppr_init_solve", "dppr_update",
    "dppr_incremental_batch_update", "dppr_execute_main_loop", "dppr_read", "dppr_write", "dppr_stats",
    "dppr_reset_stats", "dppr_inspect", "dppr_read_graph", "dppr_graph_edges", "dppr_read_out_graph", "dppr_trace_enable",
    "dppr_trace_get", "dppr_synchronize", "dppr_bench_atomics",
    "dppr_add_source_group", "dppr_group_init_solve", "dppr_group_update", "dppr_group_read", "dppr_group_stats",
    "dppr_group_reset_stats", "dppr_set_group_seeding", "dppr_seed_lists", "dppr_set_sweep_bitmap", "dppr_set_group_resident", "dppr_set_resident_slots", "dppr_set_resident_update",
    "dppr_set_renumbering", "dppr_id_space", "dppr_set_group_push", "dppr_set_binned_sweep", "dppr_device_count", "dppr_set_phase_merge", "dppr_init_solve_at", "dppr_group_init_solve_at", "dppr_set_variant", "dppr_set_batch_grouping",
    "dppr_time_batch_grouping", "dppr_debug_dump", "dppr_hint_next_batch",
    "dppr_bench_line_fills", "dppr_bench_stream_copy", "dppr_build_id", "dppr_heartbeat", "dppr_slide_concurrent", "dppr_renumbering_due", "dppr_debug_bin_tables",
]


def lib():
    """Load the HIP library; raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DpprError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(the HIP extension is mandatory, there is no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    vp, ip, dp, u8p = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_uint8)
    i64p, fp = C.POINTER(C.c_int64), C.POINTER(C.c_float)
    L.dppr_abi_version.restype = C.c_int
    L.dppr_strerror.argtypes = [C.c_int]
    L.dppr_strerror.restype = C.c_char_p
    L.dppr_last_error.argtypes = [vp]
    L.dppr_last_error.restype = C.c_char_p
    L.dppr_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int32, C.c_int32, C.c_int, C.c_int32, C.c_int32]
    L.dppr_destroy.argtypes = [vp]
    L.dppr_destroy.restype = None
    L.dppr_set_schedule.argtypes = [vp, C.c_int]
    L.dppr_set_profiling.argtypes = [vp, C.c_int]
    L.dppr_set_incremental_graph.argtypes = [vp, C.c_int]
    L.dppr_set_renumbering.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    L.dppr_set_group_push.argtypes = [vp, C.c_int, C.c_int, C.c_int64]
    L.dppr_id_space.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
    L.dppr_set_tuning.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    L.dppr_set_persistent.argtypes = [vp, C.c_int, C.c_int64]
    L.dppr_load_window.argtypes = [vp, ip, ip, C.c_int32]
    L.dppr_set_batch.argtypes = [vp, ip, ip, u8p, C.c_int32]
    L.dppr_slide.argtypes = [vp, ip, ip, C.c_int32, ip]
    L.dppr_add_source.argtypes = [vp, C.c_int32, ip]
    L.dppr_init_solve.argtypes = [vp, C.c_int32, C.c_double, fp]
    L.dppr_init_solve_at.argtypes = [vp, C.c_int32, C.c_int32, C.c_double, fp]
    L.dppr_group_init_solve_at.argtypes = [vp, C.c_int32, C.c_int32, C.c_double, fp]
    L.dppr_update.argtypes = [vp, C.c_int32, C.c_int32, C.c_double, fp]
    L.dppr_incremental_batch_update.argtypes = [vp, C.c_int32, C.c_int32]
    L.dppr_execute_main_loop.argtypes = [vp, C.c_int32, C.c_int32, C.c_int, C.c_double]
    L.dppr_read.argtypes = [vp, C.c_int32, dp, dp]
    L.dppr_write.argtypes = [vp, C.c_int32, dp, dp]
    L.dppr_stats.argtypes = [vp, C.c_int32, C.POINTER(Stats)]
    L.dppr_reset_stats.argtypes = [vp, C.c_int32]
    L.dppr_inspect.argtypes = [vp, C.c_int32, C.c_int, C.c_double, ip, ip]
    L.dppr_read_graph.argtypes = [vp, C.c_int32, ip, ip, ip]
    L.dppr_graph_edges.argtypes = [vp, C.c_int32, ip]
    L.dppr_read_out_graph.argtypes = [vp, C.c_int32, ip, ip]
    L.dppr_trace_enable.argtypes = [vp, C.c_int32, C.c_int]
    L.dppr_trace_get.argtypes = [vp, C.c_int32, i64p, i64p, i64p, ip]
    L.dppr_synchronize.argtypes = [vp]
    L.dppr_add_source_group.argtypes = [vp, ip, C.c_int32, ip]
    L.dppr_group_init_solve.argtypes = [vp, C.c_int32, C.c_double, fp]
    L.dppr_group_update.argtypes = [vp, C.c_int32, C.c_int32, C.c_double, fp]
    L.dppr_group_read.argtypes = [vp, C.c_int32, C.c_int32, dp, dp]
    L.dppr_group_stats.argtypes = [vp, C.c_int32, C.POINTER(Stats)]
    L.dppr_set_sweep_bitmap.argtypes = [vp, C.c_int]
    L.dppr_set_phase_merge.argtypes = [vp, C.c_int, C.c_int]
    L.dppr_set_variant.argtypes = [vp, C.c_int]
    L.dppr_set_batch_grouping.argtypes = [vp, C.c_int]
    L.dppr_set_binned_sweep.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int64]
    L.dppr_set_group_resident.argtypes = [vp, C.c_int]
    L.dppr_set_resident_slots.argtypes = [vp, C.c_int]
    L.dppr_set_resident_update.argtypes = [vp, C.c_int]
    L.dppr_seed_lists.argtypes = [vp, C.c_int32, C.c_int, ip, ip]
    L.dppr_group_reset_stats.argtypes = [vp, C.c_int32]
    L.dppr_set_group_seeding.argtypes = [vp, C.c_int]
    L.dppr_bench_atomics.argtypes = [C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_int, fp]
    L.dppr_time_batch_grouping.argtypes = [vp, C.c_int32, C.c_int32, fp]
    L.dppr_debug_dump.argtypes = [vp, C.c_char_p, C.c_int32]
    L.dppr_hint_next_batch.argtypes = [vp, ip, ip, C.c_int32, ip, ip, C.c_int32]
    L.dppr_bench_line_fills.argtypes = [C.c_int, C.c_int64, C.c_int64, C.c_int, fp]
    L.dppr_bench_stream_copy.argtypes = [C.c_int, C.c_int64, C.c_int, fp]
    L.dppr_build_id.restype = C.c_char_p
    L.dppr_heartbeat.argtypes = [vp]
    L.dppr_slide_concurrent.argtypes = [vp, ip, ip, C.c_int32, ip]
    L.dppr_renumbering_due.argtypes = [vp]
    u16p = C.POINTER(C.c_uint16)
    L.dppr_debug_bin_tables.argtypes = [vp, C.c_int32, ip, ip, ip, ip, ip, ip, ip, u16p, ip, ip, u16p, ip, i64p, i64p]
    L.dppr_heartbeat.restype = C.c_ulonglong
    for name in EXPORTS:
        if name not in ("dppr_strerror", "dppr_last_error", "dppr_destroy", "dppr_build_id", "dppr_heartbeat"):
            getattr(L, name).restype = C.c_int
    _lib = L
    return L


def _i32(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(C.POINTER(C.c_int32))


class Engine:
    """One device-resident window graph plus any number of source slots.

    Method names follow the reference's driver interface (gpu/PPRGPU.cuh:179-182,
    gpu/PPRRevPushGPU.cuh): ``GPUBuildSlidingGraph`` -> :meth:`slide`,
    ``IncrementalBatchUpdate`` -> :meth:`incremental_batch_update`,
    ``ExecuteMainLoop(phase)`` -> :meth:`execute_main_loop`; :meth:`update` is the
    whole timed region of ``SlidingWindowExecuteMainLoop``.
    """

    def __init__(self, V, W, directed, max_batch, n_epochs=1, device=0, schedule=SCHEDULE_EAGER,
                 hub_min_degree=None, big_row_edges=None, pull_min_frontier=None, chunk_iters=None, pull_block=None,
                 persistent=None, persist_timeout_us=None, sweep_bitmap=None, binned=None, merge_phases=None, variant=None, group_at_slide=None, resident_slots=None, resident_update=None):
        self._L = lib()
        self._h = C.c_void_p()
        self.V, self.W, self.directed, self.c = int(V), int(W), int(directed), int(max_batch)
        rc = self._L.dppr_create(C.byref(self._h), int(device), self.V, self.W, self.directed, self.c, int(n_epochs))
        if rc:
            self._h = C.c_void_p()
            raise DpprError(f"dppr_create: {self._L.dppr_strerror(rc).decode()}")
        self.set_schedule(schedule)
        if variant is not None:   # the reference's -o: sets the schedule too (1, 3: synchronous)
            self._ck(self._L.dppr_set_variant(self._h, int(variant)), "set_variant")
        if any(v is not None for v in (hub_min_degree, big_row_edges, pull_min_frontier, chunk_iters, pull_block)):
            self._ck(self._L.dppr_set_tuning(self._h, int(hub_min_degree or 256), int(big_row_edges or 512),
                                             int(pull_min_frontier or 0), int(chunk_iters or 0),
                                             int(pull_block or 0)), "set_tuning")
        if sweep_bitmap is not None:
            self._ck(self._L.dppr_set_sweep_bitmap(self._h, int(sweep_bitmap)), "set_sweep_bitmap")
        if group_at_slide is not None:
            self._ck(self._L.dppr_set_batch_grouping(self._h, int(group_at_slide)), "set_batch_grouping")
        if resident_slots is not None:
            self.set_resident_slots(resident_slots)
        if resident_update is not None:
            self._ck(self._L.dppr_set_resident_update(self._h, int(resident_update)), "set_resident_update")
        if merge_phases is not None:   # True / divisor
            self.set_phase_merge(bool(merge_phases), 0 if merge_phases is True or not merge_phases else int(merge_phases))
        if binned is not None:   # int mode, or (mode, ha_tiles, hb_tiles, target_edges, min_ids, chunk_edges, target_a_edges)
            args = (binned,) if isinstance(binned, int) else tuple(binned)
            args = tuple(int(a) for a in args) + (0,) * (7 - len(args))
            self._ck(self._L.dppr_set_binned_sweep(self._h, *args), "set_binned_sweep")
        if persistent is not None or persist_timeout_us is not None:
            self._ck(self._L.dppr_set_persistent(self._h, 1 if persistent is None else int(persistent),
                                                 int(persist_timeout_us or 0)), "set_persistent")

    def _ck(self, rc, what):
        if rc:
            raise DpprError(f"{what}: {self._L.dppr_strerror(rc).decode()} ({self._L.dppr_last_error(self._h).decode()})")

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.dppr_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_schedule(self, schedule):
        self._ck(self._L.dppr_set_schedule(self._h, int(schedule)), "set_schedule")

    def set_phase_merge(self, on, eps_divisor=0):
        """One loop for residuals of both signs, run to eps / eps_divisor (include/dppr.h); eager schedule only."""
        self._ck(self._L.dppr_set_phase_merge(self._h, int(on), int(eps_divisor)), "set_phase_merge")

    def bin_tables(self, epoch=-1, arrays=True):
        """Test hook: the binned-sweep tables of an epoch (dppr_debug_bin_tables) as a dict; None when the epoch has none."""
        na, nb, ne, nr, nt, pa, rb = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32(), C.c_int64(), C.c_int64()
        rc = self._L.dppr_debug_bin_tables(self._h, int(epoch), C.byref(na), C.byref(nb), C.byref(ne), C.byref(nr), C.byref(nt),
                                           None, None, None, None, None, None, None, C.byref(pa), C.byref(rb))
        out = {"patched": pa.value, "rebuilt": rb.value}
        if rc:
            return None if not arrays else dict(out, valid=False)
        out.update(valid=True, n_a=na.value, n_b=nb.value, n_edges=ne.value, n_runs=nr.value, n_tiles=nt.value)
        if arrays:
            n_blk, n_rb = (ne.value + 63) // 64, (nr.value + 63) // 64
            acut, bcut = np.empty(na.value + 1, np.int32), np.empty(nb.value + 1, np.int32)
            hl, dl = np.empty(max(nr.value, 1), np.uint16), np.empty(max(ne.value, 1), np.uint16)
            tdelta, tb, vb = np.empty(max(nt.value, 1), np.int32), np.empty(n_rb + 1, np.int32), np.empty(n_blk + 1, np.int32)
            i32, u16 = C.POINTER(C.c_int32), C.POINTER(C.c_uint16)
            self._ck(self._L.dppr_debug_bin_tables(self._h, int(epoch), None, None, None, None, None, acut.ctypes.data_as(i32), bcut.ctypes.data_as(i32),
                                                   hl.ctypes.data_as(u16), tdelta.ctypes.data_as(i32), tb.ctypes.data_as(i32), dl.ctypes.data_as(u16),
                                                   vb.ctypes.data_as(i32), None, None), "debug_bin_tables")
            out.update(acut=acut, bcut=bcut, hl=hl[:nr.value], tdelta=tdelta[:nt.value], tb=tb, dl=dl[:ne.value], vb=vb)
        return out

    def renumbering_due(self):
        return bool(self._L.dppr_renumbering_due(self._h))

    def set_batch_grouping(self, at_slide):
        """0 (default): CopyOutDegree + the grouping of a batch's records run inside the timed region, as the reference times them;
        1: at slide time (for epochs already built: on entry to the next update, before its event bracket opens)."""
        self._ck(self._L.dppr_set_batch_grouping(self._h, int(at_slide)), "set_batch_grouping")

    def set_incremental_graph(self, on):
        self._ck(self._L.dppr_set_incremental_graph(self._h, int(on)), "set_incremental_graph")

    def set_group_push(self, enter_pairs=-1, list_cap=0, max_edges=0):
        """Tail of a source group's loop as pushes: -1 automatic threshold, 0 never, N below N frontier pairs."""
        self._ck(self._L.dppr_set_group_push(self._h, int(enter_pairs), int(list_cap), int(max_edges)), "set_group_push")

    def set_renumbering(self, on, growth_pct=0, min_parked=0):
        """Renumbering of the internal ids at slide time (include/dppr.h); 0 keeps a threshold as it is."""
        self._ck(self._L.dppr_set_renumbering(self._h, int(on), int(growth_pct), int(min_parked)), "set_renumbering")

    def id_space(self):
        """dict(ids=swept ids, parked=ids parked with their state, renumberings=, revivals=)"""
        a, b, c, d = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int64()
        self._ck(self._L.dppr_id_space(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)), "id_space")
        return {"ids": a.value, "parked": b.value, "renumberings": c.value, "revivals": d.value}

    def set_profiling(self, on):
        self._ck(self._L.dppr_set_profiling(self._h, int(on)), "set_profiling")

    def load_window(self, e1, e2):
        a, pa = _i32(e1)
        b, pb = _i32(e2)
        self._ck(self._L.dppr_load_window(self._h, pa, pb, len(a)), "load_window")

    def set_batch(self, b1, b2, ins):
        a, pa = _i32(b1)
        b, pb = _i32(b2)
        i = np.ascontiguousarray(ins, dtype=np.uint8)
        self._ck(self._L.dppr_set_batch(self._h, pa, pb, i.ctypes.data_as(C.POINTER(C.c_uint8)), len(a)), "set_batch")

    def hint_next_batch(self, b1, b2, n1, n2):
        """Lookahead (dppr_hint_next_batch): the id lookups of the next set_batch(b1, b2, ..) / slide(n1, n2) run on helper
        threads from now on. Returns the four arrays as contiguous int32 -- pass THESE objects to set_batch / slide (the
        engine matches the hint by pointer and length) and leave them untouched until then."""
        arrs = [_i32(x) for x in (b1, b2, n1, n2)]
        if len(arrs[0][0]) != len(arrs[1][0]) or len(arrs[2][0]) != len(arrs[3][0]):
            raise DpprError("hint_next_batch: b1 / b2 and n1 / n2 must have equal lengths")
        self._hint_keep = [a for a, _ in arrs]   # (keeps the memory alive until the next hint)
        self._ck(self._L.dppr_hint_next_batch(self._h, arrs[0][1], arrs[1][1], len(arrs[0][0]), arrs[2][1], arrs[3][1], len(arrs[2][0])),
                 "hint_next_batch")
        return tuple(self._hint_keep)

    def slide(self, n1, n2, concurrent=False):
        """GPUBuildSlidingGraph. concurrent=True: dppr_slide_concurrent -- may run (from another thread) beside an update on an older,
        explicitly named epoch; needs n_epochs >= 2."""
        a, pa = _i32(n1)
        b, pb = _i32(n2)
        ep = C.c_int32(-1)
        fn = self._L.dppr_slide_concurrent if concurrent else self._L.dppr_slide
        self._ck(fn(self._h, pa, pb, len(a), C.byref(ep)), "slide")
        return ep.value

    def add_source(self, s):
        slot = C.c_int32(-1)
        self._ck(self._L.dppr_add_source(self._h, int(s), C.byref(slot)), "add_source")
        return slot.value

    def init_solve(self, slot, eps, epoch=-1):
        ms = C.c_float(0)
        self._ck(self._L.dppr_init_solve_at(self._h, slot, int(epoch), float(eps), C.byref(ms)), "init_solve")
        return ms.value

    def update(self, slot, eps, epoch=-1):
        ms = C.c_float(0)
        self._ck(self._L.dppr_update(self._h, slot, int(epoch), float(eps), C.byref(ms)), "update")
        return ms.value

    def incremental_batch_update(self, slot, epoch=-1):
        self._ck(self._L.dppr_incremental_batch_update(self._h, slot, int(epoch)), "incremental_batch_update")

    def execute_main_loop(self, slot, phase, eps, epoch=-1):
        self._ck(self._L.dppr_execute_main_loop(self._h, slot, int(epoch), int(phase), float(eps)), "execute_main_loop")

    def read(self, slot):
        p = np.empty(self.V, dtype=np.float64)
        r = np.empty(self.V, dtype=np.float64)
        dp = C.POINTER(C.c_double)
        self._ck(self._L.dppr_read(self._h, slot, p.ctypes.data_as(dp), r.ctypes.data_as(dp)), "read")
        return p, r

    def write(self, slot, p, r):
        p = np.ascontiguousarray(p, dtype=np.float64)
        r = np.ascontiguousarray(r, dtype=np.float64)
        dp = C.POINTER(C.c_double)
        self._ck(self._L.dppr_write(self._h, slot, p.ctypes.data_as(dp), r.ctypes.data_as(dp)), "write")

    def stats(self, slot):
        st = Stats()
        self._ck(self._L.dppr_stats(self._h, slot, C.byref(st)), "stats")
        return st.as_dict()

    def reset_stats(self, slot):
        self._ck(self._L.dppr_reset_stats(self._h, slot), "reset_stats")

    def inspect(self, slot, phase, eps):
        out = np.empty(self.V, dtype=np.int32)
        n = C.c_int32(0)
        self._ck(self._L.dppr_inspect(self._h, slot, int(phase), float(eps),
                                      out.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(n)), "inspect")
        return out[:n.value].copy()

    def seed_lists(self, slot, phase):
        out = np.empty(max(4 * self.c, 1), dtype=np.int32)
        n = C.c_int32(0)
        self._ck(self._L.dppr_seed_lists(self._h, slot, int(phase), out.ctypes.data_as(C.POINTER(C.c_int32)),
                                         C.byref(n)), "seed_lists")
        return out[:n.value].copy()

    def read_graph(self, epoch=-1):
        ne = C.c_int32(0)
        self._ck(self._L.dppr_graph_edges(self._h, int(epoch), C.byref(ne)), "graph_edges")
        row = np.empty(self.V + 1, dtype=np.int32)
        col = np.empty(max(ne.value, 1), dtype=np.int32)
        deg = np.empty(self.V, dtype=np.int32)
        ip = C.POINTER(C.c_int32)
        self._ck(self._L.dppr_read_graph(self._h, int(epoch), row.ctypes.data_as(ip), col.ctypes.data_as(ip),
                                         deg.ctypes.data_as(ip)), "read_graph")
        return row, col[:ne.value], deg

    def read_out_graph(self, epoch=-1):
        ne = C.c_int32(0)
        self._ck(self._L.dppr_graph_edges(self._h, int(epoch), C.byref(ne)), "graph_edges")
        row = np.empty(self.V + 1, dtype=np.int32)
        col = np.empty(max(ne.value, 1), dtype=np.int32)
        ip = C.POINTER(C.c_int32)
        self._ck(self._L.dppr_read_out_graph(self._h, int(epoch), row.ctypes.data_as(ip), col.ctypes.data_as(ip)),
                 "read_out_graph")
        return row, col[:ne.value]

    def trace_enable(self, slot, on=True):
        self._ck(self._L.dppr_trace_enable(self._h, slot, int(on)), "trace_enable")

    def trace_get(self, slot):
        ni, nd = C.c_int64(0), C.c_int64(0)
        self._ck(self._L.dppr_trace_get(self._h, slot, C.byref(ni), C.byref(nd), None, None), "trace_get")
        off = np.zeros(ni.value + 1, dtype=np.int64)
        ids = np.zeros(max(nd.value, 1), dtype=np.int32)
        self._ck(self._L.dppr_trace_get(self._h, slot, None, None, off.ctypes.data_as(C.POINTER(C.c_int64)),
                                        ids.ctypes.data_as(C.POINTER(C.c_int32))), "trace_get")
        return [ids[off[i]:off[i + 1]].copy() for i in range(ni.value)]

    # ---- source groups (multi-source batched sweeps) ----
    def add_source_group(self, sources):
        a, pa = _i32(sources)
        gid = C.c_int32(-1)
        self._ck(self._L.dppr_add_source_group(self._h, pa, len(a), C.byref(gid)), "add_source_group")
        return gid.value

    def group_init_solve(self, group, eps, epoch=-1):
        ms = C.c_float(0)
        self._ck(self._L.dppr_group_init_solve_at(self._h, group, int(epoch), float(eps), C.byref(ms)), "group_init_solve")
        return ms.value

    def group_update(self, group, eps, epoch=-1):
        ms = C.c_float(0)
        self._ck(self._L.dppr_group_update(self._h, group, int(epoch), float(eps), C.byref(ms)), "group_update")
        return ms.value

    def group_read(self, group, index):
        p = np.empty(self.V, dtype=np.float64)
        r = np.empty(self.V, dtype=np.float64)
        dp = C.POINTER(C.c_double)
        self._ck(self._L.dppr_group_read(self._h, group, int(index), p.ctypes.data_as(dp), r.ctypes.data_as(dp)),
                 "group_read")
        return p, r

    def group_stats(self, group):
        st = Stats()
        self._ck(self._L.dppr_group_stats(self._h, group, C.byref(st)), "group_stats")
        return st.as_dict()

    def group_reset_stats(self, group):
        self._ck(self._L.dppr_group_reset_stats(self._h, group), "group_reset_stats")

    def set_resident_slots(self, sorted_slots):
        """Edge slots of the single-source resident sweep: sorted by gather position (default), or CSR order."""
        self._ck(self._L.dppr_set_resident_slots(self._h, int(sorted_slots)), "set_resident_slots")

    def set_group_resident(self, on):
        self._ck(self._L.dppr_set_group_resident(self._h, int(on)), "set_group_resident")

    def set_group_seeding(self, from_tails):
        self._ck(self._L.dppr_set_group_seeding(self._h, int(from_tails)), "set_group_seeding")

    def synchronize(self):
        self._ck(self._L.dppr_synchronize(self._h), "synchronize")

    def time_batch_grouping(self, epoch=-1, reps=5):
        """ms per batch of CopyOutDegree + the grouping of the records by tail, run on their own as the timed region runs them."""
        ms = C.c_float(0)
        self._ck(self._L.dppr_time_batch_grouping(self._h, int(epoch), int(reps), C.byref(ms)), "time_batch_grouping")
        return ms.value

    def debug_dump(self):
        """Post-mortem text of the engine (dppr_debug_dump): callable from another thread than the one stuck in a call."""
        buf = C.create_string_buffer(1 << 16)
        n = self._L.dppr_debug_dump(self._h, buf, len(buf))
        return buf.raw[:max(n, 0)].decode(errors="replace")


def bench_atomics(table_elems, n, scope=0, reps=5, device=0):
    ms = C.c_float(0)
    rc = lib().dppr_bench_atomics(int(device), int(table_elems), int(n), int(scope), int(reps), C.byref(ms))
    if rc:
        raise DpprError(f"bench_atomics: {lib().dppr_strerror(rc).decode()}")
    return ms.value


def bench_line_fills(table_bytes=1 << 30, lines=1 << 26, reps=3, device=0):
    """ms for `lines` random 128-byte line fetches out of a table of table_bytes (dppr_bench_line_fills)."""
    ms = C.c_float(0)
    rc = lib().dppr_bench_line_fills(int(device), int(table_bytes), int(lines), int(reps), C.byref(ms))
    if rc:
        raise DpprError(f"bench_line_fills: {lib().dppr_strerror(rc).decode()}")
    return ms.value


def bench_stream_copy(nbytes=1 << 30, reps=5, device=0):
    """ms per streaming copy of nbytes (read + write)."""
    ms = C.c_float(0)
    rc = lib().dppr_bench_stream_copy(int(device), int(nbytes), int(reps), C.byref(ms))
    if rc:
        raise DpprError(f"bench_stream_copy: {lib().dppr_strerror(rc).decode()}")
    return ms.value


def build_id():
    """Identity of the loaded library's sources (dppr_build_id; tools/build_id.py computes the tree's)."""
    return lib().dppr_build_id().decode()
