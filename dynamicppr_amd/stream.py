"""Host-side sliding-window stream (numpy mirror of the reference's SlidingGraphVec).

Python plumbing for bench.py and the tests; the product's compiled host side is
dynamicppr_amd/host/ (C++ ``SlidingGraphVec`` / ``EdgeBatch``). Only the parts the
GPU path needs are here: workload derivation from the CLI flags
(``SlidingGraphVec.h:46-66``), the batch record layout
(``SlidingGraphVec.h:219-275``) and the window edge list handed to the device
builder (``SerializeEdgeStream``, ``SlidingGraphVec.h:201-217``). The CPU
adjacency vectors of the reference are not needed by the device path.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

SLIDE_WINDOW_RATIO = 0  # Meta.h:19-24
SLIDE_BATCH_SIZE = 1


@dataclass
class Workload:
    window: int        # sliding_window_size (stream edges)
    per_batch: int     # gStreamUpdateCountPerBatch
    batch_count: int   # gStreamBatchCount
    total: int         # gStreamUpdateCountTotal


def workload_config(stream_len: int, window_ratio: float = 0.1, cfg_type: int = SLIDE_WINDOW_RATIO,
                    ratio: float = -1.0, batch_count: int = 0, per_batch: int = 0, total: int = 0) -> Workload:
    """``SlidingGraphVec::PrepareSlidingGraph`` arithmetic, with the same truncating conversions."""
    window = int(float(stream_len) * window_ratio)            # IndexType = size_t * double
    if cfg_type == SLIDE_WINDOW_RATIO:
        if ratio < 0.0 or batch_count == 0:
            raise ValueError("invalid arguments")             # ArgumentsChecker, Arguments.h:50-52
        per_batch = int(ratio * window)                        # size_t = double * int
        total = per_batch * batch_count
    elif cfg_type == SLIDE_BATCH_SIZE:
        if per_batch == 0 or total == 0:
            raise ValueError("invalid arguments")             # Arguments.h:53-55
        batch_count = (total + per_batch - 1) // per_batch
    else:
        raise ValueError("invalid arguments")
    total = min(total, stream_len - window)                   # :64-66
    return Workload(window, per_batch, batch_count, total)


class EdgeBatch:
    """SoA batch container (``EdgeBatch.h:6-30``)."""

    def __init__(self, size: int):
        self.size = size
        self.length = 0
        self.edge1 = np.zeros(size, dtype=np.int32)
        self.edge2 = np.zeros(size, dtype=np.int32)
        self.is_insert = np.zeros(size, dtype=np.uint8)


class SlidingStream:
    """Window position + batch construction over an in-memory edge stream."""

    def __init__(self, V: int, e1: np.ndarray, e2: np.ndarray, directed: int, wl: Workload):
        self.vertex_count = int(V)
        self.directed = int(directed)
        self.e1 = np.ascontiguousarray(e1, dtype=np.int32)
        self.e2 = np.ascontiguousarray(e2, dtype=np.int32)
        self.sliding_window_size = wl.window
        self.per_batch = wl.per_batch
        self.pos = wl.window                                   # stream edges consumed
        self.edge_count = wl.window if directed else 2 * wl.window
        self.edge_batch = EdgeBatch(4 * max(wl.per_batch, 1))
        self.new_stream = EdgeBatch(2 * max(wl.per_batch, 1))

    def serialize_edge_stream(self):
        """Current window in stream order, not mirrored (``SerializeEdgeStream``)."""
        lo = self.pos - self.sliding_window_size
        return self.e1[lo:self.pos], self.e2[lo:self.pos]

    def stream_updates(self, c: int | None = None) -> bool:
        """``StreamUpdates``: fill ``new_stream`` and ``edge_batch``. True when the stream is
        over -- the remaining partial batch is dropped, as the reference does."""
        c = self.per_batch if c is None else c
        if self.pos + c > len(self.e1):
            return True
        W = self.sliding_window_size
        ns, eb = self.new_stream, self.edge_batch
        ns.edge1[:c] = self.e1[self.pos:self.pos + c]
        ns.edge2[:c] = self.e2[self.pos:self.pos + c]
        ns.is_insert[:c] = 1
        ns.length = c
        lo = self.pos - W
        eb.edge1[:c] = self.e1[lo:lo + c]
        eb.edge2[:c] = self.e2[lo:lo + c]
        eb.is_insert[:c] = 0
        eb.edge1[c:2 * c] = self.e1[self.pos:self.pos + c]
        eb.edge2[c:2 * c] = self.e2[self.pos:self.pos + c]
        eb.is_insert[c:2 * c] = 1
        eb.length = 2 * c
        self.pos += c
        if not self.directed:
            n = eb.length
            eb.edge1[n:2 * n] = eb.edge2[:n]
            eb.edge2[n:2 * n] = eb.edge1[:n]
            eb.is_insert[n:2 * n] = eb.is_insert[:n]
            eb.length = 2 * n
        return False

    def batch_arrays(self):
        n = self.edge_batch.length
        return self.edge_batch.edge1[:n], self.edge_batch.edge2[:n], self.edge_batch.is_insert[:n]

    def new_arrays(self):
        n = self.new_stream.length
        return self.new_stream.edge1[:n], self.new_stream.edge2[:n]
