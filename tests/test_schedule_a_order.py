"""Schedule A (cpu/PPRCPUMTCilkRev.h at -t 1, the path north_star names) pinned by a HAND-DERIVED run.

The reference's golden vectors pin schedule A's arithmetic only through what it shares with the FIFO schedule; the one
property no reference-held vector covers is the ORDER of its frontier: cpu/PPRCPUMTCilkRev.h:208-257 flags, in frontier x
adjacency order, every in-neighbour whose residual crosses the threshold (prer not legal, curr legal), :263 packs them
(sequence::pack is a stable compaction at one worker), :267-279 subtracts each frontier vertex's own snapshot and appends
the ones that are still legal, in frontier order. A frontier vertex reads `residual[u]` when its turn comes (:210), i.e.
INCLUDING what earlier frontier vertices of the same iteration have already added -- the eager read the GPU kernels have too.

Graph (9 window edges, src -> dst, in stream order; a reverse push from u goes to u's in-neighbours in this order):
    1->0 2->0 3->0 | 3->1 2->1 | 4->2 0->2 | 5->3 1->3
    in[0] = [1,2,3]  in[1] = [3,2]  in[2] = [4,0]  in[3] = [5,1]  in[4] = in[5] = []
    out-degrees: deg = [1, 2, 2, 2, 1, 1];  source 0, ALPHA = 0.15, eps = 0.05

Worked by hand from the cited lines (a = 0.85):
  it 0  frontier [0]: ru = 1; adds a*1/3 = .283333 to 1, 2, 3 (each crosses)            -> next [1, 2, 3]
  it 1  frontier [1, 2, 3]:
        u=1: ru = .283333; +.0802778 to 3 and to 2 (both legal already: no flag)
        u=2: ru = .363611 (its own .283333 + what 1 just added);  +.154535 to 4 (crosses), +.154535 to 0 (crosses)
        u=3: ru = .363611; +.154535 to 5 (crosses), +.103023 to 1 (legal already)
        repair: r[1] = .386356 - .283333 = .103023 > eps -> stays; r[2] = r[3] = 0        -> next [4, 0, 5, 1]
  it 2  frontier [4, 0, 5, 1]:
        u=4: no in-edges.  u=0: ru = .154535; +.0437849 to 1 (legal already), to 2 and to 3 (.0437849 < eps: no flag)
        u=5: no in-edges.  u=1: ru = .146808 (.103023 + .0437849); +.0415956 to 3 (.0437849 -> .0853805: crosses), to 2 (same)
        repair: all four back to 0                                                        -> next [3, 2]   (adjacency order of in[1], NOT ascending)
  it 3  frontier [3, 2]: ru = .0853805 each; +.0362867 to 5, +.0241911 to 1, +.0362867 to 4 and to 0: none reaches eps -> done
  p = .15 * (what each vertex pushed):  p[0] = .15 (1 + .154535), p[1] = .15 (.283333 + .146808), p[2] = p[3] = .15 (.363611 + .0853805),
  p[4] = p[5] = .15 * .154535;   r = [.0362867, .0241911, 0, 0, .0362867, .0362867]

A synchronous (snapshot-first) schedule visits [0] [1,2,3] [4,0,5,1,2,3] ... on the same graph, so this also tells the two apart.
"""
import numpy as np

from oracle import oracle as orc

E1 = np.array([1, 2, 3, 3, 2, 4, 0, 5, 1, 0], dtype=np.int32)   # (the 10th edge lies beyond the window: the stream needs one batch's worth)
E2 = np.array([0, 0, 0, 1, 1, 2, 2, 3, 3, 1], dtype=np.int32)
EXPECTED_FRONTIERS = [[0], [1, 2, 3], [4, 0, 5, 1], [3, 2]]
A = 0.85
RU2 = A / 3 + A * (A / 3) / 3                  # .363611: vertex 2 (and 3) in iteration 1
RU1B = A * RU2 / 3 + A * (A * RU2 / 2) / 3     # .146808: vertex 1 in iteration 2
RU32 = A * (A * RU2 / 2) / 3 + A * RU1B / 3   # .0853805: vertices 3 and 2 in iteration 3 (.0437849 from 0, .0415956 from 1)
EXPECTED_P = 0.15 * np.array([1 + A * RU2 / 2, A / 3 + RU1B, RU2 + RU32, RU2 + RU32, A * RU2 / 2, A * RU2 / 2])
EXPECTED_R = np.array([A * RU32 / 2, A * RU32 / 3, 0.0, 0.0, A * RU32 / 2, A * RU32 / 2])


def test_schedule_a_frontier_order_is_the_hand_derived_one():
    g = orc.Graph(6, E1, E2, 1, 9, 1)
    s = orc.State(6, 0, 0.05)
    s.trace(True)
    s.cilk_execute(g)
    got = [[int(v) for v in f] for f in s.traced_frontiers()]
    assert got == EXPECTED_FRONTIERS            # ORDERED lists: crossing targets in frontier x adjacency order, then the repaired vertices
    assert np.max(np.abs(np.array(s.p) - EXPECTED_P)) < 1e-15 and np.max(np.abs(np.array(s.r) - EXPECTED_R)) < 1e-15
    assert abs(EXPECTED_P[0] - 0.17318021) < 1e-8 and abs(EXPECTED_R[1] - 0.02419112) < 1e-8   # (the decimals of the worked example)


def test_the_synchronous_schedule_is_a_different_one_on_this_graph():
    g = orc.Graph(6, E1, E2, 1, 9, 1)
    s = orc.State(6, 0, 0.05)
    s.trace(True)
    s.sync_execute(g)
    got = [sorted(int(v) for v in f) for f in s.traced_frontiers()]
    assert got[:3] == [[0], [1, 2, 3], [0, 1, 2, 3, 4, 5]] and len(got) > len(EXPECTED_FRONTIERS)
