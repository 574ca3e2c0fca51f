/*
 * dppr_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the reference's dynamic reverse-push PPR path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link or call this. The product (dynamicppr_amd/) never does.
 *
 * Every function cites the reference file:line it follows (paths relative
 * to /root/reference).
 *
 * Parity pin (see DESIGN.md "Oracle"):
 *   - window/batch logic, the FIFO push schedule (cpu/PPRCPURev.h) and the
 *     power-iteration ground truth (cpu/PPRCPUPowVec.h) are checked
 *     BIT-FOR-BIT against the real reference headers compiled by
 *     oracle/Makefile into oracle/_ref/ref_driver (golden fixtures in
 *     tests/golden/).
 *   - the Cilk schedule (cpu/PPRCPUMTCilkRev.h at -t 1) cannot be compiled
 *     here (Cilk Plus is absent and no stand-in headers are written); it
 *     shares the pinned primitives and is checked with the reference's own
 *     Validate() criteria against the pinned ground truth.
 */
#ifndef DPPR_ORACLE_H
#define DPPR_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_ALPHA 0.15 /* Meta.h:31 */

/* ---- growable FIFO adjacency list (std::vector with erase-from-front) ---- */
typedef struct orc_vec {
    int *d;
    int head, n, cap; /* live range d[head .. head+n) */
} orc_vec;

/* ---- sliding window host graph: restates SlidingGraphVec.h:9-289 ---- */
typedef struct orc_graph {
    int V;
    int directed;
    int64_t stream_len; /* total_edge_stream_length, SlidingGraphVec.h:46 */
    int *s1, *s2;       /* owned copy of the stream (the .bin payload) */
    int W;              /* sliding_window_size */
    int64_t pos;        /* stream edges consumed so far == (file_pos-4)/8 */
    int c;              /* gStreamUpdateCountPerBatch */
    int edge_count;     /* W or 2W, SlidingGraphVec.h:68-69 */
    int *deg;           /* out-degree, GraphVec.h deg */
    orc_vec *out;       /* col_ind */
    orc_vec *in;        /* in_col_ind */
    /* edge_batch (capacity 4c) and new_stream (capacity 2c), SlidingGraphVec.h:17-18 */
    int *b1, *b2;
    uint8_t *bins;
    int blen;
    int *n1, *n2;
    int nlen;
    int *out_change, *in_change;
} orc_graph;

/* SlidingGraphVec.h:46-66 -- derive W, c, batch count, total from the flags.
 * cfg_type 0 = SLIDE_WINDOW_RATIO (-r,-b), 1 = SLIDE_BATCH_SIZE (-c,-l). */
void orc_workload_config(int64_t stream_len, double window_ratio, int cfg_type,
                         double ratio, int64_t batch_count_in, int64_t per_batch_in, int64_t total_in,
                         int *W, int64_t *per_batch, int64_t *batch_count, int64_t *total);

/* SlidingGraphVec.h:28-97 PrepareSlidingGraph on an in-memory stream. */
orc_graph *orc_graph_create(int V, const int *e1, const int *e2, int64_t stream_len,
                            int directed, int W, int c);
void orc_graph_destroy(orc_graph *g);
/* SlidingGraphVec.h:219-275. returns 1 when the stream is over (batch dropped). */
int orc_graph_stream_updates(orc_graph *g);
/* SlidingGraphVec.h:139-195. mode 0: reference-faithful (records applied in
 * batch order: all direct records, then all mirrored ones); mode 1:
 * stream-order-correct (a record and its mirror applied together), which equals
 * ScratchConstructWindowGraph order in every case. See DESIGN.md "quirk Q1". */
void orc_graph_inc_construct(orc_graph *g, int mode);
/* SlidingGraphVec.h:99-136 ScratchConstructWindowGraph. */
void orc_graph_scratch_construct(orc_graph *g);
/* flatten adjacency: row_ptr[V+1], col[edge_count]; which: 0 = out, 1 = in */
void orc_graph_flatten(const orc_graph *g, int which, int *row_ptr, int *col);

/* ---- PPR state ---- */
typedef struct orc_state {
    int V, source;
    double eps;
    double *p, *r;
    int *predeg;
    /* cilk/sync schedule frontier storage */
    int *ft, *ft2;
    double *ft_r;
    int ft_count;
    /* -t > 1 schedule scratch (edge_ind / edge_flag / vertex_offset of cpu/PPRCPUMTCilkRev.h:8-18) */
    int *edge_ind;
    unsigned char *edge_flag;
    int64_t *vertex_offset;
    int64_t edge_cap;
    int iteration_id;
    /* fifo schedule */
    int *status;
    int *q;
    int64_t qhead, qtail, qcap; /* ring queue */
    /* statistics (not in the reference; used for algorithmic-bytes accounting) */
    int64_t stat_iters, stat_F, stat_E, stat_N;
    /* optional frontier trace: concatenated per-iteration frontiers */
    int trace_on;
    int *trace_v;
    int64_t trace_len, trace_cap;
    int64_t *trace_off; /* offsets, trace_iters+1 entries */
    int64_t trace_iters, trace_off_cap;
} orc_state;

orc_state *orc_state_create(int V, int source, double eps);
void orc_state_destroy(orc_state *s);
void orc_state_trace(orc_state *s, int on); /* enable+reset / disable frontier trace */
void orc_state_reset_stats(orc_state *s);

/* gpu/PPRCommon.cuh:6-11 == cpu/PPRCPUMTCilkRev.h:75-80 */
int orc_is_legal_push(double r, int phase, double eps);

/* ---- schedule A: cpu/PPRCPUMTCilkRev.h at -t 1 (cilk_for == for) ---- */
void orc_cilk_init(orc_state *s);                                   /* :175-182 */
void orc_cilk_main_loop(orc_state *s, const orc_graph *g, int phase); /* :184-289 */
void orc_cilk_execute(orc_state *s, const orc_graph *g);            /* :38-41 ExecuteImpl */
void orc_cilk_inc_execute(orc_state *s, const orc_graph *g);        /* :43-73 IncExecuteImpl */

/* The same schedule with T > 1 workers (OpenMP in place of Cilk Plus): parallel_for over the
 * frontier, CAS-loop atomic adds (cpu/PPRCPUMTCilkRev.h:82-96), per-edge flag array + pack
 * (cpu/CilkUtil.h:246-262). Like the reference at -t > 1 the result depends on thread timing
 * (racy `ru = residual[u]`); it is held to Validate() only. This is the timed CPU baseline. */
void orc_cilk_main_loop_mt(orc_state *s, const orc_graph *g, int phase, int threads);
void orc_cilk_inc_execute_mt(orc_state *s, const orc_graph *g, int threads);
int orc_max_threads(void);

/* pieces, exposed for kernel-level parity tests */
void orc_copy_revert_out_degree(orc_state *s, const orc_graph *g);  /* cpu/PPRCPUMTCilk.h:157-174 */
void orc_stream_update(orc_state *s, const orc_graph *g);           /* cpu/PPRCPUMTCilkRev.h:108-124 */
void orc_dyn_push_init(orc_state *s, const orc_graph *g, int phase); /* :126-156 */
/* gpu/Inspect.cuh:8-48: all legal vertices, ascending id. returns count */
int orc_inspect(const orc_state *s, int phase, int *out);

/* ---- schedule B: cpu/PPRCPURev.h (deprecated single-thread FIFO) ---- */
void orc_fifo_execute(orc_state *s, const orc_graph *g);     /* :31-34 */
void orc_fifo_inc_execute(orc_state *s, const orc_graph *g); /* :36-62 */

/* ---- schedule C: synchronous (snapshot-all-then-push), frontier seeded by a
 * full Inspect as gpu/PPRRevPushGPU.cuh:97-131 does. One legal interleaving of
 * the reference GPU kernels; used to check the HIP engine's deterministic mode
 * iteration by iteration (frontier sets bit-exact). */
void orc_sync_main_loop(orc_state *s, const orc_graph *g, int phase);
void orc_sync_execute(orc_state *s, const orc_graph *g);
/* checker of the engine's merged loop (dppr_set_phase_merge): NOT a schedule of the reference, see dppr_oracle.c */
void orc_merged_main_loop(orc_state *s, const orc_graph *g, double eps);
void orc_merged_inc_execute(orc_state *s, const orc_graph *g, double eps);
void orc_sync_inc_execute(orc_state *s, const orc_graph *g);

/* ---- variants 1-3 of the CPU path at -t 1: cpu/PPRCPUMTCilkRevVariants.h (FF :224-326, Eager
 * :112-222, Vanilla :6-110), selected by -o (cpu/PPRCPUMTMain.cpp:26-32); variant 0 = schedule A ---- */
void orc_variant_main_loop(orc_state *s, const orc_graph *g, int phase, int variant);
void orc_variant_execute(orc_state *s, const orc_graph *g, int variant);
void orc_variant_inc_execute(orc_state *s, const orc_graph *g, int variant);

/* ---- ground truth: cpu/PPRCPUPowVec.h:55-83 CalPPRRev ---- */
int64_t orc_pow_rev(const orc_graph *g, int source, double alpha, double *out_p);

/* invariant of SURVEY.md section 0: max_u |p[u]+a*r[u]-a*[u==s]-(1-a)/(outdeg+1)*sum p[out]| */
double orc_invariant_max_err(const orc_state *s, const orc_graph *g);
double orc_max_abs_residual(const orc_state *s);

#ifdef __cplusplus
}
#endif
#endif
