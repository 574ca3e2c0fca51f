/*
 * dppr_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See dppr_oracle.h for the contract and the parity pin.
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off; no FMA contraction so the
 * double arithmetic is the same sequence of IEEE operations the reference's
 * g++ build performs on x86-64).
 */
#include "dppr_oracle.h"

#include <assert.h>
#include <math.h>
#include <omp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ vec */
static void vec_push(orc_vec *v, int x) {
    if (v->head + v->n == v->cap) {
        if (v->head > 0 && v->head >= v->n) { /* compact instead of growing */
            memmove(v->d, v->d + v->head, sizeof(int) * (size_t)v->n);
            v->head = 0;
        } else {
            int ncap = v->cap ? v->cap * 2 : 4;
            int *nd = (int *)malloc(sizeof(int) * (size_t)ncap);
            if (v->n) memcpy(nd, v->d + v->head, sizeof(int) * (size_t)v->n);
            free(v->d);
            v->d = nd;
            v->head = 0;
            v->cap = ncap;
        }
    }
    v->d[v->head + v->n] = x;
    v->n++;
}
static void vec_erase_front(orc_vec *v, int k) {
    assert(k <= v->n);
    v->head += k;
    v->n -= k;
    if (v->n == 0) v->head = 0;
}
static void vec_clear(orc_vec *v) { v->head = 0; v->n = 0; }
static inline int vec_at(const orc_vec *v, int j) { return v->d[v->head + j]; }

/* ------------------------------------------------------------------ workload */
/* SlidingGraphVec.h:46-66 */
void orc_workload_config(int64_t stream_len, double window_ratio, int cfg_type,
                         double ratio, int64_t batch_count_in, int64_t per_batch_in, int64_t total_in,
                         int *W, int64_t *per_batch, int64_t *batch_count, int64_t *total) {
    /* IndexType sliding_window_size = size_t * double  (truncating conversion), :47 */
    int sw = (int)((double)(uint64_t)stream_len * window_ratio);
    uint64_t pb = (uint64_t)per_batch_in, bc = (uint64_t)batch_count_in, tot = (uint64_t)total_in;
    if (cfg_type == 0) {
        pb = (uint64_t)(ratio * sw); /* :52 size_t = double * int */
        tot = pb * bc;               /* :53 */
    } else {
        bc = (tot + pb - 1) / pb;    /* :58 */
    }
    if (tot > (uint64_t)stream_len - (uint64_t)sw) tot = (uint64_t)stream_len - (uint64_t)sw; /* :64-66 */
    *W = sw;
    *per_batch = (int64_t)pb;
    *batch_count = (int64_t)bc;
    *total = (int64_t)tot;
}

/* ------------------------------------------------------------------ graph */
static void add_window_edge(orc_graph *g, int v1, int v2) {
    /* SlidingGraphVec.h:81-93 (and :118-125) */
    vec_push(&g->out[v1], v2);
    vec_push(&g->in[v2], v1);
    if (!g->directed) {
        vec_push(&g->out[v2], v1);
        vec_push(&g->in[v1], v2);
    }
}

orc_graph *orc_graph_create(int V, const int *e1, const int *e2, int64_t stream_len,
                            int directed, int W, int c) {
    orc_graph *g = (orc_graph *)calloc(1, sizeof(orc_graph));
    g->V = V;
    g->directed = directed;
    g->stream_len = stream_len;
    g->s1 = (int *)malloc(sizeof(int) * (size_t)(stream_len > 0 ? stream_len : 1));
    g->s2 = (int *)malloc(sizeof(int) * (size_t)(stream_len > 0 ? stream_len : 1));
    memcpy(g->s1, e1, sizeof(int) * (size_t)stream_len);
    memcpy(g->s2, e2, sizeof(int) * (size_t)stream_len);
    g->W = W;
    g->c = c;
    g->edge_count = directed ? W : 2 * W;
    g->deg = (int *)calloc((size_t)V, sizeof(int));
    g->out = (orc_vec *)calloc((size_t)V, sizeof(orc_vec));
    g->in = (orc_vec *)calloc((size_t)V, sizeof(orc_vec));
    g->out_change = (int *)calloc((size_t)V, sizeof(int));
    g->in_change = (int *)calloc((size_t)V, sizeof(int));
    size_t cap = (size_t)(c > 0 ? c : 1);
    g->b1 = (int *)malloc(sizeof(int) * cap * 4);
    g->b2 = (int *)malloc(sizeof(int) * cap * 4);
    g->bins = (uint8_t *)malloc(cap * 4);
    g->n1 = (int *)malloc(sizeof(int) * cap * 2);
    g->n2 = (int *)malloc(sizeof(int) * cap * 2);
    assert(W <= stream_len);
    for (int i = 0; i < W; ++i) {
        int v1 = g->s1[i], v2 = g->s2[i];
        assert(0 <= v1 && v1 < V && 0 <= v2 && v2 < V);
        g->deg[v1]++;
        if (!directed) g->deg[v2]++;
        add_window_edge(g, v1, v2);
    }
    g->pos = W;
    return g;
}

void orc_graph_destroy(orc_graph *g) {
    if (!g) return;
    for (int i = 0; i < g->V; ++i) {
        free(g->out[i].d);
        free(g->in[i].d);
    }
    free(g->out); free(g->in); free(g->deg);
    free(g->out_change); free(g->in_change);
    free(g->b1); free(g->b2); free(g->bins); free(g->n1); free(g->n2);
    free(g->s1); free(g->s2);
    free(g);
}

/* SlidingGraphVec.h:219-275 */
int orc_graph_stream_updates(orc_graph *g) {
    int64_t c = g->c;
    if (g->pos + c > g->stream_len) return 1; /* :220-221 */
    for (int64_t i = 0; i < c; ++i) {         /* new_stream, :226-233 */
        g->n1[i] = g->s1[g->pos + i];
        g->n2[i] = g->s2[g->pos + i];
    }
    g->nlen = (int)c;
    int64_t left = g->pos - g->W;             /* :237 window_left_pos */
    for (int64_t i = 0; i < c; ++i) {         /* deletes, :240-246 */
        g->b1[i] = g->s1[left + i];
        g->b2[i] = g->s2[left + i];
        g->bins[i] = 0;
    }
    for (int64_t i = 0; i < c; ++i) {         /* inserts, :252-258 */
        g->b1[c + i] = g->s1[g->pos + i];
        g->b2[c + i] = g->s2[g->pos + i];
        g->bins[c + i] = 1;
    }
    g->pos += c;
    g->blen = (int)(2 * c);
    if (!g->directed) {                       /* mirrored copy, :266-272 */
        int len = g->blen;
        memcpy(g->b1 + len, g->b2, sizeof(int) * (size_t)len);
        memcpy(g->b2 + len, g->b1, sizeof(int) * (size_t)len);
        memcpy(g->bins + len, g->bins, (size_t)len);
        g->blen = 2 * len;
    }
    return 0;
}

/* SlidingGraphVec.h:139-195 */
void orc_graph_inc_construct(orc_graph *g, int mode) {
    int L = g->blen;
    for (int i = 0; i < L; ++i) {             /* :150-155 */
        g->out_change[g->b1[i]] = 0;
        g->in_change[g->b2[i]] = 0;
    }
    for (int i = 0; i < L; ++i) {             /* :156-163 */
        if (!g->bins[i]) {
            g->out_change[g->b1[i]]++;
            g->in_change[g->b2[i]]++;
        }
    }
    for (int i = 0; i < L; ++i) {             /* :164-178 */
        if (!g->bins[i]) {
            int v1 = g->b1[i], v2 = g->b2[i];
            g->deg[v1]--;
            if (g->out_change[v1]) {
                vec_erase_front(&g->out[v1], g->out_change[v1]);
                g->out_change[v1] = 0;
            }
            if (g->in_change[v2]) {
                vec_erase_front(&g->in[v2], g->in_change[v2]);
                g->in_change[v2] = 0;
            }
        }
    }
    if (mode == 0 || g->directed) {           /* :181-189, batch order */
        for (int i = 0; i < L; ++i) {
            if (g->bins[i]) {
                int v1 = g->b1[i], v2 = g->b2[i];
                g->deg[v1]++;
                vec_push(&g->out[v1], v2);
                vec_push(&g->in[v2], v1);
            }
        }
    } else {                                  /* stream order: record + its mirror together */
        int half = L / 2;
        for (int i = 0; i < half; ++i) {
            if (g->bins[i]) {
                int v1 = g->b1[i], v2 = g->b2[i];
                g->deg[v1]++;
                g->deg[v2]++;
                add_window_edge(g, v1, v2);
            }
        }
    }
}

/* SlidingGraphVec.h:99-136 */
void orc_graph_scratch_construct(orc_graph *g) {
    for (int i = 0; i < g->V; ++i) { vec_clear(&g->in[i]); vec_clear(&g->out[i]); }
    for (int64_t k = g->pos - g->W; k < g->pos; ++k) add_window_edge(g, g->s1[k], g->s2[k]);
    for (int i = 0; i < g->V; ++i) g->deg[i] = g->out[i].n;
}

void orc_graph_flatten(const orc_graph *g, int which, int *row_ptr, int *col) {
    const orc_vec *a = which ? g->in : g->out;
    int off = 0;
    for (int u = 0; u < g->V; ++u) {
        row_ptr[u] = off;
        for (int j = 0; j < a[u].n; ++j) col[off + j] = vec_at(&a[u], j);
        off += a[u].n;
    }
    row_ptr[g->V] = off;
}

/* ------------------------------------------------------------------ state */
orc_state *orc_state_create(int V, int source, double eps) {
    orc_state *s = (orc_state *)calloc(1, sizeof(orc_state));
    s->V = V; s->source = source; s->eps = eps;
    s->p = (double *)calloc((size_t)V + 1, sizeof(double));
    s->r = (double *)calloc((size_t)V + 1, sizeof(double));
    s->predeg = (int *)calloc((size_t)V + 1, sizeof(int));
    s->ft = (int *)malloc(sizeof(int) * ((size_t)V + 1));
    s->ft2 = (int *)malloc(sizeof(int) * ((size_t)V + 1));
    s->ft_r = (double *)malloc(sizeof(double) * ((size_t)V + 1));
    s->status = (int *)calloc((size_t)V + 1, sizeof(int));
    s->qcap = (int64_t)V + 1;
    s->q = (int *)malloc(sizeof(int) * (size_t)s->qcap);
    return s;
}
void orc_state_destroy(orc_state *s) {
    if (!s) return;
    free(s->p); free(s->r); free(s->predeg); free(s->ft); free(s->ft2); free(s->ft_r);
    free(s->status); free(s->q); free(s->trace_v); free(s->trace_off);
    free(s->edge_ind); free(s->edge_flag); free(s->vertex_offset);
    free(s);
}
void orc_state_trace(orc_state *s, int on) {
    s->trace_on = on;
    s->trace_len = 0;
    s->trace_iters = 0;
}
void orc_state_reset_stats(orc_state *s) { s->stat_iters = s->stat_F = s->stat_E = s->stat_N = 0; }

static void trace_frontier(orc_state *s, const int *ft, int n) {
    if (!s->trace_on) return;
    if (s->trace_len + n > s->trace_cap) {
        s->trace_cap = (s->trace_len + n) * 2 + 1024;
        s->trace_v = (int *)realloc(s->trace_v, sizeof(int) * (size_t)s->trace_cap);
    }
    if (s->trace_iters + 2 > s->trace_off_cap) {
        s->trace_off_cap = s->trace_off_cap * 2 + 64;
        s->trace_off = (int64_t *)realloc(s->trace_off, sizeof(int64_t) * (size_t)s->trace_off_cap);
    }
    memcpy(s->trace_v + s->trace_len, ft, sizeof(int) * (size_t)n);
    s->trace_off[s->trace_iters] = s->trace_len;
    s->trace_len += n;
    s->trace_iters++;
    s->trace_off[s->trace_iters] = s->trace_len;
}

/* gpu/PPRCommon.cuh:6-11, cpu/PPRCPUMTCilkRev.h:75-80: strict inequalities */
int orc_is_legal_push(double r, int phase, double eps) {
    if ((phase == 0 && r > eps) || (phase == 1 && r < -eps)) return 1;
    return 0;
}
#define LEGAL(x) orc_is_legal_push((x), phase, s->eps)

/* ------------------------------------------------------------------ shared pieces */
/* cpu/PPRCPUMTCilk.h:166-174 CopyOutDegree then :157-165 RevertOutDegree */
void orc_copy_revert_out_degree(orc_state *s, const orc_graph *g) {
    int L = g->blen;
    for (int i = 0; i < L; ++i) {
        int u = g->b1[i], v = g->b2[i];
        s->predeg[u] = g->deg[u];
        s->predeg[v] = g->deg[v];
    }
    for (int i = 0; i < L; ++i) {
        int u = g->b1[i];
        if (g->bins[i]) s->predeg[u]--;
        else s->predeg[u]++;
    }
}

/* cpu/PPRCPUMTCilkRev.h:108-124 StreamUpdateAppData (== gpu/StreamUpdate.cuh:56-67) */
void orc_stream_update(orc_state *s, const orc_graph *g) {
    int L = g->blen;
    double *pagerank = s->p, *residual = s->r;
    for (int i = 0; i < L; ++i) {
        int u = g->b1[i], v = g->b2[i];
        double add = (1.0 - ORC_ALPHA) * pagerank[v] - pagerank[u] - ORC_ALPHA * residual[u] +
                     ORC_ALPHA * (s->source == u ? 1.0 : 0.0);
        if (g->bins[i]) {
            s->predeg[u]++;
            residual[u] += add / (s->predeg[u] + 1) / ORC_ALPHA;
        } else {
            s->predeg[u]--;
            residual[u] -= add / (s->predeg[u] + 1) / ORC_ALPHA;
        }
    }
}

/* cpu/PPRCPUMTCilkRev.h:126-156 DynPushInit: first legal occurrence wins, tails
 * (batch order) before heads. predeg doubles as vertex_map exactly as there. */
void orc_dyn_push_init(orc_state *s, const orc_graph *g, int phase) {
    int L = g->blen;
    const int *set[2] = {g->b1, g->b2};
    int *map = s->predeg;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < L; ++j) map[set[i][j]] = s->V;
    int n = 0;
    for (int i = 0; i < 2; ++i) {
        for (int j = 0; j < L; ++j) {
            int u = set[i][j];
            int off = i * L + j;
            if (LEGAL(s->r[u]) && map[u] == s->V) {
                map[u] = off;
                s->ft[n++] = u;
            }
        }
    }
    s->ft_count = n;
}

/* gpu/Inspect.cuh:8-48 InspectPureRev (as an ascending-id set) */
int orc_inspect(const orc_state *s, int phase, int *out) {
    int n = 0;
    for (int u = 0; u < s->V; ++u)
        if (LEGAL(s->r[u])) out[n++] = u;
    return n;
}

/* ------------------------------------------------------------------ schedule A (cilk -t 1) */
/* cpu/PPRCPUMTCilkRev.h:175-182 */
void orc_cilk_init(orc_state *s) {
    for (int u = 0; u < s->V; ++u) {
        s->p[u] = 0.0;
        s->r[u] = (s->source == u) ? 1.0 : 0.0;
    }
    s->ft[0] = s->source;
    s->ft_count = 1;
}

/* cpu/PPRCPUMTCilkRev.h:184-289. With one worker every parallel_for is a plain
 * loop and sequence::pack (cpu/CilkUtil.h:246-262) is a stable compaction, so the
 * next frontier is [crossing targets in frontier x adjacency order] followed by
 * [repaired vertices in frontier order]. */
void orc_cilk_main_loop(orc_state *s, const orc_graph *g, int phase) {
    double *residual = s->r, *pagerank = s->p;
    const int *deg = g->deg;
    for (;;) {
        int F = s->ft_count;
        if (F == 0) break;
        trace_frontier(s, s->ft, F);
        int n = 0;
        int64_t E = 0;
        for (int i = 0; i < F; ++i) {                     /* :208-257 */
            int u = s->ft[i];
            double ru = residual[u];
            s->ft_r[i] = ru;
            pagerank[u] += ORC_ALPHA * ru;
            const orc_vec *nb = &g->in[u];
            int indegu = nb->n;
            E += indegu;
            for (int j = 0; j < indegu; ++j) {
                int v = vec_at(nb, j);
                double add = (1.0 - ORC_ALPHA) * ru / (deg[v] + 1);
                double prer = residual[v];                /* AtomicAddResidual :82-96 */
                residual[v] = prer + add;
                double curr = prer + add;
                if (LEGAL(prer) == 0 && LEGAL(curr) == 1) s->ft2[n++] = v;
            }
        }
        int n1 = n;
        for (int i = 0; i < F; ++i) {                     /* :267-277 repair */
            int u = s->ft[i];
            residual[u] -= s->ft_r[i];
            if (LEGAL(residual[u])) s->ft2[n++] = u;
        }
        assert(n <= s->V);
        s->stat_iters++; s->stat_F += F; s->stat_E += E; s->stat_N += n;
        (void)n1;
        int *t = s->ft; s->ft = s->ft2; s->ft2 = t;       /* :282-283 */
        s->ft_count = n;
        ++s->iteration_id;
    }
}

void orc_cilk_execute(orc_state *s, const orc_graph *g) { /* :38-41 */
    orc_cilk_init(s);
    orc_cilk_main_loop(s, g, 0);
}

void orc_cilk_inc_execute(orc_state *s, const orc_graph *g) { /* :43-73 */
    orc_copy_revert_out_degree(s, g);
    orc_stream_update(s, g);
    ++s->iteration_id;
    orc_dyn_push_init(s, g, 0);
    orc_cilk_main_loop(s, g, 0);
    ++s->iteration_id;
    orc_dyn_push_init(s, g, 1);
    orc_cilk_main_loop(s, g, 1);
}

/* ------------------------------------------------------------------ schedule A with T workers */
int orc_max_threads(void) { return omp_get_max_threads(); }

/* cpu/PPRCPUMTCilkRev.h:82-96 AtomicAddResidual: CAS loop, returns the old value */
static inline double atomic_add_residual(double *addr, double add) {
    union { double d; long long i; } old_v, new_v;
    do {
        old_v.d = *(volatile double *)addr;
        new_v.d = old_v.d + add;
    } while (!__sync_bool_compare_and_swap((long long *)addr, old_v.i, new_v.i));
    return old_v.d;
}

/* stable parallel pack (sequence::pack, cpu/CilkUtil.h:246-262): out gets In[i] where Fl[i] */
static int64_t pack_mt(const int *in, const unsigned char *fl, int64_t n, int *out, int threads) {
    if (n <= 0) return 0;
    int nb = threads * 4;
    if (nb > n) nb = (int)n;
    int64_t *sums = (int64_t *)malloc(sizeof(int64_t) * ((size_t)nb + 1));
    const int64_t bs = (n + nb - 1) / nb;
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int b = 0; b < nb; ++b) {
        int64_t lo = b * bs, hi = lo + bs < n ? lo + bs : n, c = 0;
        for (int64_t i = lo; i < hi; ++i) c += fl[i];
        sums[b] = c;
    }
    int64_t tot = 0;
    for (int b = 0; b < nb; ++b) { int64_t c = sums[b]; sums[b] = tot; tot += c; }
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int b = 0; b < nb; ++b) {
        int64_t lo = b * bs, hi = lo + bs < n ? lo + bs : n, k = sums[b];
        for (int64_t i = lo; i < hi; ++i) if (fl[i]) out[k++] = in[i];
    }
    free(sums);
    return tot;
}

/* cpu/PPRCPUMTCilkRev.h:184-289 with parallel_for == omp parallel for */
void orc_cilk_main_loop_mt(orc_state *s, const orc_graph *g, int phase, int threads) {
    double *residual = s->r, *pagerank = s->p;
    const int *deg = g->deg;
    const int64_t need = (int64_t)g->edge_count + s->V + 16;
    if (s->edge_cap < need) {
        free(s->edge_ind); free(s->edge_flag); free(s->vertex_offset);
        s->edge_ind = (int *)malloc(sizeof(int) * (size_t)need);
        s->edge_flag = (unsigned char *)malloc((size_t)need);
        s->vertex_offset = (int64_t *)malloc(sizeof(int64_t) * ((size_t)s->V + 2));
        s->edge_cap = need;
    }
    for (;;) {
        const int F = s->ft_count;
        if (F == 0) break;
        int64_t total = 0; /* vertex_offset = plusScan(indeg), :200-205 */
        for (int i = 0; i < F; ++i) { s->vertex_offset[i] = total; total += g->in[s->ft[i]].n; }
#pragma omp parallel for num_threads(threads) schedule(dynamic, 64)
        for (int i = 0; i < F; ++i) { /* :208-257 */
            const int u = s->ft[i];
            const double ru = residual[u];
            s->ft_r[i] = ru;
            pagerank[u] += ORC_ALPHA * ru;
            const orc_vec *nb = &g->in[u];
            for (int j = 0; j < nb->n; ++j) {
                const int64_t off = s->vertex_offset[i] + j;
                const int v = vec_at(nb, j);
                const double add = (1.0 - ORC_ALPHA) * ru / (deg[v] + 1);
                const double prer = atomic_add_residual(&residual[v], add);
                const double curr = prer + add;
                if (LEGAL(prer) == 0 && LEGAL(curr) == 1) { s->edge_ind[off] = v; s->edge_flag[off] = 1; }
                else s->edge_flag[off] = 0;
            }
        }
        const int64_t n1 = pack_mt(s->edge_ind, s->edge_flag, total, s->ft2, threads); /* :265 */
#pragma omp parallel for num_threads(threads) schedule(static)
        for (int i = 0; i < F; ++i) { /* :267-277 */
            const int u = s->ft[i];
            residual[u] -= s->ft_r[i];
            if (LEGAL(residual[u])) { s->edge_flag[i] = 1; s->edge_ind[i] = u; }
            else s->edge_flag[i] = 0;
        }
        const int64_t n2 = pack_mt(s->edge_ind, s->edge_flag, F, s->ft2 + n1, threads); /* :279 */
        s->stat_iters++; s->stat_F += F; s->stat_E += total; s->stat_N += n1 + n2;
        int *t = s->ft; s->ft = s->ft2; s->ft2 = t;
        s->ft_count = (int)(n1 + n2);
        ++s->iteration_id;
    }
}

void orc_cilk_inc_execute_mt(orc_state *s, const orc_graph *g, int threads) { /* :43-73 */
    orc_copy_revert_out_degree(s, g);
    orc_stream_update(s, g); /* 0.3 % of the region; per-u locks of :108-124 serialise same-tail records anyway */
    ++s->iteration_id;
    orc_dyn_push_init(s, g, 0);
    orc_cilk_main_loop_mt(s, g, 0, threads);
    ++s->iteration_id;
    orc_dyn_push_init(s, g, 1);
    orc_cilk_main_loop_mt(s, g, 1, threads);
}

/* ------------------------------------------------------------------ schedule B (FIFO, cpu/PPRCPURev.h) */
static void q_push(orc_state *s, int v) {
    s->q[s->qtail % s->qcap] = v;
    s->qtail++;
    assert(s->qtail - s->qhead <= s->qcap);
}

/* cpu/PPRCPURev.h:83-109 MainLoopFIFO */
static void fifo_main_loop(orc_state *s, const orc_graph *g, int phase) {
    double *residual = s->r, *pagerank = s->p;
    const int *deg = g->deg;
    while (s->qhead != s->qtail) {
        int u = s->q[s->qhead % s->qcap];
        s->qhead++;
        s->status[u] = 0;
        const orc_vec *nb = &g->in[u];
        s->stat_F++; s->stat_E += nb->n;
        for (int j = 0; j < nb->n; ++j) {
            int v = vec_at(nb, j);
            if (v >= s->V) continue;
            residual[v] += (1.0 - ORC_ALPHA) * residual[u] / (deg[v] + 1);
            if (LEGAL(residual[v]) && s->status[v] == 0) {
                q_push(s, v);
                s->status[v] = 1;
            }
        }
        pagerank[u] += ORC_ALPHA * residual[u];
        residual[u] = 0.0;
    }
}

/* cpu/PPRCPURev.h:111-124 DynPushInit */
static void fifo_dyn_push_init(orc_state *s, const orc_graph *g, int phase) {
    for (int i = 0; i < g->blen; ++i) {
        int u = g->b1[i], v = g->b2[i];
        if (LEGAL(s->r[u]) && s->status[u] == 0) { q_push(s, u); s->status[u] = 1; }
        if (LEGAL(s->r[v]) && s->status[v] == 0) { q_push(s, v); s->status[v] = 1; }
    }
}

void orc_fifo_execute(orc_state *s, const orc_graph *g) { /* :21-34 */
    for (int u = 0; u < s->V; ++u) {
        s->p[u] = 0.0;
        s->r[u] = (s->source == u) ? 1.0 : 0.0;
    }
    memset(s->status, 0, sizeof(int) * (size_t)s->V);
    s->qhead = s->qtail = 0;
    s->status[s->source] = 1;
    q_push(s, s->source);
    fifo_main_loop(s, g, 0);
}

void orc_fifo_inc_execute(orc_state *s, const orc_graph *g) { /* :36-62 */
    int L = g->blen;
    /* cpu/PPRCPU.h:130-149 CopyOutDegree / RevertOutDegree */
    for (int i = 0; i < L; ++i) {
        int u = g->b1[i], v = g->b2[i];
        s->predeg[u] = g->deg[u];
        s->predeg[v] = g->deg[v];
    }
    for (int i = 0; i < L; ++i) {
        int u = g->b1[i];
        if (g->bins[i]) s->predeg[u]--;
        else s->predeg[u]++;
    }
    /* cpu/PPRCPURev.h:64-74 StreamUpdateAppData per record */
    for (int i = 0; i < L; ++i) {
        int u = g->b1[i], v = g->b2[i];
        if (g->bins[i]) s->predeg[u]++;
        else s->predeg[u]--;
        double add = (1.0 - ORC_ALPHA) * s->p[v] - s->p[u] - ORC_ALPHA * s->r[u] +
                     ORC_ALPHA * (s->source == u ? 1.0 : 0.0);
        if (g->bins[i]) s->r[u] += add / (s->predeg[u] + 1) / ORC_ALPHA;
        else s->r[u] -= add / (s->predeg[u] + 1) / ORC_ALPHA;
    }
    fifo_dyn_push_init(s, g, 0);
    fifo_main_loop(s, g, 0);
    fifo_dyn_push_init(s, g, 1);
    fifo_main_loop(s, g, 1);
}

/* ------------------------------------------------------------------ schedule C (synchronous) */
/* Snapshot every frontier residual first (the "ru = residual[u]; vertex_ft_r = ru;
 * pagerank[u] += ALPHA*ru" head of gpu/ExpandRev.cuh:34-42 for ALL frontier
 * vertices), then push (gpu/ExpandRev.cuh:70-77), then repair
 * (gpu/ExpandRev.cuh:708-743). Frontier seeded by Inspect over all V
 * (gpu/PPRRevPushGPU.cuh:100-104). */
void orc_sync_main_loop(orc_state *s, const orc_graph *g, int phase) {
    double *residual = s->r, *pagerank = s->p;
    const int *deg = g->deg;
    s->ft_count = orc_inspect(s, phase, s->ft);
    for (;;) {
        int F = s->ft_count;
        if (F == 0) break;
        trace_frontier(s, s->ft, F);
        for (int i = 0; i < F; ++i) {
            int u = s->ft[i];
            double ru = residual[u];
            s->ft_r[i] = ru;
            pagerank[u] += ORC_ALPHA * ru;
        }
        int n = 0;
        int64_t E = 0;
        for (int i = 0; i < F; ++i) {
            int u = s->ft[i];
            double ru = s->ft_r[i];
            const orc_vec *nb = &g->in[u];
            E += nb->n;
            for (int j = 0; j < nb->n; ++j) {
                int v = vec_at(nb, j);
                double add = (1.0 - ORC_ALPHA) * ru / (deg[v] + 1);
                double prer = residual[v];
                residual[v] = prer + add;
                double curr = prer + add;
                if (LEGAL(prer) == 0 && LEGAL(curr) == 1) s->ft2[n++] = v;
            }
        }
        for (int i = 0; i < F; ++i) {
            int u = s->ft[i];
            residual[u] -= s->ft_r[i];
            if (LEGAL(residual[u])) s->ft2[n++] = u;
        }
        assert(n <= s->V);
        s->stat_iters++; s->stat_F += F; s->stat_E += E; s->stat_N += n;
        int *t = s->ft; s->ft = s->ft2; s->ft2 = t;
        s->ft_count = n;
        ++s->iteration_id;
    }
}

void orc_sync_execute(orc_state *s, const orc_graph *g) {
    for (int u = 0; u < s->V; ++u) {
        s->p[u] = 0.0;
        s->r[u] = (s->source == u) ? 1.0 : 0.0;
    }
    orc_sync_main_loop(s, g, 0);
}

/* NOT a schedule of the reference: the checker of the engine's MERGED loop (include/dppr.h dppr_set_phase_merge).
 * One synchronous loop over residuals of BOTH signs: frontier = {u : |r[u]| > eps}; every frontier vertex is
 * snapshotted and credited (cpu/PPRCPUMTCilkRev.h:208-257's push, gpu/ExpandRev.cuh:34-42), every push lands, the
 * repair follows (gpu/ExpandRev.cuh:708-743); the next frontier is every vertex the iteration left with |r| > eps
 * (adds of both signs may take a residual across the threshold more than once: a mark keeps the list free of
 * duplicates). Same push rule, same invariant, |r| <= eps at the end. */
void orc_merged_main_loop(orc_state *s, const orc_graph *g, double eps) {
    double *residual = s->r, *pagerank = s->p;
    const int *deg = g->deg;
    int *mark = (int *)calloc((size_t)s->V + 1, sizeof(int));
    int n0 = 0;
    for (int u = 0; u < s->V; ++u)
        if (residual[u] > eps || residual[u] < -eps) s->ft[n0++] = u;
    s->ft_count = n0;
    int stamp = 0;
    for (;;) {
        int F = s->ft_count;
        if (F == 0) break;
        trace_frontier(s, s->ft, F);
        ++stamp;
        for (int i = 0; i < F; ++i) {
            int u = s->ft[i];
            double ru = residual[u];
            s->ft_r[i] = ru;
            pagerank[u] += ORC_ALPHA * ru;
        }
        int n = 0;
        int64_t E = 0;
        for (int i = 0; i < F; ++i) {
            int u = s->ft[i];
            double ru = s->ft_r[i];
            const orc_vec *nb = &g->in[u];
            E += nb->n;
            for (int j = 0; j < nb->n; ++j) {
                int v = vec_at(nb, j);
                residual[v] += (1.0 - ORC_ALPHA) * ru / (deg[v] + 1);
                if (mark[v] != stamp) { mark[v] = stamp; s->ft2[n++] = v; } /* touched: decided below */
            }
        }
        for (int i = 0; i < F; ++i) {
            int u = s->ft[i];
            residual[u] -= s->ft_r[i];
            if (mark[u] != stamp) { mark[u] = stamp; s->ft2[n++] = u; }
        }
        int m = 0;
        for (int i = 0; i < n; ++i) {
            int v = s->ft2[i];
            if (residual[v] > eps || residual[v] < -eps) s->ft2[m++] = v;
        }
        s->stat_iters++; s->stat_F += F; s->stat_E += E; s->stat_N += m;
        int *t = s->ft; s->ft = s->ft2; s->ft2 = t;
        s->ft_count = m;
        ++s->iteration_id;
    }
    free(mark);
}
void orc_merged_inc_execute(orc_state *s, const orc_graph *g, double eps) {
    orc_copy_revert_out_degree(s, g);
    orc_stream_update(s, g);
    orc_merged_main_loop(s, g, eps);
}

void orc_sync_inc_execute(orc_state *s, const orc_graph *g) {
    orc_copy_revert_out_degree(s, g);
    orc_stream_update(s, g);
    orc_sync_main_loop(s, g, 0);
    orc_sync_main_loop(s, g, 1);
}

/* ------------------------------------------------------------------ variants 1-3 (cilk -t 1) */
/* cpu/PPRCPUMTCilkRevVariants.h at one worker (parallel_for == for, sequence::pack == stable
 * compaction), the classes main() selects with -o (cpu/PPRCPUMTMain.cpp:26-32):
 *   1 FAST_FRONTIER PPRCPUMTCilkRevFF      :224-326  snapshot every frontier residual first
 *       (ft_r = r[u]; p += ALPHA*r[u]; r[u] = 0), push from the snapshot, enqueue on threshold
 *       crossing, no repair step;
 *   2 EAGER         PPRCPUMTCilkRevEager   :112-222  eager reads like variant 0, but duplicates are
 *       kept out by the status array (status[u] = iteration_id for frontier members, a target enters
 *       when its residual is legal and AtomicUpdateStatus succeeds), repair step as in variant 0;
 *   3 VANILLA       PPRCPUMTCilkRevVanilla :6-110    snapshot first (like 1) + status array (like 2).
 * status starts at -1 (cpu/PPRCPUMTCilkRev.h:12-13) and is never reset; AtomicUpdateStatus
 * (:96-106) compares with the member iteration_id. */
static int variant_update_status(orc_state *s, int v) { /* cpu/PPRCPUMTCilkRev.h:96-106 */
    if (s->status[v] < s->iteration_id) {
        s->status[v] = s->iteration_id;
        return 1;
    }
    return 0;
}

void orc_variant_main_loop(orc_state *s, const orc_graph *g, int phase, int variant) {
    double *residual = s->r, *pagerank = s->p;
    const int *deg = g->deg;
    if (variant == 0) {
        orc_cilk_main_loop(s, g, phase);
        return;
    }
    const int pre_extract = variant == 1 || variant == 3; /* FF, VANILLA */
    const int use_status = variant == 2 || variant == 3;  /* EAGER, VANILLA */
    for (;;) {
        int F = s->ft_count;
        if (F == 0) break;
        trace_frontier(s, s->ft, F);
        for (int i = 0; i < F; ++i) { /* first parallel_for of each MainLoop */
            int u = s->ft[i];
            if (pre_extract) {        /* Variants.h:24-30 / :242-248 */
                s->ft_r[i] = residual[u];
                pagerank[u] += ORC_ALPHA * residual[u];
                residual[u] = 0.0;
            } else {                  /* :130-134 */
                s->status[u] = s->iteration_id;
            }
        }
        int n = 0;
        int64_t E = 0;
        for (int i = 0; i < F; ++i) {
            int u = s->ft[i];
            double ru;
            if (pre_extract) {
                ru = s->ft_r[i];
            } else {                  /* :139-142 */
                ru = residual[u];
                s->ft_r[i] = ru;
                pagerank[u] += ORC_ALPHA * ru;
            }
            const orc_vec *nb = &g->in[u];
            E += nb->n;
            for (int j = 0; j < nb->n; ++j) {
                int v = vec_at(nb, j);
                double add = (1.0 - ORC_ALPHA) * ru / (deg[v] + 1);
                double prer = residual[v];
                residual[v] = prer + add;
                double curr = prer + add;
                int is_frontier;
                if (use_status) is_frontier = LEGAL(curr) && variant_update_status(s, v); /* :52-57 / :154-159 */
                else is_frontier = LEGAL(prer) == 0 && LEGAL(curr) == 1;                  /* :266-268 */
                if (is_frontier) s->ft2[n++] = v;
            }
        }
        if (!pre_extract) {           /* EAGER's repair step :195-206 */
            for (int i = 0; i < F; ++i) {
                int u = s->ft[i];
                residual[u] -= s->ft_r[i];
                if (LEGAL(residual[u])) s->ft2[n++] = u;
            }
        }
        assert(n <= s->V);
        s->stat_iters++; s->stat_F += F; s->stat_E += E; s->stat_N += n;
        int *t = s->ft; s->ft = s->ft2; s->ft2 = t;
        s->ft_count = n;
        ++s->iteration_id;
    }
}

void orc_variant_execute(orc_state *s, const orc_graph *g, int variant) { /* ExecuteImpl, PPRCPUMTCilkRev.h:38-41 */
    for (int u = 0; u <= s->V; ++u) s->status[u] = -1; /* the constructor's memset, :12-13 */
    s->iteration_id = 0;
    orc_cilk_init(s);
    orc_variant_main_loop(s, g, 0, variant);
}

void orc_variant_inc_execute(orc_state *s, const orc_graph *g, int variant) { /* IncExecuteImpl :43-73 */
    orc_copy_revert_out_degree(s, g);
    orc_stream_update(s, g);
    ++s->iteration_id;
    orc_dyn_push_init(s, g, 0);
    orc_variant_main_loop(s, g, 0, variant);
    ++s->iteration_id;
    orc_dyn_push_init(s, g, 1);
    orc_variant_main_loop(s, g, 1, variant);
}

/* ------------------------------------------------------------------ ground truth */
/* cpu/PPRCPUPowVec.h:55-83 CalPPRRev */
int64_t orc_pow_rev(const orc_graph *g, int source, double alpha, double *out_p) {
    int V = g->V;
    double *pr[2];
    pr[0] = (double *)malloc(sizeof(double) * (size_t)V);
    pr[1] = (double *)malloc(sizeof(double) * (size_t)V);
    for (int u = 0; u < V; ++u) pr[0][u] = (source == u) ? 1 : 0;
    int64_t iteration_count = 0;
    size_t id = 0;
    for (;;) {
        int stop = 1;
        size_t oid = 1 - id;
        for (int u = 0; u < V; ++u) {
            const orc_vec *nb = &g->out[u];
            pr[oid][u] = 0.0;
            for (int j = 0; j < nb->n; ++j) {
                int v = vec_at(nb, j);
                pr[oid][u] += pr[id][v] / ((size_t)nb->n + 1);
            }
            pr[oid][u] = (1.0 - alpha) * pr[oid][u];
            if (u == source) pr[oid][u] += alpha * 1.0;
            if (fabs(pr[oid][u] - pr[id][u]) > 1e-14) stop = 0;
        }
        if (stop) break;
        id = (id + 1) % 2;
        ++iteration_count;
    }
    memcpy(out_p, pr[id], sizeof(double) * (size_t)V);
    free(pr[0]); free(pr[1]);
    return iteration_count;
}

double orc_invariant_max_err(const orc_state *s, const orc_graph *g) {
    double worst = 0.0;
    for (int u = 0; u < s->V; ++u) {
        const orc_vec *nb = &g->out[u];
        double acc = 0.0;
        for (int j = 0; j < nb->n; ++j) acc += s->p[vec_at(nb, j)];
        double rhs = ORC_ALPHA * (u == s->source ? 1.0 : 0.0) + (1.0 - ORC_ALPHA) / (nb->n + 1) * acc;
        double lhs = s->p[u] + ORC_ALPHA * s->r[u];
        double e = fabs(lhs - rhs);
        if (e > worst) worst = e;
    }
    return worst;
}

double orc_max_abs_residual(const orc_state *s) {
    double worst = 0.0;
    for (int u = 0; u < s->V; ++u) {
        double e = fabs(s->r[u]);
        if (e > worst) worst = e;
    }
    return worst;
}
