/* mf_oracle.c - CPU ORACLE.  TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * A plain-C restatement of the blocked-SGD matrix-factorization path of
 * dtrager02/Recommender, used ONLY by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py as the checker / reported CPU baseline.  The product
 * (recommender_amd/) never imports, links or calls anything in oracle/.
 *
 * Pinned: checked against golden vectors produced by the reference's own functions
 * imported under identity stubs (oracle/gen_golden.py -> tests/golden/ *.npz), see
 * tests/test_oracle_golden.py.
 *
 * Each function cites the reference lines it restates (paths under /root/reference).
 * Build: make -C oracle      ->  oracle/libmforacle.so
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ scheduler (S1 / S2)
 * RayBlockScheduler.py:14-116.  State: update_counter[W][W], free row / col LISTS in the
 * reference's order (append on release, :101,107), completed sets.  Tie-break:
 * deterministic=1 -> first strict minimum in scan order (MPBlockScheduler.py:48-54);
 * deterministic=0 -> uniform choice among ties (RayBlockScheduler.py:58) drawn from a
 * splitmix64 stream (NOT Python's `random`; trace parity with random.seed is pinned by the
 * Python host mirror against tests/golden/scheduler.npz instead). */
typedef struct {
    int32_t W, iters, deterministic;
    int32_t *counter;            /* [W*W] */
    int32_t *rows, *cols;        /* free lists */
    int32_t n_rows, n_cols, n_done_rows, n_done_cols;
    uint64_t rng;
    int32_t *ties;               /* [W*W][2] scratch */
} orc_sched;

static uint64_t splitmix64(uint64_t *s)
{
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

orc_sched *orc_sched_new(int32_t W, int32_t iters, int32_t deterministic, uint64_t seed)
{
    orc_sched *s = (orc_sched *)calloc(1, sizeof(*s));
    if (!s) return NULL;
    s->W = W; s->iters = iters; s->deterministic = deterministic; s->rng = seed;
    s->counter = (int32_t *)calloc((size_t)W * W, sizeof(int32_t));
    s->rows = (int32_t *)malloc(sizeof(int32_t) * (size_t)W);
    s->cols = (int32_t *)malloc(sizeof(int32_t) * (size_t)W);
    s->ties = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)W * W);
    for (int i = 0; i < W; ++i) s->rows[i] = s->cols[i] = i;
    s->n_rows = s->n_cols = W;
    return s;
}

void orc_sched_free(orc_sched *s)
{
    if (!s) return;
    free(s->counter); free(s->rows); free(s->cols); free(s->ties); free(s);
}

/* RayBlockScheduler.py:74-107 */
static int orc_sched_completed(orc_sched *s, int32_t r, int32_t c)
{
    for (int i = 0; i < s->n_rows; ++i) if (s->rows[i] == r) return -1;   /* desync guard :86-87 */
    for (int i = 0; i < s->n_cols; ++i) if (s->cols[i] == c) return -1;
    s->counter[r * s->W + c] += 1;
    int rc = 0, cc = 0;
    for (int i = 0; i < s->W; ++i) {
        if (s->counter[r * s->W + i] == s->iters) ++rc;
        if (s->counter[i * s->W + c] == s->iters) ++cc;
    }
    if (cc == s->W) ++s->n_done_cols; else s->cols[s->n_cols++] = c;
    if (rc == s->W) ++s->n_done_rows; else s->rows[s->n_rows++] = r;
    return 0;
}

/* RayBlockScheduler.py:35-68; returns 1 and (*nr,*nc) or 0 for None. */
int orc_sched_get_next(orc_sched *s, int have, int32_t r, int32_t c, int32_t *nr, int32_t *nc)
{
    if (have && orc_sched_completed(s, r, c)) return 0;
    if (!s->n_cols) return 0;
    int32_t best = INT32_MAX, nt = 0;
    for (int a = 0; a < s->n_rows; ++a)
        for (int b = 0; b < s->n_cols; ++b) {
            const int32_t i = s->rows[a], j = s->cols[b], v = s->counter[i * s->W + j];
            if (v >= s->iters) continue;
            if (v < best) { best = v; nt = 0; s->ties[0] = i; s->ties[1] = j; nt = 1; }
            else if (v == best && !s->deterministic) { s->ties[2 * nt] = i; s->ties[2 * nt + 1] = j; ++nt; }
        }
    if (!nt) return 0;
    const int32_t pick = s->deterministic ? 0 : (int32_t)(splitmix64(&s->rng) % (uint64_t)nt);
    *nr = s->ties[2 * pick]; *nc = s->ties[2 * pick + 1];
    for (int a = 0; a < s->n_rows; ++a) if (s->rows[a] == *nr) {
        memmove(s->rows + a, s->rows + a + 1, sizeof(int32_t) * (size_t)(s->n_rows - a - 1)); --s->n_rows; break; }
    for (int b = 0; b < s->n_cols; ++b) if (s->cols[b] == *nc) {
        memmove(s->cols + b, s->cols + b + 1, sizeof(int32_t) * (size_t)(s->n_cols - b - 1)); --s->n_cols; break; }
    return 1;
}

int orc_sched_all_done(const orc_sched *s)
{   /* check_completion, RayBlockScheduler.py:30-32 */
    return s->n_done_rows == s->W && s->n_done_cols == s->W;
}

const int32_t *orc_sched_counter(const orc_sched *s) { return s->counter; }

/* ------------------------------------------------------------------ data prep (integer, bit-exact)
 * D2 remap, SGDRecommender.py:251-256: id -> perm[id] on columns 0 and 1. */
int orc_remap(int32_t *samples, int64_t n, int32_t stride, const int32_t *row_perm, const int32_t *col_perm)
{
    for (int64_t t = 0; t < n; ++t) {
        samples[t * stride] = row_perm[samples[t * stride]];
        samples[t * stride + 1] = col_perm[samples[t * stride + 1]];
    }
    return 0;
}

/* D2 test filter, SGDRecommender.py:259-263: keep test rows whose user AND item occur in train.
 * Stable.  Returns the number kept; out may alias nothing. */
int64_t orc_filter_test(const int32_t *test, int64_t nt, int32_t stride,
                        const int32_t *train, int64_t n, int32_t tstride,
                        int64_t n_users, int64_t n_items, int32_t *out)
{
    uint8_t *su = (uint8_t *)calloc((size_t)n_users, 1), *si = (uint8_t *)calloc((size_t)n_items, 1);
    for (int64_t t = 0; t < n; ++t) { su[train[t * tstride]] = 1; si[train[t * tstride + 1]] = 1; }
    int64_t m = 0;
    for (int64_t t = 0; t < nt; ++t) {
        const int32_t u = test[t * stride], i = test[t * stride + 1];
        if (u >= 0 && u < n_users && i >= 0 && i < n_items && su[u] && si[i]) {
            for (int c = 0; c < stride; ++c) out[m * stride + c] = test[t * stride + c];
            ++m;
        }
    }
    free(su); free(si);
    return m;
}

/* D3 bucketing, SGDRecommender.py:187-212: block (r,c) holds the rows with u//ubp==r and
 * i//ibp==c in their original relative order; blocks concatenated row-major.
 * out_samples [n][3], out_block_ptr [Pu*Pi+1]. */
int orc_bucket(const int32_t *samples, int64_t n, int32_t stride, int32_t ubp, int32_t ibp,
               int32_t Pu, int32_t Pi, int32_t *out_samples, int64_t *out_block_ptr)
{
    const int64_t nb = (int64_t)Pu * Pi;
    int64_t *cur = (int64_t *)calloc((size_t)nb + 1, sizeof(int64_t));
    if (!cur) return -1;
    for (int64_t t = 0; t < n; ++t) {
        const int64_t r = samples[t * stride] / ubp, c = samples[t * stride + 1] / ibp;
        if (r >= Pu || c >= Pi) { free(cur); return -2; }
        ++cur[r * Pi + c + 1];
    }
    for (int64_t b = 0; b < nb; ++b) cur[b + 1] += cur[b];
    memcpy(out_block_ptr, cur, sizeof(int64_t) * (size_t)(nb + 1));
    for (int64_t t = 0; t < n; ++t) {
        const int64_t r = samples[t * stride] / ubp, c = samples[t * stride + 1] / ibp;
        const int64_t d = cur[r * Pi + c]++;
        out_samples[3 * d] = samples[t * stride];
        out_samples[3 * d + 1] = samples[t * stride + 1];
        out_samples[3 * d + 2] = samples[t * stride + 2];
    }
    free(cur);
    return 0;
}

/* D4: COO -> CSR (Recommender.py:64, SGDRecommender.py:265), stable inside a row. */
int orc_csr_from_coo(const int32_t *samples, int64_t n, int32_t stride, int64_t n_users,
                     int64_t *row_ptrs, int32_t *col_inds)
{
    memset(row_ptrs, 0, sizeof(int64_t) * (size_t)(n_users + 1));
    for (int64_t t = 0; t < n; ++t) ++row_ptrs[samples[t * stride] + 1];
    for (int64_t u = 0; u < n_users; ++u) row_ptrs[u + 1] += row_ptrs[u];
    int64_t *cur = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n_users + 1));
    memcpy(cur, row_ptrs, sizeof(int64_t) * (size_t)(n_users + 1));
    for (int64_t t = 0; t < n; ++t) col_inds[cur[samples[t * stride]]++] = samples[t * stride + 1];
    free(cur);
    return 0;
}

/* M3 bold driver, SGDRecommender.py:320-324. state = {alpha, previous_mse}. */
void orc_bold_driver(double train_rmse, double *alpha, double *previous)
{
    if (train_rmse > *previous && *previous != 0.0) *alpha *= 0.6; else *alpha *= 1.06;
    *previous = train_rmse - 0.0001;
}

#define T double
#define SUF f64
#include "mf_oracle_impl.h"
#undef T
#undef SUF
#define T float
#define SUF f32
#include "mf_oracle_impl.h"
#undef T
#undef SUF

/* Recommender.renumber_column (Recommender.py:18-25): all_values = unique(), sort, value -> position. */
static int cmp_i64(const void *a, const void *b)
{
    const int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}
int64_t orc_renumber_column(const int64_t *values, int64_t n, int32_t *out)
{
    if (n <= 0) return 0;
    int64_t *u = (int64_t *)malloc((size_t)n * sizeof(int64_t));
    memcpy(u, values, (size_t)n * sizeof(int64_t));
    qsort(u, (size_t)n, sizeof(int64_t), cmp_i64);
    int64_t m = 0;
    for (int64_t t = 0; t < n; ++t)
        if (t == 0 || u[t] != u[t - 1]) u[m++] = u[t];
    for (int64_t t = 0; t < n; ++t) {
        int64_t lo = 0, hi = m;
        while (lo < hi) { const int64_t mid = (lo + hi) / 2; if (u[mid] < values[t]) lo = mid + 1; else hi = mid; }
        out[t] = (int32_t)lo;
    }
    free(u);
    return m;
}

/* Top-N of ExplicitMF.predict (SGDRecommender.py:82-83): scores[u][i] = P[u].Q[i] (+ b_u[u] + b_i[i] + b), the N
 * best items per user, best first, ties to the lower item id (a stable argsort of the negated scores).
 * float32 factors, float64 accumulation. */
void orc_topn(const float *P, const float *Q, const float *b_u, const float *b_i, double b, const int32_t *users,
              int64_t n_batch, int64_t n_items, int32_t k, int32_t with_bias, int32_t topn, int32_t *out_items,
              double *out_scores)
{
    double *sc = (double *)malloc((size_t)n_items * sizeof(double));
    for (int64_t m = 0; m < n_batch; ++m) {
        const float *p = P + (int64_t)users[m] * k;
        for (int64_t i = 0; i < n_items; ++i) {
            double acc = 0.0;
            for (int f = 0; f < k; ++f) acc += (double)p[f] * (double)Q[i * k + f];
            sc[i] = with_bias ? acc + (double)b_u[users[m]] + (double)b_i[i] + b : acc;
        }
        for (int j = 0; j < topn; ++j) {   /* selection: N passes, first maximum wins (lowest id on ties) */
            int64_t best = -1;
            for (int64_t i = 0; i < n_items; ++i)
                if (sc[i] == sc[i] && (best < 0 || sc[i] > sc[best])) best = i;
            out_items[m * topn + j] = (int32_t)best;
            out_scores[m * topn + j] = best >= 0 ? sc[best] : 0.0;
            if (best >= 0) sc[best] = 0.0 / 0.0;   /* taken */
        }
    }
    free(sc);
}
