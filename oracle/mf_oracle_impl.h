/* TEST INFRASTRUCTURE - typed body of the CPU oracle; included twice by mf_oracle.c
 * with T = double / float and SUF = f64 / f32.  See mf_oracle.c for the contract. */

#define CAT_(a, b) a##_##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUF)

/* One sequential SGD pass.
 * variant 0 = numba_methods.sgd (numba_methods.py:11-23) / block-local sgd
 *             (RayBlockScheduler.py:127-143, with u_off/i_off != 0);
 * variant 1 = numba_methods.sgd2 (numba_methods.py:31-55).
 * implicit 0 = none (y ignored), 1 = frozen (y read, never written), 2 = full.
 * P,Q,b_u,b_i are block slices when u_off/i_off != 0 (rows user-u_off, item-i_off);
 * y, row_ptrs, col_inds are always global (RayBlockScheduler.py:131-133,143).
 * z (optional, frozen only): precomputed ys/R rows indexed by GLOBAL user. */
int FN(orc_sgd)(T *P, T *Q, T *b_u, T *b_i, T b, T *y, const T *z,
                const int32_t *samples, int64_t n, int32_t stride,
                const int64_t *row_ptrs, const int32_t *col_inds,
                int32_t k, int32_t u_off, int32_t i_off,
                T alpha, T beta1, T beta2, int32_t variant, int32_t implicit)
{
    T *ys = (T *)calloc((size_t)k, sizeof(T));     /* holds y_sum / R_u */
    T *pe = (T *)calloc((size_t)k, sizeof(T));
    if (!ys || !pe) { free(ys); free(pe); return -1; }
    for (int64_t t = 0; t < n; ++t) {
        const int32_t ug = samples[t * stride + 0], ig = samples[t * stride + 1];
        const T r = (T)samples[t * stride + 2];
        T *p = P + (int64_t)(ug - u_off) * k, *q = Q + (int64_t)(ig - i_off) * k;
        T *bu = b_u + (ug - u_off), *bi = b_i + (ig - i_off);
        int64_t lo = 0, hi = 0;
        T R = 1;
        if (implicit) {
            if (implicit == 1 && z) {
                for (int f = 0; f < k; ++f) ys[f] = z[(int64_t)ug * k + f];
            } else {
                lo = row_ptrs[ug]; hi = row_ptrs[ug + 1];
                R = (T)sqrt((double)(hi - lo)) + (variant == 1 ? (T)1 : (T)0);
                for (int f = 0; f < k; ++f) ys[f] = 0;
                for (int64_t j = lo; j < hi; ++j) {
                    const T *yr = y + (int64_t)col_inds[j] * k;
                    for (int f = 0; f < k; ++f) ys[f] += yr[f];
                }
                /* sqrt(0): the reference divides by zero (numba_methods.py:15); the build
                 * defines the implicit term of a user without a CSR row as 0. */
                for (int f = 0; f < k; ++f) ys[f] = (hi > lo || variant == 1) ? ys[f] / R : (T)0;
            }
        } else {
            for (int f = 0; f < k; ++f) ys[f] = 0;
        }
        T pred = 0;
        for (int f = 0; f < k; ++f) { pe[f] = p[f] + ys[f]; pred += q[f] * pe[f]; }
        pred += *bu + *bi + b;
        const T e = r - pred;
        *bu += alpha * (e - beta1 * *bu);
        *bi += alpha * (e - beta1 * *bi);
        if (variant == 0) {
            for (int f = 0; f < k; ++f) p[f] += alpha * (e * q[f] - beta2 * p[f]);
            for (int f = 0; f < k; ++f) q[f] += alpha * (e * (p[f] + ys[f]) - beta2 * q[f]);
            if (implicit == 2) {
                const T eR = e / R;
                for (int64_t j = lo; j < hi; ++j) {
                    T *yr = y + (int64_t)col_inds[j] * k;
                    for (int f = 0; f < k; ++f) yr[f] += alpha * ((T)-1.0 * beta2 * yr[f] + eR * q[f]);
                }
            }
        } else {
            for (int f = 0; f < k; ++f) {
                const T puf = p[f], qif = q[f];
                p[f] += alpha * (e * qif - beta2 * puf);
                q[f] += alpha * (e * (puf + ys[f]) - beta2 * qif);
                if (implicit == 2) {
                    const T eRq = e / R * qif;
                    for (int64_t j = lo; j < hi; ++j) {
                        T *yr = y + (int64_t)col_inds[j] * k;
                        yr[f] += alpha * (eRq - beta2 * yr[f]);
                    }
                }
            }
        }
    }
    free(ys); free(pe);
    return 0;
}

/* One half-step of alternating least squares (ALSRecommender.py:118-155, als_step): for every row r of
 * the CSR (users when type == 'user', items - on the transposed CSR - when type == 'item') solve
 *     (F_r^T F_r + lambda * (n_r + 1) * I) x = F_r^T v_r          (ALSRecommender.py:135-139 / :148-152)
 * with F_r = fixed[col_inds of row r], v_r the row's values, n_r its length, and store x in latent[r].
 * The reference calls numpy.linalg.solve (LU); the matrix is symmetric positive definite for lambda > 0,
 * so this restatement factors it as L D L^T without pivoting.  A row without entries gives x = 0 (the
 * reference solves lambda * I x = 0).  `fixed` is read only (the reference copies it first, :127). */
int FN(orc_als_step)(T *latent, const T *fixed, const int64_t *row_ptrs, const int32_t *col_inds, const T *vals,
                     int64_t n_rows, int32_t k, T lambda)
{
    T *A = (T *)malloc(sizeof(T) * (size_t)k * k), *bv = (T *)malloc(sizeof(T) * (size_t)k);
    if (!A || !bv) { free(A); free(bv); return -1; }
    for (int64_t r = 0; r < n_rows; ++r) {
        const int64_t lo = row_ptrs[r], hi = row_ptrs[r + 1];
        for (int i = 0; i < k * k; ++i) A[i] = 0;
        for (int i = 0; i < k; ++i) bv[i] = 0;
        for (int64_t j = lo; j < hi; ++j) {
            const T *f = fixed + (int64_t)col_inds[j] * k;
            for (int a = 0; a < k; ++a) {
                for (int c = 0; c <= a; ++c) A[a * k + c] += f[a] * f[c];
                bv[a] += vals[j] * f[a];
            }
        }
        for (int a = 0; a < k; ++a) A[a * k + a] += lambda * (T)(hi - lo + 1);
        /* L D L^T in place (lower triangle), forward elimination of b alongside */
        for (int j = 0; j < k; ++j) {
            const T d = A[j * k + j];
            if (!(d > 0)) { free(A); free(bv); return -3; }
            for (int i = j + 1; i < k; ++i) {
                const T lij = A[i * k + j] / d;
                for (int c = j + 1; c <= i; ++c) A[i * k + c] -= lij * A[c * k + j];   /* column j still unscaled */
                bv[i] -= lij * bv[j];
            }
            for (int i = j + 1; i < k; ++i) A[i * k + j] /= d;
        }
        for (int j = k - 1; j >= 0; --j) {
            T x = bv[j] / A[j * k + j];
            for (int i = j + 1; i < k; ++i) x -= A[i * k + j] * latent[r * (int64_t)k + i];
            latent[r * (int64_t)k + j] = x;
        }
    }
    free(A); free(bv);
    return 0;
}

/* Exact RMSE of the numba_methods.py:94-100 predictor over ALL rows (mse2 with the
 * broken exact=True path repaired).  Accumulates in double. Global ids. */
int FN(orc_rmse)(const T *P, const T *Q, const T *b_u, const T *b_i, T b, const T *y, const T *z,
                 const int32_t *samples, int64_t n, int32_t stride,
                 const int64_t *row_ptrs, const int32_t *col_inds, int32_t k, int32_t implicit,
                 double *out_rmse)
{
    double se = 0;
    T *ys = (T *)calloc((size_t)k, sizeof(T));
    if (!ys) return -1;
    for (int64_t t = 0; t < n; ++t) {
        const int32_t u = samples[t * stride], i = samples[t * stride + 1];
        const T r = (T)samples[t * stride + 2];
        for (int f = 0; f < k; ++f) ys[f] = 0;
        if (implicit) {
            if (z) {
                for (int f = 0; f < k; ++f) ys[f] = z[(int64_t)u * k + f];
            } else {
                const int64_t lo = row_ptrs[u], hi = row_ptrs[u + 1];
                const T R = (T)sqrt((double)(hi - lo));
                for (int64_t j = lo; j < hi; ++j)
                    for (int f = 0; f < k; ++f) ys[f] += y[(int64_t)col_inds[j] * k + f];
                for (int f = 0; f < k; ++f) ys[f] = hi > lo ? ys[f] / R : (T)0;
            }
        }
        T pred = 0;
        for (int f = 0; f < k; ++f) pred += Q[(int64_t)i * k + f] * (P[(int64_t)u * k + f] + ys[f]);
        pred += b + b_u[u] + b_i[i];
        se += (double)(r - pred) * (double)(r - pred);
    }
    free(ys);
    *out_rmse = n ? sqrt(se / (double)n) : 0.0;
    return 0;
}

/* z[u] = (sum_{j in N(u)} y[j]) / sqrt(|N(u)|)  - the frozen implicit term. */
int FN(orc_implicit_z)(const T *y, const int64_t *row_ptrs, const int32_t *col_inds,
                       int64_t n_users, int32_t k, T *z)
{
    for (int64_t u = 0; u < n_users; ++u) {
        const int64_t lo = row_ptrs[u], hi = row_ptrs[u + 1];
        T *zr = z + u * k;
        for (int f = 0; f < k; ++f) zr[f] = 0;
        for (int64_t j = lo; j < hi; ++j)
            for (int f = 0; f < k; ++f) zr[f] += y[(int64_t)col_inds[j] * k + f];
        const T R = (T)sqrt((double)(hi - lo));
        for (int f = 0; f < k; ++f) zr[f] = hi > lo ? zr[f] / R : (T)0;
    }
    return 0;
}

/* ------------------------------------------------------------------ threaded FPSGD (CPU baseline)
 * RayBlockScheduler.py:148-181 (consumer loop) + :35-107 (scheduler) + :118-145 (kernel),
 * with threads instead of Ray tasks and a mutex instead of the scheduler actor.  `order_out`
 * (optional, [iters*W*W][2]) receives the blocks in the order the scheduler issued them. */
typedef struct {
    orc_sched *s;
    pthread_mutex_t *mu;
    T *P, *Q, *b_u, *b_i, *y; const T *z; T b;
    const int32_t *blocks; const int64_t *block_ptr;
    const int64_t *row_ptrs; const int32_t *col_inds;
    const int32_t *u_bins, *i_bins;
    int32_t k, implicit, W;
    T alpha, beta1, beta2;
    int32_t *order_out; int64_t *order_n;
} FN(fpsgd_ctx);

static void *FN(fpsgd_worker)(void *arg)
{
    FN(fpsgd_ctx) *c = (FN(fpsgd_ctx) *)arg;
    int32_t r = -1, col = -1, have = 0;
    for (;;) {
        int32_t nr, nc;
        pthread_mutex_lock(c->mu);
        const int got = orc_sched_get_next(c->s, have, r, col, &nr, &nc);
        if (got && c->order_out) {
            c->order_out[2 * *c->order_n] = nr; c->order_out[2 * *c->order_n + 1] = nc; ++*c->order_n;
        }
        pthread_mutex_unlock(c->mu);
        if (!got) return NULL;
        r = nr; col = nc; have = 1;
        const int64_t lo = c->block_ptr[(int64_t)r * c->W + col], hi = c->block_ptr[(int64_t)r * c->W + col + 1];
        /* global-id rows with zero offsets == block-local rows on slices (same memory) */
        FN(orc_sgd)(c->P, c->Q, c->b_u, c->b_i, c->b, c->y, c->z, c->blocks + 3 * lo, hi - lo, 3,
                    c->row_ptrs, c->col_inds, c->k, 0, 0, c->alpha, c->beta1, c->beta2, 0, c->implicit);
    }
}

int FN(orc_fpsgd)(T *P, T *Q, T *b_u, T *b_i, T b, T *y, const T *z,
                  const int32_t *blocks, const int64_t *block_ptr, int32_t W,
                  const int64_t *row_ptrs, const int32_t *col_inds, int32_t k,
                  T alpha, T beta1, T beta2, int32_t implicit,
                  int32_t iters, int32_t n_threads, int32_t deterministic, uint64_t seed,
                  int32_t *order_out)
{
    orc_sched *s = orc_sched_new(W, iters, deterministic, seed);
    if (!s) return -1;
    pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
    int64_t order_n = 0;
    FN(fpsgd_ctx) c = { s, &mu, P, Q, b_u, b_i, y, z, b, blocks, block_ptr, row_ptrs, col_inds, NULL, NULL,
                        k, implicit, W, alpha, beta1, beta2, order_out, &order_n };
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
    for (int i = 0; i < n_threads; ++i) pthread_create(&th[i], NULL, FN(fpsgd_worker), &c);
    for (int i = 0; i < n_threads; ++i) pthread_join(th[i], NULL);
    free(th);
    const int done = orc_sched_all_done(s);
    orc_sched_free(s);
    return done ? 0 : -2;
}

#undef FN
#undef CAT
#undef CAT_
