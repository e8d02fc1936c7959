/* mfsgd.h - C-ABI of libmfsgd.so: the MI355X (gfx950) blocked-SGD matrix-factorization path.
 *
 * The reference (dtrager02/Recommender) is pure Python and has no FFI layer; its boundary for this
 * path is the set of Python callables listed in SURVEY.md section 8(b).  Each entry point below
 * names the reference callable it replaces (paths under /root/reference).  The Python mirror of
 * those callables (recommender_amd/ *.py) binds exactly these symbols through ctypes;
 * INTEGRATION.md shows the stub a maintainer of the reference would add.
 *
 * Conventions
 *  - every function returns 0 on success, <0 on error; mfsgd_last_error() (thread-local) says why;
 *  - every array pointer is a DEVICE pointer (hipMalloc / torch.Tensor.data_ptr()) unless the
 *    parameter name starts with h_; the library never frees or reallocates caller arrays;
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream); calls are asynchronous on
 *    that stream unless the name ends in _sync or the comment says it synchronises;
 *  - samples are int32 [n][stride] row-major, columns 0,1,2 = user,item,rating (Recommender.py:54);
 *  - T is float (f32) or double (f64); factor matrices are row-major [rows][k];
 *  - implicit: 0 = none (biased MF), 1 = frozen (y read-only through z = sum(y[N(u)])/sqrt|N(u)|),
 *              2 = full (SVD++: y rewritten for all of N(u) on every rating, numba_methods.py:23);
 *  - variant:  0 = numba_methods.sgd order (numba_methods.py:11-23), 1 = sgd2 order (:31-55).
 *  There is no CPU fallback: every entry point fails with an error when no gfx950 device is present.
 */
#ifndef MFSGD_H
#define MFSGD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MFSGD_OK 0
#define MFSGD_ERR_ARG (-1)
#define MFSGD_ERR_HIP (-2)
#define MFSGD_ERR_UNSUPPORTED (-3)
#define MFSGD_ERR_GRID (-4)

const char *mfsgd_last_error(void);
int mfsgd_version(void);
/* number of visible devices (>=1) or <0; checks that the current device is gfx950 */
int mfsgd_device_check(void);

/* ---- update kernels: replace numba_methods.sgd / sgd2 (numba_methods.py:7-57) and the block-local
 * sgd (RayBlockScheduler.py:118-145).  One exact-order sequential pass over `samples`:
 * results equal the reference's loop order rating by rating.  u_off/i_off are the block offsets of
 * the block-local form (P,Q,b_u,b_i are then the slices; y,row_ptrs,col_inds stay global); 0,0 for
 * the global form.  y may be NULL iff implicit==0; z may be NULL (then the implicit term is gathered
 * from y through row_ptrs/col_inds on every rating, as the reference does). */
int mfsgd_hip_sgd_f32(float *P, float *Q, float *b_u, float *b_i, float b, float *y, const float *z,
                      const int32_t *samples, int64_t n, int32_t stride,
                      const int64_t *row_ptrs, const int32_t *col_inds,
                      int32_t k, int32_t u_off, int32_t i_off,
                      float alpha, float beta1, float beta2, int32_t variant, int32_t implicit, void *stream);
int mfsgd_hip_sgd_f64(double *P, double *Q, double *b_u, double *b_i, double b, double *y, const double *z,
                      const int32_t *samples, int64_t n, int32_t stride,
                      const int64_t *row_ptrs, const int32_t *col_inds,
                      int32_t k, int32_t u_off, int32_t i_off,
                      double alpha, double beta1, double beta2, int32_t variant, int32_t implicit, void *stream);

/* ---- RMSE: replaces numba_methods.mse2 / mse (numba_methods.py:59-103) with the exact (all rows)
 * evaluation of the :94-100 predictor.  Writes the SUM of squared errors (double) to *d_sse; the
 * caller takes sqrt(sse/n).  d_partials: device scratch of mfsgd_rmse_partials(n) doubles.
 * Deterministic (fixed reduction order). */
int64_t mfsgd_rmse_partials(int64_t n);
int mfsgd_hip_rmse_f32(const float *P, const float *Q, const float *b_u, const float *b_i, float b,
                       const float *y, const float *z, const int32_t *samples, int64_t n, int32_t stride,
                       const int64_t *row_ptrs, const int32_t *col_inds, int32_t k, int32_t implicit,
                       double *d_partials, double *d_sse, void *stream);
int mfsgd_hip_rmse_f64(const double *P, const double *Q, const double *b_u, const double *b_i, double b,
                       const double *y, const double *z, const int32_t *samples, int64_t n, int32_t stride,
                       const int64_t *row_ptrs, const int32_t *col_inds, int32_t k, int32_t implicit,
                       double *d_partials, double *d_sse, void *stream);

/* Allocation-free variant of mfsgd_hip_rmse_f32.  It has NO counterpart in the reference; it exists for callers
 * that cannot call hipMalloc (code under HIP graph capture, tight serving loops, harnesses that forbid it):
 * the per-block partial sums that mfsgd_hip_rmse_f32 takes as the caller's d_partials workspace live in a
 * file-scope __device__ array of the library instead (16 KB per GPU, held until the process ends).  The call
 * itself performs no hipMalloc / hipFree and calls no allocating routine; inputs and d_sse are still the
 * caller's.  Same kernels, same fixed reduction order: *d_sse is bit-identical to mfsgd_hip_rmse_f32's.
 * Limits: n <= MFSGD_RMSE_NOALLOC_MAX_N ratings per call, k <= MFSGD_RMSE_NOALLOC_MAX_K; beyond them the call
 * fails with MFSGD_ERR_ARG and touches nothing.  Every call on one GPU shares that GPU's array: calls that
 * overlap on one GPU (two streams, two threads) race - serialise them.  Asynchronous on `stream`.
 * Set-up: the array's address is resolved once per GPU (hipGetSymbolAddress, which may load the code object) by
 * mfsgd_device_check() called with that GPU current - or, failing that, by the first call of this entry on the GPU,
 * which is then NOT capture-safe: call mfsgd_device_check() once per GPU outside any capture.  The scratch used is
 * the CURRENT device's: P and d_sse must live on it, else MFSGD_ERR_ARG. */
#define MFSGD_RMSE_NOALLOC_MAX_N ((int64_t)2147483647)
#define MFSGD_RMSE_NOALLOC_MAX_K 4096
int mfsgd_hip_rmse_noalloc_f32(const float *P, const float *Q, const float *b_u, const float *b_i, float b,
                               const float *y, const float *z, const int32_t *samples, int64_t n, int32_t stride,
                               const int64_t *row_ptrs, const int32_t *col_inds, int32_t k, int32_t implicit,
                               double *d_sse, void *stream);

/* z[u] = sum_{j in N(u)} y[j] / (sqrt|N(u)| + plus_one)  (numba_methods.py:14-16, :36 for plus_one=1) */
int mfsgd_hip_implicit_z_f32(const float *y, const int64_t *row_ptrs, const int32_t *col_inds,
                             int64_t n_users, int32_t k, int32_t plus_one, float *z, void *stream);
int mfsgd_hip_implicit_z_f64(const double *y, const int64_t *row_ptrs, const int32_t *col_inds,
                             int64_t n_users, int32_t k, int32_t plus_one, double *z, void *stream);

/* ---- data prep (integer work, bit-exact with the reference).  These calls allocate and free
 * device temporaries and synchronise the stream before returning.
 * remap: ExplicitMF.random_renumber_samples id remap (SGDRecommender.py:251-256), in place. */
int mfsgd_hip_remap(int32_t *samples, int64_t n, int32_t stride, const int32_t *row_perm,
                    const int32_t *col_perm, int64_t n_users, int64_t n_items, void *stream);
/* filter_test: drop test rows whose user or item is absent from train (SGDRecommender.py:259-263).
 * Stable; writes kept rows to out[][stride] and their count to *h_count (host). */
int mfsgd_hip_filter_test(const int32_t *test, int64_t nt, int32_t stride, const int32_t *train, int64_t n,
                          int32_t tstride, int64_t n_users, int64_t n_items, int32_t *out, int64_t *h_count,
                          void *stream);
/* bucket: ExplicitMF.generate_indpendent_samples_new (SGDRecommender.py:187-212): block (r,c) = rows
 * with user/ubp==r and item/ibp==c in original relative order; blocks concatenated row-major into
 * out_samples[n][3]; out_block_ptr[Pu*Pi+1].  Fails if a row falls outside the Pu x Pi grid. */
int mfsgd_hip_bucket(const int32_t *samples, int64_t n, int32_t stride, int32_t ubp, int32_t ibp,
                     int32_t Pu, int32_t Pi, int32_t *out_samples, int64_t *out_block_ptr, void *stream);
/* csr: csr.CSR.from_coo as used by Recommender.py:64 / SGDRecommender.py:265,301-302 (rowptrs, colinds),
 * input order kept inside a row. */
int mfsgd_hip_csr_from_coo(const int32_t *samples, int64_t n, int32_t stride, int64_t n_users,
                           int64_t *row_ptrs, int32_t *col_inds, void *stream);

/* renumber_column: Recommender.renumber_column (Recommender.py:18-25), the ingestion step that turns raw ids
 * (any int64) into dense ones: out_ids[t] = rank of values[t] among the sorted distinct values; the number of
 * distinct values goes to *h_n_unique (host).  Synchronises. */
int mfsgd_hip_renumber_column(const int64_t *values, int64_t n, int32_t *out_ids, int64_t *h_n_unique, void *stream);

/* ---- batched scoring + top-N: the dense contraction of ExplicitMF.predict (SGDRecommender.py:82-83,
 * user_vecs[row,:].dot(item_vecs[col,:].T)) fused with the selection of the `topn` best items per user, so the
 * users x items score matrix is never stored.  score(u,i) = P[u].Q[i], plus b_u[u] + b_i[i] + b when with_bias
 * (the training predictor numba_methods.py:17 without the implicit term).  users[n_batch] are row ids of P;
 * out_items / out_scores are [n_batch][topn], best first, ties to the lower item id; topn in {4, 8, 16};
 * 32 <= k <= 256 (the user tiles and item chunks of a workgroup share the 160 KB of LDS: four user tiles per
 * workgroup up to k = 200, two above - BASELINE config 5 is k = 256).  Matrix-core (MFMA
 * fp32) kernel; asynchronous on `stream`. */
int mfsgd_hip_topn_f32(const float *P, const float *Q, const float *b_u, const float *b_i, float b,
                       const int32_t *users, int64_t n_batch, int64_t n_items, int32_t k, int32_t with_bias,
                       int32_t topn, int32_t *out_items, float *out_scores, void *stream);

/* ---- the block grid + on-GPU scheduler + epoch kernel: replaces the scheduler actor
 * (RayBlockScheduler.py:14-116), the worker loop (:148-181), make_subsample/update_params
 * (SGDRecommender.py:88-121) and the per-block kernel calls.  A grid holds the ratings bucketed into
 * Gr x Gc blocks (equal-width bins as SGDRecommender.py:191-204) as block-local runs of same-user
 * ratings with per-item tickets, so that a workgroup can walk one block with many waves and still
 * produce exactly the block's sequential result. */
typedef struct mfsgd_grid mfsgd_grid;
#define MFSGD_MAX_HOT 8   /* hot items (hot segments per block) an item stripe may have */

typedef struct {
    int64_t n, n_runs, n_users, n_items;
    int32_t Gr, Gc, ubp, ibp;
    int32_t order;        /* 0 = reference order inside a block, 1 = user-sorted inside a block (hot segment first) */
    int32_t user_unique;  /* 1 if no user has two runs in one segment of a block (always for order 1) */
    int64_t max_block, max_col_nnz, max_row_nnz;
    int64_t n_hot;         /* ratings stored in hot segments (order 1; 0 otherwise) */
    int32_t n_hot_stripes; /* item stripes with at least one hot item */
    int32_t n_hot_items;   /* hot items over all stripes (<= MFSGD_MAX_HOT per stripe) */
} mfsgd_grid_info;

/* order 0: rows keep the reference's relative order inside a block (deterministic parity mode);
 * order 1 (throughput mode): inside a block first the ratings of the stripe's hot item(s), one run per user (the hot
 *          segments, see mfsgd_grid_set_hot_buffers and mfsgd_grid_create_opt), then all other ratings, one run per
 *          user, inside a run the most rated item first (an item's ratings form a serial chain, and a rating that
 *          sits behind others in its run makes the whole chain wait; measured against input order and against the
 *          opposite order: 1% and 1.5% of an epoch), ties in input order: first the runs of the users with NO rating in a hot segment of this block, by
 *          user, then the runs of the users with one, by user (the pipeline that walks the hot segments holds only the
 *          latter back; the others keep the waves busy from the block's first cycle).  A block without hot ratings
 *          is simply sorted by user.  If a (user, hot item) pair occurs twice the grid is built without hot segments. */
int mfsgd_grid_create(const int32_t *samples, int64_t n, int32_t stride, int64_t n_users, int64_t n_items,
                      int32_t ubp, int32_t ibp, int32_t order, void *stream, mfsgd_grid **out);
/* The same with the hot segments at the caller's choice: up to hot_items (0 .. MFSGD_MAX_HOT) of a stripe's most rated
 * items become its hot items - item s's ratings form hot segment s in front of every block of the stripe, one
 * single-rating run per user, by user - each with at least 256 ratings and, on average, at least hot_min_per_block
 * ratings per block of the stripe (degree >= hot_min_per_block * Gr); hot_items = 0: no hot segments at all (every
 * block of an order-1 grid sorted by user only).  A rating on a hot path (mfsgd_grid_set_hot_mode: the hot-item walker,
 * 0.09 us per link of the item's chain at k = 128; the matrix-core pipeline, 0.12) costs less than half of one that goes
 * through the item tickets (0.22 us per link); hot segments that go through the tickets cost time (a block then starts
 * with a stretch in which every wave waits for the same item).  So a caller that knows a hot path will run passes
 * (1, 64) - ExplicitMF.build_grid does - and one that knows it will not passes (0, 0).  The walker runs in every
 * straight-line build: k = 64, 128 or 256 (pad the device rows with zero columns otherwise), fp32 or fp64, with or
 * without the frozen implicit term, a user-unique grid (order 1 always is) whose item stripes fit a CU's LDS
 * (mfsgd_lds_bytes), sgd's update order (variant 0), one hot item per stripe.  mfsgd_grid_create is one hot item per
 * stripe of any length, if it carries >= 1/16 of the stripe's ratings - the default for those callers; every other
 * caller should say (0, 0) here: its hot segments would go through the tickets, which is slower than no hot segments
 * (ML-25M shape: 13.0 against 11.8 ms per epoch). */
int mfsgd_grid_create_opt(const int32_t *samples, int64_t n, int32_t stride, int64_t n_users, int64_t n_items,
                          int32_t ubp, int32_t ibp, int32_t order, int32_t hot_items, int32_t hot_min_per_block,
                          void *stream, mfsgd_grid **out);
int mfsgd_grid_get_info(const mfsgd_grid *g, mfsgd_grid_info *info);
void mfsgd_grid_destroy(mfsgd_grid *g);
/* Copies the grid's arrays into caller buffers (any may be NULL): blk_run_ptr[Gr*Gc+1], run_start[n_runs+1],
 * run_user[n_runs], item[n], val[n], tick[n].  For checking the bucketing bit for bit. */
int mfsgd_grid_export(const mfsgd_grid *g, int32_t *blk_run_ptr, int32_t *run_start, int32_t *run_user,
                      int32_t *item, float *val, int32_t *tick, void *stream);
/* Hot segments (order 1 grids).  One item's updates form a serial chain - every rating needs the item row its
 * predecessor wrote - and on Netflix-shaped data the most rated items carry 10^5 ratings each, so that chain, not the
 * memory system, bounds their stripes.  A stripe's most rated item becomes its HOT item when it carries >= 1/16 of
 * the stripe's ratings (and >= 256); every block of the stripe then stores that item's ratings first, one single-
 * rating run per user, in user order.  Nothing is relaxed: the block is still walked sequentially in stored order,
 * and the free-block kernel walks a hot segment on a hot path (mfsgd_grid_set_hot_mode below) - the same chain of
 * updates on two waves of its own, or restated so that the serial part of a link is a few scalar operations; every
 * other kernel walks it through the item tickets.
 * The reference has no counterpart (its blocks are walked by one thread, RayBlockScheduler.py:127-143).
 * mfsgd_grid_set_hot_buffers: how many LDS row buffers (0..8, default 8 = as many as fit; each 32 x (k+4) floats, one per
 * pipeline wave) the matrix-core pipeline may take behind the stripe; 0 sends hot segments through the tickets whatever
 * the hot mode (same results up to rounding - the tests replay all of them on the oracle's plain sequential kernel).
 * Takes effect from the next epoch call. */
int mfsgd_grid_set_hot_buffers(mfsgd_grid *g, int32_t n);
/* How the free-block kernel walks hot segments (takes effect from the next epoch call):
 *   2 (default)  the hot-item WALKER for stripes with one hot item (csrc/epoch.hip hot_walk): a chain wave keeps the
 *                item's row in registers for the whole segment and streams the users' rows through a ring of register
 *                slots, requested straight from HBM a ring ahead - no ticket, no LDS hand-off, a link is ~200 cycles of
 *                dependent issue, the sequential arithmetic itself; an apply wave a chunk of 64 links behind forms and
 *                stores the users' rows and moves the frontier the block's other runs wait behind.  Every straight-line
 *                build: fp32 / fp64, k = 64, 128, 256, with or without the frozen implicit term, no LDS beyond the
 *                stripe.  Stripes with several hot items take the matrix-core pipeline (mode 1) where it can run, else
 *                the tickets;
 *   1            the matrix-core pipeline only (phase_a_gram; fp32, k = 64 / 128 / 256, no implicit term, row buffers
 *                in LDS - mfsgd_grid_set_hot_buffers);
 *   0            every rating through the item tickets (as mfsgd_grid_set_hot_buffers(g, 0)).
 * All three are the same sequential pass over the block's stored order. */
int mfsgd_grid_set_hot_mode(mfsgd_grid *g, int32_t mode);
/* schedule 2 of mfsgd_hip_epoch_*: of the first `n` free user stripes a workgroup finds (1..64; default 3) it takes the one
 * its item stripe has visited least often in this launch.  1 = any free stripe (passes blur most), 64 = practically the
 * least visited free stripe (passes nearly in step).  The counterpart of the reference's "minimal update count among the
 * free rows x free columns" (RayBlockScheduler.py:50-58); takes effect from the next epoch call. */
int mfsgd_grid_set_pick_width(mfsgd_grid *g, int32_t n);
/* Host-side helper for choosing a grid: bytes of LDS a free-block workgroup needs for item stripes of `ibp`
 * items, `Gr` user stripes, k factors of `elem_size` bytes (4 or 8) plus `hot_bufs` pipeline row buffers (fp32
 * only); -1 on bad arguments.  A CU has 160 KB.  Touches no device.  No counterpart in the reference (its grid is
 * (n_threads+1)^2, SGDRecommender.py:189-192). */
int64_t mfsgd_lds_bytes(int32_t ibp, int32_t Gr, int32_t k, int32_t elem_size, int32_t hot_bufs);
/* h_col_hot[Gc][MFSGD_MAX_HOT] (HOST): for every item stripe the global ids of its hot items, most rated first;
 * -1 = none.  Synchronises. */
int mfsgd_grid_col_hot(const mfsgd_grid *g, int32_t *h_col_hot, void *stream);
/* Synchronises and reports whether an epoch kernel ran out of patience on a wait (every spin in the
 * kernels is bounded); clears the flag. */
int mfsgd_grid_status(mfsgd_grid *g, void *stream);
/* Profiling aid: for every item stripe c, h_out[3c..3c+2] (HOST) = {ratings in the stripe, start, end} of its
 * tenure in the most recent free-block epoch, in ticks of the 100 MHz device realtime clock.  Synchronises. */
int mfsgd_grid_col_times(const mfsgd_grid *g, int64_t *h_out, void *stream);

/* schedule 0 = fixed: equals `iters` passes over the blocks in row-major order, each block walked in
 *               stored order (the deterministic fixed-block-order mode);
 * schedule 1 = free-block: workgroups pull column stripes (heaviest first) and lock free row stripes
 *               on the GPU; every block is done exactly once per epoch (one launch per epoch), conflicting blocks
 *               never overlap;
 * schedule 2 = free-block with the passes OVERLAPPED as the reference's scheduler overlaps them: BlockScheduler runs its
 *               `iters` passes in one session and hands out "a block of minimal update count among the free rows x free
 *               columns" (RayBlockScheduler.py:50-58) - with n_threads + 1 stripes a worker sees one or two of each, so a
 *               fifth of the blocks get a visit more than a full pass late and others their next visit early
 *               (profiles/r3_pass_overlap.txt).  Here: all `iters` passes in ONE launch, every workgroup keeps its item
 *               stripe throughout and takes, of the first few free user stripes it finds (mfsgd_grid_set_pick_width), the one
 *               it has visited least; every block still gets exactly `iters` visits and conflicting blocks never
 *               overlap.  Needs a workgroup per non-empty item stripe (Gc <= CUs); other grids run it as schedule 1.
 * order_log (nullable, int32 [iters*Gr*Gc][2], device): blocks in the order they were issued.
 * n_workgroups / waves_per_wg: 0 = auto.  implicit in {0,1} (z required for 1). */
int mfsgd_hip_epoch_f32(mfsgd_grid *g, int32_t k, float *P, float *Q, float *b_u, float *b_i, float b, const float *z,
                        float alpha, float beta1, float beta2, int32_t variant, int32_t implicit,
                        int32_t iters, int32_t schedule, uint64_t seed, int32_t n_workgroups,
                        int32_t waves_per_wg, int32_t *order_log, void *stream);
int mfsgd_hip_epoch_f64(mfsgd_grid *g, int32_t k, double *P, double *Q, double *b_u, double *b_i, double b, const double *z,
                        double alpha, double beta1, double beta2, int32_t variant, int32_t implicit,
                        int32_t iters, int32_t schedule, uint64_t seed, int32_t n_workgroups,
                        int32_t waves_per_wg, int32_t *order_log, void *stream);

/* One half-step of alternating least squares: ExplicitMF.als_step (ALSRecommender.py:118-155).  For every row r
 * of the CSR (row_ptrs[n_rows+1], col_inds, vals - the users' CSR for type == 'user', the transposed one for
 * type == 'item', :131 / :142) solves (F^T F + lambda (n_r + 1) I) x = F^T v with F = fixed[col_inds of r]
 * (:133-139) and writes x to latent[r]; a row without entries gets zeros.  `fixed` is only read (the reference
 * copies it, :127).  `fixed` has n_cols rows, the CSR nnz entries: the kernel checks every row range and column
 * index against them and reports a CSR that does not fit instead of reading outside the arrays.  k <= 256 (the
 * reference has no limit; BASELINE config 5 runs k = 256): up to k = 128 the k x k normal matrix of a row lives in the
 * registers of one 256-thread workgroup (64 entries per thread at k = 128), above in those of a 1024-thread workgroup
 * with the factor's lower triangle packed in LDS (fp32 to k = 256, fp64 to k = 192) or in a slab of device memory the
 * call allocates per workgroup (fp64, 193 <= k <= 256); k > 256 is refused with MFSGD_ERR_ARG,
 * lambda > 0 (the reference's defaults 0.0 make its solve singular for short rows).  Synchronises the stream
 * before it returns (a broken CSR or a non-positive pivot is reported as an error). */
int mfsgd_hip_als_step_f32(float *latent, const float *fixed, const int64_t *row_ptrs, const int32_t *col_inds,
                           const float *vals, int64_t n_rows, int64_t n_cols, int64_t nnz, int32_t k, float lambda,
                           void *stream);
int mfsgd_hip_als_step_f64(double *latent, const double *fixed, const int64_t *row_ptrs, const int32_t *col_inds,
                           const double *vals, int64_t n_rows, int64_t n_cols, int64_t nnz, int32_t k, double lambda,
                           void *stream);

/* ---- multi-GPU DSGD ring, host side (SURVEY.md 8(b) `mfsgd_dsgd_*`, 8(e)).  No counterpart in the reference (it has no
 * multi-device path); the rule is its FPSGD invariant (RayBlockScheduler.py:50-62) one level up: users are cut into
 * `world` contiguous super-stripes and items into `world` contiguous shards, both on boundaries of the ubp x ibp block
 * grid; rank g owns the P / b_u rows of user_edges[g] .. user_edges[g+1] for good; an epoch is `world` sub-epochs, in
 * sub-epoch `step` rank g runs mfsgd_hip_epoch_* on the grid built from ITS users' ratings of shard (g + step) mod world
 * and then sends that shard's Q / b_i rows (item_edges[q] .. item_edges[q+1]) to rank g-1 and receives the next from rank
 * g+1 (RCCL send/recv; recommender_amd/dsgd.py over torch.distributed).  These calls touch no device. */
/* user_edges[world+1], item_edges[world+1] (HOST); Gr_out / Gc_out (may be NULL): the global grid */
int mfsgd_dsgd_plan(int64_t n_users, int64_t n_items, int32_t world, int32_t ubp, int32_t ibp, int64_t *user_edges,
                    int64_t *item_edges, int32_t *Gr_out, int32_t *Gc_out);
/* the shard `rank` holds in sub-epoch `step` (steps count on across epochs), and its two ring neighbours; any may be NULL */
int mfsgd_dsgd_schedule(int32_t world, int32_t rank, int32_t step, int32_t *shard, int32_t *send_to, int32_t *recv_from);
/* Block sizes for `world` ranks: the fewest stripes per shard whose throughput term stays below the latency term of a
 * sub-epoch (the most rated item's chain + one block transition per row stripe).
 * max_item_degree <= 0: assume a netflix-like head (0.16 % of the ratings); n_cus: 256 on MI355X. */
int mfsgd_dsgd_choose_grid(int64_t n_users, int64_t n_items, int64_t n_train, int32_t world, int32_t k, int32_t elem_size,
                           int64_t max_item_degree, int32_t n_cus, int32_t *ubp_out, int32_t *ibp_out);

#ifdef __cplusplus
}
#endif
#endif
