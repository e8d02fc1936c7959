// epoch.hip - the block grid, the on-GPU free-block scheduler and the epoch kernel.
//
// Replaces the scheduler actor (RayBlockScheduler.py:14-116), the worker loop (:148-181), the
// parameter-server slicing (SGDRecommender.py:88-121) and the per-block sgd calls (:118-145).
//
// Data layout in HBM (all built on the device by mfsgd_grid_create):
//   ratings sorted by block (row-major Gr x Gc, equal-width bins, SGDRecommender.py:191-204), inside a
//   block in the reference's relative order (order 0) or additionally by user (order 1), as SoA
//   item[n] (int32 global id), val[n] (f32), tick[n] (int32);
//   runs = maximal stretches of consecutive ratings of one user inside a block:
//   run_start[n_runs+1], run_user[n_runs]; blk_run_ptr[Gr*Gc+1] indexes runs by block.
//   tick[t] = number of earlier ratings of the SAME item in the same block.
//
// How one block is walked (process_block): a workgroup owns block (r,c), i.e. P rows of user stripe r and
// Q rows of item stripe c.  Waves take whole runs (one user each: P row and b_u live in registers for the
// run).  Before a wave touches item i it waits until the item's LDS counter equals the rating's ticket and
// bumps it afterwards, so every item sees its updates in stored order and every user's ratings are taken in
// stored order by one wave: the result is exactly the sequential pass over the block (updates that share
// neither user nor item commute), whatever the timing.  Q rows (and b_i) of the stripe are staged in LDS
// when they fit (STAGE), so a hot item's chain of updates runs at LDS latency.
//
// How blocks are scheduled:
//   schedule 0 (fixed): one launch per anti-diagonal of the grid; blocks on a diagonal share no row and no
//     column, and diagonal order preserves, for every row and every column, the row-major order of its
//     blocks - so the result equals `iters` sequential row-major passes (deterministic mode).
//   schedule 1 (free-block): persistent workgroups pull item stripes heaviest-first from a queue (the Q
//     stripe then stays in LDS for the whole tenure) and lock free user stripes with an agent-scope CAS;
//     a block is done exactly once per epoch and two blocks sharing a row or a column never overlap -
//     the FPSGD invariant of RayBlockScheduler.py:50-62 with "minimal update count" = "not yet done in this
//     epoch".  Row hand-off between workgroups: release/acquire at agent scope around the lock word.
#include "common.h"
#include <hipcub/hipcub.hpp>
#include <algorithm>
#include <numeric>
#include <type_traits>
#include <vector>

namespace mfprep {  // helpers defined in prep.hip
struct DevBuf {
    void *p = nullptr;
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
    ~DevBuf() { if (p) (void)hipFree(p); }
    template <typename U> U *as() { return (U *)p; }
};
}  // namespace mfprep
using mfprep::DevBuf;

constexpr int PA_MAX_BUFS_DEFAULT = 8;
constexpr int PICK_WIDTH_DEFAULT = 3, PICK_WIDTH_MAX = 64;   // 3: held-out RMSE closest to the reference scheduler's band on both full-size shapes (profiles/r3_rmse_parity.txt)
constexpr int MAXSEG = MFSGD_MAX_HOT;   // hot segments per block
// classes of a rating inside its block: hot segment 0 .. MAXSEG-1, then the rest - first the ratings of users WITHOUT a
// rating in a hot segment of this block (class MAXSEG: nothing to wait for while the pipeline walks the segments), then
// those of the users with one (class MAXSEG + 1: behind the pipeline's user frontier)
constexpr int NCLS = MAXSEG + 2;
struct mfsgd_grid {
    mfsgd_grid_info info{};
    int32_t *blk_run_ptr = nullptr, *run_start = nullptr, *run_user = nullptr, *item = nullptr, *tick = nullptr;
    float *val = nullptr;
    int32_t *col_order = nullptr;  // item stripes, heaviest first
    int32_t *col_hot = nullptr;      // [Gc][MAXSEG] the stripe's hot items (global ids, most rated first; -1: none): item s forms hot segment s of every block of the stripe
    int32_t *blk_hot_cnt = nullptr;  // [Gr*Gc] ratings in the block's hot segments (all of them together)
    int32_t *blk_seg_cnt = nullptr;  // [Gr*Gc][MAXSEG] length of each hot segment (ratings of that item, one per user)
    int32_t *blk_free_runs = nullptr;  // [Gr*Gc] runs of class MAXSEG: users of the block without a rating in a hot segment
    int32_t max_block_runs = 0;
    int32_t hot_bufs = PA_MAX_BUFS_DEFAULT;   // LDS row buffers the hot-item pipeline may take (0: ticket path only)
    int32_t pick_width = PICK_WIDTH_DEFAULT;  // schedule 2: free user stripes among which the least visited one is taken
    int32_t hot_mode = 2;                     // how the free-block kernel walks hot segments: 0 tickets, 1 matrix-core pipeline, 2 walker
    int32_t *row_lock = nullptr;   // [Gr]
    int32_t *counters = nullptr;   // [0] next column, [1] next log slot; padded to 16 words
    long long *col_clock = nullptr;  // [Gc][2] start/end of each column tenure (100 MHz realtime), last epoch
    int32_t n_cols_nonempty = 0;
    int device = 0;
    std::vector<int64_t> col_nnz;
};

namespace {

struct GridView {
    const int32_t *blk_run_ptr, *run_start, *run_user, *item, *tick;
    const float *val;
    const int32_t *col_order;
    int32_t Gr, Gc, ubp, ibp, user_unique, n_cols;
    int64_t n_users, n_items;
    int32_t sentinel_run;  // index of an empty run behind the last real one (its user row is a valid address)
    const int32_t *col_hot, *blk_hot_cnt, *blk_seg_cnt, *blk_free_runs;   // see mfsgd_grid
};

struct SchedView {
    int32_t *row_lock, *counters, *order_log;
    int32_t log_base;  // epoch * Gr * Gc
    uint32_t seed;
    int32_t *err;      // sticky error word: 1 = ticket wait ran out, 2 = row-lock search ran out, 4 = a wait of the hot-item pipeline
    long long *col_clock;
    int32_t passes;    // visits every block gets in THIS launch (schedule 2: > 1, no barrier between the passes)
    int32_t pick;      // schedule 2: the least visited of the first `pick` free-looking user stripes is taken
};

constexpr int SPIN_TICKET_MAX = 1 << 22;  // every spin is bounded: a stuck wave flags the run and moves on
constexpr int SPIN_LOCK_SWEEPS = 1 << 20;
constexpr int META_PAD = 64;   // words behind the last rating in item / tick / val: load_meta may read (and ignore) up to 63 past a run's end
constexpr int MAX_PASSES_PER_LAUNCH = 200;   // visit counts live in one byte per user stripe (row_done in k_epoch_free)
#ifndef TICKET_NAP
#define TICKET_NAP 3   // s_sleep units (64 clocks) a waiter sleeps per hand-off in front of it
#endif

// Hand-offs between the waves of a workgroup through LDS (item tickets, the pipeline's turn and frontier words): the payload
// (an item row, a bias) is written with plain LDS stores, then the ticket word with a relaxed workgroup-scope atomic; the
// next holder reads the ticket word and then the payload.
//  * MFSGD_LDS_FENCE = 1 (make fenced): a workgroup-scope release / acquire fence restricted to the LDS address space
//    between the two ("local": hipcc emits s_waitcnt lgkmcnt(0) only, the wave's HBM prefetches stay in flight).  The
//    hand-off is then well defined in the memory model - at the price of a second LDS round trip on every link of an
//    item's chain: netflix shape 23.9 -> 24.6 ms per epoch, through the tickets alone 37.0 -> 39.2, ML-25M 9.95 -> 10.4,
//    ML-1M 1.18 -> 1.29 (profiles/r3_lds_fence_ab.txt).
//  * MFSGD_LDS_FENCE = 0 (the shipped build): the INSTRUCTION order is pinned instead - an asm statement with a memory
//    clobber is a barrier for the compiler at every level, so the payload's ds_write instructions are emitted before the
//    ticket's and the ticket's ds_read before the payload's - and the LDS serves one wave's DS instructions in issue
//    order (the hardware property every ds_read-after-ds_write of a single wave relies on).  `make` runs
//    csrc/check_lds_order.py over the ISA and fails if a hand-off is not in that order.
#ifndef MFSGD_LDS_FENCE
#define MFSGD_LDS_FENCE 0
#endif
#if MFSGD_LDS_FENCE
#define LDS_ACQUIRE() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local")
#define LDS_RELEASE() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local")
#else
#define LDS_ACQUIRE()                                              \
    do {                                                           \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");     \
        asm volatile("; mf_lds_acquire" ::: "memory");             \
    } while (0)
#define LDS_RELEASE()                                              \
    do {                                                           \
        asm volatile("; mf_lds_release" ::: "memory");             \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     \
    } while (0)
#endif

// How a user stripe's rows pass from one workgroup to the next (row lock, k_epoch_free).
//  * The general build (FAST == 0: plain loads and stores of P / b_u) brackets the lock with agent-scope fences: release =
//    write back this XCD's L2 (buffer_wbl2), acquire = invalidate the CU's L1 and the L2's non-local lines (buffer_inv) -
//    the XCDs of an MI355X have an L2 each.  That is bulk cache maintenance on every block transition: 5-6 us per block
//    (netflix shape 21.2 -> 19.5 ms per epoch, ML-25M 8.9 -> 7.5 without it - timing experiment, stale rows, -DMFSGD_NOFENCE).
//  * The straight-line builds (FAST != 0) need no fence: every access to P and b_u there is itself coherent at agent
//    scope - loads as relaxed agent-scope atomic loads (row_get_agent: global_load ... sc1), stores with the sc1 bit
//    (untracked_store, row_put_untracked, the walker's vm_* helpers) - so a row is read from and written through to the
//    memory side of the L2s, never left in a cache another XCD cannot see.  What remains of release / acquire is ordering:
//    the holder's stores have completed (s_waitcnt vmcnt(0) before the closing barrier) before the lock word is cleared,
//    and the next holder's loads are issued after its compare-and-swap has returned.  P rows have no reuse inside a block
//    (a run loads its row once and stores it once), so nothing is lost by not caching them.
#ifdef MFSGD_NOFENCE   // timing experiment for the general build: no cache maintenance at the row hand-off (stale rows)
#define MF_AGENT_RELEASE() asm volatile("" ::: "memory")
#define MF_AGENT_ACQUIRE() asm volatile("" ::: "memory")
#else
#define MF_AGENT_RELEASE() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent")
#define MF_AGENT_ACQUIRE() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent")
#endif
inline unsigned nblk(int64_t n, int bs = 256) { return (unsigned)((n + bs - 1) / bs); }

// ------------------------------------------------------------------------------------------ grid build
__global__ void k_keys(const int32_t *__restrict__ s, int64_t n, int32_t stride, int32_t ubp, int32_t ibp, int64_t n_users,
                       int64_t n_items, int32_t Gc, uint32_t *bkey, uint32_t *ukey, int32_t *idx, int32_t *err)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int32_t u = s[t * stride], i = s[t * stride + 1];
    const int32_t r = u / ubp, c = i / ibp;
    // ids are checked against the factor matrices, not against the padded last stripe: an id in
    // [n_users, Gr * ubp) would index P / b_u (and the per-item degree table) past their ends
    if (u < 0 || i < 0 || u >= n_users || i >= n_items) { *err = 1; bkey[t] = 0; ukey[t] = 0; idx[t] = (int32_t)t; return; }
    bkey[t] = (uint32_t)r * (uint32_t)Gc + (uint32_t)c;
    ukey[t] = (uint32_t)u;
    idx[t] = (int32_t)t;
}

__global__ void k_gather_u32(const uint32_t *__restrict__ src, const int32_t *__restrict__ idx, int64_t n, uint32_t *dst)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) dst[t] = src[idx[t]];
}

// sorted position t <- original row idx[t]
__global__ void k_gather_soa(const int32_t *__restrict__ s, int32_t stride, const int32_t *__restrict__ idx, int64_t n,
                             int32_t *user, int32_t *item, float *val)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int64_t o = idx[t];
    user[t] = s[o * stride];
    item[t] = s[o * stride + 1];
    val[t] = (float)s[o * stride + 2];
}

// a run = a maximal stretch of one user's ratings inside a block; in a hot segment (cls != nullptr, class < MAXSEG)
// every rating is a run of its own, and a change of class is a run boundary even if the same user follows
__global__ void k_run_flags(const uint32_t *__restrict__ bkey, const int32_t *__restrict__ user,
                            const uint8_t *__restrict__ cls, int64_t n, int32_t *flag)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    bool f = t == 0 || bkey[t] != bkey[t - 1] || user[t] != user[t - 1];
    if (cls && t > 0) f = f || cls[t] < MAXSEG || cls[t] != cls[t - 1];   // (a user is in ONE of the two rest classes)
    flag[t] = f ? 1 : 0;
}

// order 1 walks a user's ratings of a block most rated item first: key = n - degree (stable sort: ties in input order)
__global__ void k_item_rank_key(const int32_t *__restrict__ s, int64_t n, int32_t stride, const int32_t *__restrict__ deg,
                                uint32_t *key)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) key[t] = (uint32_t)(n - deg[s[t * stride + 1]]);
}

__global__ void k_item_deg_raw(const int32_t *__restrict__ s, int64_t n, int32_t stride, int32_t *deg)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) atomicAdd(&deg[s[t * stride + 1]], 1);
}

// hot_user bit (stripe c, user u): u rated one of stripe c's hot items (that rating sits in a hot segment of u's block)
__global__ void k_mark_hot_users(const int32_t *__restrict__ s, int64_t n, int32_t stride, int32_t ibp, int64_t n_users,
                                 const int32_t *__restrict__ col_hot, uint32_t *hot_user)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int32_t u = s[t * stride], item = s[t * stride + 1];
    const int32_t c = item / ibp;
    const int32_t *h = col_hot + (size_t)c * MAXSEG;
    bool is_hot = false;
#pragma unroll
    for (int j = 0; j < MAXSEG; ++j) is_hot = is_hot || h[j] == item;
    if (!is_hot) return;
    const uint64_t bit = (uint64_t)c * (uint64_t)n_users + (uint64_t)u;
    atomicOr(&hot_user[bit >> 5], 1u << (bit & 31));
}

// order 1 sorts a block's ratings of the stripe's hot items in front of the others, one segment per item:
// key = NCLS * block + class, class = s for hot item s of the stripe; every other item: MAXSEG, or MAXSEG + 1 if the
// user has a rating in a hot segment of the block (hot_user == nullptr: no hot segments at all)
__global__ void k_hot_block_key(const int32_t *__restrict__ s, int64_t n, int32_t stride, int32_t Gc, int64_t n_users,
                                const int32_t *__restrict__ col_hot, const uint32_t *__restrict__ hot_user, uint32_t *bkey)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const uint32_t b = bkey[t];
    const int32_t item = s[t * stride + 1];
    const uint32_t col = b % (uint32_t)Gc;
    const int32_t *h = col_hot + (size_t)col * MAXSEG;
    uint32_t c = MAXSEG;
    if (hot_user) {
        const uint64_t bit = (uint64_t)col * (uint64_t)n_users + (uint64_t)s[t * stride];
        c += (hot_user[bit >> 5] >> (bit & 31)) & 1u;
    }
#pragma unroll
    for (int j = 0; j < MAXSEG; ++j)
        if (h[j] == item) c = j;
    bkey[t] = NCLS * b + c;
}

__global__ void k_split_hot_key(uint32_t *bkey, int64_t n, uint8_t *cls)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const uint32_t b = bkey[t];
    cls[t] = (uint8_t)(b % NCLS);
    bkey[t] = b / NCLS;
}

// hot segments: blk_seg_cnt[block][s] = length of segment s, blk_hot_cnt[block] = their sum; *dup = 1 if a user
// occurs twice in one segment (a duplicate (user, item) pair)
__global__ void k_hot_count(const uint32_t *__restrict__ bkey, const int32_t *__restrict__ user,
                            const uint8_t *__restrict__ cls, int64_t n, int32_t *blk_seg_cnt, int32_t *blk_hot_cnt, int32_t *dup)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n || cls[t] >= MAXSEG) return;
    atomicAdd(&blk_seg_cnt[(size_t)bkey[t] * MAXSEG + cls[t]], 1);
    atomicAdd(&blk_hot_cnt[bkey[t]], 1);
    if (t > 0 && cls[t - 1] == cls[t] && bkey[t - 1] == bkey[t] && user[t - 1] == user[t]) *dup = 1;
}

__global__ void k_run_fill(const int32_t *__restrict__ flag, const int32_t *__restrict__ excl,
                           const int32_t *__restrict__ user, const uint32_t *__restrict__ bkey,
                           const uint8_t *__restrict__ cls, int64_t n, int32_t n_runs, int32_t *run_start,
                           int32_t *run_user, int32_t *blk_free_runs)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0) { run_start[n_runs] = (int32_t)n; run_start[n_runs + 1] = (int32_t)n; }
    if (t >= n || !flag[t]) return;
    run_start[excl[t]] = (int32_t)t;
    run_user[excl[t]] = user[t];
    if (cls && cls[t] == MAXSEG) atomicAdd(&blk_free_runs[bkey[t]], 1);
}

// sorted block keys -> blk_ptr (rating offsets) ; same as prep.hip's k_segment_ptr with int32 out
__global__ void k_blk_ptr(const uint32_t *__restrict__ keys, int64_t n, int64_t nb, int32_t *ptr)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t > n) return;
    const int64_t lo = t == 0 ? 0 : (int64_t)keys[t - 1] + 1;
    const int64_t hi = t == n ? nb : (int64_t)keys[t];
    for (int64_t b = lo; b <= hi; ++b) ptr[b] = (int32_t)t;
}

__global__ void k_blk_run_ptr(const int32_t *__restrict__ blk_ptr, const int32_t *__restrict__ excl, int64_t nb,
                              int64_t n, int32_t n_runs, int32_t *blk_run_ptr)
{
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b > nb) return;
    const int32_t p = blk_ptr[b];
    blk_run_ptr[b] = p < n ? excl[p] : n_runs;
}

__global__ void k_item_keys(const uint32_t *__restrict__ bkey, const int32_t *__restrict__ item, int64_t n,
                            uint64_t *key, int32_t *pos)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    key[t] = ((uint64_t)bkey[t] << 32) | (uint32_t)item[t];
    pos[t] = (int32_t)t;
}

__global__ void k_seg_start(const uint64_t *__restrict__ key, int64_t n, int32_t *start)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    start[j] = (j == 0 || key[j] != key[j - 1]) ? (int32_t)j : 0;
}

__global__ void k_ticket(const int32_t *__restrict__ segstart, const int32_t *__restrict__ pos, int64_t n, int32_t *tick)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) tick[pos[j]] = (int32_t)j - segstart[j];
}

__global__ void k_run_keys(const int32_t *__restrict__ run_start, const int32_t *__restrict__ run_user,
                           const uint32_t *__restrict__ bkey, int32_t n_runs, uint64_t *key)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n_runs) key[j] = ((uint64_t)bkey[run_start[j]] << 32) | (uint32_t)run_user[j];
}

__global__ void k_adjacent_dup(const uint64_t *__restrict__ key, int64_t n, int32_t *dup)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j + 1 < n && key[j] == key[j + 1]) *dup = 1;
}

inline int bits_for(uint64_t maxkey)
{
    int b = 1;
    while (b < 64 && (maxkey >> b)) ++b;
    return b;
}

// HOT items of a stripe: its most rated items, at most `max_hot` of them (most rated first, equal degrees by lower id),
// each with at least HOT_MIN_DEG ratings and at least `min_deg` = min_per_block * Gr (on average min_per_block
// ratings per block of the stripe, the caller's choice - mfsgd_grid_create_opt); with `need_share` an item must also
// carry 1/HOT_SHARE_DIV of the stripe's ratings, i.e. more than one wave's fair share (the rule of mfsgd_grid_create).
// Item s's ratings are stored as hot segment s in front of every block of the stripe (hot-item pipeline,
// phase_a_gram).  Segments much shorter than a pipeline batch cost more than they save (ML-1M shape, 93 x 75 blocks of
// ~130 ratings: 1.70 ms per epoch with them, 1.26 ms without).
constexpr int HOT_SHARE_DIV = 16;
constexpr int HOT_MIN_DEG = 256;
__global__ void k_col_hot(const int32_t *__restrict__ deg, int64_t n_items, int32_t ibp, int32_t Gc, int32_t max_hot,
                          int64_t min_deg, int32_t need_share, int32_t *col_hot)
{
    const int32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= Gc) return;
    int64_t total = 0;
    for (int64_t i = (int64_t)c * ibp; i < (int64_t)(c + 1) * ibp && i < n_items; ++i) total += deg[i];
    int32_t last_deg = 0x7fffffff, last_id = -1;
    bool open = true;
    for (int s = 0; s < MAXSEG; ++s) {
        int32_t best = -1, best_deg = 0;
        if (open && s < max_hot)
            for (int64_t i = (int64_t)c * ibp; i < (int64_t)(c + 1) * ibp && i < n_items; ++i) {
                const int32_t d = deg[i];
                const bool later = d < last_deg || (d == last_deg && (int32_t)i > last_id);   // not picked yet
                if (later && d > best_deg) { best_deg = d; best = (int32_t)i; }
            }
        const bool hot = best >= 0 && best_deg >= HOT_MIN_DEG && (int64_t)best_deg >= min_deg &&
                         (!need_share || (int64_t)best_deg * HOT_SHARE_DIV >= total);
        col_hot[(size_t)c * MAXSEG + s] = hot ? best : -1;
        if (!hot) open = false;   // degrees only fall from here on
        last_deg = best_deg; last_id = best;
    }
}

template <typename K, typename V>
int sort_pairs_t(K *k_in, K *k_out, V *v_in, V *v_out, int64_t n, int end_bit, hipStream_t st)
{
    size_t tb = 0;
    MF_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, k_in, k_out, v_in, v_out, (int)n, 0, end_bit, st));
    DevBuf tmp;
    MF_HIP(tmp.alloc(tb));
    MF_HIP(hipcub::DeviceRadixSort::SortPairs(tmp.p, tb, k_in, k_out, v_in, v_out, (int)n, 0, end_bit, st));
    MF_HIP(hipStreamSynchronize(st));
    return MFSGD_OK;
}

template <typename K>
int sort_keys_t(K *k_in, K *k_out, int64_t n, int end_bit, hipStream_t st)
{
    size_t tb = 0;
    MF_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, tb, k_in, k_out, (int)n, 0, end_bit, st));
    DevBuf tmp;
    MF_HIP(tmp.alloc(tb));
    MF_HIP(hipcub::DeviceRadixSort::SortKeys(tmp.p, tb, k_in, k_out, (int)n, 0, end_bit, st));
    MF_HIP(hipStreamSynchronize(st));
    return MFSGD_OK;
}

// ------------------------------------------------------------------------------------------ epoch kernel
// -DMFSGD_PROF builds a separate measuring library (profiles/Makefile): per-wave time stamps around the
// phases of a block, summed into 16 counters behind the column clocks.  The shipped library has none of it.
#ifdef MFSGD_PROF
#define PF_PARAM , unsigned long long *pf
#define PF_PASS , pf
#define PF_T(name) const unsigned long long name = __builtin_amdgcn_s_memtime()
#define PF_ADD(slot, from, to) pf[slot] += (to) - (from)
#define PF_INC(slot, v) pf[slot] += (v)
#define PF_PIN(x) asm volatile("" ::"v"(x))
#else
#define PF_PARAM
#define PF_PASS
#define PF_T(name)
#define PF_ADD(slot, from, to)
#define PF_INC(slot, v)
#define PF_PIN(x)
#endif
enum { PF_BLOCK = 0, PF_PROLOGUE, PF_ISSUE, PF_HEADWAIT, PF_POLL, PF_UPDATE, PF_TAIL, PF_DRAIN, PF_BARRIER, PF_RUNS,
       PF_RATINGS, PF_SPINS, PF_SCHED, PF_TENURE, PF_POLL0, PF_N0, PF_AHEAD, PF_POLL1, PF_N1, PF_SLOTS = 40 };

constexpr int CTL_WORDS = 16 + 128 + 5 * MAXSEG + 4;  // 16 control words + 128 sink words (see sink_ptr) + the hot segments' words
// hot-item pipeline, per hot segment s of the block in flight: CTL_TOKEN + s = batches whose chain is done (whose turn it
// is), CTL_FRONT + s = batches published, CTL_FUSER + s = "segment s is done for every user <= this" (0x7fffffff: all)
// CTL_BND + p, p = 0 .. MAXSEG + 1: run ranges of the block's passes (hot segments, then the rest; process_block)
// CTL_FMIN: a lower bound of the minimum of all CTL_FUSER words, only ever raised (atomic max): what the rest of the
// block waits behind.  The walker (one segment) uses the same words: CTL_TOKEN = groups of MFSGD_WALK_SUB links whose chain
// is done, CTL_TOKEN + 1 = "the apply wave has its copy of the item's row", CTL_FRONT = chunks of 64 links applied
enum { CTL_COL = 0, CTL_ROW = 1, CTL_RUN = 2, CTL_REMAIN = 3, CTL_FMIN = 4, CTL_SINK = 16, CTL_TOKEN = 16 + 128,
       CTL_FRONT = CTL_TOKEN + MAXSEG, CTL_FUSER = CTL_FRONT + MAXSEG, CTL_BND = CTL_FUSER + MAXSEG,
       CTL_HOT = CTL_BND + MAXSEG + 2 };   // CTL_HOT + s: hot item s of the stripe in tenure (global id, -1: none)

// "Only lane 0 does X" is written WITHOUT a branch everywhere in this file: every lane executes the store or
// atomic, lane 0 on the real address and lane l>0 on its own sink word in LDS.  Reason (seen in the gfx950
// ISA, twice): hipcc threads lanes 1..63 around an `if (lane == 0)` whose result feeds a broadcast or sits
// at a loop head/latch, re-enters the loop body with part of the wave, and the body - including its
// s_barrier, DPP and readlane instructions - then runs twice per iteration for that wave.
template <typename U>
__device__ __forceinline__ U *sink_ptr(U *real, int *ctl, int lane)
{
    return lane == 0 ? real : (U *)(ctl + CTL_SINK + 2 * (lane & 63));
}

__device__ __forceinline__ uint32_t mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

template <typename T> struct LdsPlan {  // byte offsets inside the dynamic LDS region (all multiples of 16)
    uint32_t ctl, ebuf, tick, rowdone, rowrun, qs, bis, rt_u, rt_t, abuf, abuf_bytes, total;
    int32_t n_abuf;   // row buffers of the hot-item pipeline (phase_a_gram); 0: hot segments go through the tickets
    int32_t hot_walker;   // 1: stripes with one hot item get the register-resident walker (phase_a_serial)
};
constexpr int HOT_EBUFS = 4;   // chunks of 64 errors the walker's chain wave may be ahead of its apply wave
constexpr int RT_CAP = 1024;  // runs per block the LDS run table takes (the default 288 x 256 netflix grid has ~720 per block)
constexpr int PA_B = 32;      // ratings per batch of the hot-item pipeline (one 32 x 32 MFMA tile of Gram entries)
constexpr int PA_MAX_BUFS = PA_MAX_BUFS_DEFAULT;
constexpr size_t LDS_CAP = (size_t)160 * 1024;

// `abufs`: how many row buffers of the hot-item pipeline to place behind the base plan (as many as fit, at most
// PA_MAX_BUFS); 0 = none wanted.
template <typename T>
LdsPlan<T> lds_plan(int32_t ibp, int32_t Gr, int32_t k, bool stage, bool dynamic, int abufs = 0)
{
    auto up = [](size_t x) { return (x + 15) & ~(size_t)15; };
    LdsPlan<T> p{};
    size_t o = 0;
    p.ctl = o; o = up(o + CTL_WORDS * 4);
    p.ebuf = o; o = up(o + (size_t)HOT_EBUFS * 64 * sizeof(T));   // the hot-item walker's error words, chain wave -> apply wave
    p.tick = o; o = up(o + (size_t)ibp * 4);
    p.rowdone = o; o = up(o + (dynamic ? (size_t)Gr : 0));
    p.rowrun = o; o = up(o + (dynamic ? (size_t)Gr * 16 : 0));
    p.qs = o; o = up(o + (stage ? (size_t)ibp * k * sizeof(T) : 0));
    p.bis = o; o = up(o + (stage ? (size_t)ibp * sizeof(T) : 0));
    p.rt_u = o; o = up(o + (size_t)(RT_CAP + 2) * 4);
    p.rt_t = o; o = up(o + (size_t)(RT_CAP + 2) * 4);
    p.abuf = o;
    p.abuf_bytes = (uint32_t)up((size_t)PA_B * (k + 4) * sizeof(T));
    p.n_abuf = 0;
    if (stage && abufs > 0 && o < LDS_CAP)
        p.n_abuf = (int32_t)std::min<size_t>((size_t)std::min(abufs, PA_MAX_BUFS), (LDS_CAP - o) / p.abuf_bytes);
    o += (size_t)p.n_abuf * p.abuf_bytes;
    p.total = (uint32_t)std::min<size_t>(o, 0xffffffffu);
    return p;
}

__device__ __forceinline__ void untracked_store(float *p, float v)
{
    asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void untracked_store(double *p, double v)
{
    asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}

// Lane -> factor map of the walk (load_run / walk_run): lane l holds the KC CONSECUTIVE factors KC*l .. KC*l + KC-1 of a
// row, so a row moves with ONE access per lane - global_load_dwordx2 / ds_read_b64 at k = 128, dwordx4 / b128 at k = 256
// (fp32; twice that in fp64) - instead of KC dword accesses at stride 64.  The dot product sums over all lanes and
// registers, so it does not care; the hot-item pipeline keeps its own (lane + 64 c) map for its LDS row buffers.
template <typename T, int KC> struct RowVec { typedef T type __attribute__((ext_vector_type(KC))); };
template <typename T, int KC>
__device__ __forceinline__ void row_get(const T *row, int lane, T (&x)[KC])
{
    if constexpr (KC == 2 || KC == 4) {
        const typename RowVec<T, KC>::type v = *(const typename RowVec<T, KC>::type *)(row + KC * lane);
#pragma unroll
        for (int c = 0; c < KC; ++c) x[c] = v[c];
    } else {
#pragma unroll
        for (int c = 0; c < KC; ++c) x[c] = row[KC * lane + c];
    }
}
template <typename T, int KC>
__device__ __forceinline__ void row_put(T *row, int lane, const T (&x)[KC])
{
    if constexpr (KC == 2 || KC == 4) {
        typename RowVec<T, KC>::type v;
#pragma unroll
        for (int c = 0; c < KC; ++c) v[c] = x[c];
        *(typename RowVec<T, KC>::type *)(row + KC * lane) = v;
    } else {
#pragma unroll
        for (int c = 0; c < KC; ++c) row[KC * lane + c] = x[c];
    }
}
// A row of P from device memory, coherent at agent scope (see MF_AGENT_RELEASE): relaxed agent-scope atomic loads of 4 or 8
// bytes (global_load_dword / dwordx2 ... sc1), which the compiler tracks like any other load (exact vmcnt waits).
template <typename T, int KC>
__device__ __forceinline__ void row_get_agent(const T *row, int lane, T (&x)[KC])
{
    constexpr int BYTES = KC * (int)sizeof(T);
    const T *p = row + KC * lane;
    if constexpr (BYTES % 8 == 0) {
        union { uint64_t w[BYTES / 8]; T v[KC]; } u;
#pragma unroll
        for (int i = 0; i < BYTES / 8; ++i)
            u.w[i] = __hip_atomic_load((const uint64_t *)p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int c = 0; c < KC; ++c) x[c] = u.v[c];
    } else {
        static_assert(sizeof(T) == 4, "rows of 4-byte elements");
#pragma unroll
        for (int c = 0; c < KC; ++c)
            x[c] = __uint_as_float(__hip_atomic_load((const uint32_t *)p + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
}
// the same store outside the compiler's vmcnt bookkeeping (see walk_run's tail)
template <int KC>
__device__ __forceinline__ void row_put_untracked(float *row, int lane, const float (&x)[KC])
{
    float *p = row + KC * lane;
    if constexpr (KC == 2) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        const f2 v = {x[0], x[1]};
        asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    } else if constexpr (KC == 4) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4 v = {x[0], x[1], x[2], x[3]};
        // a VMEM store of more than 64 bits must not be followed at once by a VALU write of its data registers (two wait
        // states on gfx90a-class targets); hipcc's hazard recognizer cannot see into the asm statement
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    } else {
#pragma unroll
        for (int c = 0; c < KC; ++c) asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p + c), "v"(x[c]) : "memory");
    }
}
template <int KC>
__device__ __forceinline__ void row_put_untracked(double *row, int lane, const double (&x)[KC])
{
    double *p = row + KC * lane;
    if constexpr (KC == 2) {
        typedef double d2 __attribute__((ext_vector_type(2)));
        const d2 v = {x[0], x[1]};
        // a VMEM store of more than 64 bits must not be followed at once by a VALU write of its data registers (two wait
        // states on gfx90a-class targets); hipcc's hazard recognizer cannot see into the asm statement
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    } else {
#pragma unroll
        for (int c = 0; c < KC; ++c) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p + c), "v"(x[c]) : "memory");
    }
}

// Stage 1 of a run: which run, whose, and where its ratings are (wave-uniform words).
struct RunIdx {
    int32_t j, u, t0, t1;
};

// Stage 2: the user's row (+ implicit term, bias) and the first 64 ratings of the run, one rating per lane.
template <typename T, int KC> struct RunHead {
    int32_t u, t0, t1;
    T p[KC], zz[KC], bu;
    int32_t m_item, m_tick;
    float m_val;
};

__device__ __forceinline__ void run_index(const GridView &g, int32_t j, RunIdx &x)
{
    // Loaded with VECTOR loads on purpose (atomic loads are never turned into s_load): scalar loads share
    // lgkmcnt with the LDS traffic of the run being walked and return out of order, so the first ticket poll
    // of the current run would otherwise wait for these far-ahead loads (an HBM round trip per run).
    x.j = j;
    x.u = __hip_atomic_load(&g.run_user[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    x.t0 = __hip_atomic_load(&g.run_start[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    x.t1 = __hip_atomic_load(&g.run_start[j + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

// one coalesced load per array brings the metadata of ratings [t, t+64) of a run, lane l holding rating t+l
__device__ __forceinline__ void load_meta(const GridView &g, int32_t t, int32_t t1, int lane, int32_t &m_item,
                                          int32_t &m_tick, float &m_val)
{
    // t and t1 are wave-uniform: scalar base + a lane offset below 64 - one v_min and one shift per run instead of a
    // 64-bit vector address per array (netflix shape 17.87 -> 17.72 ms per epoch, ML-25M 6.94 -> 6.85).  The same for the
    // user's row was NOT faster (its address as an opaque scalar: 18.8 ms; the stores with a scalar base: 17.9).
    // An empty run (the one behind a table's last: t == t1) reads word t + 63, which no one uses: the three arrays are
    // allocated with META_PAD words behind their last rating (mfsgd_grid_create_opt); guarding that case in here instead
    // cost more than the change gains (18.1 ms).
    const uint32_t off = (uint32_t)min((int32_t)lane, t1 - 1 - t) & 63u;
    m_item = (g.item + t)[off];
    m_tick = (g.tick + t)[off];
    m_val = (g.val + t)[off];
}

// Issues the loads of a run's head; nothing here waits, so a caller can keep two runs in flight behind the
// one it is walking and the HBM latency of the row and of the metadata hides behind that walk.
// FAST != 0 (k a multiple of 64; 1: no implicit term, 2: with the frozen term z): no lane guard and no optional load, so the loop that calls this
// is straight-line code and hipcc can count vmcnt exactly (with any conditional load in the pipeline it falls
// back to waiting for the youngest loads, i.e. exposes a full HBM round trip per run).
template <typename T, int KC, int FAST>
__device__ __forceinline__ void load_run(const GridView &g, const RunIdx &x, const T *__restrict__ P,
                                         const T *__restrict__ b_u, const T *__restrict__ z, int k, int lane,
                                         RunHead<T, KC> &h)
{
    // everything that steers control flow is made explicitly wave-uniform (SGPRs, scalar branches)
    h.u = __builtin_amdgcn_readfirstlane(x.u);
    h.t0 = __builtin_amdgcn_readfirstlane(x.t0);
    h.t1 = __builtin_amdgcn_readfirstlane(x.t1);
    if constexpr (FAST != 0) k = 64 * KC;   // the straight-line builds: a constant row width (shifts instead of multiplies)
    const T *prow = P + (int64_t)h.u * k;
    if constexpr (FAST != 0) {
        row_get_agent<T, KC>(prow, lane, h.p);
        if constexpr (FAST == 2) {
            row_get<T, KC>(z + (int64_t)h.u * k, lane, h.zz);
        } else {
#pragma unroll
            for (int c = 0; c < KC; ++c) h.zz[c] = 0;
        }
    } else {
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            const int f = KC * lane + c;
            h.p[c] = 0; h.zz[c] = 0;
            if (f < k) {
                h.p[c] = prow[f];
                if (z) h.zz[c] = z[(int64_t)h.u * k + f];
            }
        }
    }
    // biases are rewritten by other workgroups earlier in this launch: atomic loads never become scalar loads
    h.bu = __hip_atomic_load(&b_u[h.u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    load_meta(g, h.t0, h.t1, lane, h.m_item, h.m_tick, h.m_val);
}

// wave64 sum for the rating chains (walk_run, the walker): wave_sum's tree (same result bit for bit) with the two row
// broadcasts as single v_add_f32_dpp instructions (hipcc expands the masked broadcast into v_mov 0 / v_mov_dpp / v_add: 4
// instructions more per rating, on chains where every instruction is ~5 cycles)
__device__ __forceinline__ float chain_sum(float v)
{
    int x;
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false);
    v += __int_as_float(x);
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false);
    v += __int_as_float(x);
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, false);
    v += __int_as_float(x);
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, false);
    v += __int_as_float(x);
    // rows 1 and 3 += the row before, then rows 2 and 3 += row 1 / row 1 + ... : lane 63 ends with (r0 + r1) + (r2 + r3)
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
                 : "+v"(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ double chain_sum(double v) { return wave_sum(v); }

// One rating (numba_methods.py:17-22 without the implicit write-back): error, new user row `pn`, new item row
// in place, new item bias in place.  The caller settles b_u (it needs `e` only).
__device__ __forceinline__ float fma_t(float x, float y, float w) { return __builtin_fmaf(x, y, w); }
__device__ __forceinline__ double fma_t(double x, double y, double w) { return __builtin_fma(x, y, w); }
// With a = 1 - alpha beta2 (numba_methods.py:21-22):  p' = a p + (alpha e) q,   q' = a q + (alpha e)(p' + z)  - the form the
// walker uses too: a p and a q do not need the error, so they are formed while the reduction runs, and what follows the
// error is two dependent fused multiply-adds per factor (the literal form p + alpha (e q - beta2 p) cost 13 vector
// instructions per rating at k = 128 after the reduction, this one 6).
// VAR: the update order as a compile-time constant (0 sgd, 1 sgd2; -1: `variant` decides at run time - two selects per
// rating on the dependent chain behind the error)
template <typename T, int KC, bool HASZ, int VAR = -1>
__device__ __forceinline__ T sgd_step(const T (&p)[KC], const T (&zz)[KC], T (&q)[KC], T &bi, T rest, T alpha, T beta1,
                                      T beta2, int variant, T (&pn)[KC])
{
    if constexpr (VAR >= 0) variant = VAR;
    const T a = (T)1 - alpha * beta2, a1 = (T)1 - alpha * beta1;
    T part = q[0] * (HASZ ? p[0] + zz[0] : p[0]), ap[KC], aq[KC];
#pragma unroll
    for (int c = 1; c < KC; ++c) part += q[c] * (HASZ ? p[c] + zz[c] : p[c]);
#pragma unroll
    for (int c = 0; c < KC; ++c) { ap[c] = a * p[c]; aq[c] = a * q[c]; }
    const T e = rest - bi - chain_sum(part);
    const T ae = alpha * e;
#pragma unroll
    for (int c = 0; c < KC; ++c) {
        pn[c] = fma_t(ae, q[c], ap[c]);
        const T pu = variant == 0 ? pn[c] : p[c];  // sgd: new p feeds q (numba_methods.py:22); sgd2: old (:53)
        q[c] = fma_t(ae, HASZ ? pu + zz[c] : pu, aq[c]);
    }
    bi = fma_t(alpha, e, a1 * bi);               // b_i + alpha (e - beta1 b_i)
    return e;
}

#ifdef MFSGD_WALK_SINK   // A/B: lane 0 alone writes the real word
#define MF_WALK_STORE(ptr) (*sink_ptr((ptr), ctl, lane))
#else
#define MF_WALK_STORE(ptr) (*(ptr))
#endif
template <typename T, int KC, bool STAGE, int FAST, int VAR = -1>
__device__ __forceinline__ void walk_run(const GridView &g, RunHead<T, KC> &h, int32_t c_base, T *__restrict__ P,
                                         T *__restrict__ Q, T *__restrict__ b_u, T *__restrict__ b_i, T b, T alpha,
                                         T beta1, T beta2, int variant, int k, int *ctl, int *tick_done, T *qs,
                                         T *bis, int lane, int32_t *err PF_PARAM)
{
    const int32_t u = h.u, t0 = h.t0, t1 = h.t1;
    if constexpr (FAST != 0) k = 64 * KC;   // (see load_run)
    PF_T(w0);
    // The head's loads are waited for HERE, by name: hipcc then counts exactly (s_waitcnt vmcnt(N), N = the loads issued
    // since - the heads of the two runs behind this one stay in flight).  Left to the rating loop, the wait became a
    // vmcnt(0) in the loop's preheader (LLVM flushes the counter in front of a loop that only consumes loads).
    asm volatile("" ::"v"(h.m_item), "v"(h.m_tick), "v"(h.m_val), "v"(h.p[0]), "v"(h.p[KC - 1]), "v"(h.bu));
    PF_T(w1);
    PF_ADD(PF_HEADWAIT, w0, w1);
    PF_INC(PF_RUNS, 1);
    // the walk works on the head's own registers (no copies at entry: every VALU instruction counts)
    T (&p)[KC] = h.p;
    T (&zz)[KC] = h.zz;
    T q[KC];
    T &bu = h.bu;
    int32_t &m_item = h.m_item, &m_tick = h.m_tick;
    float &m_val = h.m_val;
    T *prow = P + (int64_t)u * k;
    // The head brought the metadata of the run's first 64 ratings.  A longer run reloads it every 64 ratings - OUTSIDE the
    // rating loop, and waited for on the spot: with a (conditional) load inside the loop hipcc's wait insertion merges
    // "just reloaded" into every iteration and starts each rating with s_waitcnt vmcnt(0), i.e. every run waited for the
    // head loads of the run two ahead that were issued a moment ago - a full memory round trip per run, which is what
    // the three-deep prefetch exists to hide (seen in the ISA; netflix shape 30.1 -> 24.9 ms per epoch).
    for (int32_t tc = t0;;) {
    const int32_t te = t1 - tc > 64 ? tc + 64 : t1;
    // r - b for the 64 ratings at once (fp32: the rating loop then reads it with one v_readlane instead of moving b into
    // a vector register and subtracting per rating; the value and the order of the subtractions are the same)
    float m_rb = m_val;
    if constexpr (sizeof(T) == 4) m_rb = m_val - (float)b;
    for (int32_t t = tc; t < te; ++t) {
        const int32_t slot = t - tc;
        const int32_t ig = __builtin_amdgcn_readlane(m_item, slot);
        const int32_t il = ig - c_base;
        int32_t tk = __builtin_amdgcn_readlane(m_tick, slot);
        T r = (T)__int_as_float(__builtin_amdgcn_readlane(__float_as_int(m_rb), slot));
        if constexpr (sizeof(T) != 4) r = r - b;
        T bi;
        T *qrow = Q + (int64_t)ig * k;
        PF_T(r0);
#ifdef MFSGD_PROF
        int32_t first_ahead = 0;
#endif
        if (STAGE) {
            // Ticket, Q row and b_i all live in LDS, and the LDS executes one wave's DS instructions in issue
            // order.  So each poll issues the ticket read AND the row reads back to back: if the ticket read
            // returns this rating's ticket, the row reads behind it already saw the predecessor's row (it
            // wrote the row before the ticket) - one LDS round trip per hand-off instead of two, and no
            // s_waitcnt on anything but LDS (a workgroup-scope fence over all address spaces would also
            // drain vmcnt, i.e. stall the item's chain on this wave's prefetched HBM loads).
            for (int spins = 0;; ++spins) {
                const int32_t seen = __hip_atomic_load(&tick_done[il], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                LDS_ACQUIRE();
                if constexpr (FAST != 0) {
                    row_get<T, KC>(qs + (size_t)il * k, lane, q);
                } else {
#pragma unroll
                    for (int c = 0; c < KC; ++c) {
                        const int f = KC * lane + c;
                        q[c] = f < k ? qs[(size_t)il * k + f] : (T)0;
                    }
                }
                bi = bis[il];
                int32_t ahead = tk - __builtin_amdgcn_readfirstlane(seen);
#ifdef MFSGD_PROF
                if (spins == 0) { first_ahead = ahead; }
#endif
                if (ahead == 0) break;
#ifdef MFSGD_NOWAIT   // timing experiment (make nowait): ignore the item order - results are wrong on purpose
                break;
#endif
                if (spins >= SPIN_TICKET_MAX) { atomicOr(err, 1); break; }
                PF_INC(PF_SPINS, 1);
                // `ahead` ratings of this item still have to pass.  Only the immediate successor polls at full
                // rate (row reads included); everybody further back sleeps through the hand-offs in front of it
                // and looks at the ticket word alone, so waiting waves take neither issue slots nor LDS
                // bandwidth from the wave that is on the item's critical path.
                while (ahead > 1) {
                    for (int32_t a = ahead > 9 ? 8 : ahead - 1; a > 0; --a) __builtin_amdgcn_s_sleep(TICKET_NAP);
                    ahead = tk - __builtin_amdgcn_readfirstlane(
                                     __hip_atomic_load(&tick_done[il], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                    if (++spins >= SPIN_TICKET_MAX) break;
                    PF_INC(PF_SPINS, 1);
                }
            }
            __builtin_amdgcn_s_setprio(2);   // holder of an item's ticket: the other waves of the SIMD are waiting on it
        } else {
            for (int spins = 0; __builtin_amdgcn_readfirstlane(__hip_atomic_load(
                                    &tick_done[il], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) != tk; ++spins) {
                if (spins >= SPIN_TICKET_MAX) { atomicOr(err, 1); break; }
                if (spins > 8) __builtin_amdgcn_s_sleep(1);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   // Q is in global memory: full fence
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                const int f = KC * lane + c;
                q[c] = f < k ? qrow[f] : (T)0;
            }
            bi = __hip_atomic_load(&b_i[ig], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        // the item's chain is serial across waves: only what the NEXT holder needs (the new row and b_i)
        // is computed before the ticket is passed on; b_u and the p registers are settled afterwards
        PF_PIN(q[0]); PF_PIN(q[KC - 1]); PF_PIN(bi);
        PF_T(r1);
        PF_ADD(PF_POLL, r0, r1);
#ifdef MFSGD_PROF
        if (first_ahead == 0) { pf[PF_POLL0] += r1 - r0; pf[PF_N0] += 1; }
        else if (first_ahead == 1) { pf[PF_POLL1] += r1 - r0; pf[PF_N1] += 1; }
        pf[PF_AHEAD] += first_ahead;
#endif
        PF_INC(PF_RATINGS, 1);
        T pn[KC];
        const T e = sgd_step<T, KC, FAST != 1, VAR>(p, zz, q, bi, r - bu, alpha, beta1, beta2, variant, pn);
        if (STAGE) {
            if constexpr (FAST != 0) {
                row_put<T, KC>(qs + (size_t)il * k, lane, q);
            } else {
#pragma unroll
                for (int c = 0; c < KC; ++c) {
                    const int f = KC * lane + c;
                    if (f < k) qs[(size_t)il * k + f] = q[c];
                }
            }
            // every lane stores the same value to the same word (a plain store needs no single writer; the select of
            // sink_ptr cost two vector instructions per rating)
            MF_WALK_STORE(&bis[il]) = bi;
            LDS_RELEASE();
        } else {
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                const int f = KC * lane + c;
                if (f < k) qrow[f] = q[c];
            }
            __hip_atomic_store(&b_i[ig], bi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // all lanes, same word
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        }
        __hip_atomic_store(&MF_WALK_STORE(&tick_done[il]), tk + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (STAGE) __builtin_amdgcn_s_setprio(0);
        bu = fma_t(alpha, e, ((T)1 - alpha * beta1) * bu);   // b_u + alpha (e - beta1 b_u)
#pragma unroll
        for (int c = 0; c < KC; ++c) p[c] = pn[c];
        PF_PIN(p[0]); PF_PIN(p[KC - 1]);
        PF_T(r2);
        PF_ADD(PF_UPDATE, r1, r2);
    }
    if (te >= t1) break;
    tc = te;
    load_meta(g, tc, t1, lane, m_item, m_tick, m_val);
    asm volatile("" : "+v"(m_item), "+v"(m_tick), "+v"(m_val));   // wait here, not in the loop
    }
    PF_T(w2);
    if (FAST) {
        // Stores the compiler does not see (untracked_store): with a tracked store pending, hipcc's wait
        // insertion treats vmcnt as out of order on gfx9-class targets and turns every later wait for a
        // prefetched load into vmcnt(0) - the whole pipeline would drain once per run.  Untracked stores only
        // make its counted waits a little more conservative (the real queue is longer than it thinks),
        // never less.  process_block drains vmcnt(0) by hand before the block's closing barrier.
        row_put_untracked<KC>(prow, lane, p);
        untracked_store(&b_u[u], bu);   // all lanes, same word
    } else {
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            const int f = KC * lane + c;
            if (f < k) prow[f] = p[c];
        }
        __hip_atomic_store(&b_u[u], bu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // all lanes, same word
    }
    PF_T(w3);
    PF_ADD(PF_TAIL, w2, w3);
}

// ------------------------------------------------------------------------------------ the hot-item pipeline
// A stripe's most rated item (col_hot) has its ratings FIRST in every block of the stripe, one rating per user, in
// user order (the "hot segment", see mfsgd_grid_create).  Its updates form a chain - every rating needs the item row
// its predecessor wrote - and on Netflix-shaped data the longest such chain, 140 K links of ~0.25 us through the LDS
// tickets, sets the epoch time.  This pipeline walks the SAME chain in the SAME order with the same arithmetic per
// link, but restated so that the serial part of a link is a handful of scalar operations:
//   with a = 1 - alpha*beta2, one update (numba_methods.py:17-22) is
//       e   = r - b - b_u - b_i - q.p          p' = a p + alpha e q
//       q'  = a q + alpha e p'  =  c q + d p   with c = a + alpha^2 e^2, d = alpha a e      (sgd2 order: c = a, d = alpha e)
//   so inside a batch of 32 consecutive ratings the item row stays in span{q_0, p_1 .. p_32}, and the dot products
//   the chain needs follow from the batch's Gram matrix G = P_b P_b^T (one 32x32 MFMA tile, k/2 v_mfma_f32_32x32x2
//   - computed BEFORE the wave's turn, off the chain) and g = P_b q_0 (when the turn comes):
//       s_j = q_cur . p_j  starts as g_j and becomes  c_i s_j + d_i G_ij  after link i      (one FMA pair per lane)
//   Link i itself is: read s_i and (r - b - b_u)_i from lane i, e_i, b_i, c_i, d_i - no LDS, no poll, no wave
//   reduction.  The batch's final row  q_32 = gamma q_0 + sum_i W_i p_i  is written back to the stripe and the turn
//   passes to the wave that holds the next batch; the user rows p_j' (which need every intermediate q_j) are then
//   formed off the chain by running q forward once more with the now known e_j.
// In exact arithmetic this IS the sequential pass; in fp32 it differs from it by rounding only (the tests replay
// the block order on the CPU restatement of the plain sequential kernel: factors to 1e-4, as for every other item).
// Waves 0 .. n_abuf-1 take batches round robin; each owns one LDS row buffer of PA_B x (k + 4) floats.  The other
// waves walk the rest of the block meanwhile: a rest run may load its user's row once the pipeline has published
// that the segment is done up to that user (CTL_FUSER; both are in user order), so the two never touch one row.
typedef float pa_f32x16 __attribute__((ext_vector_type(16)));
typedef float pa_f32x4 __attribute__((ext_vector_type(4)));

// min over the hot segments below `below` of "done for every user <= this" (empty segments hold 0x7fffffff)
__device__ __forceinline__ int32_t hot_frontier(const int *ctl, int below)
{
    int32_t f = 0x7fffffff;
#pragma unroll
    for (int s = 0; s < MAXSEG; ++s) {
        const int32_t v = __hip_atomic_load(&ctl[CTL_FUSER + s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (s < below && v < f) f = v;
    }
    return __builtin_amdgcn_readfirstlane(f);
}

// The hot segments of one block: segment s = ratings [t_lo + pre_s, t_lo + pre_s + cnt[s]) = single-rating runs
// [run_lo + pre_s, ...), pre_s = cnt[0] + .. + cnt[s-1], all of item hot[s], in batches of 32.  A batch waits for (a) every
// EARLIER segment to be done for its users (hot_frontier: a user's ratings are applied in stored order, segment after
// segment) before it loads their rows, and (b) the previous batch of its own segment (the item's chain, CTL_TOKEN).
// The pipeline waves are SPLIT among the segments, so that all segments move through the user range together and the
// frontier the rest of the block waits behind advances steadily: with at least as many waves as non-empty segments
// every segment gets one wave and the spare waves go where the most batches per wave are left (a segment's waves take
// its batches round robin); with fewer waves, wave w takes the non-empty segments w, w + n_abuf, .. whole, in order.
// Either way a wave only ever waits for LOWER segments or for an earlier batch of its own, and it takes its work in
// increasing (segment, batch) order: whoever holds the earliest unfinished batch can always proceed.
// MULTI = false: the stripe has ONE hot item (cnt[0] = the block's hot ratings, hot[0] = the item): no split, no look at
// other segments - a second instantiation, because the bookkeeping of the general case costs scalar registers that
// the single-item pipeline (the default) then spills (netflix shape: 32.2 ms per epoch against 30.4 ms).
template <int KC, bool MULTI>
__device__ __forceinline__ void phase_a_gram(const GridView &g, int32_t run_lo, int32_t t_lo, int32_t c_base,
                                             float *__restrict__ P, float *__restrict__ b_u, float b, float alpha,
                                             float beta1, float beta2, int variant, float *qs, float *bis, float *abuf,
                                             uint32_t abuf_floats, int n_abuf, int *ctl, int32_t *err PF_PARAM)
{
    // segment lengths and hot items come from the control words (process_block, k_epoch_free wrote them before a
    // barrier): kept in scalar registers across the run loop they were spilled
    int32_t cnt[MAXSEG], hot[MAXSEG];
#pragma unroll
    for (int s = 0; s < (MULTI ? MAXSEG : 1); ++s) {
        cnt[s] = __builtin_amdgcn_readfirstlane(ctl[CTL_BND + s + 1] - ctl[CTL_BND + s]);
        hot[s] = __builtin_amdgcn_readfirstlane(ctl[CTL_HOT + s]);
    }
    constexpr int K = 64 * KC, LD = K + 4;
    constexpr int LPR = K / 4, RPI = 64 / LPR;     // lanes per row and rows per 128-bit load instruction
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (wave >= n_abuf) return;
    float *buf = abuf + (size_t)wave * abuf_floats;
    const int col = lane & 31, kh = lane >> 5;
    const float a = 1.f - alpha * beta2;
    // this wave's share: of segment s it takes batches first[s], first[s] + step[s], ..  (step 0: none)
    int32_t first[MAXSEG], step[MAXSEG];
    if constexpr (!MULTI) {
#pragma unroll
        for (int s = 0; s < MAXSEG; ++s) { first[s] = wave; step[s] = s == 0 ? n_abuf : 0; }
    } else {
        int32_t nbat[MAXSEG], nwv[MAXSEG], nseg = 0;
#pragma unroll
        for (int s = 0; s < MAXSEG; ++s) { nbat[s] = (cnt[s] + PA_B - 1) / PA_B; nwv[s] = nbat[s] > 0 ? 1 : 0; nseg += nwv[s]; }
        if (nseg <= n_abuf) {
            for (int extra = n_abuf - nseg; extra > 0; --extra) {   // spare waves: most batches per wave first
                int best = -1;
                int64_t bn = 0, bd = 1;
#pragma unroll
                for (int s = 0; s < MAXSEG; ++s)
                    if (nwv[s] > 0 && nbat[s] > nwv[s] && (int64_t)nbat[s] * bd > bn * nwv[s]) { best = s; bn = nbat[s]; bd = nwv[s]; }
                if (best < 0) break;
#pragma unroll
                for (int s = 0; s < MAXSEG; ++s)
                    if (s == best) ++nwv[s];
            }
            int32_t before = 0;
#pragma unroll
            for (int s = 0; s < MAXSEG; ++s) {
                const bool mine = wave >= before && wave < before + nwv[s];
                first[s] = wave - before;
                step[s] = mine ? nwv[s] : 0;
                before += nwv[s];
            }
        } else {
            int32_t j = 0;
#pragma unroll
            for (int s = 0; s < MAXSEG; ++s) {
                const bool mine = nbat[s] > 0 && j % n_abuf == wave;
                first[s] = 0;
                step[s] = mine ? 1 : 0;
                j += nbat[s] > 0 ? 1 : 0;
            }
        }
    }
    PF_T(pa0);
    for (int32_t seg = 0; seg < (MULTI ? MAXSEG : 1); ++seg) {
        int32_t nh = 0, hot_item = 0, pre = 0, m0 = 0, dm = 0;
#pragma unroll
        for (int s = 0; s < (MULTI ? MAXSEG : 1); ++s) {
            if (s == seg) { nh = cnt[s]; hot_item = hot[s]; m0 = first[s]; dm = step[s]; }
            if (s < seg) pre += cnt[s];
        }
        if (dm == 0) continue;
        const int32_t nbatch = (nh + PA_B - 1) / PA_B;
    for (int32_t m = m0; m < nbatch; m += dm) {
        const int32_t *hu = g.run_user + run_lo + pre;
        const float *hv = g.val + t_lo + pre;
        float *qrow = qs + (size_t)(hot_item - c_base) * K;
        float *bip = bis + (hot_item - c_base);
        int *tok = ctl + CTL_TOKEN + seg, *front = ctl + CTL_FRONT + seg, *fuser = ctl + CTL_FUSER + seg;
        const int32_t base = m * PA_B;
        const int32_t nb = nh - base < PA_B ? nh - base : PA_B;
        // entry j of the batch lives in lanes j and j + 32 (entries past nb repeat the last one and are never applied)
        const int32_t ej = col < nb ? col : nb - 1;
        const int32_t uj = hu[base + ej];           // the segment's users and ratings come straight from the grid arrays
        if (MULTI && seg > 0) {   // the earlier segments must be done for this batch's users before their rows are loaded
            const int32_t last_user = __builtin_amdgcn_readlane(uj, PA_B - 1);
            for (int spins = 0; hot_frontier(ctl, seg) < last_user; ++spins) {
                if (spins >= SPIN_TICKET_MAX) { atomicOr(err, 4); break; }
                __builtin_amdgcn_s_sleep(2);
            }
            LDS_ACQUIRE();
        }
        const float buj = __hip_atomic_load(&b_u[uj], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float rest = hv[base + ej] - b - buj;
        PF_T(a0);
        // 1. the users' rows -> LDS, natural layout, with coalesced 128-bit loads (RPI whole rows per instruction)
        {
            const int rsub = lane / LPR, cpos = (lane % LPR) * 4;
#pragma unroll 8
            for (int t = 0; t < PA_B / RPI; ++t) {
                const int row = t * RPI + rsub;
                const int32_t ur = __builtin_amdgcn_ds_bpermute(4 * row, uj);   // lane `row` holds entry `row`
                // (agent-scope loads, see MF_AGENT_RELEASE: two 8-byte halves)
                const uint64_t *src = (const uint64_t *)(P + (int64_t)ur * K + cpos);
                union { uint64_t w[2]; pa_f32x4 v; } u;
                u.w[0] = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                u.w[1] = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *(pa_f32x4 *)(buf + row * LD + cpos) = u.v;
            }
        }
        PF_PIN(buf[lane]);
        PF_T(a1);
        PF_ADD(20, a0, a1);
        // 2. Gram tile: lane (col, kh) feeds P_b[col][8t + 4kh + c] as BOTH operands of MFMA 4t + c
        pa_f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const float *rp = buf + col * LD + 4 * kh;
#pragma unroll 4
        for (int t = 0; t < K / 8; ++t) {
            const pa_f32x4 v = *(const pa_f32x4 *)(rp + 8 * t);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.x, v.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.y, v.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.z, v.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.w, v.w, acc, 0, 0, 0);
        }
        // 3. lane (j, 0) gets the whole column j: acc[4a + b] is row 8a + b there and row 8a + 4 + b in lane (j, 1).
        //    Only the lower half's G (and s below) is meaningful; the upper half computes along and is never read.
        float G[PA_B];
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[v]), __float_as_uint(acc[v]), false, false);
            G[8 * (v / 4) + (v % 4)] = acc[v];
            G[8 * (v / 4) + 4 + (v % 4)] = __uint_as_float(kh ? sw[0] : sw[1]);
        }
        PF_PIN(G[0]); PF_PIN(G[31]);
        PF_T(a2);
        PF_ADD(21, a1, a2);
        // 4. the wave's turn
        for (int spins = 0; __builtin_amdgcn_readfirstlane(__hip_atomic_load(tok, __ATOMIC_RELAXED,
                                                                             __HIP_MEMORY_SCOPE_WORKGROUP)) != m; ++spins) {
            if (spins >= SPIN_TICKET_MAX) { atomicOr(err, 4); break; }
            __builtin_amdgcn_s_sleep(2);
        }
        LDS_ACQUIRE();
        __builtin_amdgcn_s_setprio(2);
        PF_T(a3);
        PF_ADD(22, a2, a3);
        // 5. g_j = q_0 . p_j: each half of the wave sums its chunks, the halves are added
        float q0[KC];
#pragma unroll
        for (int c = 0; c < KC; ++c) q0[c] = qrow[lane + 64 * c];
        float bi = *bip;
        float s;
        {
            float part = 0;
#pragma unroll 4
            for (int t = 0; t < K / 8; ++t) {
                const pa_f32x4 pv = *(const pa_f32x4 *)(rp + 8 * t);
                const pa_f32x4 qv = *(const pa_f32x4 *)(qrow + 8 * t + 4 * kh);
                part += pv.x * qv.x; part += pv.y * qv.y; part += pv.z * qv.z; part += pv.w * qv.w;
            }
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(part), __float_as_uint(part), false, false);
            s = part + __uint_as_float(kh ? sw[0] : sw[1]);
        }
        PF_PIN(s);
        PF_T(a4);
        PF_ADD(23, a3, a4);
        // 6. the chain
        float evec = 0, wvec = 0, gam = 1;
#pragma unroll
        for (int i = 0; i < PA_B; ++i) {
            if (i < nb) {   // wave-uniform
                const float si = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s), i));
                const float ri = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rest), i));
                const float e = ri - bi - si;
                bi += alpha * (e - beta1 * bi);
                const float c = variant == 0 ? a + (alpha * alpha) * (e * e) : a;
                const float d = variant == 0 ? (alpha * a) * e : alpha * e;
                s = c * s + d * G[i];
                wvec = col == i ? d : c * wvec;
                evec = col == i ? e : evec;
                gam *= c;
            }
        }
        PF_PIN(s); PF_PIN(wvec);
        PF_T(a5);
        PF_ADD(24, a4, a5);
        // 7. q_nb = gamma q_0 + sum_i W_i p_i, back into the stripe; the turn passes on
        {
            float qn[KC];
#pragma unroll
            for (int c = 0; c < KC; ++c) qn[c] = gam * q0[c];
            // entries past nb have W = 0 (wvec starts at 0 and is only scaled): the loop runs over the whole batch,
            // unrolled, so that the LDS reads are in flight together
#pragma unroll 8
            for (int32_t i = 0; i < PA_B; ++i) {
                const float wi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wvec), i));
#pragma unroll
                for (int c = 0; c < KC; ++c) qn[c] += wi * buf[i * LD + lane + 64 * c];
            }
#pragma unroll
            for (int c = 0; c < KC; ++c) qrow[lane + 64 * c] = qn[c];
            *sink_ptr(bip, ctl, lane) = bi;
            LDS_RELEASE();
            __hip_atomic_store(sink_ptr(tok, ctl, lane), m + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        __builtin_amdgcn_s_setprio(0);
        PF_T(a6);
        PF_ADD(25, a5, a6);
        // 8. off the chain: the users' new rows and biases (q runs forward once more, now with every e_j known)
        {
            float q[KC];
#pragma unroll
            for (int c = 0; c < KC; ++c) q[c] = q0[c];
#pragma unroll 2
            for (int32_t j = 0; j < nb; ++j) {
                const float e = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(evec), j));
                const int32_t u = __builtin_amdgcn_readlane(uj, j);
                float *prow = P + (int64_t)u * K;
#pragma unroll
                for (int c = 0; c < KC; ++c) {
                    const float pj = buf[j * LD + lane + 64 * c];
                    const float pn = pj + alpha * (e * q[c] - beta2 * pj);
                    const float pu = variant == 0 ? pn : pj;
                    q[c] = q[c] + alpha * (e * pu - beta2 * q[c]);
                    untracked_store(prow + lane + 64 * c, pn);
                }
            }
            if (lane < nb) untracked_store(&b_u[uj], buj + alpha * (evec - beta1 * buj));
        }
        // The batch's users are done once these stores are: publish it.  Batches are published in order (the frontier
        // is "every user of the segment up to this one"), so a wave may have to let an earlier batch go first.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int spins = 0; __builtin_amdgcn_readfirstlane(__hip_atomic_load(front, __ATOMIC_RELAXED,
                                                                             __HIP_MEMORY_SCOPE_WORKGROUP)) != m; ++spins) {
            if (spins >= SPIN_TICKET_MAX) { atomicOr(err, 4); break; }
            __builtin_amdgcn_s_sleep(2);
        }
        {
            const int32_t last_user = m + 1 == nbatch ? 0x7fffffff : __builtin_amdgcn_readlane(uj, PA_B - 1);
            __hip_atomic_store(sink_ptr(fuser, ctl, lane), last_user, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            LDS_RELEASE();
            __hip_atomic_store(sink_ptr(front, ctl, lane), m + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            // the rest's gate: the minimum over all segments as this wave sees it now - never above the truth (the
            // words only grow), and raised with an atomic max so that a late, lower value cannot pull it back
            if constexpr (MULTI) {
                // no hand-off: whatever these reads return is a valid lower bound (the words only grow), and the wave's
                // own store above is seen by its own reads (same wave, LDS in order); only the compiler must keep the order
                asm volatile("" ::: "memory");
                const int32_t fmin = hot_frontier(ctl, MAXSEG);
                atomicMax(sink_ptr(&ctl[CTL_FMIN], ctl, lane), fmin);
            } else {   // batches are published in order: the one word only grows
                __hip_atomic_store(sink_ptr(&ctl[CTL_FMIN], ctl, lane), last_user, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        PF_T(a7);
        PF_ADD(26, a6, a7);
        PF_INC(27, 1);
#ifdef MFSGD_PROF
        if (m + 1 == nbatch) { PF_ADD(31, pa0, a7); PF_INC(PF_ISSUE, 1); }   // from the block's start to the segment's last publish
#endif
    }   // batches
    }   // segments
}

// ------------------------------------------------------------------------------------ the hot-item walker (round 3)
// The same hot segment, walked by ONE wave that keeps the item's row in its REGISTERS for the whole segment: no
// ticket, no LDS hand-off, no Gram tile - a link is dot product, wave reduction, the two row updates (~130 cycles of
// dependent issue), and nothing of it waits for memory because the segment's users are known in advance (single-rating
// runs in user order): their rows stream through a ring of HOT_RING register slots, requested HOT_RING links ahead
// straight from HBM, the user ids / ratings / biases 64 links at a time one chunk ahead.  The arithmetic is walk_run's,
// operation by operation (same lane -> factor map, same reduction tree), so the result is the sequential pass to the
// bit the ticket path gives.  Works for every straight-line build - fp32 and fp64, k = 64 .. 256, with the frozen
// implicit term (its z row rides in a second ring) - and needs no LDS beyond the stripe, so it also runs where the
// stripes fill the LDS (ML-25M shape) and the matrix-core pipeline has no room for its row buffers.
// One hot item per stripe (the default); stripes with several hot items go through phase_a_gram or the tickets.
// The rest of the block runs meanwhile on the other waves, the runs of the segment's users behind the frontier word
// (CTL_FMIN), which the apply wave raises after every chunk of 64 links, once the chunk's row stores have drained (one
// drain of its ring of requested rows per chunk: the apply wave can afford it, the chain wave never drains).
#ifndef MFSGD_HOT_RING_REGS
#define MFSGD_HOT_RING_REGS 32   // 64 (a ring of 32 rows at k = 128): netflix shape 21.6 ms per epoch against 21.3; with the frozen term the
                                 // ring's p + z slots get twice the registers (16 slots instead of 8: 24.7 -> 22.3 ms, the chain was waiting
                                 // for HBM), so do 16-byte rows (k = 256: 34.6 -> 32.5 ms)
#endif
#ifndef MFSGD_HOT_RING_WIDE
#define MFSGD_HOT_RING_WIDE 2   // rows of 16 bytes per lane (k = 256 in fp32, 128 in fp64): twice the registers, the same number of slots
#endif
#ifndef MFSGD_HOT_RING_Z
#define MFSGD_HOT_RING_Z 2   // with the frozen term a slot is two rows (p and z): 2 = the same number of slots as without it
#endif
#ifndef MFSGD_WALK_SUB
#define MFSGD_WALK_SUB 16   // links per hand-over from the walker's chain wave to its apply wave (divides 64; see hot_walk)
#endif
#ifndef MFSGD_WALK_FRONT
// the walker's apply wave publishes the user frontier after every chunk of 64 links (0: once, at the end of the segment:
// netflix shape 24.1 ms per epoch against 22.1, frozen 31.2 against 28.1, ML-25M 9.22 against 8.79; 2: every 32 links:
// netflix 21.5 against 21.3, ML-25M 8.66 against 8.79)
#define MFSGD_WALK_FRONT 1
#endif
template <typename T, int KC, bool HASZ> struct HotRing {   // register slots: MFSGD_HOT_RING_REGS VGPRs of rows at most, 32 slots at most
    static constexpr int regs = KC * (HASZ ? 2 : 1) * (int)(sizeof(T) / 4);
    static constexpr int wanted = MFSGD_HOT_RING_REGS * (HASZ ? MFSGD_HOT_RING_Z : 1) * (KC * (int)sizeof(T) >= 16 ? MFSGD_HOT_RING_WIDE : 1);
    static constexpr int budget = wanted > 64 ? 64 : wanted;   // of the kernel's 128 registers
    // (vmcnt counts to 63: at most 32 loads of a ring in flight besides the metadata's, i.e. 16 slots of p + z)
    static constexpr int value = budget / regs >= 32 && !HASZ ? 32 : budget / regs >= 16 ? 16 : budget / regs >= 8 ? 8 : 4;
};
// The walker issues its row loads and waits for them BY HAND (hipcc's wait insertion, fed with a loop that carries 32
// loads in flight across its back edge and stores it does not see, put `s_waitcnt vmcnt(4)` in front of every link):
// a load writes its destination register whenever the memory system answers, the only interlock is the counted wait.
//   vm_load: the destination is a read-write operand, so it stays in ONE register from the request to the wait;
//   vm_wait<N>: "at most N vector memory operations still outstanding"; loads complete in the order they were issued,
//     so with N = the number of LOADS issued after the wanted one it has arrived (stores, which may complete out of
//     order relative to loads, only make the wait earlier than necessary, never later).
template <typename V> struct VmBytes { static constexpr int value = (int)sizeof(V); };
template <typename V>
__device__ __forceinline__ void vm_load(V &dst, const void *addr)
{
    if constexpr (sizeof(V) == 4) asm volatile("global_load_dword %0, %1, off" : "+v"(dst) : "v"(addr) : "memory");
    else if constexpr (sizeof(V) == 8) asm volatile("global_load_dwordx2 %0, %1, off" : "+v"(dst) : "v"(addr) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(dst) : "v"(addr) : "memory");
    static_assert(sizeof(V) == 4 || sizeof(V) == 8 || sizeof(V) == 16, "vm_load: 4, 8 or 16 bytes per lane");
}
// the same with the address as a 64-bit scalar base plus a 32-bit per-lane byte offset (one v_add instead of a 64-bit
// address computation per request)
template <typename V>
__device__ __forceinline__ void vm_load_so(V &dst, const void *sbase, uint32_t voff)
{
    if constexpr (sizeof(V) == 4) asm volatile("global_load_dword %0, %1, %2" : "+v"(dst) : "v"(voff), "s"(sbase) : "memory");
    else if constexpr (sizeof(V) == 8) asm volatile("global_load_dwordx2 %0, %1, %2" : "+v"(dst) : "v"(voff), "s"(sbase) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(dst) : "v"(voff), "s"(sbase) : "memory");
}
// ... and the forms for P and b_u, coherent at agent scope (sc1; see MF_AGENT_RELEASE)
template <typename V>
__device__ __forceinline__ void vm_load_agent(V &dst, const void *addr)
{
    static_assert(sizeof(V) == 4 || sizeof(V) == 8, "vm_load_agent: 4 or 8 bytes per lane");
    if constexpr (sizeof(V) == 4) asm volatile("global_load_dword %0, %1, off sc1" : "+v"(dst) : "v"(addr) : "memory");
    else asm volatile("global_load_dwordx2 %0, %1, off sc1" : "+v"(dst) : "v"(addr) : "memory");
}
template <typename V>
__device__ __forceinline__ void vm_load_so_agent(V &dst, const void *sbase, uint32_t voff)
{
    if constexpr (sizeof(V) == 4) asm volatile("global_load_dword %0, %1, %2 sc1" : "+v"(dst) : "v"(voff), "s"(sbase) : "memory");
    else if constexpr (sizeof(V) == 8) asm volatile("global_load_dwordx2 %0, %1, %2 sc1" : "+v"(dst) : "v"(voff), "s"(sbase) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, %2 sc1" : "+v"(dst) : "v"(voff), "s"(sbase) : "memory");
}
template <typename V>
__device__ __forceinline__ void vm_store_so(const V &src, void *sbase, uint32_t voff)
{
    if constexpr (sizeof(V) == 4) asm volatile("global_store_dword %0, %1, %2 sc1" ::"v"(voff), "v"(src), "s"(sbase) : "memory");
    else if constexpr (sizeof(V) == 8) asm volatile("global_store_dwordx2 %0, %1, %2 sc1" ::"v"(voff), "v"(src), "s"(sbase) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, %2 sc1\n\ts_nop 1" ::"v"(voff), "v"(src), "s"(sbase) : "memory");   // > 64 bits: hazard nop
}
template <int N, typename V>
__device__ __forceinline__ void vm_wait(V &slot)
{
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(slot) : "n"(N) : "memory");
}
template <typename T, int KC> struct RowReg { typedef typename RowVec<T, KC>::type type; };
template <typename T> struct RowReg<T, 1> { typedef T type; };
template <typename T, int KC> __device__ __forceinline__ T row_elem(const typename RowReg<T, KC>::type &v, int c)
{
    if constexpr (KC == 1) return v; else return v[c];
}

// ROLE 0 = the CHAIN wave: dot product, reduction, error, the item row's update - the serial part of every link and
// nothing else; it issues loads only, so its counted waits are exact and all HOT_RING rows stay in flight.  It leaves
// the errors in LDS (ebuf, 64 per chunk) and raises its turn word CTL_TOKEN after every MFSGD_WALK_SUB links
// (CTL_TOKEN = groups of MFSGD_WALK_SUB links done, counted over the whole segment).
// ROLE 1 = the APPLY wave, MFSGD_WALK_SUB links behind: with the errors known it runs the item row forward once more (the
// same operations on the same inputs: the same rows q_j) and forms, stores and publishes the users' new rows and biases.
// (Until the last step of round 3 it waited for whole chunks: ML-25M shape, 115-link segments, 7.31 -> 6.95 ms per epoch;
// 8 links: no better.)
// One update with a = 1 - alpha beta2 (numba_methods.py:21-22):  p' = a p + (alpha e) q,   q' = a q + (alpha e)(p' + z).
template <typename T, int KC, bool HASZ, int ROLE>
__device__ __forceinline__ void hot_walk(const GridView &g, int32_t run_lo, int32_t t_lo, int32_t nh, int32_t hot_item,
                                         int32_t c_base, T *__restrict__ P, T *__restrict__ b_u, const T *__restrict__ z,
                                         T b, T alpha, T beta1, T beta2, int k, T *qs, T *bis, int *ctl, T *ebuf,
                                         int32_t *err)
{
    constexpr int D = HotRing<T, KC, HASZ>::value;   // divides 64
    constexpr int L = HASZ ? 2 : 1;                  // loads per link
    constexpr int SUB = MFSGD_WALK_SUB, NS = 64 / SUB;   // the chain wave's turn word counts groups of SUB links
    static_assert(SUB == 8 || SUB == 16 || SUB == 32 || SUB == 64, "MFSGD_WALK_SUB divides 64");
    typedef typename RowReg<T, KC>::type Row;
    const int lane = threadIdx.x & 63;
    T *qrow = qs + (size_t)(hot_item - c_base) * k;
    T *bip = bis + (hot_item - c_base);
    const int32_t *hu = g.run_user + run_lo;     // the segment's users (single-rating runs) and ratings, straight from the grid
    const float *hv = g.val + t_lo;
    const T a = (T)1 - alpha * beta2, a1 = (T)1 - alpha * beta1;
    const uint32_t lane_off = (uint32_t)(KC * lane * sizeof(T)), row_bytes = (uint32_t)(k * sizeof(T));   // n_users * k * sizeof(T) < 2^32 (host check)
    T q[KC];
    row_get<T, KC>(qrow, lane, q);               // q_0: both waves start from the stripe's row
    T bi = *bip;
    if constexpr (ROLE == 1) {                   // the chain wave writes the row back only after this (short segments)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q[0]), "+v"(q[KC - 1]));
        __hip_atomic_store(sink_ptr(&ctl[CTL_TOKEN + 1], ctl, lane), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    // Metadata, 64 links per chunk, one link per lane (entries past the segment's end repeat its last link: valid
    // addresses, never applied): user ids and ratings are requested TWO chunks ahead, the users' biases - a gather that
    // needs the ids - one chunk ahead, so nothing in the loop waits for a load it has just issued.
    auto ids = [&](int32_t base, int32_t &uj, float &vj) {
        int32_t j = base + lane;
        j = j < nh ? j : nh - 1;
        vm_load(uj, hu + j);
        vm_load(vj, hv + j);
    };
    int32_t ujA = 0, ujB = 0, ujC = 0;
    float vA = 0, vB = 0, vC = 0;
    T buA = 0, buB = 0;
    ids(0, ujA, vA);
    ids(64, ujB, vB);
    ids(128, ujC, vC);
    vm_wait<0>(ujA);
    asm volatile("" : "+v"(ujB), "+v"(vA), "+v"(vB), "+v"(ujC), "+v"(vC));   // all six have arrived
    vm_load_agent(buA, b_u + ujA);
    vm_load_agent(buB, b_u + ujB);
    Row rp[D], rz[HASZ ? D : 1];
#pragma unroll
    for (int s = 0; s < D; ++s) {
        rp[s] = Row{};
        if constexpr (HASZ) rz[s] = Row{};
        const uint32_t ro = (uint32_t)__builtin_amdgcn_readlane(ujA, s) * row_bytes + lane_off;
        vm_load_so_agent(rp[s], P, ro);
        if constexpr (HASZ) vm_load_so(rz[s], z, ro);
    }
    // per-lane byte offset of every link's row: one readlane + one v_add per request in the loop
    uint32_t offA = (uint32_t)ujA * row_bytes, offB = (uint32_t)ujB * row_bytes;
    vm_wait<D * L>(buA);                       // the biases are older than the ring's rows
    asm volatile("" : "+v"(buB));
    if constexpr (ROLE == 0) __builtin_amdgcn_s_setprio(3);   // the item's chain: everything else in the block waits behind it
    int32_t chunk = 0;
    for (int32_t base = 0; base < nh; base += 64, ++chunk) {
        const int32_t nb = nh - base;                            // links left, this chunk included
        T restA = 0, evec = 0;
        T *eb = ebuf + (chunk % HOT_EBUFS) * 64;
        if constexpr (ROLE == 0) {
            restA = (T)vA - b - buA;
            // the error buffer of chunk c is free once the apply wave has read chunk c - HOT_EBUFS
            for (int spins = 0; __builtin_amdgcn_readfirstlane(__hip_atomic_load(&ctl[CTL_FRONT], __ATOMIC_RELAXED,
                                                                                 __HIP_MEMORY_SCOPE_WORKGROUP)) < chunk - (HOT_EBUFS - 1); ++spins) {
                if (spins >= SPIN_TICKET_MAX) { atomicOr(err, 4); break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        // the chunk is straight-line code: every lane index below is a constant, every load unconditional
#pragma unroll
        for (int li = 0; li < 64; ++li) {
            constexpr int DD = D;
            const int s = li % DD;
            // ROLE 1 also has a store per link in the queue: counting loads only makes its wait early, never late
            if constexpr (HASZ) { vm_wait<(DD - 1) * L>(rz[s]); asm volatile("" : "+v"(rp[s])); }
            else vm_wait<(DD - 1) * L>(rp[s]);
            if constexpr (ROLE == 1) {
                // the errors arrive MFSGD_WALK_SUB links at a time: this wave follows the chain that closely
                if (li % SUB == 0 && li < nb) {
                    for (int spins = 0; __builtin_amdgcn_readfirstlane(__hip_atomic_load(&ctl[CTL_TOKEN], __ATOMIC_RELAXED,
                                            __HIP_MEMORY_SCOPE_WORKGROUP)) <= chunk * NS + li / SUB; ++spins) {
                        if (spins >= SPIN_TICKET_MAX) { atomicOr(err, 4); break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    LDS_ACQUIRE();
                    evec = eb[lane];
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(evec));
                }
            }
            if (li < nb) {                                       // wave-uniform
                T ap[KC], aq[KC], pn[KC];
#pragma unroll
                for (int c = 0; c < KC; ++c) { ap[c] = a * row_elem<T, KC>(rp[s], c); aq[c] = a * q[c]; }
                T e;
                if constexpr (ROLE == 0) {
                    T part = q[0] * (HASZ ? row_elem<T, KC>(rp[s], 0) + row_elem<T, KC>(rz[s], 0) : row_elem<T, KC>(rp[s], 0));
#pragma unroll
                    for (int c = 1; c < KC; ++c)
                        part = fma_t(q[c], HASZ ? row_elem<T, KC>(rp[s], c) + row_elem<T, KC>(rz[s], c)
                                                : row_elem<T, KC>(rp[s], c), part);
                    T ri;
                    if constexpr (sizeof(T) == 4) ri = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(restA), li));
                    else ri = __shfl(restA, li, 64);
                    e = (ri - bi) - chain_sum(part);
                    bi = fma_t(alpha, e, a1 * bi);               // b_i' = b_i + alpha (e - beta1 b_i)
                    eb[li] = e;                                  // every lane, same word: one ds_write with an immediate offset
                } else {
                    if constexpr (sizeof(T) == 4) e = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(evec), li));
                    else e = __shfl(evec, li, 64);
                }
                const T ae = alpha * e;
#pragma unroll
                for (int c = 0; c < KC; ++c) {
                    pn[c] = fma_t(ae, q[c], ap[c]);
                    q[c] = fma_t(ae, HASZ ? pn[c] + row_elem<T, KC>(rz[s], c) : pn[c], aq[c]);
                }
                if constexpr (ROLE == 1) {
                    Row out;
                    if constexpr (KC == 1) out = pn[0];
                    else {
#pragma unroll
                        for (int c = 0; c < KC; ++c) out[c] = pn[c];
                    }
                    vm_store_so(out, P, (uint32_t)__builtin_amdgcn_readlane(offA, li) + lane_off);
                }
            }
            if constexpr (ROLE == 0 && SUB < 64) {
                if ((li + 1) % SUB == 0 && li != 63 && li < nb) {
                    LDS_RELEASE();
                    __hip_atomic_store(sink_ptr(&ctl[CTL_TOKEN], ctl, lane), chunk * NS + (li + 1) / SUB, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
#if MFSGD_WALK_FRONT >= 2
            if constexpr (ROLE == 1) {
                if (li == 31 && nb > 32) {   // half-chunk publish (see the end of the chunk)
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    const int32_t half_user = __builtin_amdgcn_readlane(ujA, 31);
                    __hip_atomic_store(sink_ptr(&ctl[CTL_FMIN], ctl, lane), half_user, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
#endif
            // the slot is free: request the row of the link HOT_RING further on
            const uint32_t ro = (uint32_t)(li + DD < 64 ? __builtin_amdgcn_readlane(offA, (li + DD) & 63)
                                                        : __builtin_amdgcn_readlane(offB, (li + DD) & 63)) + lane_off;
            vm_load_so_agent(rp[s], P, ro);
            if constexpr (HASZ) vm_load_so(rz[s], z, ro);
        }
        if constexpr (ROLE == 0) {
            LDS_RELEASE();
            __hip_atomic_store(sink_ptr(&ctl[CTL_TOKEN], ctl, lane), (chunk + 1) * NS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            // the users' biases (numba_methods.py:19); the error buffer is free again
            if (lane < nb) untracked_store(&b_u[ujA], buA + alpha * (evec - beta1 * buA));
#if MFSGD_WALK_FRONT
            // The chunk's users are done once these stores are: the frontier the gated rest waits behind moves up chunk by
            // chunk.  That costs this wave one drain of its ring per 64 links (the rows requested for the next chunk land
            // meanwhile; the chain wave, a chunk or more ahead, is not held up), and lets the runs of the segment's users
            // start while the segment is still being walked.
            if (nb > 64) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const int32_t last_user = __builtin_amdgcn_readlane(ujA, 63);
                __hip_atomic_store(sink_ptr(&ctl[CTL_FUSER], ctl, lane), last_user, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                LDS_RELEASE();
                __hip_atomic_store(sink_ptr(&ctl[CTL_FMIN], ctl, lane), last_user, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
#endif
            __hip_atomic_store(sink_ptr(&ctl[CTL_FRONT], ctl, lane), chunk + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        // The metadata moves up one chunk.  Everything copied here was requested at least a chunk ago - before ring rows
        // that have been waited for since - so it has arrived.
        ujA = ujB; vA = vB; buA = buB;
        ujB = ujC; vB = vC;
        offA = offB; offB = (uint32_t)ujB * row_bytes;
        asm volatile("" : "+v"(ujA), "+v"(vA), "+v"(buA), "+v"(ujB), "+v"(vB), "+v"(offA), "+v"(offB));   // the copies happen HERE, before the reloads
        vm_load_agent(buB, b_u + ujB);
        ids(base + 192, ujC, vC);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the ring's last requests land before their registers are reused
    if constexpr (ROLE == 0) {
        __builtin_amdgcn_s_setprio(0);
        // the item's row goes back to the stripe - once the apply wave has taken its copy of q_0
        for (int spins = 0; __builtin_amdgcn_readfirstlane(__hip_atomic_load(&ctl[CTL_TOKEN + 1], __ATOMIC_RELAXED,
                                                                             __HIP_MEMORY_SCOPE_WORKGROUP)) == 0; ++spins) {
            if (spins >= SPIN_TICKET_MAX) { atomicOr(err, 4); break; }
            __builtin_amdgcn_s_sleep(1);
        }
        row_put<T, KC>(qrow, lane, q);
        *sink_ptr(bip, ctl, lane) = bi;
    } else {
        // the segment's users are done (their stores have drained): publish
        __hip_atomic_store(sink_ptr(&ctl[CTL_FUSER], ctl, lane), 0x7fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        LDS_RELEASE();
        __hip_atomic_store(sink_ptr(&ctl[CTL_FMIN], ctl, lane), 0x7fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

// All threads of the workgroup call this; ends with a barrier after which every store of the block has been
// issued AND completed (vmcnt drained per wave).
// Runs [run_lo, run_lo + n_hot_blk) are the block's hot segments, one after the other (segment s: one single-rating run
// per user who rated hot item s of the stripe, by user; see mfsgd_grid_create_opt); the same users' other ratings
// follow in the rest of the block.  A user may sit in several segments and in the rest, so each of these PASSES is
// finished - stores drained, barrier - before the next starts.  FASTA: the hot segments go through the hot-item
// pipeline (phase_a_gram) when the launch gave it LDS row buffers, concurrently with the rest behind its user
// frontier; otherwise they are walked like any other run, through the items' tickets.
template <typename T, int KC, bool STAGE, int FAST, bool FASTA, bool SERIAL = false>
__device__ __forceinline__ void process_block(const GridView &g, int32_t r, int32_t c, T *P, T *Q, T *b_u, T *b_i, T b,
                                              const T *z, T alpha, T beta1, T beta2, int variant, int k, int *ctl,
                                              int *tick_done, T *qs, T *bis, int32_t *err, int *rt_u, int *rt_t,
                                              int32_t run_lo, int32_t run_hi, int32_t n_hot_blk, int32_t n_free_runs,
                                              char *abuf, uint32_t abuf_bytes, int n_abuf, int hot_walker, T *ebuf PF_PARAM)
{
    constexpr bool GATED = FASTA || SERIAL;   // builds in which hot segments may run beside the rest, behind the frontier
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    PF_T(b0);
    const int32_t c_base = c * g.ibp;
    int32_t width = (int32_t)(g.n_items - c_base);
    if (width > g.ibp) width = g.ibp;
    for (int i = tid; i < width; i += blockDim.x) tick_done[i] = 0;
    const bool uniq = FAST || g.user_unique;     // host side: FAST kernels are only launched on user-unique grids
    int32_t cnt[MAXSEG];                         // lengths of the block's hot segments (wave-uniform)
    // a stripe with ONE hot item needs no look at the per-segment counts
    const bool multi = __builtin_amdgcn_readfirstlane(ctl[CTL_HOT + 1]) >= 0;
#pragma unroll
    for (int s = 0; s < MAXSEG; ++s)
        cnt[s] = n_hot_blk > 0 && multi ? __builtin_amdgcn_readfirstlane(g.blk_seg_cnt[((size_t)r * g.Gc + c) * MAXSEG + s])
                                        : (s == 0 ? n_hot_blk : 0);
    // pass p < MAXSEG = hot segment p, pass MAXSEG = the rest: runs [ctl[CTL_BND + p], ctl[CTL_BND + p + 1]).  Every
    // thread writes the same words and reads back its own wave's stores (the LDS serves a wave in order).
    {
        int32_t pre = run_lo;
#pragma unroll
        for (int s = 0; s < MAXSEG; ++s) { ctl[CTL_BND + s] = pre; pre += cnt[s]; }
        ctl[CTL_BND + MAXSEG] = pre;
        ctl[CTL_BND + MAXSEG + 1] = run_hi;
    }
    bool gated = false;   // the pipeline walks the hot segments WHILE the other waves walk the rest behind its frontier
    int first_pass = 0;
    // the register-resident walker takes stripes with ONE hot item; the matrix-core pipeline needs its row buffers
    const bool serial = SERIAL && hot_walker && !multi && n_hot_blk > 0 && variant == 0 && nw >= 2;
    if constexpr (GATED) {
        if ((serial || (FASTA && n_abuf > 0)) && n_hot_blk > 0) {
            gated = true;
            first_pass = MAXSEG;
            // every thread, same values; made visible by the barrier behind the first run table below
            int32_t pre = run_lo, fmin = 0x7fffffff;
#pragma unroll
            for (int s = 0; s < MAXSEG; ++s) {
                ctl[CTL_TOKEN + s] = 0;
                ctl[CTL_FRONT + s] = 0;
                // users below the segment's first one have no rating in it; an empty segment holds nobody back
                const int32_t f0 = cnt[s] > 0 ? g.run_user[pre] - 1 : 0x7fffffff;
                ctl[CTL_FUSER + s] = f0;
                fmin = f0 < fmin ? f0 : fmin;
                pre += cnt[s];
            }
            ctl[CTL_FMIN] = fmin;
        }
    }
    if (uniq) {
    for (int pass = first_pass; pass <= MAXSEG; ++pass) {   // the hot segments (unless the pipeline took them), then the rest
    const int32_t seg_lo = __builtin_amdgcn_readfirstlane(ctl[CTL_BND + pass]);
    const int32_t seg_hi = __builtin_amdgcn_readfirstlane(ctl[CTL_BND + pass + 1]);
    const bool rest_pass = pass == MAXSEG;
    // one trip unless the pass has more than RT_CAP runs (and one trip for an empty rest behind pipelined segments)
    for (int32_t lo = seg_lo; lo < seg_hi || (gated && rest_pass && lo == seg_lo); lo += RT_CAP) {
    const int32_t nr = seg_hi - lo < RT_CAP ? seg_hi - lo : RT_CAP;
    // the chunk's run table in LDS: entries nr and nr+1 close the last run and form an empty run behind it.  Behind the
    // pipeline the rest starts with the runs of users WITHOUT a rating in a hot segment (n_free_runs of them): nothing
    // holds them back; the sign bit marks the others, who wait for the pipeline's user frontier (index() below)
    const int32_t gate_from = gated && rest_pass ? run_lo + n_hot_blk + n_free_runs : 0x7fffffff;
    for (int i = tid; i <= nr + 1; i += blockDim.x) {
        const int ii = i < nr ? i : nr;
        rt_t[i] = g.run_start[lo + ii];
        rt_u[i] = i < nr ? (g.run_user[lo + i] | (lo + i >= gate_from ? (int32_t)0x80000000 : 0)) : 0;
    }
    ctl[CTL_RUN] = 0;  // every thread, same value
    __syncthreads();
    PF_T(bs);
    PF_ADD(19, b0, bs);
    if constexpr (SERIAL) {
        if (gated && serial && lo == seg_lo && wave < 2) {   // first chunk of the rest: waves 0 and 1 walk the hot segment first
            const int32_t t_lo = __builtin_amdgcn_readfirstlane(g.run_start[run_lo]);
            const int32_t hot0 = __builtin_amdgcn_readfirstlane(ctl[CTL_HOT]);
            if (wave == 0)
                hot_walk<T, KC, FAST == 2, 0>(g, run_lo, t_lo, n_hot_blk, hot0, c_base, P, b_u, z, b, alpha, beta1, beta2, k, qs, bis,
                                              ctl, ebuf, err);
            else
                hot_walk<T, KC, FAST == 2, 1>(g, run_lo, t_lo, n_hot_blk, hot0, c_base, P, b_u, z, b, alpha, beta1, beta2, k, qs, bis,
                                              ctl, ebuf, err);
#ifdef MFSGD_PROF
            PF_T(hw1);
            if (wave == 1) { PF_ADD(31, b0, hw1); PF_INC(PF_ISSUE, 1); PF_INC(27, n_hot_blk); }   // block start -> segment published; links
            else PF_ADD(24, bs, hw1);                                                            // the chain wave's time in the segment
#endif
        }
    }
    if constexpr (FASTA) {
        if (gated && !serial && lo == seg_lo) {   // first chunk of the rest: waves 0 .. n_abuf-1 do the hot segments first
            const int32_t t_lo = __builtin_amdgcn_readfirstlane(g.run_start[run_lo]);   // single-rating runs: rating = t_lo + i
            if (multi)
                phase_a_gram<KC, true>(g, run_lo, t_lo, c_base, (float *)P, (float *)b_u, (float)b, (float)alpha,
                                       (float)beta1, (float)beta2, variant, (float *)qs, (float *)bis, (float *)abuf,
                                       abuf_bytes / 4, n_abuf < nw ? n_abuf : nw, ctl, err PF_PASS);
            else
                phase_a_gram<KC, false>(g, run_lo, t_lo, c_base, (float *)P, (float *)b_u, (float)b, (float)alpha,
                                        (float)beta1, (float)beta2, variant, (float *)qs, (float *)bis, (float *)abuf,
                                        abuf_bytes / 4, n_abuf < nw ? n_abuf : nw, ctl, err PF_PASS);
        }
    }
    {
        // A wave binds itself to TWO runs beyond the one it walks (their head loads are in flight), and claims the
        // next only after the walk.  Claiming further ahead (the index words then came from HBM, two runs early)
        // cost 7% of the epoch: a run claimed by a wave that is held up - a long run, a wait for an item - sits
        // idle, and every later rating of its items waits behind it.
        // The dequeue is branch-free on purpose: with `if (lane == 0) atomicAdd` hipcc threads lanes 1..63 around
        // the branch and re-enters the loop body without lane 0 (seen in the ISA: the run was walked twice).
        // Runs are handed out in stored order and a wave finishes a run before it walks the next one it claimed,
        // so the earliest unfinished rating is always being walked: ticket waits cannot deadlock.
        auto claim = [&]() {
            PF_T(c0);
            const int32_t jj = __builtin_amdgcn_readlane(atomicAdd(sink_ptr(&ctl[CTL_RUN], ctl, lane), 1), 0);
            PF_T(c1);
            PF_ADD(28, c0, c1);
            return jj;
        };
        auto index = [&](int32_t jj, RunIdx &x) {
            const int32_t ii = jj < nr ? jj : nr;
            x.j = jj; x.u = rt_u[ii]; x.t0 = rt_t[ii]; x.t1 = rt_t[ii + 1];
            if constexpr (GATED) {
                PF_T(i0);
                const int32_t w = __builtin_amdgcn_readfirstlane(x.u);
                PF_T(i1);
                PF_ADD(29, i0, i1);
                x.u = w & 0x7fffffff;
                if (w < 0) {   // the user's row may be loaded once every hot segment is done up to this user
                    const int32_t u = w & 0x7fffffff;
                    for (int spins = 0; __builtin_amdgcn_readfirstlane(__hip_atomic_load(
                                            &ctl[CTL_FMIN], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < u; ++spins) {
                        if (spins >= SPIN_TICKET_MAX) { atomicOr(err, 4); break; }
                        __builtin_amdgcn_s_sleep(4);
                    }
                    LDS_ACQUIRE();
                    PF_T(i2);
                    PF_ADD(30, i1, i2);
                }
            }
        };
        // The run loop is unrolled three times and the three heads change roles instead of being copied
        // (cur = nxt; nxt = far cost ~16 v_mov per run): the kernel is bound by VALU issue,
        // and this alone took the netflix epoch from 31.9 to 28.7 ms.
        auto run_loop = [&](auto var_tag) {
        constexpr int VAR = decltype(var_tag)::value;
        RunHead<T, KC> h0, h1, h2;
        RunIdx x;
        int32_t j0 = claim();
        index(j0, x);
        load_run<T, KC, FAST>(g, x, P, b_u, z, k, lane, h0);
        int32_t j1 = claim();
        index(j1, x);
        load_run<T, KC, FAST>(g, x, P, b_u, z, k, lane, h1);
        int32_t j2 = claim();
        index(j2, x);
        PF_T(b1);
        PF_ADD(PF_PROLOGUE, b0, b1);
        // one step: the head of the run claimed last goes into the free slot LOAD, the run in WALK is walked,
        // the next run is claimed (only now: DESIGN.md section 3.1) and its index words are read
#define MF_STEP(WALK, LOAD, JW, JN)                                                                                     \
        if (JW >= nr) break;                                                                                          \
        {                                                                                                             \
        PF_T(s0);                                                                                                     \
        load_run<T, KC, FAST>(g, x, P, b_u, z, k, lane, LOAD);                                                         \
        PF_T(s1);                                                                                                     \
        walk_run<T, KC, STAGE, FAST, VAR>(g, WALK, c_base, P, Q, b_u, b_i, b, alpha, beta1, beta2, variant, k, ctl,     \
                                     tick_done, qs, bis, lane, err PF_PASS);                                          \
        PF_T(s2);                                                                                                     \
        JN = claim();                                                                                                 \
        index(JN, x);                                                                                                 \
        PF_T(s3);                                                                                                     \
        PF_ADD(32, s0, s1); PF_ADD(33, s1, s2); PF_ADD(34, s0, s3);                                                     \
        }
        for (;;) {
            int32_t ja, jb, jc;
            MF_STEP(h0, h2, j0, ja)
            MF_STEP(h1, h0, j1, jb)
            MF_STEP(h2, h1, j2, jc)
            j0 = ja; j1 = jb; j2 = jc;
        }
#undef MF_STEP
        };
        // the straight-line builds get the update order as a constant (two instantiations of the run loop)
        if constexpr (FAST != 0) {
            if (variant == 0) run_loop(std::integral_constant<int, 0>{});
            else run_loop(std::integral_constant<int, 1>{});
        } else {
            run_loop(std::integral_constant<int, -1>{});
        }
    }
    if (lo + RT_CAP < seg_hi || !rest_pass) {   // another chunk or another pass follows: new table,
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // and the segment's user rows are stored before the same users' runs load them
        __syncthreads();
    }
    }   // chunks
    }   // passes
    }   // uniq
    if (!FAST) {
    if (!uniq) {
        ctl[CTL_RUN] = 0;
        __syncthreads();
        // a user may own several runs of this block: all of them go to one wave, in stored order
        for (int32_t j = run_lo; j < run_hi; ++j) {
            if ((int)(mix32((uint32_t)g.run_user[j]) % (uint32_t)nw) != wave) continue;
            RunHead<T, KC> cur;
            RunIdx x;
            run_index(g, j, x);
            load_run<T, KC, false>(g, x, P, b_u, z, k, lane, cur);
            walk_run<T, KC, STAGE, false>(g, cur, c_base, P, Q, b_u, b_i, b, alpha, beta1, beta2, variant, k, ctl,
                                          tick_done, qs, bis, lane, err PF_PASS);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the user's next run reloads this P row
        }
    }
    }
    PF_T(b2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PF_T(b3);
    __syncthreads();
    PF_T(b4);
    PF_ADD(PF_DRAIN, b2, b3);
    PF_ADD(PF_BARRIER, b3, b4);
    PF_ADD(PF_BLOCK, b0, b4);
}

// Register budget: 8 waves per SIMD (64 VGPRs) lets two 16-wave workgroups share a CU, which is what k <= 128 in
// fp32 wants; rows of more than 128 floats (or any doubles) spill at 64 VGPRs, and their stripes rarely leave
// LDS for a second workgroup anyway: 4 waves per SIMD (128 VGPRs), one workgroup per CU.
#define WAVES_PER_SIMD(T, KC) ((sizeof(T) == 8 || (KC) >= 3) ? 4 : 8)

template <typename T, bool STAGE>
__device__ __forceinline__ void stripe_load(const GridView &g, int32_t c, const T *Q, const T *b_i, int k, T *qs, T *bis)
{
    if (!STAGE) return;
    const int32_t c_base = c * g.ibp;
    int32_t width = (int32_t)(g.n_items - c_base);
    if (width > g.ibp) width = g.ibp;
    const T *src = Q + (int64_t)c_base * k;
    for (int64_t i = threadIdx.x; i < (int64_t)width * k; i += blockDim.x) qs[i] = src[i];
    for (int i = threadIdx.x; i < width; i += blockDim.x) bis[i] = b_i[c_base + i];
    __syncthreads();
}

template <typename T, bool STAGE>
__device__ __forceinline__ void stripe_store(const GridView &g, int32_t c, T *Q, T *b_i, int k, const T *qs, const T *bis)
{
    if (!STAGE) return;
    __syncthreads();
    const int32_t c_base = c * g.ibp;
    int32_t width = (int32_t)(g.n_items - c_base);
    if (width > g.ibp) width = g.ibp;
    T *dst = Q + (int64_t)c_base * k;
    for (int64_t i = threadIdx.x; i < (int64_t)width * k; i += blockDim.x) dst[i] = qs[i];
    for (int i = threadIdx.x; i < width; i += blockDim.x) b_i[c_base + i] = bis[i];
    __syncthreads();
}

// schedule 0: one launch per anti-diagonal d; workgroup w does block (r0 + w, d - r0 - w)
template <typename T, int KC, bool STAGE>
__global__ __launch_bounds__(1024, WAVES_PER_SIMD(T, KC)) void k_epoch_fixed(GridView g, T *P, T *Q, T *b_u, T *b_i, T b, const T *z, T alpha,
                                                      T beta1, T beta2, int variant, int k, int d, int r0,
                                                      LdsPlan<T> lp, SchedView s)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int *ctl = (int *)(smem + lp.ctl);
    int *tick_done = (int *)(smem + lp.tick);
    T *qs = (T *)(smem + lp.qs), *bis = (T *)(smem + lp.bis);
    const int32_t r = r0 + (int32_t)blockIdx.x, c = d - r;
    if (s.order_log) {
        // every thread, same values.  row-major position: the fixed schedule's log is the row-major order itself
        s.order_log[2 * (s.log_base + r * g.Gc + c)] = r;
        s.order_log[2 * (s.log_base + r * g.Gc + c) + 1] = c;
    }
    if (g.blk_run_ptr[r * g.Gc + c] == g.blk_run_ptr[r * g.Gc + c + 1]) return;
    stripe_load<T, STAGE>(g, c, Q, b_i, k, qs, bis);
#ifdef MFSGD_PROF
    unsigned long long pf[PF_SLOTS] = {};
#endif
    const int32_t n_hot_blk = __builtin_amdgcn_readfirstlane(g.blk_hot_cnt[r * g.Gc + c]);
    if (threadIdx.x < MAXSEG) ctl[CTL_HOT + threadIdx.x] = g.col_hot[(size_t)c * MAXSEG + threadIdx.x];
    __syncthreads();
    process_block<T, KC, STAGE, false, false>(g, r, c, P, Q, b_u, b_i, b, z, alpha, beta1, beta2, variant, k, ctl, tick_done,
                                              qs, bis, s.err, (int *)(smem + lp.rt_u), (int *)(smem + lp.rt_t),
                                              __builtin_amdgcn_readfirstlane(g.blk_run_ptr[r * g.Gc + c]),
                                              __builtin_amdgcn_readfirstlane(g.blk_run_ptr[r * g.Gc + c + 1]), n_hot_blk, 0,
                                              nullptr, 0, 0, 0, nullptr PF_PASS);
    stripe_store<T, STAGE>(g, c, Q, b_i, k, qs, bis);
}

// schedule 1: persistent workgroups, column tenure + row locks
// WPS = waves per SIMD the kernel is built for: 8 (64 VGPRs) lets two 16-wave workgroups share a CU, 4 (128 VGPRs)
// is one workgroup per CU without the register squeeze - 3-8% faster whenever the grid has no more stripes than
// the GPU has CUs (measured: netflix 33.7 -> 32.3 ms, frozen 44.0 -> 40.7 ms), so both are built for k <= 128 fp32.
// HOTK = how this build walks hot segments: 0 item tickets, 1 the matrix-core pipeline (phase_a_gram), 2 the walker
// (hot_walk).  Separate builds: with both in one kernel the register allocator spills 36 VGPRs in the rest's walk
// (8 with either alone).
template <typename T, int KC, bool STAGE, int FAST, int WPS, int HOTK>
__global__ __launch_bounds__(1024, WPS) void k_epoch_free(GridView g, T *P, T *Q, T *b_u, T *b_i, T b, const T *z, T alpha,
                                                     T beta1, T beta2, int variant, int k, LdsPlan<T> lp, SchedView s)
{
    // the hot-item pipeline needs fp32 MFMA, k = 64 / 128 / 256, 128 registers and row buffers in LDS (lp.n_abuf > 0)
    constexpr bool FASTA = HOTK == 1 && FAST == 1 && STAGE && sizeof(T) == 4 && (KC <= 2 || KC == 4) && WPS == 4;
    // the register-resident hot-item walker: every straight-line build with the stripe in LDS and 128 registers
    constexpr bool SERIAL = HOTK == 2 && FAST != 0 && STAGE && WPS == 4 && KC * sizeof(T) <= 16 && KC != 3;   // a row slot is one 4/8/16-byte load per lane
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int *ctl = (int *)(smem + lp.ctl);
    int *tick_done = (int *)(smem + lp.tick);
    unsigned char *row_done = (unsigned char *)(smem + lp.rowdone);
    int *rowrun = (int *)(smem + lp.rowrun);   // [Gr][4] of block (r, c): first run, end, ratings in hot segments, free runs
    T *qs = (T *)(smem + lp.qs), *bis = (T *)(smem + lp.bis);
    const int tid = threadIdx.x, lane = tid & 63;
    // The scheduler runs on wave 0 with all 64 lanes executing the same (wave-uniform) code; the branch that
    // selects wave 0 is scalar, and nothing inside is predicated per lane (sink_ptr).
    const bool sched_wave = __builtin_amdgcn_readfirstlane(tid >> 6) == 0;
    const int32_t passes = s.passes;   // visits per block in this launch (1 unless schedule 2)
#ifdef MFSGD_PROF
    unsigned long long pf[PF_SLOTS] = {};
    PF_T(k0);
#endif
    for (;;) {
        if (sched_wave) {
            const int32_t ci = __builtin_amdgcn_readlane(
                __hip_atomic_fetch_add(sink_ptr(&s.counters[0], ctl, lane), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), 0);
            int32_t col = -1;
            if (ci < g.n_cols) {
                col = __builtin_amdgcn_readfirstlane(g.col_order[ci]);
                s.col_clock[2 * col] = (long long)__builtin_amdgcn_s_memrealtime();
            }
            ctl[CTL_COL] = col;
            ctl[CTL_REMAIN] = 0;
        }
        __syncthreads();
        const int32_t c = __builtin_amdgcn_readfirstlane(ctl[CTL_COL]);
        if (c < 0) break;
        // rows with an empty block need no visit
        int32_t mine = 0;
        for (int r = tid; r < g.Gr; r += blockDim.x) {
            const int32_t lo = g.blk_run_ptr[r * g.Gc + c], hi = g.blk_run_ptr[r * g.Gc + c + 1];
            const bool empty = lo == hi;
            rowrun[4 * r] = lo;
            rowrun[4 * r + 1] = hi;
            rowrun[4 * r + 2] = g.blk_hot_cnt[r * g.Gc + c];
            rowrun[4 * r + 3] = g.blk_free_runs[r * g.Gc + c];
            row_done[r] = empty ? passes : 0;    // visits of block (r, c) so far in this launch
            mine += empty ? 0 : passes;
        }
        atomicAdd(&ctl[CTL_REMAIN], mine);
        if (tid < MAXSEG) ctl[CTL_HOT + tid] = g.col_hot[(size_t)c * MAXSEG + tid];   // the stripe's hot items
        stripe_load<T, STAGE>(g, c, Q, b_i, k, qs, bis);
        __syncthreads();
        int32_t held = -1;   // scheduler wave: the row of the block just finished, released AFTER the next one is found

        for (;;) {
            if (sched_wave) {
                int32_t found = -1;
                const int32_t remain = __builtin_amdgcn_readfirstlane(ctl[CTL_REMAIN]);
                auto release = [&]() {
                    // every wave drained its stores before the barrier that ended process_block
                    if (held >= 0) {
                        if constexpr (FAST == 0) MF_AGENT_RELEASE();
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        __hip_atomic_store(&s.row_lock[held], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        held = -1;
                    }
                };
                for (int sweeps = 0; remain > 0 && found < 0; ++sweeps) {
                    if (sweeps >= SPIN_LOCK_SWEEPS) { atomicOr(s.err, 2); break; }
                    // The search starts at a pseudo-random row every time (workgroups that look at the same stretch see the
                    // same free rows and would all race for the first), 64 rows per step: every lane looks at one row
                    // of the rotation that starts at `pos`
                    const int32_t pos = (int32_t)(mix32((uint32_t)c * 0x9E3779B9u + (uint32_t)(remain + sweeps) * 0x85EBCA6Bu + s.seed) %
                                                  (uint32_t)g.Gr);
                    // peek at the lock words with one coalesced agent-scope load per 64 rows (two cache lines from L2; as
                    // a vector atomic this was 64 read-modify-writes on those lines per peek, and with short blocks the
                    // lock search grew to 40% of the epoch) and collect the first free-looking rows of the rotation.
                    // One pass per launch: the first one is taken.  Several passes in one launch (schedule 2): of the
                    // FIRST TWO free-looking rows the one this column has visited less often goes first.  That is the
                    // reference's rule - "a block of minimal update count among the free rows x free columns",
                    // RayBlockScheduler.py:50-58, where with n_threads + 1 stripes a worker sees one or two free rows and
                    // columns - and like there a block can get its second visit while others still wait for their
                    // first: the passes overlap (profiles/r3_pass_overlap.txt compares the two schedulers' visit times).
                    const int want = passes > 1 ? s.pick : 1;
                    int32_t r1 = -1, r2 = -1, v1 = 0x7fffffff, v2 = 0x7fffffff, got = 0;   // least and second least visited
                    for (int32_t base = 0; base < g.Gr && got < want; base += 64) {
                        int32_t r = pos + base + lane;
                        while (r >= g.Gr) r -= g.Gr;
                        const int32_t visits = row_done[r];
                        bool cand = base + lane < g.Gr && visits < passes;
                        if (__ballot(cand) == 0) continue;
                        const int32_t lockv = __hip_atomic_load(&s.row_lock[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        cand = cand && lockv == 0;
                        unsigned long long mask = __ballot(cand);
                        while (mask != 0 && got < want) {
                            const int l = __builtin_ctzll(mask);
                            mask &= mask - 1;
                            ++got;
                            const int32_t rc = __builtin_amdgcn_readlane(r, l), vc = __builtin_amdgcn_readlane(visits, l);
                            if (vc < v1) { r2 = r1; v2 = v1; r1 = rc; v1 = vc; }      // ties keep the rotation's order
                            else if (vc < v2) { r2 = rc; v2 = vc; }
                        }
                    }
#ifdef MFSGD_SCHED_DBG
                    atomicAdd(sink_ptr(&s.counters[2], ctl, lane), 1);                       // sweeps
                    if (r1 < 0) atomicAdd(sink_ptr(&s.counters[3], ctl, lane), 1);           // ... that saw no candidate
                    if (r1 >= 0 && r2 < 0) atomicAdd(sink_ptr(&s.counters[4], ctl, lane), 1);  // ... exactly one
                    if (r2 >= 0 && v2 != v1) atomicAdd(sink_ptr(&s.counters[6], ctl, lane), 1);  // two with different counts
#endif
                    for (int attempt = 0; attempt < 2 && found < 0; ++attempt) {
                        const int32_t rc = attempt == 0 ? r1 : r2;
                        if (rc < 0) break;
                        // one real CAS (lane 0); the other lanes hit their own sink words
                        int32_t expect = 0;
                        int32_t *w = sink_ptr(&s.row_lock[rc], ctl, lane);
                        if (lane != 0) expect = 0x7fffffff;   // value select, not control flow
                        __hip_atomic_compare_exchange_strong(w, &expect, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_AGENT);
                        if (__builtin_amdgcn_readlane(expect, 0) == 0) found = rc;   // unchanged <=> swapped
#ifdef MFSGD_SCHED_DBG
                        else atomicAdd(sink_ptr(&s.counters[5], ctl, lane), 1);             // CAS lost
#endif
                    }
                    if (found < 0) {
                        release();   // never wait for a row while holding one
                        __builtin_amdgcn_s_sleep(32);
                    }
                }
                if (found >= 0) {
                    row_done[found] = (unsigned char)(row_done[found] + 1);   // every lane, same value
                    ctl[CTL_REMAIN] = remain - 1;
                    // acquire: drop this CU's L1 so that the previous holder's P rows are re-read
                    if constexpr (FAST == 0) MF_AGENT_ACQUIRE();
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (s.order_log) {
                        const int32_t slot = __builtin_amdgcn_readlane(
                            __hip_atomic_fetch_add(sink_ptr(&s.counters[1], ctl, lane), 1, __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT), 0);
                        s.order_log[2 * (s.log_base + slot)] = found;
                        s.order_log[2 * (s.log_base + slot) + 1] = c;
                    }
                }
                ctl[CTL_ROW] = found;
            }
            __syncthreads();
            const int32_t r = __builtin_amdgcn_readfirstlane(ctl[CTL_ROW]);
            if (sched_wave) {
                // the finished block's row goes back while the other waves already start on the next block
                if (held >= 0) {
                    if constexpr (FAST == 0) MF_AGENT_RELEASE();
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __hip_atomic_store(&s.row_lock[held], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                held = r;
            }
            if (r < 0) break;
            process_block<T, KC, STAGE, FAST, FASTA, SERIAL>(g, r, c, P, Q, b_u, b_i, b, z, alpha, beta1, beta2, variant, k,
                                                     ctl, tick_done, qs, bis, s.err, (int *)(smem + lp.rt_u),
                                                     (int *)(smem + lp.rt_t), __builtin_amdgcn_readfirstlane(rowrun[4 * r]),
                                                     __builtin_amdgcn_readfirstlane(rowrun[4 * r + 1]),
                                                     __builtin_amdgcn_readfirstlane(rowrun[4 * r + 2]),
                                                     __builtin_amdgcn_readfirstlane(rowrun[4 * r + 3]),
                                                     smem + lp.abuf, lp.abuf_bytes, lp.n_abuf, lp.hot_walker,
                                                     (T *)(smem + lp.ebuf) PF_PASS);
        }
        stripe_store<T, STAGE>(g, c, Q, b_i, k, qs, bis);
        if (sched_wave) s.col_clock[2 * c + 1] = (long long)__builtin_amdgcn_s_memrealtime();
    }
#ifdef MFSGD_PROF
    PF_T(k1);
    pf[PF_TENURE] = k1 - k0;
    pf[PF_SCHED] = pf[PF_TENURE] - pf[PF_BLOCK];
    unsigned long long *out = (unsigned long long *)(s.col_clock + 2 * g.Gc);
    for (int i = 0; i < PF_SLOTS; ++i)
        atomicAdd((unsigned long long *)sink_ptr(out + i, ctl, lane), pf[i]);
#endif
}

template <typename T, int KC, bool STAGE>
int launch_epochs(mfsgd_grid *G, const GridView &gv, T *P, T *Q, T *b_u, T *b_i, T b, const T *z, T alpha, T beta1,
                  T beta2, int variant, int k, int iters, int schedule, uint64_t seed, int n_wg, int waves,
                  int32_t *order_log, hipStream_t st)
{
    const bool dynamic = schedule >= 1;
    LdsPlan<T> lp = lds_plan<T>(gv.ibp, gv.Gr, k, STAGE, dynamic);
    if (waves <= 0) waves = 16;
    if (waves > 16) waves = 16;
    const int threads = waves * 64;
    const int64_t nb = (int64_t)gv.Gr * gv.Gc;
    if (!dynamic) {
        auto kern = k_epoch_fixed<T, KC, STAGE>;
        MF_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lp.total));
        for (int it = 0; it < iters; ++it) {
            SchedView s{G->row_lock, G->counters, order_log, (int32_t)(it * nb), (uint32_t)seed, G->counters + 8, G->col_clock, 1, 1};
            for (int d = 0; d <= gv.Gr + gv.Gc - 2; ++d) {
                const int r0 = std::max(0, d - (gv.Gc - 1)), r1 = std::min(gv.Gr - 1, d);
                hipLaunchKernelGGL(kern, dim3(r1 - r0 + 1), dim3(threads), lp.total, st, gv, P, Q, b_u, b_i, b, z,
                                   alpha, beta1, beta2, variant, k, d, r0, lp, s);
                MF_LAUNCH_CHECK();
            }
        }
        return MFSGD_OK;
    }
    // 0: general kernel; 1: k a multiple of 64, user-unique grid, no implicit term; 2: the same with the frozen term z
    const int fast = k == 64 * KC && gv.user_unique ? (z == nullptr ? 1 : 2) : 0;
    hipDeviceProp_t prop;
    MF_HIP(hipGetDeviceProperties(&prop, G->device));
    // two workgroups per CU only pay when there are more stripes than CUs and two stripes fit the LDS of one
    constexpr bool can8 = WAVES_PER_SIMD(T, KC) == 8;
    // ... unless the grid has hot segments and their pipeline can run: it needs the 128-register build, and hot segments
    // that go through the tickets cost more than the second workgroup gains
    // hot segments: mode 2 (default) = the register-resident walker for stripes with one hot item (every straight-line
    // build), the matrix-core pipeline for stripes with several; mode 1 = the pipeline only; mode 0 or hot_bufs 0 = tickets
    const bool walker = fast != 0 && STAGE && G->info.n_hot > 0 && G->hot_bufs > 0 && G->hot_mode == 2 &&
                        G->info.n_hot_items == G->info.n_hot_stripes && KC * sizeof(T) <= 16 && KC != 3 &&
                        (uint64_t)gv.n_users * (uint64_t)k * sizeof(T) < ((uint64_t)1 << 32);   // 32-bit row offsets in the walker
    const bool pipe = !walker && G->hot_mode >= 1 &&
                      fast == 1 && STAGE && sizeof(T) == 4 && (KC <= 2 || KC == 4) && G->info.n_hot > 0 && G->hot_bufs > 0 &&
                      lds_plan<T>(gv.ibp, gv.Gr, k, STAGE, dynamic, G->hot_bufs).n_abuf >= 2;   // one wave alone is no pipeline (ExplicitMF._pipeline_runs applies the same rule)
    const bool wps8 = can8 && !pipe && !walker && G->n_cols_nonempty > prop.multiProcessorCount && 2 * (size_t)lp.total <= (size_t)160 * 1024;
    auto kern = fast == 1 ? k_epoch_free<T, KC, STAGE, 1, 4, 0> : fast == 2 ? k_epoch_free<T, KC, STAGE, 2, 4, 0> : k_epoch_free<T, KC, STAGE, 0, 4, 0>;
    if constexpr (STAGE && sizeof(T) == 4 && (KC <= 2 || KC == 4)) {
        if (pipe) kern = k_epoch_free<T, KC, STAGE, 1, 4, 1>;
    }
    if constexpr (STAGE && KC * sizeof(T) <= 16 && KC != 3) {
        if (walker) kern = fast == 1 ? k_epoch_free<T, KC, STAGE, 1, 4, 2> : k_epoch_free<T, KC, STAGE, 2, 4, 2>;
    }
    if constexpr (can8) {
        if (wps8) kern = fast == 1 ? k_epoch_free<T, KC, STAGE, 1, 8, 0> : fast == 2 ? k_epoch_free<T, KC, STAGE, 2, 8, 0> : k_epoch_free<T, KC, STAGE, 0, 8, 0>;
    }
    // the hot-item pipeline (phase_a_gram): fp32, k = 64, 128 or 256, the 128-register build, and as many row buffers as
    // the LDS left by the stripe holds (at most G->hot_bufs); without a buffer the hot segments take the ticket path
    if (pipe) lp = lds_plan<T>(gv.ibp, gv.Gr, k, STAGE, dynamic, G->hot_bufs);
    lp.hot_walker = walker ? 1 : 0;
    MF_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lp.total));
    if (n_wg <= 0) {
        int per_cu = (int)((size_t)160 * 1024 / std::max<size_t>(lp.total, 1));
        per_cu = std::max(1, std::min(per_cu, std::max(1, (wps8 ? 32 : 16) / waves)));
        n_wg = prop.multiProcessorCount * per_cu;
    }
    // schedule 2: all `iters` passes in ONE launch, no barrier between them - every workgroup keeps its item stripe and
    // visits each user stripe `iters` times, least visited of two first (k_epoch_free) - when every non-empty stripe has a
    // workgroup of its own; with more stripes than workgroups a tenure of several passes would give a stripe all its
    // visits in one burst, so such grids run pass by pass like schedule 1
    const bool all_resident = G->n_cols_nonempty <= n_wg;
    n_wg = std::max(1, std::min(n_wg, G->n_cols_nonempty > 0 ? G->n_cols_nonempty : 1));
    const int per_launch = schedule == 2 && all_resident ? MAX_PASSES_PER_LAUNCH : 1;
    for (int it = 0; it < iters; it += per_launch) {
        const int passes = std::min(per_launch, iters - it);
        MF_HIP(hipMemsetAsync(G->counters, 0, 32, st));
        MF_HIP(hipMemsetAsync(G->row_lock, 0, sizeof(int32_t) * (((size_t)gv.Gr + 3) & ~(size_t)3), st));
        SchedView s{G->row_lock, G->counters, order_log, (int32_t)(it * nb), (uint32_t)(seed + 0x9E37u * it), G->counters + 8, G->col_clock,
                    passes, G->pick_width};
        hipLaunchKernelGGL(kern, dim3(n_wg), dim3(threads), lp.total, st, gv, P, Q, b_u, b_i, b, z, alpha, beta1, beta2,
                           variant, k, lp, s);
        MF_LAUNCH_CHECK();
    }
    return MFSGD_OK;
}

template <typename T>
int epoch_impl(mfsgd_grid *G, T *P, T *Q, T *b_u, T *b_i, T b, const T *z, T alpha, T beta1, T beta2, int32_t variant,
               int32_t implicit, int32_t iters, int32_t schedule, uint64_t seed, int32_t n_wg, int32_t waves,
               int32_t *order_log, int32_t k, void *stream)
{
    MF_ARG(G && P && Q && b_u && b_i, "epoch: null pointer");
    MF_ARG(k >= 1 && k <= 256, "epoch: k=%d outside [1,256]", k);
    MF_ARG(variant == 0 || variant == 1, "epoch: variant must be 0 or 1");
    MF_ARG(implicit == 0 || (implicit == 1 && z), "epoch: implicit must be 0, or 1 with z (2 needs the serial pass)");
    MF_ARG(schedule >= 0 && schedule <= 2, "epoch: schedule must be 0 (fixed), 1 (free-block) or 2 (free-block, passes overlapped)");
    MF_ARG(iters >= 0, "epoch: iters < 0");
    if (implicit == 0) z = nullptr;
    const mfsgd_grid_info &in = G->info;
    if (in.n == 0 || iters == 0) return MFSGD_OK;
    GridView gv{G->blk_run_ptr, G->run_start, G->run_user, G->item, G->tick, G->val, G->col_order,
                in.Gr, in.Gc, in.ubp, in.ibp, in.user_unique, G->n_cols_nonempty, in.n_users, in.n_items,
                (int32_t)in.n_runs, G->col_hot, G->blk_hot_cnt, G->blk_seg_cnt, G->blk_free_runs};
    const bool dynamic = schedule >= 1;
    const size_t cap = LDS_CAP;
    const bool stage = lds_plan<T>(in.ibp, in.Gr, k, true, dynamic).total <= cap;
    if (!stage && lds_plan<T>(in.ibp, in.Gr, k, false, dynamic).total > cap) {
        mfsgd_set_error("epoch: item stripes of %d items (or %d row stripes) do not fit LDS; use a finer grid",
                        in.ibp, in.Gr);
        return MFSGD_ERR_GRID;
    }
    hipStream_t st = as_stream(stream);
    const int kc = (k + 63) / 64;
#define MF_GO(KC_)                                                                                                    \
    return stage ? launch_epochs<T, KC_, true>(G, gv, P, Q, b_u, b_i, b, z, alpha, beta1, beta2, variant, k, iters,   \
                                               schedule, seed, n_wg, waves, order_log, st)                            \
                 : launch_epochs<T, KC_, false>(G, gv, P, Q, b_u, b_i, b, z, alpha, beta1, beta2, variant, k, iters,  \
                                                schedule, seed, n_wg, waves, order_log, st)
#ifdef MFSGD_DEV_KC2      // development builds: only the float32 k = 65..128 kernels (a tenth of the compile time)
    if constexpr (sizeof(T) == 4) { if (kc == 2) { MF_GO(2); } }
    mfsgd_set_error("epoch: development build (MFSGD_DEV_KC2) has float32 k = 65..128 only");
    return MFSGD_ERR_UNSUPPORTED;
#else
    switch (kc) {
        case 1: MF_GO(1);
        case 2: MF_GO(2);
        case 3: MF_GO(3);
        default: MF_GO(4);
    }
#endif
#undef MF_GO
}

}  // namespace

extern "C" {

void mfsgd_grid_destroy(mfsgd_grid *g)
{
    if (!g) return;
    void *ptrs[] = {g->blk_run_ptr, g->run_start, g->run_user, g->item, g->tick, g->val, g->col_order, g->row_lock,
                    g->counters, g->col_clock, g->col_hot, g->blk_hot_cnt, g->blk_seg_cnt, g->blk_free_runs};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    delete g;
}

int mfsgd_grid_status(mfsgd_grid *g, void *stream)
{
    MF_ARG(g, "grid_status: null grid");
    int32_t h = 0;
    hipStream_t st = as_stream(stream);
    MF_HIP(hipMemcpyAsync(&h, g->counters + 8, 4, hipMemcpyDeviceToHost, st));
    MF_HIP(hipStreamSynchronize(st));
#ifdef MFSGD_SCHED_DBG
    {
        int32_t c[8];
        MF_HIP(hipMemcpy(c, g->counters, 32, hipMemcpyDeviceToHost));
        fprintf(stderr, "sched dbg (last launch): blocks %d, sweeps %d, no candidate %d, one candidate %d, CAS lost %d, two with different counts %d\n",
                c[1], c[2], c[3], c[4], c[5], c[6]);
    }
#endif
    if (h) {
        MF_HIP(hipMemsetAsync(g->counters + 8, 0, 4, st));
        mfsgd_set_error("epoch kernel gave up waiting (%s%s%s): results of that run are invalid",
                        (h & 1) ? "item ticket " : "", (h & 2) ? "row lock " : "",
                        (h & 4) ? "hot-item pipeline: user frontier / turn / publish order" : "");
        return MFSGD_ERR_GRID;
    }
    return MFSGD_OK;
}

int64_t mfsgd_lds_bytes(int32_t ibp, int32_t Gr, int32_t k, int32_t elem_size, int32_t hot_bufs)
{
    if (ibp < 1 || Gr < 1 || k < 1 || hot_bufs < 0 || (elem_size != 4 && elem_size != 8)) return -1;
    // not clipped to the LDS capacity: the caller compares with 160 KB
    const size_t extra = (size_t)std::min<int32_t>(hot_bufs, PA_MAX_BUFS) * lds_plan<float>(ibp, Gr, k, true, true).abuf_bytes;
    return elem_size == 8 ? (int64_t)lds_plan<double>(ibp, Gr, k, true, true).total
                          : (int64_t)(lds_plan<float>(ibp, Gr, k, true, true).total + extra);
}

int mfsgd_grid_set_hot_buffers(mfsgd_grid *g, int32_t n)
{
    MF_ARG(g, "grid_set_hot_buffers: null grid");
    MF_ARG(n >= 0 && n <= PA_MAX_BUFS, "grid_set_hot_buffers: n=%d outside [0,%d]", n, PA_MAX_BUFS);
    g->hot_bufs = n;
    return MFSGD_OK;
}

int mfsgd_grid_set_hot_mode(mfsgd_grid *g, int32_t mode)
{
    MF_ARG(g, "grid_set_hot_mode: null grid");
    MF_ARG(mode >= 0 && mode <= 2, "grid_set_hot_mode: mode=%d outside [0,2]", mode);
    g->hot_mode = mode;
    return MFSGD_OK;
}

int mfsgd_grid_set_pick_width(mfsgd_grid *g, int32_t n)
{
    MF_ARG(g, "grid_set_pick_width: null grid");
    MF_ARG(n >= 1 && n <= PICK_WIDTH_MAX, "grid_set_pick_width: n=%d outside [1,%d]", n, PICK_WIDTH_MAX);
    g->pick_width = n;
    return MFSGD_OK;
}

int mfsgd_grid_col_hot(const mfsgd_grid *g, int32_t *h_col_hot, void *stream)
{
    MF_ARG(g && h_col_hot, "grid_col_hot: null pointer");
    hipStream_t st = as_stream(stream);
    MF_HIP(hipMemcpyAsync(h_col_hot, g->col_hot, sizeof(int32_t) * (size_t)g->info.Gc * MAXSEG, hipMemcpyDeviceToHost, st));
    MF_HIP(hipStreamSynchronize(st));
    return MFSGD_OK;
}

int mfsgd_grid_export(const mfsgd_grid *g, int32_t *blk_run_ptr, int32_t *run_start, int32_t *run_user, int32_t *item,
                      float *val, int32_t *tick, void *stream)
{
    MF_ARG(g, "grid_export: null grid");
    hipStream_t st = as_stream(stream);
    const size_t nb = (size_t)g->info.Gr * g->info.Gc, n = (size_t)g->info.n, nr = (size_t)g->info.n_runs;
    if (blk_run_ptr) MF_HIP(hipMemcpyAsync(blk_run_ptr, g->blk_run_ptr, (nb + 1) * 4, hipMemcpyDeviceToDevice, st));
    if (run_start) MF_HIP(hipMemcpyAsync(run_start, g->run_start, (nr + 1) * 4, hipMemcpyDeviceToDevice, st));
    if (run_user && nr) MF_HIP(hipMemcpyAsync(run_user, g->run_user, nr * 4, hipMemcpyDeviceToDevice, st));
    if (item && n) MF_HIP(hipMemcpyAsync(item, g->item, n * 4, hipMemcpyDeviceToDevice, st));
    if (val && n) MF_HIP(hipMemcpyAsync(val, g->val, n * 4, hipMemcpyDeviceToDevice, st));
    if (tick && n) MF_HIP(hipMemcpyAsync(tick, g->tick, n * 4, hipMemcpyDeviceToDevice, st));
    return MFSGD_OK;
}

int mfsgd_grid_col_times(const mfsgd_grid *g, int64_t *h_nnz_start_end, void *stream)
{
    MF_ARG(g && h_nnz_start_end, "grid_col_times: null pointer");
    hipStream_t st = as_stream(stream);
    const size_t Gc = (size_t)g->info.Gc;
    std::vector<long long> clk(2 * Gc);
    MF_HIP(hipMemcpyAsync(clk.data(), g->col_clock, sizeof(long long) * 2 * Gc, hipMemcpyDeviceToHost, st));
    MF_HIP(hipStreamSynchronize(st));
#ifdef MFSGD_PROF
    {
        static const char *names[PF_SLOTS] = {"block", "prologue", "pipe: segments", "headwait", "poll", "update", "tail", "drain",
                                              "barrier", "runs", "ratings", "spins", "sched+stripe", "tenure", "poll|ahead=0", "n|ahead=0",
                                              "sum ahead", "poll|ahead=1", "n|ahead=1", "setup+table", "pipe: rows->LDS", "pipe: gram+cols", "pipe: turn wait", "pipe: g",
                                              "pipe: chain", "pipe: q_nb+pass", "pipe: apply", "pipe: batches", "claim", "index", "gate wait", "pipe: segment done at",
                                              "load_run issue", "walk_run", "step", nullptr, nullptr, nullptr, nullptr, nullptr};
        unsigned long long pf[PF_SLOTS];
        MF_HIP(hipMemcpy(pf, g->col_clock + 2 * Gc, sizeof(pf), hipMemcpyDeviceToHost));
        MF_HIP(hipMemset(g->col_clock + 2 * Gc, 0, sizeof(pf)));
        for (int i = 0; i < PF_SLOTS; ++i)
            fprintf(stderr, "slot %2d %14llu\n", i, pf[i]);
        for (int i = 0; i < PF_SLOTS; ++i)
            if (names[i]) fprintf(stderr, "prof %-13s %14llu  (%.1f%% of tenure)\n", names[i], pf[i], 100.0 * pf[i] / (double)pf[PF_TENURE]);
    }
#endif
    for (size_t c = 0; c < Gc; ++c) {
        h_nnz_start_end[3 * c] = g->col_nnz.empty() ? 0 : g->col_nnz[c];
        h_nnz_start_end[3 * c + 1] = clk[2 * c];
        h_nnz_start_end[3 * c + 2] = clk[2 * c + 1];
    }
    return MFSGD_OK;
}

int mfsgd_grid_get_info(const mfsgd_grid *g, mfsgd_grid_info *info)
{
    MF_ARG(g && info, "grid_get_info: null pointer");
    *info = g->info;
    return MFSGD_OK;
}

// hot_items > 0: order 1 only - store each block's ratings of the stripe's hot items as segments of their own in front.
// Returns 1 (no error) when that layout is impossible because a (user, hot item) pair occurs twice: the caller
// then builds without hot segments.
static int grid_build(const int32_t *samples, int64_t n, int32_t stride, int64_t n_users, int64_t n_items, int32_t ubp,
                      int32_t ibp, int32_t order, int32_t hot_items, int32_t hot_min_per_block, bool need_share,
                      void *stream, mfsgd_grid **out)
{
    const bool allow_hot = hot_items > 0;
    MF_ARG(out, "grid_create: null out");
    *out = nullptr;
    MF_ARG(n >= 0 && n < ((int64_t)1 << 31) - 1, "grid_create: n=%lld must fit int32", (long long)n);
    MF_ARG(stride >= 3 && n_users > 0 && n_items > 0 && ubp > 0 && ibp > 0, "grid_create: bad shape argument");
    MF_ARG(order == 0 || order == 1, "grid_create: order must be 0 or 1");
    MF_ARG(n == 0 || samples, "grid_create: null samples");
    const int64_t Gr = (n_users + ubp - 1) / ubp, Gc = (n_items + ibp - 1) / ibp, nb = Gr * Gc;
    MF_ARG(nb < ((int64_t)1 << 28), "grid_create: %lld x %lld blocks is too many", (long long)Gr, (long long)Gc);
    hipStream_t st = as_stream(stream);
    mfsgd_grid *G = new mfsgd_grid();
    struct Guard { mfsgd_grid *g; ~Guard() { if (g) mfsgd_grid_destroy(g); } } guard{G};
    MF_HIP(hipGetDevice(&G->device));
    G->info.n = n; G->info.n_users = n_users; G->info.n_items = n_items;
    G->info.Gr = (int32_t)Gr; G->info.Gc = (int32_t)Gc; G->info.ubp = ubp; G->info.ibp = ibp; G->info.order = order;
    G->info.user_unique = 1;
    auto dalloc = [&](auto **p, size_t count) { return hipMalloc((void **)p, (count ? count : 1) * sizeof(**p)); };
    MF_HIP(dalloc(&G->blk_run_ptr, (size_t)nb + 1));
    MF_HIP(dalloc(&G->item, (size_t)n + META_PAD)); MF_HIP(dalloc(&G->tick, (size_t)n + META_PAD)); MF_HIP(dalloc(&G->val, (size_t)n + META_PAD));
    MF_HIP(dalloc(&G->col_order, (size_t)Gc));
    MF_HIP(dalloc(&G->col_hot, (size_t)Gc * MAXSEG)); MF_HIP(dalloc(&G->blk_hot_cnt, (size_t)nb));
    MF_HIP(dalloc(&G->blk_seg_cnt, (size_t)nb * MAXSEG)); MF_HIP(dalloc(&G->blk_free_runs, (size_t)nb));
    MF_HIP(hipMemsetAsync(G->blk_free_runs, 0, sizeof(int32_t) * (size_t)nb, st));
    MF_HIP(hipMemsetAsync(G->col_hot, 0xff, sizeof(int32_t) * (size_t)Gc * MAXSEG, st));
    MF_HIP(hipMemsetAsync(G->blk_hot_cnt, 0, sizeof(int32_t) * (size_t)nb, st));
    MF_HIP(hipMemsetAsync(G->blk_seg_cnt, 0, sizeof(int32_t) * (size_t)nb * MAXSEG, st));
    MF_HIP(dalloc(&G->row_lock, ((size_t)Gr + 3) & ~(size_t)3));
    MF_HIP(dalloc(&G->counters, 16));
    MF_HIP(dalloc(&G->col_clock, (size_t)Gc * 2 + PF_SLOTS));
    MF_HIP(hipMemsetAsync(G->col_clock, 0, sizeof(long long) * (2 * (size_t)Gc + PF_SLOTS), st));
    MF_HIP(hipMemsetAsync(G->counters, 0, 64, st));
    std::vector<int32_t> h_blk((size_t)nb + 1, 0);
    if (n == 0) {
        MF_HIP(hipMemsetAsync(G->blk_run_ptr, 0, sizeof(int32_t) * (size_t)(nb + 1), st));
        MF_HIP(dalloc(&G->run_start, 2)); MF_HIP(dalloc(&G->run_user, 1));
        MF_HIP(hipMemsetAsync(G->run_start, 0, 8, st));
        MF_HIP(hipMemsetAsync(G->run_user, 0, 4, st));
        MF_HIP(hipStreamSynchronize(st));
        guard.g = nullptr; *out = G;
        return MFSGD_OK;
    }
    // 1. keys and the stable sort(s)
    DevBuf bk0, bk1, uk0, uk1, ix0, ix1, err;
    MF_HIP(bk0.alloc((size_t)n * 4)); MF_HIP(bk1.alloc((size_t)n * 4));
    MF_HIP(uk0.alloc((size_t)n * 4)); MF_HIP(uk1.alloc((size_t)n * 4));
    MF_HIP(ix0.alloc((size_t)n * 4)); MF_HIP(ix1.alloc((size_t)n * 4));
    MF_HIP(err.alloc(4));
    MF_HIP(hipMemsetAsync(err.p, 0, 4, st));
    hipLaunchKernelGGL(k_keys, dim3(nblk(n)), dim3(256), 0, st, samples, n, stride, ubp, ibp, n_users, n_items, (int32_t)Gc,
                       bk0.as<uint32_t>(), uk0.as<uint32_t>(), ix0.as<int32_t>(), err.as<int32_t>());
    MF_LAUNCH_CHECK();
    int32_t herr = 0;
    MF_HIP(hipMemcpyAsync(&herr, err.p, 4, hipMemcpyDeviceToHost, st));
    MF_HIP(hipStreamSynchronize(st));
    if (herr) { mfsgd_set_error("grid_create: a sample lies outside [0,%lld) x [0,%lld)", (long long)n_users, (long long)n_items); return MFSGD_ERR_GRID; }
    int rc;
    int32_t *idx_sorted;
    DevBuf hot;            // order 1: hot[t] = class of rating t (sorted position) inside its block, see NCLS
    DevBuf hot_user;       // bit (stripe, user): the user rated one of the stripe's hot items
    const bool with_hot = order == 1 && allow_hot;
    if (order == 1) {
        DevBuf deg;
        MF_HIP(deg.alloc((size_t)n_items * 4));
        MF_HIP(hipMemsetAsync(deg.p, 0, (size_t)n_items * 4, st));
        hipLaunchKernelGGL(k_item_deg_raw, dim3(nblk(n)), dim3(256), 0, st, samples, n, stride, deg.as<int32_t>());
        MF_LAUNCH_CHECK();
        if (with_hot) {
            hipLaunchKernelGGL(k_col_hot, dim3(nblk(Gc, 64)), dim3(64), 0, st, deg.as<int32_t>(), n_items, ibp, (int32_t)Gc,
                               std::min<int32_t>(hot_items, MAXSEG), (int64_t)hot_min_per_block * Gr, need_share ? 1 : 0,
                               G->col_hot);
            MF_LAUNCH_CHECK();
            const size_t words = (size_t)(((uint64_t)Gc * (uint64_t)n_users + 31) / 32);
            MF_HIP(hot_user.alloc(words * 4));
            MF_HIP(hipMemsetAsync(hot_user.p, 0, words * 4, st));
            hipLaunchKernelGGL(k_mark_hot_users, dim3(nblk(n)), dim3(256), 0, st, samples, n, stride, ibp, n_users,
                               G->col_hot, hot_user.as<uint32_t>());
            MF_LAUNCH_CHECK();
            MF_HIP(hipStreamSynchronize(st));
        }
        // block key with the class (without hot segments col_hot is all -1 and every rating is of class MAXSEG)
        hipLaunchKernelGGL(k_hot_block_key, dim3(nblk(n)), dim3(256), 0, st, samples, n, stride, (int32_t)Gc, n_users,
                           G->col_hot, with_hot ? hot_user.as<uint32_t>() : (const uint32_t *)nullptr, bk0.as<uint32_t>());
        MF_LAUNCH_CHECK();
        // LSD: stable by the item's degree (most rated first), then stable by user, then stable by (block, class)
        hipLaunchKernelGGL(k_item_rank_key, dim3(nblk(n)), dim3(256), 0, st, samples, n, stride, deg.as<int32_t>(),
                           bk1.as<uint32_t>());
        MF_LAUNCH_CHECK();
        rc = sort_pairs_t(bk1.as<uint32_t>(), uk1.as<uint32_t>(), ix0.as<int32_t>(), ix1.as<int32_t>(), n,
                          bits_for((uint64_t)n + 1), st);
        if (rc) return rc;
        hipLaunchKernelGGL(k_gather_u32, dim3(nblk(n)), dim3(256), 0, st, uk0.as<uint32_t>(), ix1.as<int32_t>(), n,
                           uk1.as<uint32_t>());
        MF_LAUNCH_CHECK();
        rc = sort_pairs_t(uk1.as<uint32_t>(), uk0.as<uint32_t>(), ix1.as<int32_t>(), ix0.as<int32_t>(), n,
                          bits_for((uint64_t)n_users), st);
        if (rc) return rc;
        hipLaunchKernelGGL(k_gather_u32, dim3(nblk(n)), dim3(256), 0, st, bk0.as<uint32_t>(), ix0.as<int32_t>(), n,
                           bk1.as<uint32_t>());
        MF_LAUNCH_CHECK();
        rc = sort_pairs_t(bk1.as<uint32_t>(), bk0.as<uint32_t>(), ix0.as<int32_t>(), ix1.as<int32_t>(), n,
                          bits_for((uint64_t)(NCLS * nb)), st);
        if (rc) return rc;
        idx_sorted = ix1.as<int32_t>();   // sorted keys now in bk0
        MF_HIP(hot.alloc((size_t)n));
        hipLaunchKernelGGL(k_split_hot_key, dim3(nblk(n)), dim3(256), 0, st, bk0.as<uint32_t>(), n, hot.as<uint8_t>());
        MF_LAUNCH_CHECK();
    } else {
        rc = sort_pairs_t(bk0.as<uint32_t>(), bk1.as<uint32_t>(), ix0.as<int32_t>(), ix1.as<int32_t>(), n,
                          bits_for((uint64_t)nb), st);
        if (rc) return rc;
        MF_HIP(hipMemcpyAsync(bk0.p, bk1.p, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
        idx_sorted = ix1.as<int32_t>();
    }
    uint32_t *bkey = bk0.as<uint32_t>();  // sorted block ids
    // 2. SoA payload in sorted order
    DevBuf user;
    MF_HIP(user.alloc((size_t)n * 4));
    hipLaunchKernelGGL(k_gather_soa, dim3(nblk(n)), dim3(256), 0, st, samples, stride, idx_sorted, n,
                       user.as<int32_t>(), G->item, G->val);
    MF_LAUNCH_CHECK();
    if (with_hot) {
        DevBuf dup;
        MF_HIP(dup.alloc(4));
        MF_HIP(hipMemsetAsync(dup.p, 0, 4, st));
        hipLaunchKernelGGL(k_hot_count, dim3(nblk(n)), dim3(256), 0, st, bkey, user.as<int32_t>(), hot.as<uint8_t>(), n,
                           G->blk_seg_cnt, G->blk_hot_cnt, dup.as<int32_t>());
        MF_LAUNCH_CHECK();
        int32_t hd = 0;
        MF_HIP(hipMemcpyAsync(&hd, dup.p, 4, hipMemcpyDeviceToHost, st));
        MF_HIP(hipStreamSynchronize(st));
        if (hd) return 1;   // a (user, hot item) pair occurs twice: two runs of one user would share a segment
    }
    // 3. runs
    DevBuf flag, excl, tmp;
    MF_HIP(flag.alloc((size_t)n * 4)); MF_HIP(excl.alloc((size_t)n * 4));
    hipLaunchKernelGGL(k_run_flags, dim3(nblk(n)), dim3(256), 0, st, bkey, user.as<int32_t>(),
                       with_hot ? hot.as<uint8_t>() : (const uint8_t *)nullptr, n, flag.as<int32_t>());
    MF_LAUNCH_CHECK();
    {
        size_t tb = 0;
        MF_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, flag.as<int32_t>(), excl.as<int32_t>(), (int)n, st));
        MF_HIP(tmp.alloc(tb));
        MF_HIP(hipcub::DeviceScan::ExclusiveSum(tmp.p, tb, flag.as<int32_t>(), excl.as<int32_t>(), (int)n, st));
    }
    int32_t last_excl = 0, last_flag = 0;
    MF_HIP(hipMemcpyAsync(&last_excl, excl.as<int32_t>() + (n - 1), 4, hipMemcpyDeviceToHost, st));
    MF_HIP(hipMemcpyAsync(&last_flag, flag.as<int32_t>() + (n - 1), 4, hipMemcpyDeviceToHost, st));
    MF_HIP(hipStreamSynchronize(st));
    const int32_t n_runs = last_excl + last_flag;
    G->info.n_runs = n_runs;
    MF_HIP(dalloc(&G->run_start, (size_t)n_runs + 2)); MF_HIP(dalloc(&G->run_user, (size_t)n_runs + 1));
    MF_HIP(hipMemsetAsync(G->run_user + n_runs, 0, 4, st));   // the sentinel run belongs to user 0 and is empty
    hipLaunchKernelGGL(k_run_fill, dim3(nblk(n)), dim3(256), 0, st, flag.as<int32_t>(), excl.as<int32_t>(),
                       user.as<int32_t>(), bkey, with_hot ? hot.as<uint8_t>() : (const uint8_t *)nullptr, n, n_runs,
                       G->run_start, G->run_user, G->blk_free_runs);
    MF_LAUNCH_CHECK();
    DevBuf blk_ptr;
    MF_HIP(blk_ptr.alloc(((size_t)nb + 1) * 4));
    hipLaunchKernelGGL(k_blk_ptr, dim3(nblk(n + 1)), dim3(256), 0, st, bkey, n, nb, blk_ptr.as<int32_t>());
    MF_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_blk_run_ptr, dim3(nblk(nb + 1)), dim3(256), 0, st, blk_ptr.as<int32_t>(), excl.as<int32_t>(),
                       nb, n, n_runs, G->blk_run_ptr);
    MF_LAUNCH_CHECK();
    MF_HIP(hipMemcpyAsync(h_blk.data(), G->blk_run_ptr, ((size_t)nb + 1) * 4, hipMemcpyDeviceToHost, st));
    MF_HIP(hipStreamSynchronize(st));
    for (int64_t i = 0; i < nb; ++i) G->max_block_runs = std::max(G->max_block_runs, h_blk[i + 1] - h_blk[i]);
    MF_HIP(hipMemcpyAsync(h_blk.data(), blk_ptr.p, ((size_t)nb + 1) * 4, hipMemcpyDeviceToHost, st));
    // 4. tickets: rank of a rating among the ratings of its (block, item), in stored order
    {
        DevBuf k0, k1, p0, p1, seg;
        MF_HIP(k0.alloc((size_t)n * 8)); MF_HIP(k1.alloc((size_t)n * 8));
        MF_HIP(p0.alloc((size_t)n * 4)); MF_HIP(p1.alloc((size_t)n * 4)); MF_HIP(seg.alloc((size_t)n * 4));
        hipLaunchKernelGGL(k_item_keys, dim3(nblk(n)), dim3(256), 0, st, bkey, G->item, n, k0.as<uint64_t>(),
                           p0.as<int32_t>());
        MF_LAUNCH_CHECK();
        rc = sort_pairs_t(k0.as<uint64_t>(), k1.as<uint64_t>(), p0.as<int32_t>(), p1.as<int32_t>(), n,
                          32 + bits_for((uint64_t)nb), st);
        if (rc) return rc;
        hipLaunchKernelGGL(k_seg_start, dim3(nblk(n)), dim3(256), 0, st, k1.as<uint64_t>(), n, seg.as<int32_t>());
        MF_LAUNCH_CHECK();
        size_t tb = 0;
        DevBuf tmp2;
        MF_HIP(hipcub::DeviceScan::InclusiveScan(nullptr, tb, seg.as<int32_t>(), p0.as<int32_t>(), hipcub::Max(), (int)n, st));
        MF_HIP(tmp2.alloc(tb));
        MF_HIP(hipcub::DeviceScan::InclusiveScan(tmp2.p, tb, seg.as<int32_t>(), p0.as<int32_t>(), hipcub::Max(), (int)n, st));
        hipLaunchKernelGGL(k_ticket, dim3(nblk(n)), dim3(256), 0, st, p0.as<int32_t>(), p1.as<int32_t>(), n, G->tick);
        MF_LAUNCH_CHECK();
        MF_HIP(hipStreamSynchronize(st));
    }
    // 5. does any user own two runs of one block?  (never for order 1)
    if (order == 0 && n_runs > 1) {
        DevBuf rk0, rk1, dup;
        MF_HIP(rk0.alloc((size_t)n_runs * 8)); MF_HIP(rk1.alloc((size_t)n_runs * 8)); MF_HIP(dup.alloc(4));
        MF_HIP(hipMemsetAsync(dup.p, 0, 4, st));
        hipLaunchKernelGGL(k_run_keys, dim3(nblk(n_runs)), dim3(256), 0, st, G->run_start, G->run_user, bkey, n_runs,
                           rk0.as<uint64_t>());
        MF_LAUNCH_CHECK();
        rc = sort_keys_t(rk0.as<uint64_t>(), rk1.as<uint64_t>(), n_runs, 32 + bits_for((uint64_t)nb), st);
        if (rc) return rc;
        hipLaunchKernelGGL(k_adjacent_dup, dim3(nblk(n_runs)), dim3(256), 0, st, rk1.as<uint64_t>(), (int64_t)n_runs,
                           dup.as<int32_t>());
        MF_LAUNCH_CHECK();
        int32_t hd = 0;
        MF_HIP(hipMemcpyAsync(&hd, dup.p, 4, hipMemcpyDeviceToHost, st));
        MF_HIP(hipStreamSynchronize(st));
        G->info.user_unique = hd ? 0 : 1;
    }
    // 5b. hot segments: how many stripes have one, how many ratings they hold
    if (with_hot) {
        std::vector<int32_t> h_hot((size_t)Gc * MAXSEG), h_cnt((size_t)nb);
        MF_HIP(hipMemcpyAsync(h_hot.data(), G->col_hot, sizeof(int32_t) * (size_t)Gc * MAXSEG, hipMemcpyDeviceToHost, st));
        MF_HIP(hipMemcpyAsync(h_cnt.data(), G->blk_hot_cnt, sizeof(int32_t) * (size_t)nb, hipMemcpyDeviceToHost, st));
        MF_HIP(hipStreamSynchronize(st));
        for (int64_t c = 0; c < Gc; ++c) G->info.n_hot_stripes += h_hot[(size_t)c * MAXSEG] >= 0;
        for (int32_t v : h_hot) G->info.n_hot_items += v >= 0;
        for (int32_t v : h_cnt) G->info.n_hot += v;
    }
    MF_HIP(hipStreamSynchronize(st));
    // 6. stripe weights (host, Gr*Gc words): heaviest item stripe first
    std::vector<int64_t> col_nnz((size_t)Gc, 0), row_nnz((size_t)Gr, 0);
    int64_t max_block = 0;
    for (int64_t r = 0; r < Gr; ++r)
        for (int64_t c = 0; c < Gc; ++c) {
            const int64_t m = (int64_t)h_blk[r * Gc + c + 1] - h_blk[r * Gc + c];
            col_nnz[c] += m; row_nnz[r] += m; max_block = std::max(max_block, m);
        }
    std::vector<int32_t> ord((size_t)Gc);
    std::iota(ord.begin(), ord.end(), 0);
    std::stable_sort(ord.begin(), ord.end(), [&](int32_t a, int32_t b2) { return col_nnz[a] > col_nnz[b2]; });
    int32_t ne = 0;
    while (ne < Gc && col_nnz[ord[ne]] > 0) ++ne;
    G->n_cols_nonempty = ne;
    G->col_nnz = col_nnz;
    G->info.max_block = max_block;
    G->info.max_col_nnz = *std::max_element(col_nnz.begin(), col_nnz.end());
    G->info.max_row_nnz = *std::max_element(row_nnz.begin(), row_nnz.end());
    MF_HIP(hipMemcpyAsync(G->col_order, ord.data(), sizeof(int32_t) * (size_t)Gc, hipMemcpyHostToDevice, st));
    MF_HIP(hipStreamSynchronize(st));
    guard.g = nullptr;
    *out = G;
    return MFSGD_OK;
}

int mfsgd_grid_create_opt(const int32_t *samples, int64_t n, int32_t stride, int64_t n_users, int64_t n_items, int32_t ubp,
                          int32_t ibp, int32_t order, int32_t hot_items, int32_t hot_min_per_block, void *stream,
                          mfsgd_grid **out)
{
    MF_ARG(hot_items >= 0 && hot_items <= MAXSEG, "grid_create: hot_items=%d outside [0,%d]", hot_items, MAXSEG);
    MF_ARG(hot_min_per_block >= 1 || hot_items == 0, "grid_create: hot_min_per_block < 1");
    int rc = hot_items > 0 ? grid_build(samples, n, stride, n_users, n_items, ubp, ibp, order, hot_items, hot_min_per_block,
                                        false, stream, out) : 1;
    if (rc == 1) rc = grid_build(samples, n, stride, n_users, n_items, ubp, ibp, order, 0, 0, false, stream, out);
    return rc;
}

int mfsgd_grid_create(const int32_t *samples, int64_t n, int32_t stride, int64_t n_users, int64_t n_items, int32_t ubp,
                      int32_t ibp, int32_t order, void *stream, mfsgd_grid **out)
{
    int rc = grid_build(samples, n, stride, n_users, n_items, ubp, ibp, order, 1, 1, true, stream, out);
    if (rc == 1) rc = grid_build(samples, n, stride, n_users, n_items, ubp, ibp, order, 0, 0, false, stream, out);
    return rc;
}

int mfsgd_hip_epoch_f32(mfsgd_grid *g, int32_t k, float *P, float *Q, float *b_u, float *b_i, float b, const float *z,
                        float alpha, float beta1, float beta2, int32_t variant, int32_t implicit, int32_t iters,
                        int32_t schedule, uint64_t seed, int32_t n_workgroups, int32_t waves_per_wg,
                        int32_t *order_log, void *stream)
{
    return epoch_impl<float>(g, P, Q, b_u, b_i, b, z, alpha, beta1, beta2, variant, implicit, iters, schedule, seed,
                             n_workgroups, waves_per_wg, order_log, k, stream);
}

int mfsgd_hip_epoch_f64(mfsgd_grid *g, int32_t k, double *P, double *Q, double *b_u, double *b_i, double b,
                        const double *z, double alpha, double beta1, double beta2, int32_t variant, int32_t implicit,
                        int32_t iters, int32_t schedule, uint64_t seed, int32_t n_workgroups, int32_t waves_per_wg,
                        int32_t *order_log, void *stream)
{
    return epoch_impl<double>(g, P, Q, b_u, b_i, b, z, alpha, beta1, beta2, variant, implicit, iters, schedule, seed,
                              n_workgroups, waves_per_wg, order_log, k, stream);
}

}  // extern "C"
