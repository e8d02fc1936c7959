// topn.hip - batched scoring + top-N on gfx950 (MI355X): SURVEY.md section 8(f).3.
//
// Replaces the dense contraction behind ExplicitMF.predict (reference SGDRecommender.py:82-83,
// `user_vecs[row,:].dot(item_vecs[col,:].T)`) when what the caller wants from it is the N best items per user:
// scores = P[users] . Q^T (+ b_i[item] + b_u[user] + b, the training predictor of numba_methods.py:17 without the
// implicit term) are produced tile by tile on the matrix cores and reduced to a top-N on the fly - the
// users x items score matrix (34 GB for the netflix shape) is never written.
//
// Layout of one workgroup (4 waves): each wave owns 32 users, whose factor rows it keeps in registers; all four walk
// the same 64-item chunks of Q, staged once per workgroup in LDS (double-buffered).  Per 32-item tile a wave computes
// S^T = Q_tile . P_tile^T  with v_mfma_f32_32x32x2_f32: with items as MFMA rows and users as MFMA columns a lane ends
// up holding 16 candidate items of ONE user (its column), so the running top-N is private to a lane (registers); the
// two lanes of a user are merged at the end.
#include <algorithm>
#include "common.h"

namespace {

constexpr int TN_TILE = 32;     // users per wave, items per chunk
constexpr int TN_MAX = 16;      // largest N
#ifndef TN_SCHED
#define TN_SCHED 8
#endif
#ifndef TN_OCC2_MAX
#define TN_OCC2_MAX 128   // widest padded k built for two workgroups per CU (256 registers per lane)
#endif

typedef float f16v __attribute__((ext_vector_type(16)));

// Sorted insertion into a lane-private descending list (branch-free per lane: the wave only enters when some
// lane has a candidate above its current N-th best).
template <int N>
__device__ __forceinline__ void insert_top(float (&sc)[N], int32_t (&id)[N], float s, int32_t i)
{
#pragma unroll
    for (int j = 0; j < N; ++j) {
        // ties: the lower item id wins (what a stable argsort of -scores gives)
        const bool better = s > sc[j] || (s == sc[j] && i < id[j]);
        const float ts = better ? sc[j] : s;
        const int32_t ti = better ? id[j] : i;
        sc[j] = better ? s : sc[j];
        id[j] = better ? i : id[j];
        s = ts; i = ti;
    }
}

// KP = k rounded up to a multiple of 32 (zero columns).  One workgroup = 4 waves = 4 x 32 users.
//  * The users' rows live in REGISTERS for the whole kernel: lane (col, kh) holds P[user col][2 j + kh], j < KP / 2 -
//    exactly the B operand of MFMA j - so the multiply reads only the item operand from LDS (round 2 kept the user tiles
//    in LDS: two LDS reads per MFMA, 66 KB of the CU's LDS, one workgroup per CU).
//  * A step takes 64 items: two 32-item tiles with independent accumulators, their MFMAs interleaved (one wave has two
//    dependent chains in flight instead of one; the user operand is shared), one barrier per 64 items.
//  * The item chunks are staged once per workgroup, HBM -> registers before the multiply of the previous step and
//    registers -> LDS after it.  LDS: 2 x 64 x (KP + 1) floats = 66 KB at k = 128 (two workgroups per CU: two waves per
//    SIMD), 132 KB at k = 256.
template <int N, int KP>
__global__ __launch_bounds__(256, (KP <= 96 || (KP <= TN_OCC2_MAX && N <= 8)) ? 2 : 1) void k_topn(const float *__restrict__ P, const float *__restrict__ Q,
                                                                const float *__restrict__ b_u, const float *__restrict__ b_i,
                                                                float b, const int32_t *__restrict__ users, int64_t n_batch,
                                                                int64_t n_items, int k, int with_bias,
                                                                int32_t *__restrict__ out_items, float *__restrict__ out_scores)
{
    constexpr int W = 4, CH = 2 * TN_TILE, LD = KP + 1;   // LD odd: a column walk over 32 items hits 32 banks
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *qc = lds;                       // item chunks [2][CH][LD]
    float *bc = qc + 2 * CH * LD;          // their biases [2][CH]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, kh = lane >> 5;          // MFMA operand coordinates of this lane
    const int64_t u0 = ((int64_t)blockIdx.x * W + wave) * TN_TILE;
    const bool live_u = u0 + col < n_batch;

    // this lane's user row -> registers (rows beyond the batch and columns beyond k are zero)
    float preg[KP / 2];
    {
        // every load is unconditional (a clamped address, the value masked afterwards): a conditional load becomes a
        // branch, and 64 of them with their exec masks cost more registers than the rows themselves
        const float *prow = P + (int64_t)users[live_u ? u0 + col : 0] * k;
#pragma unroll
        for (int j = 0; j < KP / 2; ++j) {
            const int f = 2 * j + kh;
            const float v = prow[f < k ? f : 0];
            preg[j] = (live_u && f < k) ? v : 0.f;
        }
    }
    constexpr int STG = CH * KP / 256;     // floats of a chunk per thread (KP is a multiple of 32: exact)
    float stg[STG];
    float stg_b = 0.f;
    auto fetch = [&](int64_t c) {
        const int64_t i0 = c * CH;
        const float *src = Q + i0 * k;
        const int32_t rows = (int32_t)std::min<int64_t>(CH, n_items - i0);   // rows beyond the last item read as 0
#pragma unroll
        for (int t = 0; t < STG; ++t) {
            const int e = tid + t * 256;
            const int r = e / KP, f = e - r * KP;
            const bool in = r < rows && f < k;
            const float v = src[in ? r * k + f : 0];          // unconditional, see above (i0 < n_items: src[0] exists)
            stg[t] = in ? v : 0.f;
        }
        if (tid < CH) stg_b = (with_bias && i0 + tid < n_items) ? b_i[i0 + tid] : 0.f;
    };
    auto commit = [&](int buf) {
        float *dst = qc + (size_t)buf * CH * LD;
#pragma unroll
        for (int t = 0; t < STG; ++t) {
            const int e = tid + t * 256;
            const int r = e / KP, f = e - r * KP;
            dst[r * LD + f] = stg[t];
        }
        if (tid < CH) bc[buf * CH + tid] = stg_b;
    };
    float sc[N];
    int32_t id[N];
#pragma unroll
    for (int j = 0; j < N; ++j) { sc[j] = -INFINITY; id[j] = 0x7fffffff; }
    // lane holds S^T[item row][user = col] of a tile for rows 8*(j/4) + 4*kh + (j%4), j = 0..15
    auto select = [&](const f16v &acc, int64_t i0, const float *bias) {
        // A user's two lanes each keep the best N of their own half of the items; the user's N-th best over
        // both halves is at least the larger of the two lanes' N-th bests, so that is the bar a candidate has
        // to pass (strictly: equal scores are settled by the list itself).
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(sc[N - 1]), __float_as_uint(sc[N - 1]), false, false);
        const float other = __uint_as_float(kh ? sw[0] : sw[1]);
        const float bar = fmaxf(sc[N - 1], other);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int row = 8 * (j / 4) + 4 * kh + (j % 4);
            const int64_t item = i0 + row;
            const float s = acc[j] + bias[row];
            const bool cand = item < n_items && s >= bar &&
                              (s > sc[N - 1] || (s == sc[N - 1] && (int32_t)item < id[N - 1]));
#ifdef TN_NOSELECT   // timing experiment: multiply only (one value is kept so the MFMAs are not dead code)
            if (s == 12345.678f) sc[0] = s;
#else
            if (__ballot(cand)) insert_top<N>(sc, id, cand ? s : -INFINITY, cand ? (int32_t)item : 0x7fffffff);
#endif
        }
    };
    const int64_t n_steps = (n_items + CH - 1) / CH;
    fetch(0);
    commit(0);
    __syncthreads();
    for (int64_t c = 0; c < n_steps; ++c) {
        const int buf = (int)(c & 1);
        if (c + 1 < n_steps) fetch(c + 1);              // in flight during the multiply
        const float *qa0 = qc + (size_t)buf * CH * LD + col * LD + kh;   // A[item = col][2 j + kh] of the first tile
        const float *qa1 = qa0 + TN_TILE * LD;                           // ... of the second
        f16v acc0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        f16v acc1 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < KP / 2; ++j) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa0[2 * j], preg[j], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa1[2 * j], preg[j], acc1, 0, 0, 0);
            // the loop is fully unrolled (preg is a register array); keep the scheduler from hoisting all KP item
            // operand reads in front of it: at k = 128 that spilled 45 registers of the 256 two workgroups per CU leave
            if (j % TN_SCHED == TN_SCHED - 1) __builtin_amdgcn_sched_barrier(0);
        }
        if (c + 1 < n_steps) commit(buf ^ 1);           // the other buffer: nobody reads it before the barrier below
        select(acc0, c * CH, bc + buf * CH);
        select(acc1, c * CH + TN_TILE, bc + buf * CH + TN_TILE);
        __syncthreads();
    }
    // merge the two lanes of a user (lane, lane+32) through LDS (the item chunks are no longer needed), lane < 32 writes
    float *ms = lds + (size_t)wave * (2 * TN_TILE * N);     // [32][N] scores of the upper lanes, then [32][N] ids
    int32_t *mi = (int32_t *)(ms + TN_TILE * N);
    if (kh == 1) {
#pragma unroll
        for (int j = 0; j < N; ++j) { ms[col * N + j] = sc[j]; mi[col * N + j] = id[j]; }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (kh == 0 && live_u) {
#pragma unroll
        for (int j = 0; j < N; ++j) insert_top<N>(sc, id, ms[col * N + j], mi[col * N + j]);
        const float add = with_bias ? b_u[users[u0 + col]] + b : 0.f;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            out_items[(u0 + col) * N + j] = id[j] == 0x7fffffff ? -1 : id[j];
            out_scores[(u0 + col) * N + j] = id[j] == 0x7fffffff ? -INFINITY : sc[j] + add;
        }
    }
}

template <int N, int KP>
int launch_topn_k(const float *P, const float *Q, const float *b_u, const float *b_i, float b, const int32_t *users,
                  int64_t n_batch, int64_t n_items, int k, int with_bias, int32_t *out_items, float *out_scores,
                  hipStream_t st)
{
    constexpr size_t bytes = ((size_t)2 * 2 * TN_TILE * (KP + 1) + 2 * 2 * TN_TILE) * sizeof(float);
    static_assert(bytes <= (size_t)160 * 1024, "item chunks must fit a CU's LDS");
    static_assert((size_t)2 * 2 * TN_TILE * (KP + 1) >= (size_t)4 * 2 * TN_TILE * N, "the final merge reuses the chunk buffers");
    auto kern = k_topn<N, KP>;
    MF_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    const int64_t per_wg = (int64_t)4 * TN_TILE;
    const unsigned grid = (unsigned)((n_batch + per_wg - 1) / per_wg);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), bytes, st, P, Q, b_u, b_i, b, users, n_batch, n_items, k, with_bias,
                       out_items, out_scores);
    MF_LAUNCH_CHECK();
    return MFSGD_OK;
}

template <int N>
int launch_topn(const float *P, const float *Q, const float *b_u, const float *b_i, float b, const int32_t *users,
                int64_t n_batch, int64_t n_items, int k, int with_bias, int32_t *out_items, float *out_scores,
                hipStream_t st)
{
#define MF_TN(KP_) case KP_: return launch_topn_k<N, KP_>(P, Q, b_u, b_i, b, users, n_batch, n_items, k, with_bias, out_items, out_scores, st);
    switch ((k + 31) & ~31) {
        MF_TN(32) MF_TN(64) MF_TN(96) MF_TN(128) MF_TN(160) MF_TN(192) MF_TN(224) MF_TN(256)
    }
#undef MF_TN
    mfsgd_set_error("topn: k=%d outside [32,256]", k);
    return MFSGD_ERR_ARG;
}

}  // namespace

extern "C" int mfsgd_hip_topn_f32(const float *P, const float *Q, const float *b_u, const float *b_i, float b,
                                  const int32_t *users, int64_t n_batch, int64_t n_items, int32_t k,
                                  int32_t with_bias, int32_t topn, int32_t *out_items, float *out_scores, void *stream)
{
    MF_ARG(P && Q && users && out_items && out_scores, "topn: null pointer");
    MF_ARG(!with_bias || (b_u && b_i), "topn: with_bias needs b_u and b_i");
    MF_ARG(n_batch >= 0 && n_items > 0 && k >= 32 && k <= 256, "topn: bad shape (k in [32,256])");
    MF_ARG(topn >= 1 && topn <= TN_MAX, "topn: N=%d outside [1,%d]", topn, TN_MAX);
    if (n_batch == 0) return MFSGD_OK;
    hipStream_t st = as_stream(stream);
    // N is a compile-time size of the lane-private lists: rounded up to 4, 8 or 16, the caller's N first entries
    // are what it asked for only when N is one of those - the Python side asks for the rounded size and cuts.
    switch (topn) {
        case 4: return launch_topn<4>(P, Q, b_u, b_i, b, users, n_batch, n_items, k, with_bias, out_items, out_scores, st);
        case 8: return launch_topn<8>(P, Q, b_u, b_i, b, users, n_batch, n_items, k, with_bias, out_items, out_scores, st);
        case 16: return launch_topn<16>(P, Q, b_u, b_i, b, users, n_batch, n_items, k, with_bias, out_items, out_scores, st);
        default: mfsgd_set_error("topn: N must be 4, 8 or 16 (got %d)", topn); return MFSGD_ERR_ARG;
    }
}
