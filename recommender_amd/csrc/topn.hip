// topn.hip - batched scoring + top-N on gfx950 (MI355X): SURVEY.md section 8(f).3.
//
// Replaces the dense contraction behind ExplicitMF.predict (reference SGDRecommender.py:82-83,
// `user_vecs[row,:].dot(item_vecs[col,:].T)`) when what the caller wants from it is the N best items per user:
// scores = P[users] . Q^T (+ b_i[item] + b_u[user] + b, the training predictor of numba_methods.py:17 without the
// implicit term) are produced tile by tile on the matrix cores and reduced to a top-N on the fly - the
// users x items score matrix (34 GB for the netflix shape) is never written.
//
// Layout of one workgroup (4 waves; 2 waves for k > 200, where four user tiles plus the item chunks no longer fit the
// 160 KB of LDS): each wave owns 32 users (their factor rows sit in LDS, padded against bank
// conflicts), all four walk the same 32-item chunks of Q, staged once per workgroup in LDS (double-buffered).
// Per chunk a wave computes the 32x32 tile  S^T = Q_chunk . P_tile^T  with v_mfma_f32_32x32x2_f32: with items as
// MFMA rows and users as MFMA columns a lane ends up holding 16 candidate items of ONE user (its column), so the
// running top-N is private to a lane (registers); the two lanes of a user are merged at the end.
#include <algorithm>
#include "common.h"

namespace {

// waves per workgroup = user tiles per workgroup: a template parameter W (4 up to k = 200, 2 up to k = 256)
constexpr int TN_TILE = 32;     // users per wave, items per chunk
constexpr int TN_MAX = 16;      // largest N

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

template <int N, int TN_WAVES>
__global__ __launch_bounds__(TN_WAVES * 64) void k_topn(const float *__restrict__ P, const float *__restrict__ Q,
                                                        const float *__restrict__ b_u, const float *__restrict__ b_i,
                                                        float b, const int32_t *__restrict__ users, int64_t n_batch,
                                                        int64_t n_items, int k, int kpad, int with_bias,
                                                        int32_t *__restrict__ out_items, float *__restrict__ out_scores)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ld = kpad + 1;                           // row stride in LDS: odd, so a column walk hits 32 banks
    float *pt = lds + (size_t)wave * TN_TILE * ld;     // this wave's user tile [32][ld]
    float *qc = lds + (size_t)TN_WAVES * TN_TILE * ld; // item chunks [2][32][ld]
    float *bc = qc + (size_t)2 * TN_TILE * ld;         // item biases of the chunks [2][32]
    const int64_t u0 = ((int64_t)blockIdx.x * TN_WAVES + wave) * TN_TILE;

    // user tile -> LDS (rows beyond the batch and columns beyond k are zero)
    for (int e = lane; e < TN_TILE * kpad; e += 64) {
        const int r = e / kpad, f = e - r * kpad;
        float v = 0.f;
        if (u0 + r < n_batch && f < k) v = P[(int64_t)users[u0 + r] * k + f];
        pt[r * ld + f] = v;
    }
    // A chunk of Q goes HBM -> registers before the multiply of the previous chunk and registers -> LDS after it,
    // so the load latency hides behind the MFMA loop (one wave per SIMD: nothing else would hide it).
    constexpr int KPAD_MAX = TN_WAVES == 4 ? 208 : 256;
    constexpr int STG = (TN_TILE * KPAD_MAX + TN_WAVES * 64 - 1) / (TN_WAVES * 64);   // floats per thread at the widest row
    float stg[STG];
    float stg_b = 0.f;
    int32_t src_off[STG], dst_off[STG];   // where this thread's elements of a chunk come from and go to (-1: none)
#pragma unroll
    for (int t = 0; t < STG; ++t) {
        const int e = tid + t * TN_WAVES * 64;
        const int r = e / kpad, f = e - r * kpad;
        const bool in = e < TN_TILE * kpad;
        dst_off[t] = in ? r * ld + f : -1;
        src_off[t] = in && f < k ? r * k + f : -1;
    }
    auto fetch = [&](int64_t c) {
        const int64_t i0 = c * TN_TILE;
        const float *src = Q + i0 * k;
        const int32_t lim = (int32_t)std::min<int64_t>(TN_TILE, n_items - i0) * k;   // rows beyond the last item read as 0
#pragma unroll
        for (int t = 0; t < STG; ++t) stg[t] = (src_off[t] >= 0 && src_off[t] < lim) ? src[src_off[t]] : 0.f;
        if (tid < TN_TILE) stg_b = (with_bias && i0 + tid < n_items) ? b_i[i0 + tid] : 0.f;
    };
    auto commit = [&](int buf) {
        float *dst = qc + (size_t)buf * TN_TILE * ld;
#pragma unroll
        for (int t = 0; t < STG; ++t)
            if (dst_off[t] >= 0) dst[dst_off[t]] = stg[t];
        if (tid < TN_TILE) bc[buf * TN_TILE + tid] = stg_b;
    };
    float sc[N];
    int32_t id[N];
#pragma unroll
    for (int j = 0; j < N; ++j) { sc[j] = -INFINITY; id[j] = 0x7fffffff; }
    const int64_t n_chunks = (n_items + TN_TILE - 1) / TN_TILE;
    fetch(0);
    commit(0);
    __syncthreads();
    const int col = lane & 31, kh = lane >> 5;          // MFMA operand coordinates of this lane
    for (int64_t c = 0; c < n_chunks; ++c) {
        const int buf = (int)(c & 1);
        if (c + 1 < n_chunks) fetch(c + 1);             // in flight during the multiply
        const float *qa = qc + (size_t)buf * TN_TILE * ld + col * ld + kh;   // A[item = col][kk + kh]
        const float *pb = pt + col * ld + kh;                               // B[kk + kh][user = col]
        f16v acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        // kpad is a multiple of 8 (zero columns beyond k); the operands of the next four MFMAs are read from LDS
        // while the current four run (one wave per SIMD: nothing else hides the LDS latency)
        float a0 = qa[0], a1 = qa[2], a2 = qa[4], a3 = qa[6];
        float p0 = pb[0], p1 = pb[2], p2 = pb[4], p3 = pb[6];
        for (int kk = 8; kk < kpad; kk += 8) {
            const float n0 = qa[kk], n1 = qa[kk + 2], n2 = qa[kk + 4], n3 = qa[kk + 6];
            const float m0 = pb[kk], m1 = pb[kk + 2], m2 = pb[kk + 4], m3 = pb[kk + 6];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, p0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, p1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, p2, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a3, p3, acc, 0, 0, 0);
            a0 = n0; a1 = n1; a2 = n2; a3 = n3;
            p0 = m0; p1 = m1; p2 = m2; p3 = m3;
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, p0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, p1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, p2, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a3, p3, acc, 0, 0, 0);
        if (c + 1 < n_chunks) commit(buf ^ 1);          // the other buffer: nobody reads it before the barrier below
        // lane holds S^T[item row][user = col] for rows 8*(j/4) + 4*kh + (j%4), j = 0..15
        const int64_t i0 = c * TN_TILE;
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
            const float s = acc[j] + bc[buf * TN_TILE + row];
            const bool cand = item < n_items && s >= bar &&
                              (s > sc[N - 1] || (s == sc[N - 1] && (int32_t)item < id[N - 1]));
#ifdef TN_NOSELECT   // timing experiment: multiply only (one value is kept so the MFMAs are not dead code)
            if (s == 12345.678f) sc[0] = s;
#else
            if (__ballot(cand)) insert_top<N>(sc, id, cand ? s : -INFINITY, cand ? (int32_t)item : 0x7fffffff);
#endif
        }
        __syncthreads();
    }
    // merge the two lanes of a user (lane, lane+32) through LDS (the user tile is no longer needed), lane < 32 writes
    float *ms = pt;                       // [32][N] scores of the upper lanes
    int32_t *mi = (int32_t *)(pt + TN_TILE * N);
    if (kh == 1) {
#pragma unroll
        for (int j = 0; j < N; ++j) { ms[col * N + j] = sc[j]; mi[col * N + j] = id[j]; }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (kh == 0 && u0 + col < n_batch) {
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

template <int N, int TN_WAVES>
int launch_topn_w(const float *P, const float *Q, const float *b_u, const float *b_i, float b, const int32_t *users,
                int64_t n_batch, int64_t n_items, int k, int with_bias, int32_t *out_items, float *out_scores,
                hipStream_t st)
{
    const int kpad = (k + 7) & ~7;
    const size_t ld = (size_t)kpad + 1;
    const size_t bytes = ((size_t)(TN_WAVES + 2) * TN_TILE * ld + 2 * TN_TILE) * sizeof(float);
    MF_ARG(bytes <= (size_t)160 * 1024, "topn: k=%d needs %zu bytes of LDS", k, bytes);
    MF_ARG((size_t)TN_TILE * ld >= (size_t)2 * TN_TILE * N, "topn: k=%d too small for N=%d", k, N);
    auto kern = k_topn<N, TN_WAVES>;
    MF_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    const int64_t per_wg = (int64_t)TN_WAVES * TN_TILE;
    const unsigned grid = (unsigned)((n_batch + per_wg - 1) / per_wg);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(TN_WAVES * 64), bytes, st, P, Q, b_u, b_i, b, users, n_batch, n_items, k,
                       kpad, with_bias, out_items, out_scores);
    MF_LAUNCH_CHECK();
    return MFSGD_OK;
}

template <int N>
int launch_topn(const float *P, const float *Q, const float *b_u, const float *b_i, float b, const int32_t *users,
                int64_t n_batch, int64_t n_items, int k, int with_bias, int32_t *out_items, float *out_scores,
                hipStream_t st)
{
    if (k <= 200) return launch_topn_w<N, 4>(P, Q, b_u, b_i, b, users, n_batch, n_items, k, with_bias, out_items, out_scores, st);
    return launch_topn_w<N, 2>(P, Q, b_u, b_i, b, users, n_batch, n_items, k, with_bias, out_items, out_scores, st);
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
