// als.hip - one half-step of alternating least squares for gfx950.
//
// Replaces ExplicitMF.als_step of the reference (ALSRecommender.py:118-155): for every row r of a CSR (users, or
// items on the transposed CSR)
//     (F_r^T F_r + lambda (n_r + 1) I) x = F_r^T v_r,   F_r = fixed[col_inds of r],  v_r = the row's ratings,
// solved by the reference with numpy.linalg.solve, one row per Numba thread.
//
// Here one workgroup of 256 threads (a 16 x 16 arrangement) owns a row at a time (rows are handed out through a
// global counter: row lengths differ by four orders of magnitude).  Thread (ty, tx) keeps the entries
// A[ty + 16a][tx + 16b] of the k x k normal matrix in REGISTERS, cyclically, so that the work stays balanced
// while the factorisation shrinks the active part; only the tiles b <= a (on and below the diagonal) are ever
// computed - the matrix is symmetric and the elimination reads columns from the diagonal down:
//   1. Gram: the row's factor rows are gathered 16 at a time into LDS (the next 16 are in flight meanwhile).
//      fp32 with k > 64: the four waves multiply the 32 x 32 tiles on and below the diagonal on the matrix cores
//      (v_mfma_f32_32x32x2_f32) and hand the result to the cyclic register layout through LDS.  Otherwise every
//      thread adds the outer-product entries it owns (vector FMAs, 2*TS LDS reads per chunk row, TS = ceil(k/16)).
//      F^T v is built alongside.
//   2. L D L^T without pivoting (the matrix is symmetric positive definite for lambda > 0), still in registers:
//      step j publishes column j (and the j-th entry of the eliminated right-hand side) to LDS, one barrier, and
//      everybody subtracts c_i c_l / d from what it owns.  Columns stay unscaled - no square roots, no second
//      barrier - and the LDS copy of the columns is the factor the substitution needs.
//   3. Back substitution by wave 0 on the LDS copy (column reads at an odd stride: conflict-free).
// fp32 and fp64 share the code (the reference is float64; fp64 is what the 1e-9 parity test runs).
// k > 128 (BASELINE config 5's k = 256; the reference has no limit, ALSRecommender.py:118-155): the same scheme on a
// 32 x 32 arrangement (1024 threads, at most 8 x 8 entries per thread), vector Gram, and the factor stored PACKED (lower
// triangle only, k (k + 1) / 2 entries): 132 KB of LDS in fp32 at k = 256; in fp64 it does not fit a CU's LDS above
// k = 176 and lives in a per-workgroup slab of device memory instead (the workgroup's own L1/L2-resident scratch).
#include "common.h"
#include <algorithm>

namespace {

#ifndef MFSGD_ALS_CH
#define MFSGD_ALS_CH 16
#endif
constexpr int ALS_RPT = MFSGD_ALS_CH / 16;   // factor rows every thread row of the G x G arrangement fetches per step
#ifndef MFSGD_ALS_EXP
#define MFSGD_ALS_EXP 0   // timing experiments (results wrong on purpose): 1 = no elimination, 2 = no substitution
#endif

struct AlsPlan {                 // byte offsets in dynamic LDS (multiples of 16)
    uint32_t chunk, cvals, lmat, zfin, ctl, total;
};

// g = side of the thread arrangement (16 or 32); packed = the factor keeps its lower triangle only; factor_in_lds = false:
// the factor lives in device memory (als_impl allocates the slabs) and only the gather buffers are in LDS
template <typename T> AlsPlan als_plan(int k, int ts, int g = 16, bool packed = false, bool factor_in_lds = true)
{
    auto up = [](size_t x) { return (x + 15) & ~(size_t)15; };
    const size_t kp = (size_t)ts * g, ld = kp + 1, ch = (size_t)g * ALS_RPT;
    AlsPlan p{};
    size_t o = 0;
    // the gather buffers are only alive during the Gram phase, the factor only after it: they share one region
    const size_t stage = up(2 * ch * kp * sizeof(T)) + up(2 * ch * sizeof(T));
    const size_t factor = !factor_in_lds ? 0 : packed ? kp * (kp + 1) / 2 * sizeof(T) : kp * ld * sizeof(T);
    p.chunk = (uint32_t)o;
    p.cvals = (uint32_t)(o + up(2 * ch * kp * sizeof(T)));
    p.lmat = (uint32_t)o;
    o = up(o + std::max(stage, factor));   // kp rows: the matrix-core path parks the whole padded Gram matrix here
    p.zfin = (uint32_t)o;  o = up(o + 2 * kp * sizeof(T));   // eliminated right-hand side, reciprocal pivots
    p.ctl = (uint32_t)o;   o = up(o + 16);
    p.total = (uint32_t)o;
    return p;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));

// value of `v` in lane `lane` (wave-uniform index): v_readlane, no LDS crossbar
__device__ __forceinline__ float lane_bcast(float v, int lane)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ double lane_bcast(double v, int lane)
{
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// 1 / d for the pivots: the hardware reciprocal (1 ulp) in fp32, a division in fp64
__device__ __forceinline__ float pivot_inv(float d) { return __builtin_amdgcn_rcpf(d); }
__device__ __forceinline__ double pivot_inv(double d) { return 1.0 / d; }

// MFMA (fp32, TS even): the Gram matrix is accumulated on the matrix cores (v_mfma_f32_32x32x2_f32) in 32 x 32
// tiles, the tiles on and below the diagonal dealt round-robin to the four waves, and handed over to the cyclic
// register layout of the elimination through LDS.  Otherwise (fp64, or k <= 64 where the padding to 32 costs more
// than it saves) the threads accumulate their own entries with vector FMAs.
// G = side of the thread arrangement: 16 (256 threads, k <= 128) or 32 (1024 threads, k <= 256: PACKED factor, in LDS or -
// GLOBAL_L - in the workgroup's slab of `lscratch`)
template <typename T, int G, int TS, bool MFMA, bool PACKED, bool GLOBAL_L>
__global__ __launch_bounds__(G * G) void k_als_rows(T *__restrict__ latent, const T *__restrict__ fixed,
                                                          const int64_t *__restrict__ row_ptrs,
                                                          const int32_t *__restrict__ col_inds,
                                                          const T *__restrict__ vals, int64_t n_rows, int64_t n_cols,
                                                          int64_t nnz, int k, T lambda, int32_t *next_row,
                                                          int32_t *err, AlsPlan lp, T *lscratch)
{
    static_assert(G == 16 || G == 32, "16 x 16 or 32 x 32 threads");
    static_assert(!MFMA || (G == 16 && !PACKED && !GLOBAL_L), "the matrix-core Gram path is the 16 x 16 kernel's");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ALS_THREADS = G * G, ALS_CH = G * ALS_RPT;
    constexpr int KP = G * TS, LD = KP + 1;
    T *chunk = (T *)(smem + lp.chunk);    // [2][ALS_CH][KP], columns k..KP-1 are zero
    T *cvals = (T *)(smem + lp.cvals);    // [2][ALS_CH]
    // the factor: column j, unscaled, entry (j, j) = d_j.  Full layout: [k][LD], row j holds column j.  PACKED: column j
    // holds its rows j .. KP-1 only, the columns back to back (lidx)
    T *lmat;
    if constexpr (GLOBAL_L) lmat = lscratch + (size_t)blockIdx.x * ((size_t)KP * (KP + 1) / 2);
    else lmat = (T *)(smem + lp.lmat);
    auto lidx = [](int j, int i) { return PACKED ? j * KP - j * (j - 1) / 2 + (i - j) : j * LD + i; };   // column j, row i (>= j)
    T *zfin = (T *)(smem + lp.zfin);      // [KP]: eliminated right-hand side
    T *dinv = zfin + KP;                  // [KP]: 1 / d_j
    int *ctl = (int *)(smem + lp.ctl);
    const int tid = threadIdx.x, ty = tid / G, tx = tid % G;
    // a barrier that also orders the workgroup's device-memory traffic when the factor lives there
    auto sync = [] {
        if constexpr (GLOBAL_L) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        if constexpr (GLOBAL_L) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    for (;;) {
        __syncthreads();                  // the previous row's LDS is no longer read
        if (tid == 0) ctl[0] = atomicAdd(next_row, 1);
        __syncthreads();
        const int64_t r = ctl[0];
        if (r >= n_rows) break;
        const int64_t lo = row_ptrs[r], hi = row_ptrs[r + 1];
        const bool broken = lo < 0 || hi < lo || hi > nnz;     // a CSR that is not one: flag it, touch nothing of it
        if (broken && tid == 0) atomicOr(err, 2);
        if (hi == lo || broken) {         // lambda I x = 0 (ALSRecommender.py:135-139 with an empty fv)
            for (int c = tid; c < k; c += ALS_THREADS) latent[r * k + c] = 0;
            continue;
        }
        T acc[TS][TS], zacc[TS];
#pragma unroll
        for (int a = 0; a < TS; ++a) {
            zacc[a] = 0;
#pragma unroll
            for (int b = 0; b < TS; ++b) acc[a][b] = 0;
        }
        // ---- 1. Gram matrix and right-hand side.  The gather of chunk c+1 (HBM latency) is in flight while chunk c
        // is multiplied out of LDS; two LDS buffers, one barrier per chunk.
        T pre[ALS_RPT][TS], prev[ALS_RPT];
        auto fetch = [&](int64_t base) {          // thread row ty of the 16 x 16 arrangement fetches factor rows ty, ty + 16, ..
#pragma unroll
            for (int q = 0; q < ALS_RPT; ++q) {
                const int64_t t = base + ty + G * q;
                bool live = t < hi;
                const int64_t col = live ? col_inds[t] : 0;
                if (live && (col < 0 || col >= n_cols)) { live = false; atomicOr(err, 2); }   // never read outside `fixed`
                const T *src = fixed + (live ? col : 0) * k;
#pragma unroll
                for (int b = 0; b < TS; ++b) {
                    const int c = tx + G * b;
                    pre[q][b] = (live && c < k) ? src[c] : (T)0;
                }
                prev[q] = (live && tx == 0) ? vals[t] : (T)0;
            }
        };
        auto commit = [&](int buf) {
#pragma unroll
            for (int q = 0; q < ALS_RPT; ++q) {
#pragma unroll
                for (int b = 0; b < TS; ++b) chunk[buf * (ALS_CH * KP) + (ty + G * q) * KP + tx + G * b] = pre[q][b];
                if (tx == 0) cvals[buf * ALS_CH + ty + G * q] = prev[q];
            }
        };
        fetch(lo);
        commit(0);
        __syncthreads();
        int buf = 0;
        // matrix-core path: tile t = wave + 4 s of the lower triangle, t = ti (ti + 1) / 2 + tj
        constexpr int NT = TS / 2, NTL = NT * (NT + 1) / 2, SLOTS = MFMA ? (NTL + 3) / 4 : 1;
        f32x16 macc[SLOTS];
        int offa[SLOTS], offb[SLOTS];
        bool own[SLOTS];
        T zt = 0;
        if (MFMA) {
            const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
            for (int sl = 0; sl < SLOTS; ++sl) {
                const int t = wave + 4 * sl;
                int ti = 0;
                while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
                own[sl] = t < NTL;
                offa[sl] = own[sl] ? 32 * ti : 0;
                offb[sl] = own[sl] ? 32 * (t - ti * (ti + 1) / 2) : 0;
#pragma unroll
                for (int e = 0; e < 16; ++e) macc[sl][e] = 0;
            }
        }
        for (int64_t base = lo; base < hi; base += ALS_CH, buf ^= 1) {
            const bool more = base + ALS_CH < hi;
            if (more) fetch(base + ALS_CH);
            const T *ch = chunk + buf * (ALS_CH * KP);
            if constexpr (MFMA) {
                const int colm = tid & 31, kh = (tid >> 5) & 1;   // MFMA operand coordinates of this lane
#pragma unroll
                for (int kk = 0; kk < ALS_CH; kk += 2) {
#pragma unroll
                    for (int sl = 0; sl < SLOTS; ++sl) {
                        // no `if (own[sl])` here: a branch per MFMA puts every one of them behind its own LDS round
                        // trip (seen in the ISA).  A wave without a tile for this slot multiplies tile (0, 0) again
                        // on its own SIMD - nobody waits for it - and drops the result at the hand-over.
                        const float av = ch[(kk + kh) * KP + offa[sl] + colm];
                        const float bv = ch[(kk + kh) * KP + offb[sl] + colm];
                        macc[sl] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, macc[sl], 0, 0, 0);
                    }
                }
                if (tid < KP) {
#pragma unroll
                    for (int m = 0; m < ALS_CH; ++m) zt += cvals[buf * ALS_CH + m] * ch[m * KP + tid];
                }
            } else {
#pragma unroll 4
            for (int m = 0; m < ALS_CH; ++m) {
                T ai[TS], al[TS];
#pragma unroll
                for (int a = 0; a < TS; ++a) { ai[a] = ch[m * KP + ty + G * a]; al[a] = ch[m * KP + tx + G * a]; }
#pragma unroll
                for (int a = 0; a < TS; ++a)
#pragma unroll
                    for (int b = 0; b <= a; ++b) acc[a][b] += ai[a] * al[b];   // tiles b > a lie above the diagonal
                if (ty == 0) {
                    const T v = cvals[buf * ALS_CH + m];
#pragma unroll
                    for (int b = 0; b < TS; ++b) zacc[b] += v * al[b];
                }
            }
            }
            if (more) commit(buf ^ 1);
            __syncthreads();
        }
        if constexpr (MFMA) {
            // hand-over: MFMA tile layout (lane = column, 16 rows in registers) -> LDS -> the threads' cyclic entries
            const int colm = tid & 31, kh = (tid >> 5) & 1;
#pragma unroll
            for (int sl = 0; sl < SLOTS; ++sl) {
                if (own[sl]) {
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        lmat[(offa[sl] + 8 * (e / 4) + 4 * kh + (e % 4)) * LD + offb[sl] + colm] = macc[sl][e];
                }
            }
            if (tid < KP) zfin[tid] = zt;
            __syncthreads();
#pragma unroll
            for (int a = 0; a < TS; ++a)
#pragma unroll
                for (int b = 0; b <= a; ++b) acc[a][b] = lmat[(ty + 16 * a) * LD + tx + 16 * b];
            if (ty == 0) {
#pragma unroll
                for (int b = 0; b < TS; ++b) zacc[b] = zfin[tx + 16 * b];
            }
            __syncthreads();          // the elimination reuses this memory for the factor
        }
        if (ty == tx) {
            const T reg = lambda * (T)(hi - lo + 1);
#pragma unroll
            for (int a = 0; a < TS; ++a) acc[a][a] += reg;
        }
        // ---- 2. L D L^T in registers; column j and z_j go to LDS when they are final.  The loop over the 16-column
        // panels is unrolled, so the panel index jb is a constant in the body: the published column is a fixed
        // register (no select chain), and tiles above or left of the panel - finished rows and columns - are skipped.
        bool bad = false;
#pragma unroll
        for (int jb = 0; jb < TS; ++jb) {
            for (int jt = 0; jt < G; ++jt) {
                const int j = G * jb + jt;
                if (j >= ((MFSGD_ALS_EXP & 1) ? 0 : k)) break;
                if (tx == jt) {
#pragma unroll
                    for (int a = jb; a < TS; ++a)
                        if (!PACKED || ty + G * a >= j) lmat[lidx(j, ty + G * a)] = acc[a][jb];   // packed columns start at their diagonal
                    if (ty == 0) zfin[j] = zacc[jb];
                }
                if constexpr (PACKED) sync();
                else __syncthreads();
                const T d = lmat[lidx(j, j)];
                bad = bad || !(d > 0);
                const T inv = pivot_inv(d);
                if (tid == 0) dinv[j] = inv;
                T ci[TS], cl[TS];
                // rows above the diagonal (inside the panel's first tile) are finished: whatever is read for them - in the
                // packed layout the tail of the previous column - only reaches entries that are never read again
#pragma unroll
                for (int a = jb; a < TS; ++a) { ci[a] = lmat[lidx(j, ty + G * a)] * inv; cl[a] = lmat[lidx(j, tx + G * a)]; }
                // inside the panel's own tiles, entries in rows or columns <= j are finished and never read again, so
                // the update runs unguarded there; the right-hand side is live and is guarded
#pragma unroll
                for (int a = jb; a < TS; ++a)
#pragma unroll
                    for (int b = jb; b <= a; ++b) acc[a][b] -= ci[a] * cl[b];
                if (ty == 0) {
                    const T zj = zfin[j] * inv;
#pragma unroll
                    for (int b = jb; b < TS; ++b)
                        if (tx + G * b > j) zacc[b] -= cl[b] * zj;
                }
            }
        }
        if (bad && tid == 0) atomicOr(err, 1);
        if constexpr (PACKED) sync();
        else __syncthreads();
        // ---- 3. back substitution: x_j = w_j / d_j, then w_l -= A[j][l] x_j for l < j (A[j][l] = column l at row j)
        if (tid < 64) {
            // One wave, k dependent steps: everything a step needs except x_j itself is fetched four steps ahead
            // (column j of the transposed factor, a strided LDS read), w_j and 1/d_j come out of registers with
            // v_readlane (j is wave-uniform), so a step is a handful of ALU instructions, not an LDS round trip.
            constexpr int WC = (KP + 63) / 64;
            T w[WC], x[WC], dv[WC], col[WC];
#pragma unroll
            for (int c = 0; c < WC; ++c) {
                const int l = tid + 64 * c;
                w[c] = l < k ? zfin[l] : (T)0;
                dv[c] = l < k ? dinv[l] : (T)0;
                x[c] = 0;
            }
            const int k_sub = (MFSGD_ALS_EXP & 2) ? 0 : k;
            auto column = [&](int j, T (&out)[WC]) {      // A[j][l] for this lane's l < j: column l at row j
#pragma unroll
                for (int c = 0; c < WC; ++c) {
                    const int l = tid + 64 * c;
                    out[c] = (j >= 0 && l < j) ? lmat[lidx(l, j)] : (T)0;
                }
            };
            constexpr int AHEAD = 4;                      // columns in flight (LDS latency / step time)
            T ring[AHEAD][WC];
#pragma unroll
            for (int q = 0; q < AHEAD; ++q) column(k_sub - 1 - q, ring[q]);
            for (int j0 = k_sub - 1; j0 >= 0; j0 -= AHEAD) {
#pragma unroll
                for (int q = 0; q < AHEAD; ++q) {
                    const int j = j0 - q;
                    if (j < 0) break;                     // wave-uniform
#pragma unroll
                    for (int c = 0; c < WC; ++c) col[c] = ring[q][c];
                    column(j - AHEAD, ring[q]);
                    T wj = 0, dj = 0;
#pragma unroll
                    for (int c = 0; c < WC; ++c)
                        if ((j >> 6) == c) { wj = lane_bcast(w[c], j & 63); dj = lane_bcast(dv[c], j & 63); }
                    const T xj = wj * dj;
#pragma unroll
                    for (int c = 0; c < WC; ++c) {
                        const int l = tid + 64 * c;
                        if (l == j) x[c] = xj;
                        w[c] -= col[c] * xj;              // col is zero for l >= j
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < WC; ++c) { const int l = tid + 64 * c; if (l < k) latent[r * k + l] = x[c]; }
        }
    }
}

template <typename T>
int als_impl(T *latent, const T *fixed, const int64_t *row_ptrs, const int32_t *col_inds, const T *vals, int64_t n_rows,
             int64_t n_cols, int64_t nnz, int32_t k, T lambda, void *stream)
{
    MF_ARG(n_rows >= 0 && n_rows < (int64_t)1 << 31, "als_step: n_rows=%lld out of range", (long long)n_rows);
    if (n_rows == 0) return MFSGD_OK;
    MF_ARG(latent && fixed && row_ptrs, "als_step: null pointer");
    MF_ARG(n_cols >= 1 && nnz >= 0 && (nnz == 0 || (col_inds && vals)), "als_step: n_cols=%lld, nnz=%lld or null CSR arrays",
           (long long)n_cols, (long long)nnz);
    MF_ARG(k >= 1 && k <= 256, "als_step: k=%d outside [1,256]", k);
    MF_ARG(lambda > 0, "als_step: lambda must be positive (the normal matrix has to be positive definite)");
    hipStream_t st = as_stream(stream);
    int32_t *work = nullptr;             // [0] next row, [1] error flag
    MF_HIP(hipMalloc((void **)&work, 16));
    struct Guard { void *p; ~Guard() { if (p) (void)hipFree(p); } } guard{work};
    MF_HIP(hipMemsetAsync(work, 0, 16, st));
    int dev = 0, n_cu = 256;
    MF_HIP(hipGetDevice(&dev));
    MF_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    constexpr size_t LDS_BYTES = (size_t)160 * 1024;
#define MF_ALS_GO(G_, TS_, M_, PK_, GL_)                                                                              \
    {                                                                                                                 \
        auto kern = k_als_rows<T, G_, TS_, M_, PK_, GL_>;                                                             \
        MF_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lp.total));   \
        hipLaunchKernelGGL(kern, dim3((unsigned)n_wg), dim3(G_ * G_), lp.total, st, latent, fixed, row_ptrs,          \
                           col_inds, vals, n_rows, n_cols, nnz, (int)k, lambda, work, work + 1, lp, lscratch);        \
        break;                                                                                                        \
    }
    T *lscratch = nullptr;
    Guard guard2{nullptr};
    if (k <= 128) {
        // fp32 above k = 64: Gram on the matrix cores, which wants the padded size to be a multiple of 32
        const bool mfma = sizeof(T) == 4 && k > 64;
        const int ts = mfma ? 2 * ((k + 31) / 32) : (k + 15) / 16;
        const AlsPlan lp = als_plan<T>(k, ts);
        const int per_cu = std::max(1, std::min(8, (int)(LDS_BYTES / lp.total)));
        const int64_t n_wg = std::min<int64_t>(n_rows, (int64_t)n_cu * per_cu);
#define MF_ALS(TS_, M_) MF_ALS_GO(16, TS_, M_, false, false)
        if constexpr (sizeof(T) == 4) {
            switch (mfma ? 100 + ts : ts) {
                case 1: MF_ALS(1, false) case 2: MF_ALS(2, false) case 3: MF_ALS(3, false) case 4: MF_ALS(4, false)
                case 106: MF_ALS(6, true) case 108: MF_ALS(8, true)
            }
        } else {
            switch (ts) {
                case 1: MF_ALS(1, false) case 2: MF_ALS(2, false) case 3: MF_ALS(3, false) case 4: MF_ALS(4, false)
                case 5: MF_ALS(5, false) case 6: MF_ALS(6, false) case 7: MF_ALS(7, false) case 8: MF_ALS(8, false)
            }
        }
#undef MF_ALS
    } else {
        // 129 <= k <= 256: 32 x 32 threads, the factor packed; in LDS when its triangle fits next to nothing else (the
        // gather buffers share its region), else in a slab of device memory per workgroup
        const int ts = (k + 31) / 32;
        const bool in_lds = als_plan<T>(k, ts, 32, true, true).total <= LDS_BYTES;
        const AlsPlan lp = als_plan<T>(k, ts, 32, true, in_lds);
        MF_ARG(lp.total <= LDS_BYTES, "als_step: k=%d does not fit", k);
        const int64_t n_wg = std::min<int64_t>(n_rows, (int64_t)n_cu);   // 1024 threads: one workgroup per CU
        if (!in_lds) {
            const size_t slab = (size_t)(32 * ts) * (32 * ts + 1) / 2;
            MF_HIP(hipMalloc((void **)&lscratch, (size_t)n_wg * slab * sizeof(T)));
            guard2.p = lscratch;
        }
#define MF_ALS(TS_) { if (in_lds) MF_ALS_GO(32, TS_, false, true, false) else MF_ALS_GO(32, TS_, false, true, true) }
        switch (ts) {
            case 5: MF_ALS(5) break; case 6: MF_ALS(6) break; case 7: MF_ALS(7) break; case 8: MF_ALS(8) break;
        }
#undef MF_ALS
    }
#undef MF_ALS_GO
    MF_LAUNCH_CHECK();
    int32_t h[2] = {0, 0};
    MF_HIP(hipMemcpyAsync(h, work, 8, hipMemcpyDeviceToHost, st));
    MF_HIP(hipStreamSynchronize(st));
    if (h[1]) {
        mfsgd_set_error("als_step: %s%s", (h[1] & 2) ? "row_ptrs / col_inds do not describe a CSR with n_cols columns and nnz entries; " : "",
                        (h[1] & 1) ? "a normal matrix was not positive definite (pivot <= 0 or not finite)" : "");
        return MFSGD_ERR_ARG;
    }
    return MFSGD_OK;
}

}  // namespace

extern "C" {

int mfsgd_hip_als_step_f32(float *latent, const float *fixed, const int64_t *row_ptrs, const int32_t *col_inds,
                           const float *vals, int64_t n_rows, int64_t n_cols, int64_t nnz, int32_t k, float lambda,
                           void *stream)
{
    return als_impl<float>(latent, fixed, row_ptrs, col_inds, vals, n_rows, n_cols, nnz, k, lambda, stream);
}

int mfsgd_hip_als_step_f64(double *latent, const double *fixed, const int64_t *row_ptrs, const int32_t *col_inds,
                           const double *vals, int64_t n_rows, int64_t n_cols, int64_t nnz, int32_t k, double lambda,
                           void *stream)
{
    return als_impl<double>(latent, fixed, row_ptrs, col_inds, vals, n_rows, n_cols, nnz, k, lambda, stream);
}

}  // extern "C"
