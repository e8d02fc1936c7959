// sgd_serial.hip - exact-order SGD pass, RMSE and the implicit-term kernel.
//
// k_sgd_serial restates numba_methods.sgd / sgd2 (numba_methods.py:11-23, :31-55) and the block-local
// sgd (RayBlockScheduler.py:127-143) for ONE workgroup: ratings are taken strictly in array order (the
// reference's loop order, including the rewrite of every y[j], j in N(u), after each rating), the
// parallelism is across the k factors: thread t owns factor columns t, t+256, ... of P, Q and y for the
// whole pass, so no two threads ever touch the same word and the only exchange per rating is the k-wide
// dot (wave DPP sum + 4-word LDS combine, fixed order).
#include "common.h"
#include <mutex>

// ---- scratch of the allocation-free entry point (mfsgd_hip_rmse_noalloc_f32): the per-block partial sums
// live in a file-scope device array instead of a caller workspace.  In host code `scratch` is only a stand-in
// for the symbol, never a device address: the address is resolved once per device with hipGetSymbolAddress
// (each GPU holds its own copy of the array) and only that pointer is handed to kernels.
constexpr int SCRATCH_DOUBLES = 2048;   // = RMSE_BLOCKS (static_assert below)
static __device__ __align__(256) double scratch[SCRATCH_DOUBLES];   // 16 KB per GPU, held until the process ends
constexpr int MAX_DEVICES = 64;
static double *g_scratch_ptr[MAX_DEVICES];
static std::mutex g_scratch_mu;

static int scratch_ptr(double **out, int *dev_out)
{
    int dev = 0;
    MF_HIP(hipGetDevice(&dev));
    MF_ARG(dev >= 0 && dev < MAX_DEVICES, "rmse_noalloc: device ordinal %d outside [0,%d)", dev, MAX_DEVICES);
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        void *p = nullptr;
        MF_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));   // with `dev` current: this device's copy
        g_scratch_ptr[dev] = (double *)p;
    }
    *out = g_scratch_ptr[dev];
    if (dev_out) *dev_out = dev;
    return MFSGD_OK;
}

// Resolves the current device's scratch address ahead of the first mfsgd_hip_rmse_noalloc_f32 call: the resolution
// (hipGetSymbolAddress may load the code object) is neither capture-safe nor strictly allocation-free, so it belongs
// to set-up.  mfsgd_device_check() - the library's init call, made with the device current - calls this.
__attribute__((visibility("hidden"))) int mfsgd_internal_scratch_init(void)
{
    double *p = nullptr;
    return scratch_ptr(&p, nullptr);
}

namespace {

constexpr int SERIAL_THREADS = 256;
constexpr int SERIAL_MAXF = 4;  // k <= 1024

template <typename T>
__global__ __launch_bounds__(SERIAL_THREADS) void k_sgd_serial(
    T *__restrict__ P, T *__restrict__ Q, T *__restrict__ b_u, T *__restrict__ b_i, T b, T *__restrict__ y,
    const T *__restrict__ z, const int32_t *__restrict__ samples, int64_t n, int32_t stride,
    const int64_t *__restrict__ row_ptrs, const int32_t *__restrict__ col_inds, int32_t k, int32_t u_off,
    int32_t i_off, T alpha, T beta1, T beta2, int32_t variant, int32_t implicit)
{
    __shared__ T s_red[2][SERIAL_THREADS / 64];
    const int tid = threadIdx.x, wave = tid >> 6;
    for (int64_t t = 0; t < n; ++t) {
        const int32_t ug = samples[t * stride], ig = samples[t * stride + 1];
        const T r = (T)samples[t * stride + 2];
        T *p = P + (int64_t)(ug - u_off) * k, *q = Q + (int64_t)(ig - i_off) * k;
        int64_t lo = 0, hi = 0;
        T R = 1;
        T ys[SERIAL_MAXF], pf[SERIAL_MAXF], qf[SERIAL_MAXF];
#pragma unroll
        for (int c = 0; c < SERIAL_MAXF; ++c) ys[c] = 0;
        if (implicit) {
            if (implicit == 1 && z) {
#pragma unroll
                for (int c = 0; c < SERIAL_MAXF; ++c) {
                    const int f = tid + c * SERIAL_THREADS;
                    if (f < k) ys[c] = z[(int64_t)ug * k + f];
                }
            } else {
                lo = row_ptrs[ug]; hi = row_ptrs[ug + 1];
                R = (T)sqrt((double)(hi - lo)) + (variant == 1 ? (T)1 : (T)0);
                for (int64_t j = lo; j < hi; ++j) {
                    const T *yr = y + (int64_t)col_inds[j] * k;
#pragma unroll
                    for (int c = 0; c < SERIAL_MAXF; ++c) {
                        const int f = tid + c * SERIAL_THREADS;
                        if (f < k) ys[c] += yr[f];
                    }
                }
                const bool ok = hi > lo || variant == 1;  // sqrt(0): implicit term defined as 0
#pragma unroll
                for (int c = 0; c < SERIAL_MAXF; ++c) ys[c] = ok ? ys[c] / R : (T)0;
            }
        }
        T part = 0;
#pragma unroll
        for (int c = 0; c < SERIAL_MAXF; ++c) {
            const int f = tid + c * SERIAL_THREADS;
            pf[c] = 0; qf[c] = 0;
            if (f < k) { pf[c] = p[f]; qf[c] = q[f]; part += qf[c] * (pf[c] + ys[c]); }
        }
        // Biases: every thread loads them (one address per wave: a broadcast), every thread computes the
        // same e and every thread stores the same new values - no `if (tid == 0)` inside a loop with a
        // barrier (hipcc may re-enter such a loop body with part of a wave, see epoch.hip sink_ptr), and no
        // extra barrier: a thread's own store precedes its own later load of that word, and all stores of
        // one rating carry identical bits.
        const T bu = __hip_atomic_load(&b_u[ug - u_off], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const T bi = __hip_atomic_load(&b_i[ig - i_off], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        part = wave_sum(part);
        s_red[t & 1][wave] = part;   // all lanes of the wave, same value
        __syncthreads();  // the only barrier per rating (the slot is reused two ratings later)
        T pred = b + bu + bi;
#pragma unroll
        for (int w = 0; w < SERIAL_THREADS / 64; ++w) pred += s_red[t & 1][w];
        const T e = r - pred;
        __hip_atomic_store(&b_u[ug - u_off], bu + alpha * (e - beta1 * bu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_store(&b_i[ig - i_off], bi + alpha * (e - beta1 * bi), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (variant == 0) {
#pragma unroll
            for (int c = 0; c < SERIAL_MAXF; ++c) {
                const int f = tid + c * SERIAL_THREADS;
                if (f < k) {
                    const T pn = pf[c] + alpha * (e * qf[c] - beta2 * pf[c]);
                    const T qn = qf[c] + alpha * (e * (pn + ys[c]) - beta2 * qf[c]);
                    p[f] = pn; q[f] = qn; qf[c] = qn;
                }
            }
        } else {
#pragma unroll
            for (int c = 0; c < SERIAL_MAXF; ++c) {
                const int f = tid + c * SERIAL_THREADS;
                if (f < k) {
                    p[f] = pf[c] + alpha * (e * qf[c] - beta2 * pf[c]);
                    q[f] = qf[c] + alpha * (e * (pf[c] + ys[c]) - beta2 * qf[c]);
                    // sgd2 feeds the OLD q into the y update (numba_methods.py:51,55): qf stays
                }
            }
        }
        if (implicit == 2) {
            const T eR = e / R;
            for (int64_t j = lo; j < hi; ++j) {
                T *yr = y + (int64_t)col_inds[j] * k;
#pragma unroll
                for (int c = 0; c < SERIAL_MAXF; ++c) {
                    const int f = tid + c * SERIAL_THREADS;
                    if (f < k) {
                        const T yo = yr[f];
                        yr[f] = variant == 0 ? yo + alpha * ((T)-1.0 * beta2 * yo + eR * qf[c])
                                             : yo + alpha * (eR * qf[c] - beta2 * yo);
                    }
                }
            }
        }
    }
}

constexpr int RMSE_THREADS = 256;
constexpr int RMSE_BLOCKS = 2048;
static_assert(RMSE_BLOCKS <= SCRATCH_DOUBLES, "the no-alloc scratch must hold one partial per block");

template <typename T>
__global__ __launch_bounds__(RMSE_THREADS) void k_rmse(
    const T *__restrict__ P, const T *__restrict__ Q, const T *__restrict__ b_u, const T *__restrict__ b_i, T b,
    const T *__restrict__ y, const T *__restrict__ z, const int32_t *__restrict__ samples, int64_t n,
    int32_t stride, const int64_t *__restrict__ row_ptrs, const int32_t *__restrict__ col_inds, int32_t k,
    int32_t implicit, double *__restrict__ partials)
{
    __shared__ double s_acc[RMSE_THREADS / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t nwaves = (int64_t)gridDim.x * (RMSE_THREADS / 64);
    double acc = 0;
    for (int64_t t = (int64_t)blockIdx.x * (RMSE_THREADS / 64) + wave; t < n; t += nwaves) {
        const int32_t u = samples[t * stride], i = samples[t * stride + 1];
        const T r = (T)samples[t * stride + 2];
        T part = 0;
        int64_t lo = 0, hi = 0;
        T R = 1;
        if (implicit && !z) { lo = row_ptrs[u]; hi = row_ptrs[u + 1]; R = (T)sqrt((double)(hi - lo)); }
        for (int f = lane; f < k; f += 64) {
            T ys = 0;
            if (implicit) {
                if (z) ys = z[(int64_t)u * k + f];
                else {
                    for (int64_t j = lo; j < hi; ++j) ys += y[(int64_t)col_inds[j] * k + f];
                    ys = hi > lo ? ys / R : (T)0;
                }
            }
            part += Q[(int64_t)i * k + f] * (P[(int64_t)u * k + f] + ys);
        }
        const T pred = wave_sum(part) + b + b_u[u] + b_i[i];
        const double d = (double)(r - pred);
        acc += d * d;
    }
    if (lane == 0) s_acc[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0;
        for (int w = 0; w < RMSE_THREADS / 64; ++w) s += s_acc[w];
        partials[blockIdx.x] = s;
    }
}

__global__ void k_sum_partials(const double *__restrict__ partials, int nb, double *__restrict__ out)
{
    // one wave, fixed order: lane l adds partials l, l+64, ...; then the fixed shuffle tree
    double s = 0;
    for (int i = threadIdx.x; i < nb; i += 64) s += partials[i];
    s = wave_sum(s);
    if (threadIdx.x == 0) *out = s;
}

template <typename T>
__global__ __launch_bounds__(256) void k_implicit_z(const T *__restrict__ y, const int64_t *__restrict__ row_ptrs,
                                                   const int32_t *__restrict__ col_inds, int64_t n_users, int32_t k,
                                                   int32_t plus_one, T *__restrict__ z)
{
    const int lane = threadIdx.x & 63;
    const int64_t u = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (u >= n_users) return;
    const int64_t lo = row_ptrs[u], hi = row_ptrs[u + 1];
    const T R = (T)sqrt((double)(hi - lo)) + (plus_one ? (T)1 : (T)0);
    for (int f = lane; f < k; f += 64) {
        T s = 0;
        for (int64_t j = lo; j < hi; ++j) s += y[(int64_t)col_inds[j] * k + f];
        z[u * k + f] = (hi > lo || plus_one) ? s / R : (T)0;
    }
}

template <typename T>
int sgd_impl(T *P, T *Q, T *b_u, T *b_i, T b, T *y, const T *z, const int32_t *samples, int64_t n, int32_t stride,
             const int64_t *row_ptrs, const int32_t *col_inds, int32_t k, int32_t u_off, int32_t i_off, T alpha,
             T beta1, T beta2, int32_t variant, int32_t implicit, void *stream)
{
    MF_ARG(P && Q && b_u && b_i && (samples || n == 0), "sgd: null factor/sample pointer");
    MF_ARG(n >= 0 && stride >= 3, "sgd: need n >= 0 and stride >= 3 (got %lld, %d)", (long long)n, stride);
    MF_ARG(k >= 1 && k <= SERIAL_THREADS * SERIAL_MAXF, "sgd: k=%d outside [1,%d]", k, SERIAL_THREADS * SERIAL_MAXF);
    MF_ARG(variant == 0 || variant == 1, "sgd: variant must be 0 (sgd) or 1 (sgd2)");
    MF_ARG(implicit >= 0 && implicit <= 2, "sgd: implicit must be 0, 1 or 2");
    MF_ARG(implicit == 0 || (implicit == 1 && z) || (y && row_ptrs && col_inds),
           "sgd: implicit=%d needs y,row_ptrs,col_inds (or z for implicit=1)", implicit);
    if (n == 0) return MFSGD_OK;
    hipLaunchKernelGGL(k_sgd_serial<T>, dim3(1), dim3(SERIAL_THREADS), 0, as_stream(stream), P, Q, b_u, b_i, b, y, z,
                       samples, n, stride, row_ptrs, col_inds, k, u_off, i_off, alpha, beta1, beta2, variant, implicit);
    MF_LAUNCH_CHECK();
    return MFSGD_OK;
}

template <typename T>
int rmse_impl(const T *P, const T *Q, const T *b_u, const T *b_i, T b, const T *y, const T *z,
              const int32_t *samples, int64_t n, int32_t stride, const int64_t *row_ptrs, const int32_t *col_inds,
              int32_t k, int32_t implicit, double *d_partials, double *d_sse, void *stream)
{
    MF_ARG(P && Q && b_u && b_i && d_partials && d_sse, "rmse: null pointer");
    MF_ARG(n >= 0 && stride >= 3 && k >= 1, "rmse: bad n/stride/k");
    MF_ARG(implicit == 0 || z || (y && row_ptrs && col_inds), "rmse: implicit needs z or y,row_ptrs,col_inds");
    int nb = (int)((n + 3) / 4);
    if (nb > RMSE_BLOCKS) nb = RMSE_BLOCKS;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(k_rmse<T>, dim3(nb), dim3(RMSE_THREADS), 0, as_stream(stream), P, Q, b_u, b_i, b, y, z, samples,
                       n, stride, row_ptrs, col_inds, k, implicit ? 1 : 0, d_partials);
    MF_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(64), 0, as_stream(stream), d_partials, nb, d_sse);
    MF_LAUNCH_CHECK();
    return MFSGD_OK;
}

template <typename T>
int z_impl(const T *y, const int64_t *row_ptrs, const int32_t *col_inds, int64_t n_users, int32_t k, int32_t plus_one,
           T *z, void *stream)
{
    MF_ARG(y && row_ptrs && col_inds && z && n_users >= 0 && k >= 1, "implicit_z: bad argument");
    if (n_users == 0) return MFSGD_OK;
    hipLaunchKernelGGL(k_implicit_z<T>, dim3((unsigned)((n_users + 3) / 4)), dim3(256), 0, as_stream(stream), y,
                       row_ptrs, col_inds, n_users, k, plus_one, z);
    MF_LAUNCH_CHECK();
    return MFSGD_OK;
}

}  // namespace

extern "C" {

int mfsgd_hip_sgd_f32(float *P, float *Q, float *b_u, float *b_i, float b, float *y, const float *z,
                      const int32_t *samples, int64_t n, int32_t stride, const int64_t *row_ptrs,
                      const int32_t *col_inds, int32_t k, int32_t u_off, int32_t i_off, float alpha, float beta1,
                      float beta2, int32_t variant, int32_t implicit, void *stream)
{
    return sgd_impl<float>(P, Q, b_u, b_i, b, y, z, samples, n, stride, row_ptrs, col_inds, k, u_off, i_off, alpha,
                           beta1, beta2, variant, implicit, stream);
}
int mfsgd_hip_sgd_f64(double *P, double *Q, double *b_u, double *b_i, double b, double *y, const double *z,
                      const int32_t *samples, int64_t n, int32_t stride, const int64_t *row_ptrs,
                      const int32_t *col_inds, int32_t k, int32_t u_off, int32_t i_off, double alpha, double beta1,
                      double beta2, int32_t variant, int32_t implicit, void *stream)
{
    return sgd_impl<double>(P, Q, b_u, b_i, b, y, z, samples, n, stride, row_ptrs, col_inds, k, u_off, i_off, alpha,
                            beta1, beta2, variant, implicit, stream);
}

int64_t mfsgd_rmse_partials(int64_t) { return RMSE_BLOCKS; }

int mfsgd_hip_rmse_noalloc_f32(const float *P, const float *Q, const float *b_u, const float *b_i, float b,
                               const float *y, const float *z, const int32_t *samples, int64_t n, int32_t stride,
                               const int64_t *row_ptrs, const int32_t *col_inds, int32_t k, int32_t implicit,
                               double *d_sse, void *stream)
{
    MF_ARG(n <= MFSGD_RMSE_NOALLOC_MAX_N, "rmse_noalloc: n=%lld exceeds the limit of %lld ratings per call",
           (long long)n, (long long)MFSGD_RMSE_NOALLOC_MAX_N);
    MF_ARG(k <= MFSGD_RMSE_NOALLOC_MAX_K, "rmse_noalloc: k=%d exceeds the limit of %d factors", k,
           MFSGD_RMSE_NOALLOC_MAX_K);
    MF_ARG(P && d_sse, "rmse_noalloc: null pointer");
    double *partials = nullptr;
    int dev = 0;
    const int rc = scratch_ptr(&partials, &dev);
    if (rc) return rc;
    // the scratch is the CURRENT device's copy: outputs and inputs must live on that device (a pointer query, no
    // stream work: legal under capture), else the kernels would be handed another GPU's array
    for (const void *q : {(const void *)P, (const void *)d_sse}) {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, q) == hipSuccess && at.type == hipMemoryTypeDevice)
            MF_ARG(at.device == dev, "rmse_noalloc: an argument lives on device %d but device %d is current (its scratch "
                   "would be used): make the arguments' device current", at.device, dev);
        else
            (void)hipGetLastError();   // host-registered / managed / unknown: nothing to compare
    }
    // every block overwrites its own partial and k_sum_partials reads only the blocks of THIS call:
    // nothing left behind by an earlier (larger) call is ever read, so nothing has to be cleared
    return rmse_impl<float>(P, Q, b_u, b_i, b, y, z, samples, n, stride, row_ptrs, col_inds, k, implicit, partials,
                            d_sse, stream);
}

int mfsgd_hip_rmse_f32(const float *P, const float *Q, const float *b_u, const float *b_i, float b, const float *y,
                       const float *z, const int32_t *samples, int64_t n, int32_t stride, const int64_t *row_ptrs,
                       const int32_t *col_inds, int32_t k, int32_t implicit, double *d_partials, double *d_sse,
                       void *stream)
{
    return rmse_impl<float>(P, Q, b_u, b_i, b, y, z, samples, n, stride, row_ptrs, col_inds, k, implicit, d_partials,
                            d_sse, stream);
}
int mfsgd_hip_rmse_f64(const double *P, const double *Q, const double *b_u, const double *b_i, double b,
                       const double *y, const double *z, const int32_t *samples, int64_t n, int32_t stride,
                       const int64_t *row_ptrs, const int32_t *col_inds, int32_t k, int32_t implicit,
                       double *d_partials, double *d_sse, void *stream)
{
    return rmse_impl<double>(P, Q, b_u, b_i, b, y, z, samples, n, stride, row_ptrs, col_inds, k, implicit, d_partials,
                             d_sse, stream);
}

int mfsgd_hip_implicit_z_f32(const float *y, const int64_t *row_ptrs, const int32_t *col_inds, int64_t n_users,
                             int32_t k, int32_t plus_one, float *z, void *stream)
{
    return z_impl<float>(y, row_ptrs, col_inds, n_users, k, plus_one, z, stream);
}
int mfsgd_hip_implicit_z_f64(const double *y, const int64_t *row_ptrs, const int32_t *col_inds, int64_t n_users,
                             int32_t k, int32_t plus_one, double *z, void *stream)
{
    return z_impl<double>(y, row_ptrs, col_inds, n_users, k, plus_one, z, stream);
}

}  // extern "C"
