// prep.hip - integer data-prep kernels (bit-exact with the reference): id remap, test filter,
// P x P bucketing, COO -> CSR.  HBM-bound byte work; the stable orderings come from an LSD radix
// sort of (key, original index) pairs (rocPRIM device sort, which is stable), everything else is
// hand-written.  These entry points allocate temporaries and synchronise; they are not on the
// per-epoch path.
#include "common.h"
#include <hipcub/hipcub.hpp>
#include <stdarg.h>

static thread_local char g_err[512] = "";

void mfsgd_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *mfsgd_last_error(void) { return g_err; }
extern "C" int mfsgd_version(void) { return 100; }

int mfsgd_internal_scratch_init(void);

extern "C" int mfsgd_device_check(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) {
        mfsgd_set_error("no HIP device visible: libmfsgd has no CPU fallback");
        return MFSGD_ERR_HIP;
    }
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        mfsgd_set_error("hipGetDeviceProperties failed");
        return MFSGD_ERR_HIP;
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        mfsgd_set_error("device %d is %s; libmfsgd is built for gfx950 only", dev, prop.gcnArchName);
        return MFSGD_ERR_UNSUPPORTED;
    }
    // set-up of the allocation-free entry point for this device (csrc/sgd_serial.hip)
    if (const int rc = mfsgd_internal_scratch_init()) return rc;
    return n;
}

namespace mfprep {

struct DevBuf {  // RAII device temporary
    void *p = nullptr;
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
    ~DevBuf() { if (p) (void)hipFree(p); }
    template <typename U> U *as() { return (U *)p; }
};

__global__ void k_remap(int32_t *samples, int64_t n, int32_t stride, const int32_t *__restrict__ row_perm,
                        const int32_t *__restrict__ col_perm, int64_t n_users, int64_t n_items, int32_t *err)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int32_t u = samples[t * stride], i = samples[t * stride + 1];
    if (u < 0 || u >= n_users || i < 0 || i >= n_items) { *err = 1; return; }
    samples[t * stride] = row_perm[u];
    samples[t * stride + 1] = col_perm[i];
}

__global__ void k_mark_seen(const int32_t *__restrict__ train, int64_t n, int32_t stride, int64_t n_users,
                            int64_t n_items, uint8_t *seen_u, uint8_t *seen_i, int32_t *err)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int32_t u = train[t * stride], i = train[t * stride + 1];
    if (u < 0 || u >= n_users || i < 0 || i >= n_items) { *err = 1; return; }   // never write outside the tables
    seen_u[u] = 1;  // benign same-value race
    seen_i[i] = 1;
}

// read-only pass in front of k_remap: a bad id must be reported before anything is rewritten in place
__global__ void k_check_ids(const int32_t *__restrict__ samples, int64_t n, int32_t stride, int64_t n_users,
                            int64_t n_items, int32_t *err)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int32_t u = samples[t * stride], i = samples[t * stride + 1];
    if (u < 0 || u >= n_users || i < 0 || i >= n_items) *err = 1;
}

__global__ void k_flag_test(const int32_t *__restrict__ test, int64_t nt, int32_t stride,
                            const uint8_t *__restrict__ seen_u, const uint8_t *__restrict__ seen_i, int64_t n_users,
                            int64_t n_items, uint8_t *flags, int64_t *idx)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nt) return;
    const int32_t u = test[t * stride], i = test[t * stride + 1];
    flags[t] = (u >= 0 && u < n_users && i >= 0 && i < n_items && seen_u[u] && seen_i[i]) ? 1 : 0;
    idx[t] = t;
}

__global__ void k_gather_rows(const int32_t *__restrict__ src, int32_t sstride, const int64_t *__restrict__ idx,
                              const int64_t *__restrict__ d_count, int64_t n_max, int32_t *dst, int32_t dstride,
                              int32_t ncols)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t m = d_count ? *d_count : n_max;
    if (t >= m) return;
    const int64_t s = idx[t];
    for (int c = 0; c < ncols; ++c) dst[t * dstride + c] = src[s * sstride + c];
}

// key = block id (or user id); idx = original position
__global__ void k_block_keys(const int32_t *__restrict__ samples, int64_t n, int32_t stride, int32_t ubp, int32_t ibp,
                             int32_t Pu, int32_t Pi, uint32_t *keys, int64_t *idx, int32_t *err)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int32_t u = samples[t * stride], i = samples[t * stride + 1];
    const int32_t r = u / ubp, c = i / ibp;
    if (u < 0 || i < 0 || r >= Pu || c >= Pi) { *err = 1; keys[t] = 0; idx[t] = t; return; }
    keys[t] = (uint32_t)r * (uint32_t)Pi + (uint32_t)c;
    idx[t] = t;
}

__global__ void k_user_keys(const int32_t *__restrict__ samples, int64_t n, int32_t stride, int64_t n_users,
                            uint32_t *keys, int64_t *idx, int32_t *err)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int32_t u = samples[t * stride];
    if (u < 0 || u >= n_users) { *err = 1; keys[t] = 0; idx[t] = t; return; }
    keys[t] = (uint32_t)u;
    idx[t] = t;
}

// sorted keys -> segment pointers: ptr[b] = first position whose key >= b, ptr[nb] = n
__global__ void k_segment_ptr(const uint32_t *__restrict__ keys, int64_t n, int64_t nb, int64_t *ptr)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t > n) return;
    const int64_t lo = t == 0 ? 0 : (int64_t)keys[t - 1] + 1;
    const int64_t hi = t == n ? nb : (int64_t)keys[t];
    for (int64_t b = lo; b <= hi; ++b) ptr[b] = t;
}

__global__ void k_gather_cols(const int32_t *__restrict__ samples, int32_t stride, const int64_t *__restrict__ idx,
                              int64_t n, int32_t *col_inds)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) col_inds[t] = samples[idx[t] * stride + 1];
}

inline unsigned nblk(int64_t n, int bs = 256) { return (unsigned)((n + bs - 1) / bs); }

inline int bits_for(uint64_t maxkey)
{
    int b = 1;
    while (b < 32 && (maxkey >> b)) ++b;
    return b;
}

// stable sort of (keys, idx) by key; out arrays distinct from in
int sort_pairs(uint32_t *k_in, uint32_t *k_out, int64_t *v_in, int64_t *v_out, int64_t n, int end_bit, hipStream_t st)
{
    MF_ARG(n < (int64_t)1 << 31, "sort: n=%lld does not fit int32", (long long)n);
    size_t tmp_bytes = 0;
    MF_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, k_in, k_out, v_in, v_out, (int)n, 0, end_bit, st));
    DevBuf tmp;
    MF_HIP(tmp.alloc(tmp_bytes));
    MF_HIP(hipcub::DeviceRadixSort::SortPairs(tmp.p, tmp_bytes, k_in, k_out, v_in, v_out, (int)n, 0, end_bit, st));
    MF_HIP(hipStreamSynchronize(st));
    return MFSGD_OK;
}

__global__ void k_rank_of(const int64_t *__restrict__ values, int64_t n, const int64_t *__restrict__ uniq,
                          const int64_t *__restrict__ n_uniq, int32_t *__restrict__ out)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int64_t v = values[t];
    int64_t lo = 0, hi = *n_uniq;   // first position with uniq[pos] >= v; v is present
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (uniq[mid] < v) lo = mid + 1; else hi = mid;
    }
    out[t] = (int32_t)lo;
}

}  // namespace mfprep

using namespace mfprep;

extern "C" int mfsgd_hip_remap(int32_t *samples, int64_t n, int32_t stride, const int32_t *row_perm,
                               const int32_t *col_perm, int64_t n_users, int64_t n_items, void *stream)
{
    MF_ARG(n >= 0 && stride >= 2 && (n == 0 || (samples && row_perm && col_perm)), "remap: bad argument");
    if (n == 0) return MFSGD_OK;
    hipStream_t st = as_stream(stream);
    DevBuf err;
    MF_HIP(err.alloc(4));
    MF_HIP(hipMemsetAsync(err.p, 0, 4, st));
    // the remap is in place: ids are validated in a read-only pass first, so a failed call changes nothing
    hipLaunchKernelGGL(k_check_ids, dim3(nblk(n)), dim3(256), 0, st, samples, n, stride, n_users, n_items,
                       err.as<int32_t>());
    MF_LAUNCH_CHECK();
    int32_t h = 0;
    MF_HIP(hipMemcpyAsync(&h, err.p, 4, hipMemcpyDeviceToHost, st));
    MF_HIP(hipStreamSynchronize(st));
    MF_ARG(h == 0, "remap: an id is outside [0,n_users) x [0,n_items); nothing was rewritten");
    hipLaunchKernelGGL(k_remap, dim3(nblk(n)), dim3(256), 0, st, samples, n, stride, row_perm, col_perm, n_users,
                       n_items, err.as<int32_t>());
    MF_LAUNCH_CHECK();
    MF_HIP(hipStreamSynchronize(st));
    return MFSGD_OK;
}

extern "C" int mfsgd_hip_filter_test(const int32_t *test, int64_t nt, int32_t stride, const int32_t *train, int64_t n,
                                     int32_t tstride, int64_t n_users, int64_t n_items, int32_t *out, int64_t *h_count,
                                     void *stream)
{
    MF_ARG(nt >= 0 && n >= 0 && stride >= 2 && tstride >= 2 && h_count, "filter_test: bad argument");
    *h_count = 0;
    if (nt == 0) return MFSGD_OK;
    hipStream_t st = as_stream(stream);
    MF_ARG(n_users > 0 && n_items > 0, "filter_test: n_users and n_items must be positive");
    DevBuf su, si, flags, idx, sel, cnt, tmp, err;
    MF_HIP(su.alloc((size_t)n_users)); MF_HIP(si.alloc((size_t)n_items));
    MF_HIP(err.alloc(4));
    MF_HIP(hipMemsetAsync(err.p, 0, 4, st));
    MF_HIP(flags.alloc((size_t)nt)); MF_HIP(idx.alloc((size_t)nt * 8)); MF_HIP(sel.alloc((size_t)nt * 8));
    MF_HIP(cnt.alloc(8));
    MF_HIP(hipMemsetAsync(su.p, 0, (size_t)n_users ? (size_t)n_users : 1, st));
    MF_HIP(hipMemsetAsync(si.p, 0, (size_t)n_items ? (size_t)n_items : 1, st));
    if (n) {
        hipLaunchKernelGGL(k_mark_seen, dim3(nblk(n)), dim3(256), 0, st, train, n, tstride, n_users, n_items,
                           su.as<uint8_t>(), si.as<uint8_t>(), err.as<int32_t>());
        MF_LAUNCH_CHECK();
        int32_t h = 0;
        MF_HIP(hipMemcpyAsync(&h, err.p, 4, hipMemcpyDeviceToHost, st));
        MF_HIP(hipStreamSynchronize(st));
        MF_ARG(h == 0, "filter_test: a train id is outside [0,n_users) x [0,n_items)");
    }
    hipLaunchKernelGGL(k_flag_test, dim3(nblk(nt)), dim3(256), 0, st, test, nt, stride, su.as<uint8_t>(),
                       si.as<uint8_t>(), n_users, n_items, flags.as<uint8_t>(), idx.as<int64_t>());
    MF_LAUNCH_CHECK();
    MF_ARG(nt < (int64_t)1 << 31, "filter_test: nt does not fit int32");
    size_t tb = 0;
    MF_HIP(hipcub::DeviceSelect::Flagged(nullptr, tb, idx.as<int64_t>(), flags.as<uint8_t>(), sel.as<int64_t>(),
                                         cnt.as<int64_t>(), (int)nt, st));
    MF_HIP(tmp.alloc(tb));
    MF_HIP(hipcub::DeviceSelect::Flagged(tmp.p, tb, idx.as<int64_t>(), flags.as<uint8_t>(), sel.as<int64_t>(),
                                         cnt.as<int64_t>(), (int)nt, st));
    hipLaunchKernelGGL(k_gather_rows, dim3(nblk(nt)), dim3(256), 0, st, test, stride, sel.as<int64_t>(),
                       cnt.as<int64_t>(), nt, out, stride, stride);
    MF_LAUNCH_CHECK();
    MF_HIP(hipMemcpyAsync(h_count, cnt.p, 8, hipMemcpyDeviceToHost, st));
    MF_HIP(hipStreamSynchronize(st));
    return MFSGD_OK;
}

extern "C" int mfsgd_hip_bucket(const int32_t *samples, int64_t n, int32_t stride, int32_t ubp, int32_t ibp,
                                int32_t Pu, int32_t Pi, int32_t *out_samples, int64_t *out_block_ptr, void *stream)
{
    MF_ARG(n >= 0 && stride >= 3 && ubp > 0 && ibp > 0 && Pu > 0 && Pi > 0 && out_block_ptr, "bucket: bad argument");
    MF_ARG((int64_t)Pu * Pi < ((int64_t)1 << 31), "bucket: grid %d x %d too large", Pu, Pi);
    hipStream_t st = as_stream(stream);
    const int64_t nb = (int64_t)Pu * Pi;
    if (n == 0) {
        MF_HIP(hipMemsetAsync(out_block_ptr, 0, sizeof(int64_t) * (size_t)(nb + 1), st));
        MF_HIP(hipStreamSynchronize(st));
        return MFSGD_OK;
    }
    MF_ARG(samples && out_samples, "bucket: null sample pointer");
    DevBuf k0, k1, v0, v1, err;
    MF_HIP(k0.alloc((size_t)n * 4)); MF_HIP(k1.alloc((size_t)n * 4));
    MF_HIP(v0.alloc((size_t)n * 8)); MF_HIP(v1.alloc((size_t)n * 8));
    MF_HIP(err.alloc(4));
    MF_HIP(hipMemsetAsync(err.p, 0, 4, st));
    hipLaunchKernelGGL(k_block_keys, dim3(nblk(n)), dim3(256), 0, st, samples, n, stride, ubp, ibp, Pu, Pi,
                       k0.as<uint32_t>(), v0.as<int64_t>(), err.as<int32_t>());
    MF_LAUNCH_CHECK();
    int32_t h = 0;
    MF_HIP(hipMemcpyAsync(&h, err.p, 4, hipMemcpyDeviceToHost, st));
    MF_HIP(hipStreamSynchronize(st));
    if (h) { mfsgd_set_error("bucket: a sample falls outside the %d x %d grid (ubp=%d, ibp=%d)", Pu, Pi, ubp, ibp); return MFSGD_ERR_GRID; }
    int rc = sort_pairs(k0.as<uint32_t>(), k1.as<uint32_t>(), v0.as<int64_t>(), v1.as<int64_t>(), n,
                        bits_for((uint64_t)nb), st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_segment_ptr, dim3(nblk(n + 1)), dim3(256), 0, st, k1.as<uint32_t>(), n, nb, out_block_ptr);
    MF_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_gather_rows, dim3(nblk(n)), dim3(256), 0, st, samples, stride, v1.as<int64_t>(),
                       (const int64_t *)nullptr, n, out_samples, 3, 3);
    MF_LAUNCH_CHECK();
    MF_HIP(hipStreamSynchronize(st));
    return MFSGD_OK;
}

extern "C" int mfsgd_hip_csr_from_coo(const int32_t *samples, int64_t n, int32_t stride, int64_t n_users,
                                      int64_t *row_ptrs, int32_t *col_inds, void *stream)
{
    MF_ARG(n >= 0 && stride >= 2 && n_users >= 0 && row_ptrs, "csr_from_coo: bad argument");
    hipStream_t st = as_stream(stream);
    if (n == 0) {
        MF_HIP(hipMemsetAsync(row_ptrs, 0, sizeof(int64_t) * (size_t)(n_users + 1), st));
        MF_HIP(hipStreamSynchronize(st));
        return MFSGD_OK;
    }
    MF_ARG(samples && col_inds, "csr_from_coo: null pointer");
    DevBuf k0, k1, v0, v1, err;
    MF_HIP(k0.alloc((size_t)n * 4)); MF_HIP(k1.alloc((size_t)n * 4));
    MF_HIP(v0.alloc((size_t)n * 8)); MF_HIP(v1.alloc((size_t)n * 8));
    MF_HIP(err.alloc(4));
    MF_HIP(hipMemsetAsync(err.p, 0, 4, st));
    hipLaunchKernelGGL(k_user_keys, dim3(nblk(n)), dim3(256), 0, st, samples, n, stride, n_users, k0.as<uint32_t>(),
                       v0.as<int64_t>(), err.as<int32_t>());
    MF_LAUNCH_CHECK();
    int32_t h = 0;
    MF_HIP(hipMemcpyAsync(&h, err.p, 4, hipMemcpyDeviceToHost, st));
    MF_HIP(hipStreamSynchronize(st));
    MF_ARG(h == 0, "csr_from_coo: a user id is outside [0,n_users)");
    int rc = sort_pairs(k0.as<uint32_t>(), k1.as<uint32_t>(), v0.as<int64_t>(), v1.as<int64_t>(), n,
                        bits_for((uint64_t)n_users), st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_segment_ptr, dim3(nblk(n + 1)), dim3(256), 0, st, k1.as<uint32_t>(), n, n_users, row_ptrs);
    MF_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_gather_cols, dim3(nblk(n)), dim3(256), 0, st, samples, stride, v1.as<int64_t>(), n, col_inds);
    MF_LAUNCH_CHECK();
    MF_HIP(hipStreamSynchronize(st));
    return MFSGD_OK;
}

// Recommender.renumber_column (Recommender.py:18-25): every value -> its rank among the sorted distinct values.
extern "C" int mfsgd_hip_renumber_column(const int64_t *values, int64_t n, int32_t *out_ids, int64_t *h_n_unique,
                                         void *stream)
{
    MF_ARG(n >= 0 && h_n_unique && (n == 0 || (values && out_ids)), "renumber_column: bad argument");
    MF_ARG(n < ((int64_t)1 << 31), "renumber_column: n=%lld does not fit int32", (long long)n);
    *h_n_unique = 0;
    if (n == 0) return MFSGD_OK;
    hipStream_t st = as_stream(stream);
    DevBuf sorted, uniq, cnt, tmp;
    MF_HIP(sorted.alloc((size_t)n * 8)); MF_HIP(uniq.alloc((size_t)n * 8)); MF_HIP(cnt.alloc(8));
    size_t tb = 0;
    MF_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, tb, values, sorted.as<int64_t>(), (int)n, 0, 64, st));
    MF_HIP(tmp.alloc(tb));
    MF_HIP(hipcub::DeviceRadixSort::SortKeys(tmp.p, tb, values, sorted.as<int64_t>(), (int)n, 0, 64, st));
    size_t tb2 = 0;
    DevBuf tmp2;
    MF_HIP(hipcub::DeviceSelect::Unique(nullptr, tb2, sorted.as<int64_t>(), uniq.as<int64_t>(), cnt.as<int64_t>(), (int)n, st));
    MF_HIP(tmp2.alloc(tb2));
    MF_HIP(hipcub::DeviceSelect::Unique(tmp2.p, tb2, sorted.as<int64_t>(), uniq.as<int64_t>(), cnt.as<int64_t>(), (int)n, st));
    hipLaunchKernelGGL(k_rank_of, dim3(nblk(n)), dim3(256), 0, st, values, n, uniq.as<int64_t>(), cnt.as<int64_t>(), out_ids);
    MF_LAUNCH_CHECK();
    MF_HIP(hipMemcpyAsync(h_n_unique, cnt.p, 8, hipMemcpyDeviceToHost, st));
    MF_HIP(hipStreamSynchronize(st));
    return MFSGD_OK;
}
