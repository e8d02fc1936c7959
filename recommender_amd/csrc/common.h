// common.h - shared helpers for libmfsgd (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/mfsgd.h"

void mfsgd_set_error(const char *fmt, ...);

#define MF_HIP(call)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            mfsgd_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
            return MFSGD_ERR_HIP;                                                            \
        }                                                                                    \
    } while (0)

#define MF_ARG(cond, ...)                   \
    do {                                    \
        if (!(cond)) {                      \
            mfsgd_set_error(__VA_ARGS__);   \
            return MFSGD_ERR_ARG;           \
        }                                   \
    } while (0)

#define MF_LAUNCH_CHECK() MF_HIP(hipGetLastError())

// ---- wave64 sum.  f32: DPP butterfly inside each 16-lane row, two row broadcasts (gfx9-family DPP), and one
// readlane: the total comes back wave-uniform (an SGPR), the tree is fixed, so the result is the same on every run.
__device__ __forceinline__ float wave_sum(float v)
{
    int x;
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false);  // quad_perm [1,0,3,2]
    v += __int_as_float(x);
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false);  // quad_perm [2,3,0,1]
    v += __int_as_float(x);
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, false); // row_half_mirror
    v += __int_as_float(x);
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, false); // row_mirror
    v += __int_as_float(x);
    // every lane of a 16-lane row now holds its row's sum
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xA, 0xF, false); // row_bcast15 -> rows 1,3
    v += __int_as_float(x);
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xC, 0xF, false); // row_bcast31 -> rows 2,3
    v += __int_as_float(x);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));        // (r0+r1)+(r2+r3)
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
    return v;
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

static inline hipStream_t as_stream(void *s) { return (hipStream_t)s; }
