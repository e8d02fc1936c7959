// row_rmw.hip - what the memory system serves for the epoch kernel's access pattern, without the SGD around it:
// every wave walks its own list of random 512-byte rows of a 246 MB matrix (the netflix-shape P), reads a row, adds
// to it, writes it back, with DEPTH row loads in flight.  Same geometry as k_epoch_free's default (256 workgroups
// of 16 waves).  Build: hipcc -O3 --offload-arch=gfx950 -o profiles/_prof/row_rmw profiles/row_rmw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int DEPTH>
__global__ __launch_bounds__(1024, 4) void k_rmw(float *P, const int *rows, int per_wave, int k)
{
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int *mine = rows + (size_t)wave * per_wave;
    float2 buf[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) buf[d] = *(const float2 *)(P + (size_t)mine[d] * k + 2 * lane);
    for (int i = 0; i + DEPTH < per_wave; i += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            float2 v = buf[d];
            const int r = mine[i + d];
            buf[d] = *(const float2 *)(P + (size_t)mine[i + d + DEPTH] * k + 2 * lane);
            v.x = v.x * 0.999f + 1e-3f; v.y = v.y * 0.999f + 1e-3f;
            *(float2 *)(P + (size_t)r * k + 2 * lane) = v;
        }
    }
}

int main()
{
    const int U = 480189, k = 128, n_wg = 256, waves = n_wg * 16, per_wave = 12800;   // 52 M row visits, like an epoch
    float *P; int *rows;
    hipMalloc(&P, (size_t)U * k * 4); hipMemset(P, 0, (size_t)U * k * 4);
    std::vector<int> h((size_t)waves * per_wave);
    unsigned s = 12345;
    for (auto &x : h) { s = s * 1664525u + 1013904223u; x = (int)((s >> 8) % U); }
    for (int i = 0; i < 64; ++i) h.push_back(0);      // the last wave prefetches a few entries past its list
    hipMalloc(&rows, h.size() * 4); hipMemcpy(rows, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto run = [&](auto kern, int depth) {
        hipLaunchKernelGGL(kern, dim3(n_wg), dim3(1024), 0, 0, P, rows, per_wave, k);
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL(kern, dim3(n_wg), dim3(1024), 0, 0, P, rows, per_wave, k);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
        const double visits = (double)waves * (per_wave - depth);
        printf("depth %d: %.2f ms, %.2f G row read-modify-writes/s, %.2f TB/s (read + write)\n", depth, ms,
               visits / ms / 1e6, visits * 1024 / ms / 1e9);
    };
    run(k_rmw<1>, 1); run(k_rmw<2>, 2); run(k_rmw<3>, 3); run(k_rmw<4>, 4); run(k_rmw<6>, 6); run(k_rmw<8>, 8);
    const hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("HIP error: %s\n", hipGetErrorString(e)); return 1; }
    return 0;
}
