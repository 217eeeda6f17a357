// transform.hip -- K3: per-cluster rigid transform and its backward (calculate_pc, mlp_reg.py:155-170).
#include "creg_common.h"
#include "creg_dev.h"

namespace creg {

__device__ __forceinline__ int find_segment(const int* __restrict__ off, int k, int n) {
    int lo = 0, hi = k;                         // largest s with off[s] <= n
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (off[mid] <= n) lo = mid; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(256) void k_cluster_transform(const float* __restrict__ pts,
                                                           const int* __restrict__ off, int k,
                                                           const float* __restrict__ M,
                                                           float* __restrict__ out) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= off[k]) return;
    const float* T = M + 16 * find_segment(off, k, n);
    const float p0 = pts[3 * (size_t)n], p1 = pts[3 * (size_t)n + 1], p2 = pts[3 * (size_t)n + 2];
#pragma unroll
    for (int a = 0; a < 3; ++a)
        out[3 * (size_t)n + a] = fmaf(p2, T[4 * a + 2], fmaf(p1, T[4 * a + 1], p0 * T[4 * a])) + T[4 * a + 3];
}

// one block per cluster; fixed-order reduction of [g (x) p | g]
__global__ __launch_bounds__(256) void k_cluster_transform_bwd(const float* __restrict__ pts,
                                                               const int* __restrict__ off,
                                                               const float* __restrict__ g,
                                                               float* __restrict__ gM) {
    __shared__ float sc[4];
    const int k = blockIdx.x, b = off[k], e = off[k + 1];
    float acc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = 0.f;
    for (int n = b + threadIdx.x; n < e; n += 256) {
        const float p[3] = {pts[3 * (size_t)n], pts[3 * (size_t)n + 1], pts[3 * (size_t)n + 2]};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float ga = g[3 * (size_t)n + a];
            acc[4 * a + 3] += ga;
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[4 * a + c] = fmaf(ga, p[c], acc[4 * a + c]);
        }
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        const float r = block_sum<float, 256>(acc[i], sc);
        if (threadIdx.x == 0) gM[16 * k + i] = r;
    }
    if (threadIdx.x < 4) gM[16 * k + 12 + threadIdx.x] = 0.f;
}

}  // namespace creg
using namespace creg;

extern "C" int creg_cluster_transform_f32(const float* pts, int64_t n, const int32_t* seg_offsets, int32_t k,
                                          const float* M, float* out, creg_stream_t stream) {
    CREG_REQUIRE(pts && seg_offsets && M && out && k >= 1 && n >= 0, "creg_cluster_transform_f32: bad argument");
    if (n == 0) return CREG_OK;
    hipLaunchKernelGGL(k_cluster_transform, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, pts,
                       seg_offsets, k, M, out);
    CREG_LAUNCH_CHECK();
    return CREG_OK;
}

extern "C" int creg_cluster_transform_bwd_f32(const float* pts, const int32_t* seg_offsets, int32_t k,
                                              const float* grad_out, float* grad_M, creg_stream_t stream) {
    CREG_REQUIRE(pts && seg_offsets && grad_out && grad_M && k >= 1, "creg_cluster_transform_bwd_f32: bad argument");
    hipLaunchKernelGGL(k_cluster_transform_bwd, dim3(k), dim3(256), 0, (hipStream_t)stream, pts, seg_offsets,
                       grad_out, grad_M);
    CREG_LAUNCH_CHECK();
    return CREG_OK;
}
