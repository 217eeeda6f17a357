// nn_l1.hip -- K1: L1 nearest neighbour both ways, its backward, and the Chamfer reduction.
// Replaces pytorch3d knn_points / chamfer_distance(norm=1) as reached from mlp_reg.py:96.
#include <type_traits>
#include "creg_common.h"
#include "nn_l1.h"

namespace creg {

// grad_x[i][d] = gx * s(x_i, y[ix_i]) + gy * cnt[i][d],  cnt accumulated with integer atomics
// (exact, order independent) from the y side:  cnt[iy_j][d] -= s(y_j, x[iy_j]).
__global__ __launch_bounds__(256) void k_nn_bwd_scatter(const float* __restrict__ x,
                                                        const float* __restrict__ y, int ny,
                                                        const int64_t* __restrict__ iy,
                                                        int* __restrict__ cnt) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= ny) return;
    const int64_t i = iy[j];
    if (i < 0) return;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int s = (y[(size_t)j * 3 + d] > x[(size_t)i * 3 + d]) ? 1 : -1;
        atomicAdd(&cnt[(size_t)i * 3 + d], -s);
    }
}

__global__ __launch_bounds__(256) void k_nn_bwd_finish(const float* __restrict__ x, int nx,
                                                       const float* __restrict__ y,
                                                       const int64_t* __restrict__ ix,
                                                       const int* __restrict__ cnt, float gx, float gy,
                                                       float* __restrict__ grad) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nx) return;
    const int64_t j = ix ? ix[i] : -1;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        float g = 0.f;
        if (j >= 0) g = (x[(size_t)i * 3 + d] > y[(size_t)j * 3 + d]) ? gx : -gx;
        grad[(size_t)i * 3 + d] = g + gy * (float)cnt[(size_t)i * 3 + d];
    }
}

// single block: fixed-order sums (thread-strided partials, then a fixed tree)
__global__ __launch_bounds__(1024) void k_chamfer_reduce(const float* __restrict__ dx, int nx,
                                                         const float* __restrict__ dy, int ny,
                                                         float* __restrict__ loss) {
    __shared__ float sc[16];
    float a = 0.f, b = 0.f;
    for (int i = threadIdx.x; i < nx; i += 1024) a += dx[i];
    for (int i = threadIdx.x; i < ny; i += 1024) b += dy[i];
    a = block_sum<float, 1024>(a, sc);
    b = block_sum<float, 1024>(b, sc);
    if (threadIdx.x == 0) loss[0] = a / (float)nx + b / (float)ny;
}

}  // namespace creg

using namespace creg;

extern "C" int creg_nn_l1_bidir_f32(const float* x, int64_t nx, const float* y, int64_t ny, float* dx,
                                    int64_t* ix, float* dy, int64_t* iy, creg_stream_t stream) {
    CREG_REQUIRE(x && y, "creg_nn_l1_bidir_f32: null input");
    CREG_REQUIRE(nx >= 1 && ny >= 1 && nx < (1ll << 31) && ny < (1ll << 31),
                 "creg_nn_l1_bidir_f32: sizes must be in [1, 2^31)");
    CREG_REQUIRE((dx != nullptr) == (ix != nullptr) && (dy != nullptr) == (iy != nullptr),
                 "creg_nn_l1_bidir_f32: pass both or neither output of a direction");
    launch_nn_l1<int64_t>(x, (int)nx, 3, y, (int)ny, 3, dx, ix, dy, iy, dx != nullptr, dy != nullptr, NnEpilogueNone{},
                          (hipStream_t)stream);
    CREG_LAUNCH_CHECK();
    return CREG_OK;
}

extern "C" size_t creg_nn_l1_bwd_scratch_bytes(int64_t nx) { return (size_t)nx * 3 * sizeof(int); }

extern "C" int creg_nn_l1_bwd_f32(const float* x, int64_t nx, const float* y, int64_t ny,
                                  const int64_t* ix, const int64_t* iy, float gx_scale, float gy_scale,
                                  float* grad_x, void* scratch, creg_stream_t stream) {
    CREG_REQUIRE(x && y && grad_x && scratch, "creg_nn_l1_bwd_f32: null pointer");
    CREG_REQUIRE(nx >= 1 && ny >= 1, "creg_nn_l1_bwd_f32: empty cloud");
    hipStream_t s = (hipStream_t)stream;
    int* cnt = (int*)scratch;
    CREG_HIP(hipMemsetAsync(cnt, 0, (size_t)nx * 3 * sizeof(int), s));
    if (iy) hipLaunchKernelGGL(k_nn_bwd_scatter, dim3(cdiv(ny, 256)), dim3(256), 0, s, x, y, (int)ny, iy, cnt);
    hipLaunchKernelGGL(k_nn_bwd_finish, dim3(cdiv(nx, 256)), dim3(256), 0, s, x, (int)nx, y, ix, cnt,
                       gx_scale, gy_scale, grad_x);
    CREG_LAUNCH_CHECK();
    return CREG_OK;
}

extern "C" int creg_chamfer_l1_reduce_f32(const float* dx, int64_t nx, const float* dy, int64_t ny,
                                          float* loss, creg_stream_t stream) {
    CREG_REQUIRE(dx && dy && loss && nx >= 1 && ny >= 1, "creg_chamfer_l1_reduce_f32: bad argument");
    hipLaunchKernelGGL(k_chamfer_reduce, dim3(1), dim3(1024), 0, (hipStream_t)stream, dx, (int)nx, dy,
                       (int)ny, loss);
    CREG_LAUNCH_CHECK();
    return CREG_OK;
}
