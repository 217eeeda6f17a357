// sample.hip -- N4: area-weighted surface sampling of an articulated triangle-mesh robot, fp64.
// The data-generation side of the path (reference Sim/sim_data.py:246-370 renders depth images of the
// PyBullet model from 20 cameras and fuses them; this build samples the same URDF + mesh surfaces
// directly): one thread per output point picks a triangle by inverting the cumulative-area table,
// a uniform barycentric point on it, and moves it by its link's pose of the current joint state.
//   f   = first triangle with cum_area[f] > u0 * cum_area[F-1]
//   s   = sqrt(u1);  p = (1 - s) A + (s (1 - u2)) B + (s u2) C         (uniform on the triangle)
//   out = R_link p + t_link
// Every operation is written out in a fixed order without contraction, so the host restatement
// (oracle/sim_data.py, numpy) reproduces the points bit for bit.
#include "creg_common.h"
#include "creg_dev.h"

namespace creg {

__global__ __launch_bounds__(256) void k_sample_mesh(const double* __restrict__ tri, const double* __restrict__ cum_area,
                                                     const int* __restrict__ tri_link, const double* __restrict__ link_T,
                                                     const double* __restrict__ u, int64_t n, int F,
                                                     double* __restrict__ out, int* __restrict__ link_out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double u0 = u[3 * i], u1 = u[3 * i + 1], u2 = u[3 * i + 2];
    const double target = u0 * cum_area[F - 1];
    int lo = 0, hi = F - 1;                              // upper bound: first f with cum_area[f] > target
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cum_area[mid] > target) hi = mid; else lo = mid + 1;
    }
    const double* t = tri + 9 * (size_t)lo;
    const double s = sqrt(u1);
    const double b0 = 1.0 - s, b1 = s * (1.0 - u2), b2 = s * u2;
    double p[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) p[d] = (b0 * t[d] + b1 * t[3 + d]) + b2 * t[6 + d];
    const int l = tri_link[lo];
    const double* T = link_T + 16 * (size_t)l;
#pragma unroll
    for (int d = 0; d < 3; ++d) out[3 * i + d] = ((T[4 * d] * p[0] + T[4 * d + 1] * p[1]) + T[4 * d + 2] * p[2]) + T[4 * d + 3];
    if (link_out) link_out[i] = l;
}

}  // namespace creg
using namespace creg;

extern "C" int creg_sample_mesh_f64(const double* tri, const double* cum_area, const int32_t* tri_link, int32_t n_tri,
                                    const double* link_T, int32_t n_links, const double* u, int64_t n, double* out,
                                    int32_t* link_out, creg_stream_t stream) {
    CREG_REQUIRE(tri && cum_area && tri_link && link_T && u && out, "creg_sample_mesh_f64: null pointer");
    CREG_REQUIRE(n_tri >= 1 && n_links >= 1 && n >= 1, "creg_sample_mesh_f64: bad size");
    hipLaunchKernelGGL(k_sample_mesh, dim3((unsigned)cdiv(n, (int64_t)256)), dim3(256), 0, (hipStream_t)stream, tri, cum_area,
                       tri_link, link_T, u, n, (int)n_tri, out, link_out);
    CREG_LAUNCH_CHECK();
    return CREG_OK;
}
