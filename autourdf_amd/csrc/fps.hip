// fps.hip -- N1: farthest-point down-sampling in fp64, replacing open3d
// PointCloud::farthest_point_down_sample as used by Segments._load_pc (reference cluster_icp.py:43)
// and the data generator (Sim/sim_data.py:347-350).  Start at index 0; repeatedly take the point with
// the largest squared distance to the selected set (strict '>' : the FIRST maximum wins).
// Inherently sequential over the m selections: one workgroup, dmin kept in registers when the cloud
// is small enough (<= 8 points per thread), two barriers per selection.
#include "creg_common.h"
#include "creg_dev.h"

namespace creg {

constexpr int FPS_NT = 1024;
constexpr int FPS_PT = 8;          // points per thread held in registers (n <= 8192); else global dmin

template <bool REG>
__global__ __launch_bounds__(FPS_NT) void k_fps(const double* __restrict__ X, int n, int m,
                                                int64_t* __restrict__ sel, double* __restrict__ dmin_g) {
    __shared__ double s_val[FPS_NT / 64];
    __shared__ int s_idx[FPS_NT / 64];
    __shared__ double s_c[3];
    __shared__ int s_cur;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    double px[FPS_PT], py[FPS_PT], pz[FPS_PT], dm[FPS_PT];
    if (REG) {
#pragma unroll
        for (int q = 0; q < FPS_PT; ++q) {
            const int i = min(q * FPS_NT + tid, n - 1);
            px[q] = X[3 * (size_t)i]; py[q] = X[3 * (size_t)i + 1]; pz[q] = X[3 * (size_t)i + 2]; dm[q] = INFINITY;
        }
    } else {
        for (int i = tid; i < n; i += FPS_NT) dmin_g[i] = INFINITY;
    }
    if (tid == 0) s_cur = 0;
    __syncthreads();
    for (int s = 0; s < m; ++s) {
        const int cur = s_cur;
        if (tid == 0) sel[s] = cur;
        if (tid < 3) s_c[tid] = X[3 * (size_t)cur + tid];
        __syncthreads();
        const double cx = s_c[0], cy = s_c[1], cz = s_c[2];
        double bv = -1.0; int bi = 0x7fffffff;
        if (REG) {
#pragma unroll
            for (int q = 0; q < FPS_PT; ++q) {
                const int i = q * FPS_NT + tid;
                const double a = px[q] - cx, b = py[q] - cy, c = pz[q] - cz;
                const double d = (a * a + b * b) + c * c;
                dm[q] = d < dm[q] ? d : dm[q];
                if (i < n && dm[q] > bv) { bv = dm[q]; bi = i; }       // ascending i per thread: first max kept
            }
        } else {
            for (int i = tid; i < n; i += FPS_NT) {
                const double a = X[3 * (size_t)i] - cx, b = X[3 * (size_t)i + 1] - cy, c = X[3 * (size_t)i + 2] - cz;
                const double d = (a * a + b * b) + c * c;
                const double old = dmin_g[i];
                const double nv = d < old ? d : old;
                dmin_g[i] = nv;
                if (nv > bv) { bv = nv; bi = i; }
            }
        }
        // (max value, min index) over the block
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const double ov = __shfl_xor(bv, off, 64); const int oi = __shfl_xor(bi, off, 64);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) { s_val[wv] = bv; s_idx[wv] = bi; }
        __syncthreads();
        if (tid == 0) {
            double v = s_val[0]; int i = s_idx[0];
            for (int w = 1; w < FPS_NT / 64; ++w) if (s_val[w] > v || (s_val[w] == v && s_idx[w] < i)) { v = s_val[w]; i = s_idx[w]; }
            s_cur = i;
        }
        __syncthreads();
    }
}

}  // namespace creg
using namespace creg;

extern "C" size_t creg_fps_scratch_bytes(int64_t n) { return n > (int64_t)FPS_NT * FPS_PT ? sizeof(double) * (size_t)n : 8; }

extern "C" int creg_fps_f64(const double* X, int64_t n, int64_t m, int64_t* sel, void* scratch, creg_stream_t stream) {
    CREG_REQUIRE(X && sel && scratch && n >= 1 && n < (1ll << 31) && m >= 1 && m <= n, "creg_fps_f64: need 1 <= m <= n < 2^31");
    if (n <= (int64_t)FPS_NT * FPS_PT)
        hipLaunchKernelGGL((k_fps<true>), dim3(1), dim3(FPS_NT), 0, (hipStream_t)stream, X, (int)n, (int)m, sel, (double*)scratch);
    else
        hipLaunchKernelGGL((k_fps<false>), dim3(1), dim3(FPS_NT), 0, (hipStream_t)stream, X, (int)n, (int)m, sel, (double*)scratch);
    CREG_LAUNCH_CHECK();
    return CREG_OK;
}
