// coord_map.hip -- N2: pose-sequence distance maps, fp64.  Replaces the Python triple loops of
// CoordMap.coord_dist_map (reference PointCloud/coord_map.py:230-307, both `diff` branches) and the
// pose -> xyz + quaternion step of load_matrix (:204-219), including the four roma functions the loops
// call (rotmat_to_rotvec, utils.rotvec_geodesic_distance, rotmat_geodesic_distance; restated from the
// published algorithm, see oracle/coord_map.py).
//
//   diff = 1 (coord_map.py:250-281), per step i < T-1:
//     trans_diff_k = t_{i+1,k} - t_{i,k};  rot_diff_k = rotvec(R_{i,k}^T R_{i+1,k})
//     d_xyz[j][k] = |trans_diff_j - trans_diff_k| / (2 bbox);  d_rpy[j][k] = geodesic(rot_diff_j, rot_diff_k) / pi
//     map[j][k][i] = |d_xyz[j][:] - d_xyz[k][:]|_2 + |d_rpy[j][:] - d_rpy[k][:]|_2      (distance between ROWS)
//   diff = 0 (:283-301), per step i < T:
//     map[j][k][i] = |t_j - t_k| / (2 bbox) + acos(clamp((tr(R_j^T R_k) - 1) / 2)) / pi
//   sum_map[j][k] = sum_i |map[j][k][i]|                                                   (:304-305)
//
// One workgroup per step; the two K x K pair matrices live in LDS (K <= 64) or in the workspace.
// O(T K^3) flops on a few KB: latency-bound by construction, a handful of microseconds per call.
#include "creg_common.h"
#include "creg_dev.h"

namespace creg {

constexpr int CM_NT = 1024;         // launch bound; small K launches 256 threads
constexpr int CM_LDS_K = 64;           // pair matrices in LDS up to this K (2 * 64 * 64 * 8 B = 64 KB)

// roma.rotmat_to_unitquat (SciPy's decision-matrix form), xyzw
__device__ __forceinline__ void rotmat_to_unitquat_xyzw(const double m[9], double q[4]) {
    const double tr = (m[0] + m[4]) + m[8];
    const double dec[4] = {m[0], m[4], m[8], tr};
    int c = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i) if (dec[i] > dec[c]) c = i;            // argmax, first maximum
    // (i, j, k) = (c, c+1, c+2) mod 3:  q[i] = 1 - tr + 2 m_ii, q[j] = m_ji + m_ij, q[k] = m_ki + m_ik, q[3] = m_kj - m_jk
    if (c == 3)      { q[0] = m[7] - m[5]; q[1] = m[2] - m[6]; q[2] = m[3] - m[1]; q[3] = 1 + tr; }
    else if (c == 0) { q[0] = 1 - tr + 2 * m[0]; q[1] = m[3] + m[1]; q[2] = m[6] + m[2]; q[3] = m[7] - m[5]; }
    else if (c == 1) { q[1] = 1 - tr + 2 * m[4]; q[2] = m[7] + m[5]; q[0] = m[1] + m[3]; q[3] = m[2] - m[6]; }
    else             { q[2] = 1 - tr + 2 * m[8]; q[0] = m[2] + m[6]; q[1] = m[5] + m[7]; q[3] = m[3] - m[1]; }
    const double n = sqrt(((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3]);
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] /= n;
}

// roma.unitquat_to_rotvec(shortest_arc=True) followed by roma.rotvec_to_unitquat: the quaternion the
// geodesic distance of two rotation VECTORS is evaluated on (coord_map.py:262,267)
__device__ __forceinline__ void rotvec_roundtrip(double q[4]) {
    if (q[3] < 0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
    const double half = atan2(sqrt((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]), q[3]);
    const double angle = 2 * half;
    const double a2 = angle * angle;
    const double scale = fabs(angle) <= 1e-3 ? 2 + a2 / 12 + 7 * (a2 * a2) / 2880 : angle / sin(half);
    const double v[3] = {scale * q[0], scale * q[1], scale * q[2]};
    const double nv = sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
    const double n2 = nv * nv;
    const double s2 = nv <= 1e-3 ? 0.5 - n2 / 48 + (n2 * n2) / 3840 : sin(nv / 2) / nv;
    q[0] = s2 * v[0]; q[1] = s2 * v[1]; q[2] = s2 * v[2]; q[3] = cos(nv / 2);
}

__global__ __launch_bounds__(CM_NT) void k_coord_dist_map(const double* __restrict__ M, int T, int K, double lam_bbox,
                                                          double lam_rot, int diff, double* __restrict__ d_map,
                                                          double* __restrict__ ws) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int i = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
    const int Tn = diff ? T - 1 : T;
    double* sv = (double*)smem;                           // [K][8]: translation(-difference) xyz, quaternion xyzw / unused
    double* sR = sv + 8 * (size_t)K;                      // [K][9] rotation (diff = 0)
    double* A = K <= CM_LDS_K ? sR + 9 * (size_t)K : ws + (size_t)i * 2 * K * K;     // [2][K][K] pair matrices
    double* B = A + (size_t)K * K;
    const double* M0 = M + (size_t)i * K * 16;
    for (int k = tid; k < K; k += nthr) {
        const double* a = M0 + 16 * (size_t)k;
        if (diff) {
            const double* b = a + (size_t)K * 16;         // step i + 1
            double rel[9];                                // R_i^T R_{i+1}
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) rel[3 * r + c] = (a[r] * b[c] + a[4 + r] * b[4 + c]) + a[8 + r] * b[8 + c];
            double q[4];
            rotmat_to_unitquat_xyzw(rel, q);
            rotvec_roundtrip(q);
#pragma unroll
            for (int d = 0; d < 3; ++d) sv[8 * k + d] = b[4 * d + 3] - a[4 * d + 3];
#pragma unroll
            for (int d = 0; d < 4; ++d) sv[8 * k + 3 + d] = q[d];
        } else {
#pragma unroll
            for (int d = 0; d < 3; ++d) sv[8 * k + d] = a[4 * d + 3];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) sR[9 * k + 3 * r + c] = a[4 * r + c];
        }
    }
    __syncthreads();
    for (int p = tid; p < K * K; p += nthr) {
        const int j = p / K, k = p % K;
        const double dx = sv[8 * j] - sv[8 * k], dy = sv[8 * j + 1] - sv[8 * k + 1], dz = sv[8 * j + 2] - sv[8 * k + 2];
        const double dxyz = lam_bbox * sqrt((dx * dx + dy * dy) + dz * dz);
        double drpy;
        if (diff) {
            double sm = 0, sp = 0;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const double a = sv[8 * j + 3 + d], b = sv[8 * k + 3 + d];
                sm += (b - a) * (b - a); sp += (b + a) * (b + a);
            }
            drpy = lam_rot * (4 * asin(0.5 * fmin(sqrt(sm), sqrt(sp))));
            A[p] = dxyz; B[p] = drpy;
        } else {
            double tr = 0;                                // trace(R_j^T R_k)
#pragma unroll
            for (int c = 0; c < 3; ++c) tr += (sR[9 * j + c] * sR[9 * k + c] + sR[9 * j + 3 + c] * sR[9 * k + 3 + c]) + sR[9 * j + 6 + c] * sR[9 * k + 6 + c];
            const double cs = fmin(fmax(0.5 * (tr - 1.0), -1.0), 1.0);
            drpy = lam_rot * acos(cs);
            d_map[(size_t)p * Tn + i] = dxyz + drpy;
        }
    }
    if (!diff) return;
    __syncthreads();                                      // (global A/B: same workgroup wrote them; the barrier orders them)
    for (int p = tid; p < K * K; p += nthr) {
        const int j = p / K, k = p % K;
        double s1 = 0, s2 = 0;
#pragma unroll 8
        for (int m = 0; m < K; ++m) {      // loads of 8 steps in flight; the two fma chains keep their order
            // both pair matrices are exactly symmetric: row k is read as column k, which consecutive lanes
            // (consecutive k) fetch from consecutive addresses
            const double a = A[(size_t)j * K + m] - A[(size_t)m * K + k], b = B[(size_t)j * K + m] - B[(size_t)m * K + k];
            s1 = fma(a, a, s1); s2 = fma(b, b, s2);
        }
        d_map[(size_t)p * Tn + i] = sqrt(s1) + sqrt(s2);
    }
}

__global__ void k_sum_map(const double* __restrict__ d_map, int KK, int Tn, double* __restrict__ sum_map) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= KK) return;
    double s = 0;
    for (int i = 0; i < Tn; ++i) s += fabs(d_map[(size_t)p * Tn + i]);
    sum_map[p] = s;
}

// load_matrix coord_map.py:204-219: pose -> [x, y, z, qw, qx, qy, qz] (pytorch3d matrix_to_quaternion, fp64)
__global__ void k_pose_coords(const double* __restrict__ M, int64_t n, double* __restrict__ coords) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const double* a = M + 16 * r;
    const double R[9] = {a[0], a[1], a[2], a[4], a[5], a[6], a[8], a[9], a[10]};
    double q[4];
    matrix_to_quat(R, q);
    double* o = coords + 7 * r;
    o[0] = a[3]; o[1] = a[7]; o[2] = a[11]; o[3] = q[0]; o[4] = q[1]; o[5] = q[2]; o[6] = q[3];
}

}  // namespace creg
using namespace creg;

extern "C" size_t creg_coord_dist_map_workspace_bytes(int32_t T, int32_t K) {
    if (T < 1 || K < 1) return 0;
    return K <= CM_LDS_K ? 256 : sizeof(double) * 2 * (size_t)K * K * T;
}

extern "C" int creg_coord_dist_map_f64(const double* M, int32_t T, int32_t K, double bounding_box, int32_t diff,
                                       double* d_map, double* sum_map, void* workspace, size_t workspace_bytes,
                                       creg_stream_t stream) {
    CREG_REQUIRE(M && d_map && sum_map && workspace, "creg_coord_dist_map_f64: null pointer");
    CREG_REQUIRE(K >= 1 && K <= 1024 && T >= (diff ? 2 : 1) && T <= (1 << 20), "creg_coord_dist_map_f64: bad size (T=%d, K=%d)", T, K);
    CREG_REQUIRE(bounding_box > 0, "creg_coord_dist_map_f64: bounding_box must be positive");
    CREG_REQUIRE(workspace_bytes >= creg_coord_dist_map_workspace_bytes(T, K), "creg_coord_dist_map_f64: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int Tn = diff ? T - 1 : T;
    const size_t smem = sizeof(double) * ((size_t)17 * K + (K <= CM_LDS_K ? 2 * (size_t)K * K : 0));
    // per device, not per process: set on every call (a cached flag would leave a second GPU at the 64 KB default)
    CREG_HIP(hipFuncSetAttribute((const void*)k_coord_dist_map, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
    hipLaunchKernelGGL(k_coord_dist_map, dim3(Tn), dim3(K <= 32 ? 256 : CM_NT), smem, s, M, T, K, 1.0 / (bounding_box * 2.0),
                       1.0 / 3.14159265358979323846, diff ? 1 : 0, d_map, (double*)workspace);
    CREG_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_sum_map, dim3(cdiv(K * K, 256)), dim3(256), 0, s, d_map, K * K, Tn, sum_map);
    CREG_LAUNCH_CHECK();
    return CREG_OK;
}

extern "C" int creg_pose_coords_f64(const double* M, int64_t n, double* coords, creg_stream_t stream) {
    CREG_REQUIRE(M && coords, "creg_pose_coords_f64: null pointer");
    CREG_REQUIRE(n >= 1, "creg_pose_coords_f64: n must be positive");
    hipLaunchKernelGGL(k_pose_coords, dim3((unsigned)cdiv(n, (int64_t)256)), dim3(256), 0, (hipStream_t)stream, M, n, coords);
    CREG_LAUNCH_CHECK();
    return CREG_OK;
}
