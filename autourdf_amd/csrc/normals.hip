// normals.hip -- the `--normal` branch of the reference (PointCloud/mlp_reg.py:190-192, cluster_icp.py:49-51):
//   pc.estimate_normals(search_param=KDTreeSearchParamHybrid(radius=0.1, max_nn=30))
//   pc.orient_normals_consistent_tangent_plane(30)
// open3d 0.18 is not vendored by the reference and absent from this image: what is restated here is its published
// behaviour.  estimate_normals: per point the (up to) max_nn nearest points with squared distance < radius^2 (the point
// itself included; KDTreeFlann::SearchHybrid = knnSearch + cut at the radius), their covariance from the nine raw moments
// (sum x, ..., sum zz) / count in neighbour order, the unit eigenvector of its smallest eigenvalue (Eberly's non-iterative
// symmetric 3x3 eigensolver: trigonometric eigenvalues of the scaled matrix, the best-conditioned eigenvector from row cross
// products, the others by deflation), (0,0,1) with fewer than three neighbours.  The sign of an eigenvector is the solver's
// business; orient_normals_consistent_tangent_plane fixes it afterwards (host: autourdf_amd/normals.py) and needs the k
// nearest neighbours of every point: the same search without the radius, exported as index lists.
//
// The search is exhaustive (n = 4096-16384 points per frame, once per frame): a block of KNN_Q queries, one thread each,
// sweeps the cloud through LDS tiles; a thread keeps its (distance, index)-sorted list of the max_nn best in LDS.
#include <cfloat>
#include "creg_common.h"
#include "creg_dev.h"

namespace creg {

constexpr int KNN_Q = 128;            // queries (threads) per block
constexpr int KNN_TILE = 512;         // candidates staged per LDS tile
constexpr int KNN_MAX = 32;           // most neighbours a list holds

__device__ __forceinline__ void cross3(const double* a, const double* b, double* o) {
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
// unit eigenvector of A for eigenvalue ev from the largest of the three row cross products of A - ev I
__device__ void eigvec_by_rows(const double A[6], double ev, double* v) {   // A = (a00, a01, a02, a11, a12, a22)
    const double r0[3] = {A[0] - ev, A[1], A[2]}, r1[3] = {A[1], A[3] - ev, A[4]}, r2[3] = {A[2], A[4], A[5] - ev};
    double c01[3], c02[3], c12[3];
    cross3(r0, r1, c01); cross3(r0, r2, c02); cross3(r1, r2, c12);
    const double d0 = c01[0] * c01[0] + c01[1] * c01[1] + c01[2] * c01[2];
    const double d1 = c02[0] * c02[0] + c02[1] * c02[1] + c02[2] * c02[2];
    const double d2 = c12[0] * c12[0] + c12[1] * c12[1] + c12[2] * c12[2];
    const double* c = c01; double dm = d0;
    if (d1 > dm) { dm = d1; c = c02; }
    if (d2 > dm) { dm = d2; c = c12; }
    const double inv = dm > 0 ? 1.0 / sqrt(dm) : 0.0;
    v[0] = c[0] * inv; v[1] = c[1] * inv; v[2] = c[2] * inv;
}
// second eigenvector (eigenvalue ev1) in the plane orthogonal to the unit eigenvector e0
__device__ void eigvec_deflated(const double A[6], const double* e0, double ev1, double* v) {
    double U[3], V[3];
    if (fabs(e0[0]) > fabs(e0[1])) { const double il = 1.0 / sqrt(e0[0] * e0[0] + e0[2] * e0[2]); U[0] = -e0[2] * il; U[1] = 0; U[2] = e0[0] * il; }
    else { const double il = 1.0 / sqrt(e0[1] * e0[1] + e0[2] * e0[2]); U[0] = 0; U[1] = e0[2] * il; U[2] = -e0[1] * il; }
    cross3(e0, U, V);
    const double AU[3] = {A[0] * U[0] + A[1] * U[1] + A[2] * U[2], A[1] * U[0] + A[3] * U[1] + A[4] * U[2], A[2] * U[0] + A[4] * U[1] + A[5] * U[2]};
    const double AV[3] = {A[0] * V[0] + A[1] * V[1] + A[2] * V[2], A[1] * V[0] + A[3] * V[1] + A[4] * V[2], A[2] * V[0] + A[4] * V[1] + A[5] * V[2]};
    double m00 = U[0] * AU[0] + U[1] * AU[1] + U[2] * AU[2] - ev1, m01 = U[0] * AV[0] + U[1] * AV[1] + U[2] * AV[2],
           m11 = V[0] * AV[0] + V[1] * AV[1] + V[2] * AV[2] - ev1;
    const double a00 = fabs(m00), a01 = fabs(m01), a11 = fabs(m11);
    double cu, cv;                                      // v = cu U + cv V in the null space of the 2x2 [[m00, m01], [m01, m11]]
    if (a00 >= a11) {
        if (fmax(a00, a01) > 0) {
            if (a00 >= a01) { m01 /= m00; m00 = 1.0 / sqrt(1.0 + m01 * m01); m01 *= m00; } else { m00 /= m01; m01 = 1.0 / sqrt(1.0 + m00 * m00); m00 *= m01; }
            cu = m01; cv = -m00;
        } else { cu = 1; cv = 0; }
    } else {
        if (fmax(a11, a01) > 0) {
            if (a11 >= a01) { m01 /= m11; m11 = 1.0 / sqrt(1.0 + m01 * m01); m01 *= m11; } else { m11 /= m01; m01 = 1.0 / sqrt(1.0 + m11 * m11); m11 *= m01; }
            cu = m11; cv = -m01;
        } else { cu = 1; cv = 0; }
    }
    for (int a = 0; a < 3; ++a) v[a] = cu * U[a] + cv * V[a];
}
// unit eigenvector of the SMALLEST eigenvalue of the symmetric 3x3 C = (c00, c01, c02, c11, c12, c22); zero vector for C = 0
__device__ void smallest_eigvec(const double Cin[6], double* nrm) {
    double mx = 0;
    for (int i = 0; i < 6; ++i) mx = fmax(mx, fabs(Cin[i]));
    if (!(mx > 0)) { nrm[0] = nrm[1] = nrm[2] = 0; return; }
    double A[6];
    for (int i = 0; i < 6; ++i) A[i] = Cin[i] / mx;
    const double off2 = A[1] * A[1] + A[2] * A[2] + A[4] * A[4];
    if (off2 > 0) {
        const double q = (A[0] + A[3] + A[5]) / 3.0;
        const double b00 = A[0] - q, b11 = A[3] - q, b22 = A[5] - q;
        const double p = sqrt((b00 * b00 + b11 * b11 + b22 * b22 + 2.0 * off2) / 6.0);
        const double c00 = b11 * b22 - A[4] * A[4], c01 = A[1] * b22 - A[4] * A[2], c02 = A[1] * A[4] - b11 * A[2];
        const double det = (b00 * c00 - A[1] * c01 + A[2] * c02) / (p * p * p);
        const double hd = fmin(fmax(det * 0.5, -1.0), 1.0);
        const double ang = acos(hd) / 3.0;
        const double beta2 = cos(ang) * 2.0, beta0 = cos(ang + 2.09439510239319549) * 2.0, beta1 = -(beta0 + beta2);
        const double e0 = q + p * beta0, e1 = q + p * beta1, e2 = q + p * beta2;        // e0 <= e1 <= e2
        if (hd >= 0) {                                  // e2 is the well-separated one: start there, deflate, finish by a cross product
            double v2[3], v1[3];
            eigvec_by_rows(A, e2, v2);
            eigvec_deflated(A, v2, e1, v1);
            cross3(v1, v2, nrm);
        } else eigvec_by_rows(A, e0, nrm);
    } else {                                            // diagonal matrix: the axis of the smallest entry (z on ties, as the branch order gives)
        nrm[0] = (A[0] < A[3] && A[0] < A[5]) ? 1.0 : 0.0;
        nrm[1] = (nrm[0] == 0.0 && A[3] < A[0] && A[3] < A[5]) ? 1.0 : 0.0;
        nrm[2] = (nrm[0] == 0.0 && nrm[1] == 0.0) ? 1.0 : 0.0;
    }
}

// grid = ceil(n / KNN_Q) blocks of KNN_Q threads.  r2 < 0: no radius (plain k nearest).  idx_out (n, max_nn) / cnt_out (n)
// optional; normals (n,3) optional.
__global__ __launch_bounds__(KNN_Q) void k_knn_normals(const double* __restrict__ X, int n, double r2, int max_nn,
                                                       int* __restrict__ idx_out, int* __restrict__ cnt_out, double* __restrict__ normals) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* tile = (double*)smem;                                 // [KNN_TILE][3]
    double* ld = tile + 3 * KNN_TILE;                             // [KNN_MAX][KNN_Q] distances, list position major (no bank conflicts)
    int* li = (int*)(ld + KNN_MAX * KNN_Q);                       // [KNN_MAX][KNN_Q] indices
    const int tid = threadIdx.x, q = blockIdx.x * KNN_Q + tid;
    const bool live = q < n;
    const double qx = live ? X[3 * (size_t)q] : 0, qy = live ? X[3 * (size_t)q + 1] : 0, qz = live ? X[3 * (size_t)q + 2] : 0;
    int cnt = 0;
    double worst = r2 >= 0 ? r2 : DBL_MAX;                        // a candidate must beat this (strictly, like the radius cut) to enter
    for (int t0 = 0; t0 < n; t0 += KNN_TILE) {
        const int nt = min(KNN_TILE, n - t0);
        __syncthreads();
        for (int i = tid; i < 3 * nt; i += KNN_Q) tile[i] = X[3 * (size_t)t0 + i];
        __syncthreads();
        if (!live) continue;
        for (int j = 0; j < nt; ++j) {
            const double dx = tile[3 * j] - qx, dy = tile[3 * j + 1] - qy, dz = tile[3 * j + 2] - qz;
            const double d = dx * dx + dy * dy + dz * dz;
            if (cnt == max_nn ? !(d < worst) : !(r2 < 0 || d < r2)) continue;
            // insert (d, t0 + j) into the sorted list (ascending distance, then index: candidates arrive in index order, so an
            // equal distance goes BEHIND the entries already there)
            int p = cnt < max_nn ? cnt : max_nn - 1;
            while (p > 0 && ld[(p - 1) * KNN_Q + tid] > d) { ld[p * KNN_Q + tid] = ld[(p - 1) * KNN_Q + tid]; li[p * KNN_Q + tid] = li[(p - 1) * KNN_Q + tid]; --p; }
            ld[p * KNN_Q + tid] = d; li[p * KNN_Q + tid] = t0 + j;
            if (cnt < max_nn) ++cnt;
            if (cnt == max_nn) worst = ld[(max_nn - 1) * KNN_Q + tid];
        }
    }
    if (!live) return;
    if (cnt_out) cnt_out[q] = cnt;
    if (idx_out) for (int p = 0; p < max_nn; ++p) idx_out[(size_t)q * max_nn + p] = p < cnt ? li[p * KNN_Q + tid] : -1;
    if (!normals) return;
    double nv[3] = {0.0, 0.0, 1.0};
    if (cnt >= 3) {
        double m[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int p = 0; p < cnt; ++p) {
            const int i = li[p * KNN_Q + tid];
            const double x = X[3 * (size_t)i], y = X[3 * (size_t)i + 1], z = X[3 * (size_t)i + 2];
            m[0] += x; m[1] += y; m[2] += z; m[3] += x * x; m[4] += x * y; m[5] += x * z; m[6] += y * y; m[7] += y * z; m[8] += z * z;
        }
        for (int a = 0; a < 9; ++a) m[a] /= (double)cnt;
        const double C[6] = {m[3] - m[0] * m[0], m[4] - m[0] * m[1], m[5] - m[0] * m[2], m[6] - m[1] * m[1], m[7] - m[1] * m[2], m[8] - m[2] * m[2]};
        smallest_eigvec(C, nv);
        const double l2 = nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2];
        if (!(l2 > 0)) { nv[0] = 0; nv[1] = 0; nv[2] = 1; }
        else { const double il = 1.0 / sqrt(l2); nv[0] *= il; nv[1] *= il; nv[2] *= il; }
    }
    normals[3 * (size_t)q] = nv[0]; normals[3 * (size_t)q + 1] = nv[1]; normals[3 * (size_t)q + 2] = nv[2];
}

}  // namespace creg
using namespace creg;

extern "C" int creg_knn_normals_f64(const double* X, int64_t n, double radius, int32_t max_nn, int32_t* idx_out, int32_t* cnt_out,
                                    double* normals, creg_stream_t stream) {
    CREG_REQUIRE(X && (idx_out || cnt_out || normals), "creg_knn_normals_f64: null pointer");
    CREG_REQUIRE(n >= 1 && n < (1ll << 31) && max_nn >= 1 && max_nn <= KNN_MAX, "creg_knn_normals_f64: needs 1 <= max_nn <= %d", KNN_MAX);
    const int smem = (int)(sizeof(double) * 3 * KNN_TILE + (sizeof(double) + sizeof(int)) * KNN_MAX * KNN_Q);
    hipLaunchKernelGGL(k_knn_normals, dim3(cdiv(n, KNN_Q)), dim3(KNN_Q), smem, (hipStream_t)stream, X, (int)n,
                       radius > 0 ? radius * radius : -1.0, (int)max_nn, idx_out, cnt_out, normals);
    CREG_LAUNCH_CHECK();
    return CREG_OK;
}
