/*
 * creg_oracle.c -- CPU restatement (TEST INFRASTRUCTURE, never shipped in the product path)
 * of the third-party native arithmetic AutoURDF's registration path reaches:
 *
 *   - pytorch3d==0.7.7 knn_points (K=1, norm=1) forward/backward, reached from
 *     chamfer_distance(pred, y, norm=1) at reference PointCloud/mlp_reg.py:96.
 *     pytorch3d is NOT vendored under /root/reference; this restates its published CPU
 *     algorithm (pytorch3d/csrc/knn/knn_cpu.cpp): brute force over p2, strict `<` so the
 *     FIRST minimum wins, L1 accumulated d=0,1,2 from 0.0f, backward sign rule
 *     (p1 > p2 ? +1 : -1).  Parity status: unpinned by the reference (it has no tests).
 *   - scikit-learn k_means Lloyd iteration (reference PointCloud/mlp_reg.py:204,
 *     cluster_icp.py:67), following sklearn/cluster/_kmeans.py:699-750 and
 *     _k_means_lloyd.pyx:168-218, _k_means_common.pyx:167-311.  Pinned against live
 *     sklearn 1.7.2 in tests/test_oracle_kmeans.py.
 *
 * Canonical floating-point orders (shared bit-for-bit with the HIP kernels):
 *   L1:     d = (|x0-y0| + |x1-y1|) + |x2-y2|                       (fp32, no fma possible)
 *   kmeans: d = fma(x2,b2, fma(x1,b1, fma(x0,b0, csq))),  b = -2c,   (fp64)
 *           csq = fma(c2,c2, fma(c1,c1, c0*c0))
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

/* team size of the parallel loops below (bench.py's cpu_baseline times the port at 16, 1 and all host threads: the environment
 * variable is only read when the OpenMP runtime starts) */
void oracle_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }

/* ---------------------------------------------------------------- NN, L1, K=1 */
void oracle_nn_l1_f32(const float* x, int64_t nx, const float* y, int64_t ny,
                      float* dist, int64_t* idx) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < nx; ++i) {
        const float a0 = x[3 * i], a1 = x[3 * i + 1], a2 = x[3 * i + 2];
        float best = INFINITY;
        int64_t bj = -1;
        for (int64_t j = 0; j < ny; ++j) {
            float d = fabsf(a0 - y[3 * j]);
            d = d + fabsf(a1 - y[3 * j + 1]);
            d = d + fabsf(a2 - y[3 * j + 2]);
            if (d < best || bj < 0) { best = d; bj = j; }
        }
        dist[i] = (ny > 0) ? best : 0.0f;
        idx[i] = bj;
    }
}

/* ---------------------------------------------------------------- NN, squared L2, K=1, float64 (open3d registration_icp's search)
 * d2 = ((dx*dx + dy*dy) + dz*dz): numpy's ((a - b) ** 2).sum(-1) of oracle/icp.py:_correspond, first minimum in target order
 * (np.argmin).  For pairs too large for the dense (n, m, 3) array of the numpy form (Sim/evaluation.py:358-362 registers whole
 * robot clouds); tests/test_oracle_golden.py holds the two forms to identical indices and distances. */
void oracle_nn_l2_f64(const double* x, int64_t nx, const double* y, int64_t ny, double* d2, int64_t* idx) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < nx; ++i) {
        const double a0 = x[3 * i], a1 = x[3 * i + 1], a2 = x[3 * i + 2];
        double best = INFINITY;
        int64_t bj = -1;
        for (int64_t j = 0; j < ny; ++j) {
            const double e0 = a0 - y[3 * j], e1 = a1 - y[3 * j + 1], e2 = a2 - y[3 * j + 2];
            const double d = (e0 * e0 + e1 * e1) + e2 * e2;
            if (d < best || bj < 0) { best = d; bj = j; }
        }
        d2[i] = best;
        idx[i] = bj;
    }
}

/* grad wrt x of  sum_i gx*|x_i - y[ix_i]|_1 + sum_j gy*|y_j - x[iy_j]|_1  (pytorch3d sign rule).
 * part_x / part_y are returned separately (autograd adds them afterwards). */
void oracle_nn_l1_bwd_f32(const float* x, int64_t nx, const float* y, int64_t ny,
                          const int64_t* ix, const int64_t* iy, float gx, float gy,
                          float* part_x, float* part_y) {
    memset(part_x, 0, sizeof(float) * 3 * (size_t)nx);
    memset(part_y, 0, sizeof(float) * 3 * (size_t)nx);
    for (int64_t i = 0; i < nx; ++i) {           /* knn(p1=x, p2=y): grad_p1 */
        int64_t j = ix[i];
        if (j < 0) continue;
        for (int d = 0; d < 3; ++d) {
            float s = (x[3 * i + d] > y[3 * j + d]) ? 1.0f : -1.0f;
            part_x[3 * i + d] += gx * s;
        }
    }
    for (int64_t j = 0; j < ny; ++j) {           /* knn(p1=y, p2=x): grad_p2 -= diff */
        int64_t i = iy[j];
        if (i < 0) continue;
        for (int d = 0; d < 3; ++d) {
            float s = (y[3 * j + d] > x[3 * i + d]) ? 1.0f : -1.0f;
            part_y[3 * i + d] -= gy * s;
        }
    }
}

/* ---------------------------------------------------------------- Lloyd k-means, fp64, D=3 */
static inline double kdist(const double* p, const double* b, double csq) {
    double d = fma(p[0], b[0], csq);
    d = fma(p[1], b[1], d);
    d = fma(p[2], b[2], d);
    return d;
}

static void assign_labels(const double* X, int64_t n, const double* C, int k, int32_t* labels) {
    double* B = (double*)malloc(sizeof(double) * 4 * (size_t)k);
    for (int j = 0; j < k; ++j) {
        const double* c = C + 3 * j;
        B[4 * j] = -2.0 * c[0]; B[4 * j + 1] = -2.0 * c[1]; B[4 * j + 2] = -2.0 * c[2];
        B[4 * j + 3] = fma(c[2], c[2], fma(c[1], c[1], c[0] * c[0]));
    }
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const double* p = X + 3 * i;
        double best = kdist(p, B, B[3]);
        int32_t lab = 0;
        for (int j = 1; j < k; ++j) {
            double d = kdist(p, B + 4 * j, B[4 * j + 3]);
            if (d < best) { best = d; lab = j; }
        }
        labels[i] = lab;
    }
    free(B);
}

void oracle_kmeans_assign_f64(const double* X, int64_t n, const double* C, int k, int32_t* labels) {
    assign_labels(X, n, C, k, labels);
}

/* One full k_means(X, init=C0, n_init=1) run (sample_weight == 1).
 * X is NOT modified; centring by the mean is done on a copy like KMeans.fit does.
 * Returns n_iter. centers_out (k,3) are un-centred, labels_out (n), inertia_out scalar. */
int oracle_kmeans_lloyd_f64(const double* X_in, int64_t n, const double* C0, int k, int max_iter,
                            double tol_rel, double* centers_out, int32_t* labels_out,
                            double* inertia_out) {
    double* X = (double*)malloc(sizeof(double) * 3 * (size_t)n);
    double mean[3] = {0, 0, 0}, var[3] = {0, 0, 0};
    /* np.mean(axis=0) uses pairwise summation; the difference to a plain loop is O(1e-16)
     * relative and only shifts the frame, it cancels in every distance comparison. */
    for (int d = 0; d < 3; ++d) {
        long double s = 0;
        for (int64_t i = 0; i < n; ++i) s += X_in[3 * i + d];
        mean[d] = (double)(s / (long double)n);
    }
    for (int64_t i = 0; i < n; ++i)
        for (int d = 0; d < 3; ++d) X[3 * i + d] = X_in[3 * i + d] - mean[d];
    /* tol = mean(var(X, axis=0)) * tol_rel  computed on the ORIGINAL X (sklearn _kmeans.py:1436) */
    for (int d = 0; d < 3; ++d) {
        long double s = 0;
        for (int64_t i = 0; i < n; ++i) { long double t = X_in[3 * i + d] - mean[d]; s += t * t; }
        var[d] = (double)(s / (long double)n);
    }
    const double tol = ((var[0] + var[1] + var[2]) / 3.0) * tol_rel;

    double* C = (double*)malloc(sizeof(double) * 3 * (size_t)k);
    double* Cn = (double*)malloc(sizeof(double) * 3 * (size_t)k);
    double* w = (double*)malloc(sizeof(double) * (size_t)k);
    int32_t* labels = labels_out;
    int32_t* labels_old = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
    double* dist_far = NULL;
    for (int j = 0; j < k; ++j)
        for (int d = 0; d < 3; ++d) C[3 * j + d] = C0[3 * j + d] - mean[d];
    for (int64_t i = 0; i < n; ++i) { labels[i] = -1; labels_old[i] = -1; }

    int strict = 0, it = 0;
    for (it = 0; it < max_iter; ++it) {
        assign_labels(X, n, C, k, labels);
        memset(Cn, 0, sizeof(double) * 3 * (size_t)k);
        memset(w, 0, sizeof(double) * (size_t)k);
        for (int64_t i = 0; i < n; ++i) {
            int32_t l = labels[i];
            w[l] += 1.0;
            Cn[3 * l] += X[3 * i]; Cn[3 * l + 1] += X[3 * i + 1]; Cn[3 * l + 2] += X[3 * i + 2];
        }
        /* empty-cluster relocation (_k_means_common.pyx:167-211); farthest points taken in
         * DESCENDING distance order, ties to the lower index. */
        int n_empty = 0;
        for (int j = 0; j < k; ++j) n_empty += (w[j] == 0.0);
        if (n_empty > 0) {
            if (!dist_far) dist_far = (double*)malloc(sizeof(double) * (size_t)n);
            double dmax = 0;
            for (int64_t i = 0; i < n; ++i) {
                const double* c = C + 3 * labels[i];
                double a = X[3 * i] - c[0], b = X[3 * i + 1] - c[1], e = X[3 * i + 2] - c[2];
                dist_far[i] = (a * a + b * b) + e * e;
                if (dist_far[i] > dmax) dmax = dist_far[i];
            }
            if (dmax > 0) {
                for (int j = 0; j < k; ++j) {
                    if (w[j] != 0.0) continue;
                    int64_t far = 0; double best = -1;
                    for (int64_t i = 0; i < n; ++i)
                        if (dist_far[i] > best) { best = dist_far[i]; far = i; }
                    dist_far[far] = -2;               /* consumed */
                    int32_t old = labels[far];
                    for (int d = 0; d < 3; ++d) {
                        Cn[3 * old + d] -= X[3 * far + d];
                        Cn[3 * j + d] = X[3 * far + d];
                    }
                    w[j] = 1.0; w[old] -= 1.0;
                }
            }
        }
        int argmax = 0;
        for (int j = 1; j < k; ++j) if (w[j] > w[argmax]) argmax = j;
        for (int j = 0; j < k; ++j) {
            if (w[j] > 0) {
                double alpha = 1.0 / w[j];
                for (int d = 0; d < 3; ++d) Cn[3 * j + d] *= alpha;
            } else {
                for (int d = 0; d < 3; ++d) Cn[3 * j + d] = Cn[3 * argmax + d];
            }
        }
        double shift_tot = 0;
        for (int j = 0; j < k; ++j) {
            double s = 0;
            for (int d = 0; d < 3; ++d) { double t = Cn[3 * j + d] - C[3 * j + d]; s += t * t; }
            double sh = sqrt(s);
            shift_tot += sh * sh;
        }
        { double* t = C; C = Cn; Cn = t; }
        int same = 1;
        for (int64_t i = 0; i < n; ++i) if (labels[i] != labels_old[i]) { same = 0; break; }
        if (same) { strict = 1; ++it; break; }
        if (shift_tot <= tol) { ++it; break; }
        memcpy(labels_old, labels, sizeof(int32_t) * (size_t)n);
    }
    if (!strict) assign_labels(X, n, C, k, labels);
    double inertia = 0;
    for (int64_t i = 0; i < n; ++i) {
        const double* c = C + 3 * labels[i];
        double a = X[3 * i] - c[0], b = X[3 * i + 1] - c[1], e = X[3 * i + 2] - c[2];
        inertia += (a * a + b * b) + e * e;
    }
    *inertia_out = inertia;
    for (int j = 0; j < k; ++j)
        for (int d = 0; d < 3; ++d) centers_out[3 * j + d] = C[3 * j + d] + mean[d];
    free(X); free(C); free(Cn); free(w); free(labels_old); free(dist_far);
    return it;
}

/* ---------------------------------------------------------------- farthest point sampling (fp64)
 * open3d 0.18 PointCloud::FarthestPointDownSample (reference cluster_icp.py:43): start at index 0,
 * repeatedly select the point with the largest squared distance to the selected set; first max wins.
 * open3d is not vendored: restated from its published algorithm; parity unpinned. */
void oracle_fps_f64(const double* X, int64_t n, int64_t m, int64_t* sel) {
    double* dmin = (double*)malloc(sizeof(double) * (size_t)n);
    for (int64_t i = 0; i < n; ++i) dmin[i] = INFINITY;
    int64_t cur = 0;
    for (int64_t s = 0; s < m; ++s) {
        sel[s] = cur;
        const double* c = X + 3 * cur;
        int64_t far = 0; double best = -1;
        for (int64_t i = 0; i < n; ++i) {
            double a = X[3 * i] - c[0], b = X[3 * i + 1] - c[1], e = X[3 * i + 2] - c[2];
            double d = (a * a + b * b) + e * e;
            if (d < dmin[i]) dmin[i] = d;
            if (dmin[i] > best) { best = dmin[i]; far = i; }
        }
        cur = far;
    }
    free(dmin);
}
