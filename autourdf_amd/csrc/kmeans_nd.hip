// kmeans_nd.hip -- Lloyd k-means over DIM-dimensional features, sklearn.cluster.k_means(X, init=array, n_init=1) semantics,
// for the `--normal` branch of the reference: resample_cluster / Segments.k_means_cluster cluster [xyz | 0.5 * normal]
// (PointCloud/mlp_reg.py:190-203, cluster_icp.py:49-62).  One workgroup runs the whole k_means() -- mean / tol, centring,
// up to max_iter Lloyd iterations with the strict-convergence / tol test, empty-cluster relocation, final E-step, inertia --
// with no host involvement, like k_km_small of kmeans.hip (whose 3-D arithmetic it generalises step by step):
//   * distances as |c|^2 + sum_d x_d * (-2 c_d), one fma chain in feature order, first minimum wins;
//   * the M-step sums as EXACT int64 fixed point (scale chosen from the data's range), kept incrementally: a point whose
//     label changed moves from the old cluster's sums to the new one's -- order independent, bit reproducible;
//   * centres = sums * (1 / count) (sklearn's multiplication with the reciprocal), empty clusters take the points farthest
//     from their centres (_relocate_empty_clusters_dense), centre shift = sum_j |c_new - c_old|^2 against tol = mean
//     feature variance * tol_rel.
// The frame is read from global memory on every use (n * DIM * 8 B: L2-resident) and centred on the fly.
#include "creg_common.h"
#include "creg_dev.h"

namespace creg {

constexpr int KND_NT = 1024;
constexpr int KND_MAXK = 128;

__device__ __forceinline__ double knd_fix_scale(double range, int n) {      // 2^s with n * range * 2^s < 2^62
    if (!(range > 0.0)) return 1.0;
    int e;
    frexp(range, &e);
    const int nb = 32 - __clz(n);
    return ldexp(1.0, 62 - nb - e);
}

// GL (round 5): the two label buffers live in the caller's workspace instead of LDS -- frames above 16384 points (the reference's
// sklearn call has no cap; one workgroup still runs the whole k_means(): a 32768-point frame with 45 clusters takes ~0.1 ms per
// Lloyd iteration, which the `--normal` branch, dominated by its O(N^2) neighbour search, does not notice).  Stores and loads of one
// workgroup to global memory are ordered by its barriers.
template <int DIM, bool GL = false>
__global__ __launch_bounds__(KND_NT) void k_km_nd(const double* __restrict__ X, int n, const double* __restrict__ init, int k,
                                                 int max_iter, double tol_rel, double* __restrict__ centers, int* __restrict__ labels_out,
                                                 double* __restrict__ inertia, int* __restrict__ n_iter, double* __restrict__ far_d) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BW = DIM + 1;
    double* Bm = (double*)smem;                                   // [k][DIM + 1]: -2 c | |c|^2
    double* C2 = Bm + (size_t)k * BW;                             // [2][k][DIM] centre ping-pong
    double* Cw = C2 + 2 * (size_t)k * DIM;                        // [k][DIM + 1] sums | count
    unsigned long long* accI = (unsigned long long*)(Cw + (size_t)k * BW);   // [k][DIM + 1] exact sums | count
    double* s_sh = (double*)(accI + (size_t)k * BW);              // [k]
    unsigned short* lab0 = GL ? (unsigned short*)(far_d + n) : (unsigned short*)(s_sh + k);
    unsigned short* lab[2] = {lab0, lab0 + n};
    __shared__ double sc[16], s_mean[DIM], s_tol, s_fscale, s_finv, s_dmax, s_fv[16];
    __shared__ int s_changed, s_done, s_strict, s_it, s_nempty, s_argmax, s_fi[16];
    __shared__ unsigned char s_emp[128];                          // (k <= 128) the clusters that were empty when the relocation started
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // ---- mean, variance -> tol, range -> fixed-point scale
    for (int d = 0; d < DIM; ++d) {
        double s = 0;
        for (int i = tid; i < n; i += KND_NT) s += X[(size_t)i * DIM + d];
        s = block_sum<double, KND_NT>(s, sc);
        if (tid == 0) s_mean[d] = s / (double)n;
        __syncthreads();
    }
    double var = 0, amax = 0;
    for (int d = 0; d < DIM; ++d) {
        const double m = s_mean[d];
        double s = 0;
        for (int i = tid; i < n; i += KND_NT) { const double t = X[(size_t)i * DIM + d] - m; s = fma(t, t, s); amax = fmax(amax, fabs(t)); }
        s = block_sum<double, KND_NT>(s, sc);
        if (tid == 0) var += s / (double)n;
        __syncthreads();
    }
    for (int off = 32; off >= 1; off >>= 1) amax = fmax(amax, __shfl_xor(amax, off, 64));
    if (lane == 0) s_fv[wv] = amax;
    for (int j = tid; j < k * BW; j += KND_NT) accI[j] = 0ull;
    __syncthreads();
    if (tid == 0) {
        double r = 0;
        for (int q = 0; q < 16; ++q) r = fmax(r, s_fv[q]);
        s_fscale = knd_fix_scale(r, n); s_finv = 1.0 / s_fscale;
        s_tol = (var / (double)DIM) * tol_rel; s_done = 0; s_strict = 0; s_it = 0; s_changed = 0; s_nempty = 0;
    }
    for (int i = tid; i < n; i += KND_NT) lab[1][i] = 0xFFFF;      // iteration 0 compares against "no label"
    auto make_row = [&](const double* c, double* b) {
        double n2 = 0;
#pragma unroll
        for (int d = 0; d < DIM; ++d) { b[d] = -2.0 * c[d]; n2 = fma(c[d], c[d], n2); }
        b[DIM] = n2;
    };
    if (tid < k) {
        double c[DIM];
#pragma unroll
        for (int d = 0; d < DIM; ++d) { c[d] = init[(size_t)tid * DIM + d] - s_mean[d]; C2[(size_t)tid * DIM + d] = c[d]; }
        make_row(c, Bm + (size_t)tid * BW);
    }
    __syncthreads();
    double mean[DIM];
#pragma unroll
    for (int d = 0; d < DIM; ++d) mean[d] = s_mean[d];
    const double fscale = s_fscale, finv = s_finv;
    auto load_x = [&](int i, double* x) {
#pragma unroll
        for (int d = 0; d < DIM; ++d) x[d] = X[(size_t)i * DIM + d] - mean[d];
    };
    auto nearest = [&](const double* x) -> int {
        double best; int lb = 0;
        {
            double dd = Bm[DIM];
#pragma unroll
            for (int d = 0; d < DIM; ++d) dd = fma(x[d], Bm[d], dd);
            best = dd;
        }
        for (int j = 1; j < k; ++j) {
            const double* b = Bm + (size_t)j * BW;
            double dd = b[DIM];
#pragma unroll
            for (int d = 0; d < DIM; ++d) dd = fma(x[d], b[d], dd);
            if (dd < best) { best = dd; lb = j; }
        }
        return lb;
    };
    int cur = 0;
    for (int it = 0; it < max_iter; ++it) {
        unsigned short* lcur = lab[it & 1];
        const unsigned short* lprev = lab[(it + 1) & 1];
        // ---- E-step + incremental exact M-step sums
        int diff = 0;
        for (int i = tid; i < n; i += KND_NT) {
            double x[DIM];
            load_x(i, x);
            const int lb = nearest(x), pv = lprev[i];
            lcur[i] = (unsigned short)lb;
            if (pv != lb) {
                ++diff;
#pragma unroll
                for (int d = 0; d < DIM; ++d) {
                    const unsigned long long v = (unsigned long long)__double2ll_rn(x[d] * fscale);
                    atomicAdd(&accI[(size_t)lb * BW + d], v);
                    if (pv != 0xFFFF) atomicAdd(&accI[(size_t)pv * BW + d], 0ull - v);
                }
                atomicAdd(&accI[(size_t)lb * BW + DIM], 1ull);
                if (pv != 0xFFFF) atomicAdd(&accI[(size_t)pv * BW + DIM], ~0ull);
            }
        }
        if (diff) atomicAdd(&s_changed, diff);
        __syncthreads();
        // ---- M-step: thread j owns cluster j
        const double* Cold = C2 + (size_t)cur * k * DIM;
        double* Cnew = C2 + (size_t)(cur ^ 1) * k * DIM;
        if (tid < k) {
#pragma unroll
            for (int d = 0; d < DIM; ++d) Cw[(size_t)tid * BW + d] = (double)(long long)accI[(size_t)tid * BW + d] * finv;
            const double cnt = (double)(long long)accI[(size_t)tid * BW + DIM];
            Cw[(size_t)tid * BW + DIM] = cnt;
            if (cnt == 0.0) atomicAdd(&s_nempty, 1);
        }
        __syncthreads();
        if (s_nempty > 0) {                                      // block-uniform; rare: _relocate_empty_clusters_dense
            double dmax = 0;
            for (int i = tid; i < n; i += KND_NT) {
                double x[DIM];
                load_x(i, x);
                const double* c = Cold + (size_t)lcur[i] * DIM;
                double dd = 0;
#pragma unroll
                for (int d = 0; d < DIM; ++d) { const double t = x[d] - c[d]; dd += t * t; }
                far_d[i] = dd;
                dmax = fmax(dmax, dd);
            }
            for (int off = 32; off >= 1; off >>= 1) dmax = fmax(dmax, __shfl_xor(dmax, off, 64));
            __syncthreads();
            if (lane == 0) sc[wv] = dmax;
            __syncthreads();
            if (tid == 0) { double m = 0; for (int q = 0; q < 16; ++q) m = fmax(m, sc[q]); s_dmax = m; }
            __syncthreads();
            if (s_dmax > 0) {
                // the list of empty clusters is FIXED before anything moves (sklearn's _relocate_empty_clusters_dense takes
                // np.where(weight_in_clusters == 0) first): a cluster that a relocation empties -- its only point was the farthest one --
                // is not relocated in this pass (round 3 re-tested the count while walking and relocated it too)
                if (tid < k) s_emp[tid] = Cw[(size_t)tid * BW + DIM] == 0.0;
                __syncthreads();
                for (int j = 0; j < k; ++j) {
                    if (!s_emp[j]) continue;                            // block-uniform (LDS value)
                    double bv = -1; int bi = 0x7fffffff;
                    for (int i = tid; i < n; i += KND_NT) if (far_d[i] > bv) { bv = far_d[i]; bi = i; }
                    for (int off = 32; off >= 1; off >>= 1) {
                        const double ov = __shfl_xor(bv, off, 64); const int oi = __shfl_xor(bi, off, 64);
                        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
                    }
                    __syncthreads();
                    if (lane == 0) { s_fv[wv] = bv; s_fi[wv] = bi; }
                    __syncthreads();
                    if (tid == 0) {
                        for (int q = 0; q < 16; ++q) if (s_fv[q] > bv || (s_fv[q] == bv && s_fi[q] < bi)) { bv = s_fv[q]; bi = s_fi[q]; }
                        far_d[bi] = -2;
                        const int old = lcur[bi];
                        double x[DIM];
                        load_x(bi, x);
#pragma unroll
                        for (int d = 0; d < DIM; ++d) { Cw[(size_t)old * BW + d] -= x[d]; Cw[(size_t)j * BW + d] = x[d]; }
                        Cw[(size_t)j * BW + DIM] = 1.0; Cw[(size_t)old * BW + DIM] -= 1.0;
                    }
                    __threadfence_block();
                    __syncthreads();
                }
            }
            if (tid == 0) { int am = 0; for (int j = 1; j < k; ++j) if (Cw[(size_t)j * BW + DIM] > Cw[(size_t)am * BW + DIM]) am = j; s_argmax = am; }
            __syncthreads();
        }
        if (tid < k) {
            const int j = tid;
            const double cnt = Cw[(size_t)j * BW + DIM];
            const int src = cnt > 0 ? j : s_argmax;
            const double alpha = 1.0 / Cw[(size_t)src * BW + DIM];
            double c[DIM], s = 0;
#pragma unroll
            for (int d = 0; d < DIM; ++d) {
                c[d] = Cw[(size_t)src * BW + d] * alpha;
                Cnew[(size_t)j * DIM + d] = c[d];
                const double t = c[d] - Cold[(size_t)j * DIM + d];
                s += t * t;
            }
            const double sh = sqrt(s);
            s_sh[j] = sh * sh;
            make_row(c, Bm + (size_t)j * BW);
        }
        __syncthreads();
        if (tid == 0) {
            double tot = 0;
            for (int j = 0; j < k; ++j) tot += s_sh[j];
            s_it = it + 1;
            if (s_changed == 0) { s_strict = 1; s_done = 1; }
            else if (tot <= s_tol) s_done = 1;
            s_changed = 0; s_nempty = 0;
        }
        cur ^= 1;
        __syncthreads();
        if (s_done) break;
    }
    // ---- final E-step when not strictly converged, labels out, inertia, un-centred centres
    unsigned short* last = lab[(s_it - 1) & 1];
    if (!s_strict) {
        for (int i = tid; i < n; i += KND_NT) { double x[DIM]; load_x(i, x); last[i] = (unsigned short)nearest(x); }
        __syncthreads();
    }
    const double* C = C2 + (size_t)cur * k * DIM;
    double s = 0;
    for (int i = tid; i < n; i += KND_NT) {
        labels_out[i] = (int)last[i];
        double x[DIM];
        load_x(i, x);
        const double* c = C + (size_t)last[i] * DIM;
        double dd = 0;
#pragma unroll
        for (int d = 0; d < DIM; ++d) { const double t = x[d] - c[d]; dd += t * t; }
        s += dd;
    }
    s = block_sum<double, KND_NT>(s, sc);
    if (tid == 0) { inertia[0] = s; n_iter[0] = s_it; }
    for (int j = tid; j < k * DIM; j += KND_NT) centers[j] = C[j] + s_mean[j % DIM];
}

}  // namespace creg
using namespace creg;

constexpr int64_t KND_LDS_N = 16384;       // labels of frames up to this size stay in LDS
extern "C" size_t creg_kmeans_nd_workspace_bytes(int64_t n) {
    const size_t nn = (size_t)(n > 0 ? n : 1);
    return align_up(sizeof(double) * nn + (n > KND_LDS_N ? 2 * sizeof(unsigned short) * nn : 0), 256);
}

extern "C" int creg_kmeans_lloyd_nd_f64(const double* X, int64_t n, int32_t dim, const double* init, int32_t k, int32_t max_iter,
                                        double tol_rel, double* centers, int32_t* labels, double* inertia, int32_t* n_iter,
                                        void* workspace, size_t workspace_bytes, creg_stream_t stream) {
    CREG_REQUIRE(X && init && centers && labels && inertia && n_iter && workspace, "creg_kmeans_lloyd_nd_f64: null pointer");
    CREG_REQUIRE(dim == 6 || dim == 3, "creg_kmeans_lloyd_nd_f64: dim must be 6 ([xyz | 0.5 normal], the --normal branch) or 3");
    CREG_REQUIRE(n >= 1 && n < (1ll << 24) && k >= 1 && k <= KND_MAXK && max_iter >= 1,
                 "creg_kmeans_lloyd_nd_f64: needs n < 2^24 and k <= %d (one workgroup, centres in LDS)", KND_MAXK);
    CREG_REQUIRE(workspace_bytes >= creg_kmeans_nd_workspace_bytes(n), "creg_kmeans_lloyd_nd_f64: workspace too small");
    const int bw = dim + 1;
    const bool gl = n > KND_LDS_N;
    const int smem = (int)(sizeof(double) * ((size_t)k * bw + 2 * (size_t)k * dim + (size_t)k * bw + (size_t)k * bw + k) + (gl ? 0 : 2 * sizeof(unsigned short) * (size_t)n));
    auto go = [&](auto kern) -> int {
        CREG_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048));
        hipLaunchKernelGGL(kern, dim3(1), dim3(KND_NT), smem, (hipStream_t)stream, X, (int)n, init, k, max_iter, tol_rel, centers, labels,
                           inertia, n_iter, (double*)workspace);
        return CREG_OK;
    };
    const int rc = dim == 6 ? (gl ? go(k_km_nd<6, true>) : go(k_km_nd<6>)) : (gl ? go(k_km_nd<3, true>) : go(k_km_nd<3>));
    if (rc) return rc;
    CREG_LAUNCH_CHECK();
    return CREG_OK;
}
