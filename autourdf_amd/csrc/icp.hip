// icp.hip -- K4: masked point-to-point ICP per cluster in fp64, one workgroup per cluster, the
// whole ICP loop inside one launch.  Replaces masked_icp (reference cluster_icp.py:118-191) and
// the open3d registration_icp it calls (point-to-point, relative_fitness = relative_rmse = 1e-6):
//   1. float32 AABB of the predicted world cluster, scaled about its centre, strict inequalities
//      (cluster_icp.py:133-146) -> ordered compaction of the frame points inside it
//   2. loop: nearest target (squared L2, first minimum, within th) for every source point ->
//      fitness / inlier RMSE -> best rigid update (Horn's closed form: dominant eigenvector of the
//      4x4 profile matrix by cyclic Jacobi; equals Umeyama/Kabsch with the det correction) ->
//      compose on the left, move the source incrementally like open3d does
//   3. stop when |d fitness| < 1e-6 and |d rmse| < 1e-6, or after max_iteration
#include "creg_common.h"
#include "creg_dev.h"

namespace creg {

// dominant eigenvector of a symmetric 4x4 (cyclic Jacobi), returned as a unit quaternion
__device__ void sym4_max_eigvec(double A[4][4], double q[4]) {
    double V[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0, diag = 0;
        for (int i = 0; i < 4; ++i) { diag += A[i][i] * A[i][i]; for (int j = i + 1; j < 4; ++j) off += A[i][j] * A[i][j]; }
        if (off <= 1e-60 + 1e-34 * diag) break;
        for (int p = 0; p < 3; ++p)
            for (int r = p + 1; r < 4; ++r) {
                if (A[p][r] == 0.0) continue;
                const double theta = (A[r][r] - A[p][p]) / (2.0 * A[p][r]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 4; ++k) { const double akp = A[k][p], akr = A[k][r]; A[k][p] = c * akp - s * akr; A[k][r] = s * akp + c * akr; }
                for (int k = 0; k < 4; ++k) { const double apk = A[p][k], ark = A[r][k]; A[p][k] = c * apk - s * ark; A[r][k] = s * apk + c * ark; }
                for (int k = 0; k < 4; ++k) { const double vkp = V[k][p], vkr = V[k][r]; V[k][p] = c * vkp - s * vkr; V[k][r] = s * vkp + c * vkr; }
            }
    }
    int m = 0;
    for (int i = 1; i < 4; ++i) if (A[i][i] > A[m][m]) m = i;
    double n = 0;
    for (int i = 0; i < 4; ++i) n += V[i][m] * V[i][m];
    n = sqrt(n);
    const double sg = V[0][m] < 0 ? -1.0 : 1.0;
    for (int i = 0; i < 4; ++i) q[i] = sg * V[i][m] / n;
}

struct IcpLayout { size_t srcw, tidx, nn, total; };
static IcpLayout icp_layout(int64_t n, int64_t nf, int k) {
    IcpLayout L; size_t o = 0;
    auto take = [&](size_t b) { size_t r = o; o = align_up(o + b, 256); return r; };
    L.srcw = take(sizeof(double) * 3 * n);          // source points in the world frame (moved incrementally)
    L.tidx = take(sizeof(int) * (size_t)k * nf);    // per cluster: frame indices inside its box, ascending
    L.nn = take(sizeof(int) * n);                   // per source point: matched target (frame index) or -1
    L.total = o;
    return L;
}

template <int NT>
__device__ __forceinline__ double bsum(double v, double* sc) {
    const double r = block_sum<double, NT>(v, sc);
    __shared__ double bc;
    if (threadIdx.x == 0) bc = r;
    __syncthreads();
    const double out = bc;
    __syncthreads();
    return out;
}

__global__ __launch_bounds__(256) void k_masked_icp(
    const double* __restrict__ local, const float* __restrict__ world, const int* __restrict__ off,
    const double* __restrict__ frame, int nf, const double* __restrict__ Min, float half_scale, double th,
    int max_iter, int keep_t, double* __restrict__ Mout, double* __restrict__ world_out,
    int* __restrict__ n_iter_out, double* __restrict__ srcw, int* __restrict__ tidx_all, int* __restrict__ nn,
    int lds_cap) {
    __shared__ double sc[4];
    __shared__ float s_lo[3], s_hi[3];
    __shared__ int s_cnt, s_wofs[4];
    __shared__ double T[16], U[16];
    const int k = blockIdx.x, b = off[k], e = off[k + 1], ns = e - b;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int* tidx = tidx_all + (size_t)k * nf;

    // ---- 1. box in float32, exactly as numpy evaluates it on the float32 cluster ----
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = b + tid; i < e; i += 256)
        for (int d = 0; d < 3; ++d) { const float v = world[3 * (size_t)i + d]; lo[d] = fminf(lo[d], v); hi[d] = fmaxf(hi[d], v); }
    for (int d = 0; d < 3; ++d) {
        for (int o = 32; o >= 1; o >>= 1) { lo[d] = fminf(lo[d], __shfl_xor(lo[d], o, 64)); hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], o, 64)); }
        __shared__ float wl[4], wh[4];
        if (lane == 0) { wl[wv] = lo[d]; wh[wv] = hi[d]; }
        __syncthreads();
        if (tid == 0) {
            const float l = fminf(fminf(wl[0], wl[1]), fminf(wl[2], wl[3])), h = fmaxf(fmaxf(wh[0], wh[1]), fmaxf(wh[2], wh[3]));
            const float c = (l + h) / 2.0f, sz = h - l;
            s_lo[d] = c - half_scale * sz; s_hi[d] = c + half_scale * sz;
        }
        __syncthreads();
    }
    // ---- ordered compaction of the frame points strictly inside the box ----
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    for (int base = 0; base < nf; base += 256) {
        const int j = base + tid;
        bool in = false;
        if (j < nf && ns > 0) {
            const double x = frame[3 * (size_t)j], y = frame[3 * (size_t)j + 1], z = frame[3 * (size_t)j + 2];
            in = x > (double)s_lo[0] && x < (double)s_hi[0] && y > (double)s_lo[1] && y < (double)s_hi[1] &&
                 z > (double)s_lo[2] && z < (double)s_hi[2];
        }
        const unsigned long long m = __ballot(in);
        if (lane == 0) s_wofs[wv] = __popcll(m);
        __syncthreads();
        int before = s_cnt;
        for (int w = 0; w < wv; ++w) before += s_wofs[w];
        if (in) tidx[before + __popcll(m & ((1ull << lane) - 1ull))] = j;
        __syncthreads();
        if (tid == 0) s_cnt += s_wofs[0] + s_wofs[1] + s_wofs[2] + s_wofs[3];
        __syncthreads();
    }
    const int nt = s_cnt;
    // masked target coordinates into LDS (all threads sweep the same target at the same time, so the
    // nearest-neighbour loop below becomes broadcast LDS reads instead of dependent global gathers)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* sT = (double*)smem;                       // [lds_cap][3]
    int* sI = (int*)(sT + 3 * (size_t)lds_cap);       // [lds_cap]
    const bool in_lds = nt <= lds_cap;
    if (in_lds)
        for (int t = tid; t < nt; t += 256) {
            const int j = tidx[t];
            sT[3 * t] = frame[3 * (size_t)j]; sT[3 * t + 1] = frame[3 * (size_t)j + 1]; sT[3 * t + 2] = frame[3 * (size_t)j + 2];
            sI[t] = j;
        }
    __syncthreads();

    // ---- 2. ICP ----
    if (tid < 16) T[tid] = Min[16 * k + tid];
    __syncthreads();
    for (int i = tid; i < ns; i += 256) {
        const double* p = local + 3 * (size_t)(b + i);
        for (int a = 0; a < 3; ++a)
            srcw[3 * (size_t)(b + i) + a] = fma(T[4 * a + 2], p[2], fma(T[4 * a + 1], p[1], T[4 * a] * p[0])) + T[4 * a + 3];
    }
    __syncthreads();
    const double th2 = th * th;
    double fit = 0, rmse = 0;
    int it = 0;
    auto correspond = [&](double& fitness, double& rm) {
        double cnt = 0, err = 0;
        for (int i = tid; i < ns; i += 256) {
            const double* s = srcw + 3 * (size_t)(b + i);
            const double s0 = s[0], s1 = s[1], s2 = s[2];
            double best = INFINITY; int bj = -1;
            if (in_lds) {
                int bt = -1;
                for (int t = 0; t < nt; ++t) {
                    const double dx = s0 - sT[3 * t], dy = s1 - sT[3 * t + 1], dz = s2 - sT[3 * t + 2];
                    const double d2 = (dx * dx + dy * dy) + dz * dz;
                    if (d2 < best) { best = d2; bt = t; }
                }
                if (bt >= 0) bj = sI[bt];
            } else {
                for (int t = 0; t < nt; ++t) {
                    const int j = tidx[t];
                    const double dx = s0 - frame[3 * (size_t)j], dy = s1 - frame[3 * (size_t)j + 1], dz = s2 - frame[3 * (size_t)j + 2];
                    const double d2 = (dx * dx + dy * dy) + dz * dz;
                    if (d2 < best) { best = d2; bj = j; }
                }
            }
            if (bj >= 0 && best <= th2) { nn[b + i] = bj; cnt += 1.0; err += best; } else nn[b + i] = -1;
        }
        cnt = bsum<256>(cnt, sc); err = bsum<256>(err, sc);
        fitness = ns > 0 ? cnt / (double)ns : 0.0;
        rm = cnt > 0 ? sqrt(err / cnt) : 0.0;
        return cnt;
    };
    double ncorr = correspond(fit, rmse);
    for (it = 1; it <= max_iter; ++it) {
        // best rigid update from the current correspondences
        double ms[3] = {0, 0, 0}, md[3] = {0, 0, 0};
        for (int i = tid; i < ns; i += 256) {
            const int j = nn[b + i];
            if (j < 0) continue;
            for (int a = 0; a < 3; ++a) { ms[a] += srcw[3 * (size_t)(b + i) + a]; md[a] += frame[3 * (size_t)j + a]; }
        }
        for (int a = 0; a < 3; ++a) { ms[a] = bsum<256>(ms[a], sc); md[a] = bsum<256>(md[a], sc); }
        if (ncorr > 0) for (int a = 0; a < 3; ++a) { ms[a] /= ncorr; md[a] /= ncorr; }
        double S[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};            // S[a][c] = sum (src-ms)_a (dst-md)_c
        for (int i = tid; i < ns; i += 256) {
            const int j = nn[b + i];
            if (j < 0) continue;
            double sv[3], dv[3];
            for (int a = 0; a < 3; ++a) { sv[a] = srcw[3 * (size_t)(b + i) + a] - ms[a]; dv[a] = frame[3 * (size_t)j + a] - md[a]; }
            for (int a = 0; a < 3; ++a) for (int c = 0; c < 3; ++c) S[3 * a + c] = fma(sv[a], dv[c], S[3 * a + c]);
        }
        for (int i = 0; i < 9; ++i) S[i] = bsum<256>(S[i], sc);
        if (tid == 0) {
            for (int i = 0; i < 16; ++i) U[i] = (i % 5 == 0) ? 1.0 : 0.0;
            if (ncorr > 0) {
                const double Sxx = S[0], Sxy = S[1], Sxz = S[2], Syx = S[3], Syy = S[4], Syz = S[5], Szx = S[6], Szy = S[7], Szz = S[8];
                double N[4][4] = {{Sxx + Syy + Szz, Syz - Szy, Szx - Sxz, Sxy - Syx},
                                  {Syz - Szy, Sxx - Syy - Szz, Sxy + Syx, Szx + Sxz},
                                  {Szx - Sxz, Sxy + Syx, -Sxx + Syy - Szz, Syz + Szy},
                                  {Sxy - Syx, Szx + Sxz, Syz + Szy, -Sxx - Syy + Szz}};
                double q[4], R[9];
                sym4_max_eigvec(N, q);
                quat_to_matrix(q, R);
                for (int a = 0; a < 3; ++a) {
                    U[4 * a] = R[3 * a]; U[4 * a + 1] = R[3 * a + 1]; U[4 * a + 2] = R[3 * a + 2];
                    U[4 * a + 3] = md[a] - (R[3 * a] * ms[0] + R[3 * a + 1] * ms[1] + R[3 * a + 2] * ms[2]);
                }
            }
            double Tn[16];
            for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) {
                double s = 0;
                for (int m = 0; m < 4; ++m) s = fma(U[4 * r + m], T[4 * m + c], s);
                Tn[4 * r + c] = s;
            }
            for (int i = 0; i < 16; ++i) T[i] = Tn[i];
        }
        __syncthreads();
        for (int i = tid; i < ns; i += 256) {
            double* s = srcw + 3 * (size_t)(b + i);
            const double p0 = s[0], p1 = s[1], p2 = s[2];
            for (int a = 0; a < 3; ++a) s[a] = fma(U[4 * a + 2], p2, fma(U[4 * a + 1], p1, U[4 * a] * p0)) + U[4 * a + 3];
        }
        __syncthreads();
        const double pf = fit, pr = rmse;
        ncorr = correspond(fit, rmse);
        if (fabs(pf - fit) < 1e-6 && fabs(pr - rmse) < 1e-6) break;
    }
    // ---- 3. outputs: icp matrix (optionally with the old translation), cluster moved by it ----
    if (tid == 0) {
        if (keep_t) { T[3] = Min[16 * k + 3]; T[7] = Min[16 * k + 7]; T[11] = Min[16 * k + 11]; }
        for (int i = 0; i < 16; ++i) Mout[16 * k + i] = T[i];
        n_iter_out[k] = it > max_iter ? max_iter : it;
    }
    __syncthreads();
    for (int i = tid; i < ns; i += 256) {
        const double* p = local + 3 * (size_t)(b + i);
        for (int a = 0; a < 3; ++a)
            world_out[3 * (size_t)(b + i) + a] = fma(T[4 * a + 2], p[2], fma(T[4 * a + 1], p[1], T[4 * a] * p[0])) + T[4 * a + 3];
    }
}

}  // namespace creg
using namespace creg;

extern "C" size_t creg_icp_workspace_bytes(int64_t n, int64_t nf, int32_t k) {
    if (n < 1 || nf < 1 || k < 1) return 0;
    return icp_layout(n, nf, k).total;
}

extern "C" int creg_masked_icp_f64(const double* local, const float* world, int64_t n, const int32_t* seg_offsets, int32_t k,
                                   const double* frame, int64_t nf, const double* M, double scale, double th,
                                   int32_t max_iteration, int32_t keep_translation, double* M_out, double* world_out,
                                   int32_t* n_iter_out, void* workspace, size_t workspace_bytes, creg_stream_t stream) {
    CREG_REQUIRE(local && world && seg_offsets && frame && M && M_out && world_out && n_iter_out && workspace,
                 "creg_masked_icp_f64: null pointer");
    CREG_REQUIRE(k >= 1 && nf >= 1 && nf < (1ll << 31) && max_iteration >= 1, "creg_masked_icp_f64: bad size");
    CREG_REQUIRE(n >= 1 && n < (1ll << 31), "creg_masked_icp_f64: no source points");
    hipStream_t s = (hipStream_t)stream;
    const IcpLayout L = icp_layout(n, nf, k);
    CREG_REQUIRE(workspace_bytes >= L.total, "creg_masked_icp_f64: workspace too small (%zu < %zu)", workspace_bytes, L.total);
    char* w = (char*)workspace;
    const int lds_cap = (int)(nf < 4096 ? nf : 4096);             // masked targets kept in LDS (28 B each, <= 112 KB)
    const int smem = lds_cap * 28;
    static bool attr_set = false;
    if (!attr_set) {
        CREG_HIP(hipFuncSetAttribute((const void*)k_masked_icp, hipFuncAttributeMaxDynamicSharedMemorySize, 4096 * 28));
        attr_set = true;
    }
    hipLaunchKernelGGL(k_masked_icp, dim3(k), dim3(256), smem, s, local, world, seg_offsets, frame, (int)nf, M,
                       (float)(0.5 * scale), th, max_iteration, keep_translation, M_out, world_out, n_iter_out,
                       (double*)(w + L.srcw), (int*)(w + L.tidx), (int*)(w + L.nn), lds_cap);
    CREG_LAUNCH_CHECK();
    return CREG_OK;
}
