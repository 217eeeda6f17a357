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
#include <type_traits>
#include <vector>
#include "creg_common.h"
#include "creg_dev.h"

namespace creg {

// 1/x and 1/sqrt(x) from the hardware seeds (v_rcp_f64 / v_rsq_f64, ~2^-23) + two Newton steps (~2^-90
// before rounding): a few ulp, ~6 instructions instead of the ~30 of the IEEE division / sqrt expansions.
// Used only inside the Jacobi rotations (x finite, away from 0 and inf), where a few-ulp angle is as good
// as a correctly rounded one: every rotation is re-orthogonalising by construction.
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    return fma(fma(-x, r, 1.0), r, r);
}
__device__ __forceinline__ double fast_rsqrt(double x) {
    double r = __builtin_amdgcn_rsq(x);
    r = fma(0.5 * r, fma(-x * r, r, 1.0), r);
    return fma(0.5 * r, fma(-x * r, r, 1.0), r);
}

// dominant eigenvector of a symmetric 4x4 (cyclic Jacobi with the usual small-element skip, warm started),
// returned as a unit quaternion.  Runs on one lane: the rotation chain is serial, so it is written for few
// instructions (fast_rcp / fast_rsqrt, skipped negligible rotations, eps-level stopping rule).
// Vp (LDS, in/out): the eigenvector basis of the previous call.  Successive ICP iterations have nearly
// the same profile matrix, so Vp^T N Vp is almost diagonal and one or two sweeps finish the job instead
// of five or six; the first call passes the identity.
__device__ void sym4_max_eigvec(const double N[4][4], double q[4], double* Vp) {
    double V[4][4], A[4][4];
    {
        double M[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) V[i][j] = Vp[4 * i + j];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) M[i][j] = fma(N[i][3], V[3][j], fma(N[i][2], V[2][j], fma(N[i][1], V[1][j], N[i][0] * V[0][j])));
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = i; j < 4; ++j) {
                A[i][j] = fma(V[3][i], M[3][j], fma(V[2][i], M[2][j], fma(V[1][i], M[1][j], V[0][i] * M[0][j])));
                A[j][i] = A[i][j];
            }
    }
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0, diag = 0;
        for (int i = 0; i < 4; ++i) { diag += A[i][i] * A[i][i]; for (int j = i + 1; j < 4; ++j) off += A[i][j] * A[i][j]; }
        if (off <= 1e-60 + 1e-31 * diag) break;            // |off-diagonal| <= 3e-16 |diagonal|
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int r = p + 1; r < 4; ++r) {
                const double apr = A[p][r];
                if (fabs(apr) <= 1e-22 * (fabs(A[p][p]) + fabs(A[r][r]))) { A[p][r] = 0.0; A[r][p] = 0.0; continue; }
                const double theta = 0.5 * (A[r][r] - A[p][p]) * fast_rcp(apr);
                const double th2p1 = fma(theta, theta, 1.0);
                const double t = (theta >= 0 ? 1.0 : -1.0) * fast_rcp(fabs(theta) + th2p1 * fast_rsqrt(th2p1));
                const double c = fast_rsqrt(fma(t, t, 1.0)), s = t * c;
#pragma unroll
                for (int k = 0; k < 4; ++k) { const double akp = A[k][p], akr = A[k][r]; A[k][p] = c * akp - s * akr; A[k][r] = s * akp + c * akr; }
#pragma unroll
                for (int k = 0; k < 4; ++k) { const double apk = A[p][k], ark = A[r][k]; A[p][k] = c * apk - s * ark; A[r][k] = s * apk + c * ark; }
#pragma unroll
                for (int k = 0; k < 4; ++k) { const double vkp = V[k][p], vkr = V[k][r]; V[k][p] = c * vkp - s * vkr; V[k][r] = s * vkp + c * vkr; }
            }
    }
    // column of the largest eigenvalue, selected with static indices only (a runtime column index would
    // push A and V to scratch)
    double lam = A[0][0], v0 = V[0][0], v1 = V[1][0], v2 = V[2][0], v3 = V[3][0];
#pragma unroll
    for (int c = 1; c < 4; ++c) {
        const bool gt = A[c][c] > lam;
        lam = gt ? A[c][c] : lam;
        v0 = gt ? V[0][c] : v0; v1 = gt ? V[1][c] : v1; v2 = gt ? V[2][c] : v2; v3 = gt ? V[3][c] : v3;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) Vp[4 * i + j] = V[i][j];
    const double n = sqrt(((v0 * v0 + v1 * v1) + v2 * v2) + v3 * v3);
    const double sg = v0 < 0 ? -1.0 : 1.0;
    q[0] = sg * v0 / n; q[1] = sg * v1 / n; q[2] = sg * v2 / n; q[3] = sg * v3 / n;
}

// adjugate of a symmetric 4x4 (Laplace expansion over the 2x2 minors of its row pairs): adj(M) = det(M) inv(M), and for a
// singular M with a one-dimensional null space, adj(M) = (product of the non-zero eigenvalues) v v^T with v the null vector.
__device__ __forceinline__ void adj4(const double (&a)[4][4], double (&r)[4][4]) {
    const double s0 = a[0][0] * a[1][1] - a[0][1] * a[1][0], s1 = a[0][0] * a[1][2] - a[0][2] * a[1][0], s2 = a[0][0] * a[1][3] - a[0][3] * a[1][0];
    const double s3 = a[0][1] * a[1][2] - a[0][2] * a[1][1], s4 = a[0][1] * a[1][3] - a[0][3] * a[1][1], s5 = a[0][2] * a[1][3] - a[0][3] * a[1][2];
    const double c5 = a[2][2] * a[3][3] - a[2][3] * a[3][2], c4 = a[2][1] * a[3][3] - a[2][3] * a[3][1], c3 = a[2][1] * a[3][2] - a[2][2] * a[3][1];
    const double c2 = a[2][0] * a[3][3] - a[2][3] * a[3][0], c1 = a[2][0] * a[3][2] - a[2][2] * a[3][0], c0 = a[2][0] * a[3][1] - a[2][1] * a[3][0];
    r[0][0] = (a[1][1] * c5 - a[1][2] * c4) + a[1][3] * c3;  r[0][1] = (-a[0][1] * c5 + a[0][2] * c4) - a[0][3] * c3;
    r[0][2] = (a[3][1] * s5 - a[3][2] * s4) + a[3][3] * s3;  r[0][3] = (-a[2][1] * s5 + a[2][2] * s4) - a[2][3] * s3;
    r[1][0] = (-a[1][0] * c5 + a[1][2] * c2) - a[1][3] * c1; r[1][1] = (a[0][0] * c5 - a[0][2] * c2) + a[0][3] * c1;
    r[1][2] = (-a[3][0] * s5 + a[3][2] * s2) - a[3][3] * s1; r[1][3] = (a[2][0] * s5 - a[2][2] * s2) + a[2][3] * s1;
    r[2][0] = (a[1][0] * c4 - a[1][1] * c2) + a[1][3] * c0;  r[2][1] = (-a[0][0] * c4 + a[0][1] * c2) - a[0][3] * c0;
    r[2][2] = (a[3][0] * s4 - a[3][1] * s2) + a[3][3] * s0;  r[2][3] = (-a[2][0] * s4 + a[2][1] * s2) - a[2][3] * s0;
    r[3][0] = (-a[1][0] * c3 + a[1][1] * c1) - a[1][2] * c0; r[3][1] = (a[0][0] * c3 - a[0][1] * c1) + a[0][2] * c0;
    r[3][2] = (-a[3][0] * s3 + a[3][1] * s1) - a[3][2] * s0; r[3][3] = (a[2][0] * s3 - a[2][1] * s1) + a[2][2] * s0;
}

// Dominant eigenvector of Horn's symmetric, traceless 4x4 profile matrix as a unit quaternion (q0 >= 0).
// Fast path (a few hundred mostly independent flops instead of the serial Jacobi chain, which was 6-7 us of every ICP iteration
// on one lane): the largest root of the characteristic polynomial  l^4 + c2 l^2 + c1 l + c0  by Newton from the upper bound
// sqrt(-1.5 c2) (monotone from above), then  v = a column of adj(N - l I),  refined by Rayleigh-quotient rounds
// (l <- v^T N v / v^T v, new adjugate) until l stops moving at the 1e-15 level: each round squares the eigenvalue error, so
// two or three rounds reach what the conditioning of the eigenvector (eps / gap) allows.  When the dominant eigenvalue is
// (nearly) repeated -- too few or degenerate correspondences -- the adjugate vanishes and the warm-started Jacobi sweep
// below takes over.
__device__ void horn_max_eigvec(const double N[4][4], double q[4], double* Vp) {
    double nn2 = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) nn2 = fma(N[i][j], N[i][j], nn2);
    bool ok = nn2 > 0 && nn2 < 1e300;
    if (ok) {
        double A[4][4], R[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) A[i][j] = N[i][j];
        // characteristic polynomial of a traceless symmetric matrix: c2 = -tr(N^2)/2, c1 = -tr(N^3)/3, c0 = det N
        const double c2 = -0.5 * nn2;
        double t3 = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const double n2ij = fma(N[i][3], N[3][j], fma(N[i][2], N[2][j], fma(N[i][1], N[1][j], N[i][0] * N[0][j])));
                t3 = fma(n2ij, N[j][i], t3);
            }
        const double c1 = -t3 / 3.0;
        adj4(A, R);
        const double c0 = fma(N[0][3], R[3][0], fma(N[0][2], R[2][0], fma(N[0][1], R[1][0], N[0][0] * R[0][0])));   // det = row 0 . column 0 of adj
        double lam = sqrt(-1.5 * c2) * (1.0 + 1e-12);
        for (int it = 0; it < 16; ++it) {                       // ~8 steps; the Rayleigh rounds below finish the job
            const double l2 = lam * lam;
            const double P = fma(fma(l2 + c2, lam, c1), lam, c0), dP = fma(fma(4.0, l2, 2.0 * c2), lam, c1);
            if (!(dP > 0)) break;
            const double step = P * fast_rcp(dP);             // Newton corrects itself: a few-ulp quotient is as good
            lam -= step;
            if (fabs(step) <= 1e-11 * fabs(lam)) break;
        }
        const double thr = 1e-9 * nn2 * sqrt(nn2);          // |adj| ~ (product of the gaps to the other three eigenvalues)
        double v0 = 0, v1 = 0, v2 = 0, v3 = 0;
        for (int round = 0; round < 6 && ok; ++round) {
#pragma unroll
            for (int i = 0; i < 4; ++i) A[i][i] = N[i][i] - lam;
            adj4(A, R);
            // the column of the largest diagonal cofactor (static indices only)
            double d = fabs(R[0][0]);
            v0 = R[0][0]; v1 = R[1][0]; v2 = R[2][0]; v3 = R[3][0];
#pragma unroll
            for (int c = 1; c < 4; ++c) {
                const bool gt = fabs(R[c][c]) > d;
                d = gt ? fabs(R[c][c]) : d;
                v0 = gt ? R[0][c] : v0; v1 = gt ? R[1][c] : v1; v2 = gt ? R[2][c] : v2; v3 = gt ? R[3][c] : v3;
            }
            if (!(d > thr)) { ok = false; break; }
            const double w0 = fma(N[0][3], v3, fma(N[0][2], v2, fma(N[0][1], v1, N[0][0] * v0)));
            const double w1 = fma(N[1][3], v3, fma(N[1][2], v2, fma(N[1][1], v1, N[1][0] * v0)));
            const double w2 = fma(N[2][3], v3, fma(N[2][2], v2, fma(N[2][1], v1, N[2][0] * v0)));
            const double w3 = fma(N[3][3], v3, fma(N[3][2], v2, fma(N[3][1], v1, N[3][0] * v0)));
            const double vv = fma(v3, v3, fma(v2, v2, fma(v1, v1, v0 * v0)));
            const double ray = fma(v3, w3, fma(v2, w2, fma(v1, w1, v0 * w0))) * fast_rcp(vv);
            const bool settled = fabs(ray - lam) <= 1e-15 * fabs(ray);
            lam = ray;
            if (settled) break;
        }
        if (ok) {
            const double rn = (v0 < 0 ? -1.0 : 1.0) * fast_rsqrt(((v0 * v0 + v1 * v1) + v2 * v2) + v3 * v3);
            q[0] = v0 * rn; q[1] = v1 * rn; q[2] = v2 * rn; q[3] = v3 * rn;
            return;
        }
    }
    sym4_max_eigvec(N, q, Vp);
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

constexpr int ICP_NT = 512;                           // threads per workgroup
constexpr int ICP_WAVES = ICP_NT / 64;                // 8 waves
constexpr int ICP_ROWS = 256;                         // fallback path: threads that own source points ("row" threads)
constexpr int ICP_PARTS = ICP_NT / ICP_ROWS;          // fallback path: the target list is split this many ways per source point
constexpr int ICP_BATCH_MAX = 16;
constexpr int ICP_SRC_LDS = 1024;                     // source points of a cluster kept in LDS (28 B each)
constexpr int ICP_PAD = 72;                           // far-away entries behind the LDS target list (the split scan overshoots by < 64 + 4)
constexpr int ICP_NSLAB = 64;                         // slabs along the cluster's longest axis (targets and sources are binned by them)

// Sum N values held by the threads over the block, result in every thread.  `mine` (wave-uniform): this wave holds
// contributions; a wave without any contributes exact zeros and skips its DPP sums.
// Fixed association: DPP wave sum per wave, then 0 + w0 + w1 + ... + w7 (by N lanes of wave 0), read back by all.
template <int N>
__device__ __forceinline__ void bsum_n(double (&v)[N], double* sc /* [ICP_WAVES + 1][N] */, bool mine = true) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (mine) {
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = wave_sum_fast(v[i]);
    }
    __syncthreads();                                   // previous readers of sc are done
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < N; ++i) sc[wv * N + i] = mine ? v[i] : 0.0;
    }
    __syncthreads();
    if (threadIdx.x < N) { double r = 0.0; for (int w = 0; w < ICP_WAVES; ++w) r += sc[w * N + threadIdx.x]; sc[ICP_WAVES * N + threadIdx.x] = r; }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = sc[ICP_WAVES * N + i];
}

// one v_min_f64 (fmin() lowers to canonicalising v_max pairs around it); neither operand is ever NaN here
__device__ __forceinline__ double vmin_f64(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// wave-uniform minimum / maximum through the DPP row moves of wave_sum_fast (no LDS-crossbar permutes)
__device__ __forceinline__ double vmax_f64(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
#define CREG_DPP_MINMAX_F64(v, ctrl, mask, OP)                                                                                    \
    {                                                                                                                             \
        const int lo_ = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), ctrl, mask, 0xF, false);                  \
        const int hi_ = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), ctrl, mask, 0xF, false);                  \
        v = OP(v, __hiloint2double(hi_, lo_));                                                                                    \
    }
__device__ __forceinline__ double wave_min_fast(double v) {
    CREG_DPP_MINMAX_F64(v, 0xB1, 0xF, vmin_f64) CREG_DPP_MINMAX_F64(v, 0x4E, 0xF, vmin_f64) CREG_DPP_MINMAX_F64(v, 0x141, 0xF, vmin_f64)
    CREG_DPP_MINMAX_F64(v, 0x140, 0xF, vmin_f64) CREG_DPP_MINMAX_F64(v, 0x142, 0xA, vmin_f64) CREG_DPP_MINMAX_F64(v, 0x143, 0xC, vmin_f64)
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
__device__ __forceinline__ double wave_max_fast(double v) {
    CREG_DPP_MINMAX_F64(v, 0xB1, 0xF, vmax_f64) CREG_DPP_MINMAX_F64(v, 0x4E, 0xF, vmax_f64) CREG_DPP_MINMAX_F64(v, 0x141, 0xF, vmax_f64)
    CREG_DPP_MINMAX_F64(v, 0x140, 0xF, vmax_f64) CREG_DPP_MINMAX_F64(v, 0x142, 0xA, vmax_f64) CREG_DPP_MINMAX_F64(v, 0x143, 0xC, vmax_f64)
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
__device__ __forceinline__ double wave_min_f64(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_max_f64(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}
// slab of a coordinate along the binning axis: monotone in x (the same expression bins targets, sources and search bounds)
__device__ __forceinline__ int icp_slab(double x, double x0, double inv_w) {
    return (int)fmin(fmax((x - x0) * inv_w, 0.0), (double)(ICP_NSLAB - 1));
}

#ifdef CREG_STAMPS
// debug build only (CREG_EXTRA_FLAGS=-DCREG_STAMPS, tests/measure/icp_stamps.py): shader-clock cycles of the phases of
// every workgroup, accumulated over its iterations: [0] mask + setup [1] NN scan [2] combine + sums of the means [3]
// covariance sums [4] Horn on lane 0 [5] move [6] iterations [7] targets scanned per wave and iteration (fast path)
__device__ unsigned long long g_icp_wall[8];      // tail launches: sums of [0] first live block start - launch start [1] last search end - launch start [2] fit end - launch start (100 MHz ticks) [3] launches
__device__ unsigned long long g_icp_wmin[4];      // scratch of the running launch: launch start, first live start, last search end, fit end
__device__ unsigned long long g_icp_stamps[512 * 16];
__global__ void k_icp_wall_fold() {
    if (g_icp_wmin[0] != ~0ull && g_icp_wmin[3] != 0ull) {
        g_icp_wall[0] += g_icp_wmin[1] - g_icp_wmin[0]; g_icp_wall[1] += g_icp_wmin[2] - g_icp_wmin[0]; g_icp_wall[2] += g_icp_wmin[3] - g_icp_wmin[0]; g_icp_wall[3] += 1;
    }
    g_icp_wmin[0] = ~0ull; g_icp_wmin[1] = ~0ull; g_icp_wmin[2] = 0ull; g_icp_wmin[3] = 0ull;
}
     // [workgroup (x + gridDim.x * y) % 512][slot]; 8.. = finer split of the NN scan (wave 0's view): [8] bounds + range [9] scan [10] rescans [11] wait for the other waves
#define ICP_STAMP(slot) do { if (threadIdx.x == 0) { const unsigned long long now_ = clock64(); g_icp_stamps[stamp_b * 16 + slot] += now_ - stamp_t; stamp_t = now_; } } while (0)
#else
#define ICP_STAMP(slot) do { } while (0)
#endif

struct IcpBatch {
    const double* local[ICP_BATCH_MAX]; const float* world[ICP_BATCH_MAX]; const int* off[ICP_BATCH_MAX];
    const int* woff[ICP_BATCH_MAX];                    // segment offsets of `world` (the box clouds); null = same as `off`
    const double* frame[ICP_BATCH_MAX]; const double* Min[ICP_BATCH_MAX];
    const int* toff[ICP_BATCH_MAX];                    // point-to-point mode: target segment offsets into `frame`; null = masked mode
    double* Mout[ICP_BATCH_MAX]; double* world_out[ICP_BATCH_MAX]; int* n_iter_out[ICP_BATCH_MAX];
};

// grid (k, batch): one 512-thread workgroup per cluster per problem; the whole ICP loop of the cluster runs in it.
//
// Fast path (<= lds_cap masked targets and <= ICP_SRC_LDS source points: every cluster of the reference's frames): targets
// and the moving source points live in LDS, both BINNED into ICP_NSLAB slabs along the longest axis of the mask box
// (targets: counting sort while compacting; sources: stable counting sort, once, at their initial pose -- a rigid motion
// of a few degrees keeps them nearly sorted).  A wave owns 64 consecutive sorted sources; the nearest target of a source is
// at most as far as the target it matched in the previous iteration, so the wave scans only the slabs its sources'
// [x - r, x + r] intervals touch -- a fraction of the list once the cluster is near its pose.  The search is exact: every
// target at least as near as the previous match lies in the scanned slabs; among equidistant candidates the lowest frame
// index wins (the sequential scan's first minimum), detected lazily (an equality seen during the scan triggers a rescan
// with the lexicographic comparison).  The sources of a small cluster are split over several waves by target range.
// Fallback (larger clusters of a ragged segmentation, or more masked targets than fit): the unbinned lists, sources or
// targets read through flat pointers from the workspace.
__global__ __launch_bounds__(ICP_NT) void k_masked_icp(IcpBatch P, int nf, float half_scale, double th, int max_iter,
                                                       int keep_t, char* __restrict__ ws, size_t ws_stride, size_t o_srcw,
                                                       size_t o_tidx, size_t o_nn, int lds_cap) {
    __shared__ double sc[(ICP_WAVES + 1) * 20];
    __shared__ float s_lo[3], s_hi[3];
    __shared__ int s_wofs[ICP_WAVES];
    __shared__ double T[16], U[16], Vp[16];
    __shared__ double sB[ICP_NT];                      // per (part, source): best squared distance of the round
    __shared__ int sM[ICP_NT];                         //                     and its target
    __shared__ int tstart[ICP_NSLAB + 1], tfill[ICP_NSLAB], s_sbase[ICP_NSLAB];
    __shared__ int wcnt[2 * ICP_WAVES][ICP_NSLAB];     // source sort: [chunk of 64][slab]
    __shared__ double s_x0, s_inv;
    __shared__ int s_axis;
    const int z = blockIdx.y;
#ifdef CREG_STAMPS
    const int stamp_b = (blockIdx.x + gridDim.x * blockIdx.y) % 512;
    unsigned long long stamp_t = clock64();
#endif
    const double* __restrict__ local = P.local[z]; const float* __restrict__ world = P.world[z];
    const int* __restrict__ off = P.off[z]; const double* __restrict__ frame = P.frame[z];
    const double* __restrict__ Min = P.Min[z];
    char* wz = ws + (size_t)z * ws_stride;
    const int k = blockIdx.x, b = off[k], e = off[k + 1], ns = e - b;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const bool row = tid < ICP_ROWS;
    int* tidx = (int*)(wz + o_tidx) + (size_t)k * nf;

    const int* __restrict__ toff = P.toff[z];         // block-uniform
    // ---- 1. box in float32, exactly as numpy evaluates it on the float32 cluster (min / max: any order) ----
    if (!toff) {
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        if (world) {
            // the box cloud of cluster k may have another size than its ICP source (match() --mlp_icp: frame-0 clusters
            // are the source, the trained clouds of the CURRENT segmentation give the boxes, mlp_reg.py:248,325)
            const int* __restrict__ woff = P.woff[z] ? P.woff[z] : off;
            const int wb = woff[k], we = woff[k + 1];
            for (int i = wb + tid; i < we; i += ICP_NT)
                for (int d = 0; d < 3; ++d) { const float v = world[3 * (size_t)i + d]; lo[d] = fminf(lo[d], v); hi[d] = fmaxf(hi[d], v); }
        } else {
            // no world clouds given: the cluster in its current pose, evaluated in float32 exactly as
            // creg_cluster_transform_f32 does on the float32 casts of `local` and `M` (same fma order)
            float Tf[12];
            for (int q = 0; q < 12; ++q) Tf[q] = (float)Min[16 * k + q];
            for (int i = b + tid; i < e; i += ICP_NT) {
                const float p0 = (float)local[3 * (size_t)i], p1 = (float)local[3 * (size_t)i + 1], p2 = (float)local[3 * (size_t)i + 2];
                for (int d = 0; d < 3; ++d) {
                    const float v = fmaf(p2, Tf[4 * d + 2], fmaf(p1, Tf[4 * d + 1], p0 * Tf[4 * d])) + Tf[4 * d + 3];
                    lo[d] = fminf(lo[d], v); hi[d] = fmaxf(hi[d], v);
                }
            }
        }
        __shared__ float wl[ICP_WAVES][3], wh[ICP_WAVES][3];
        for (int d = 0; d < 3; ++d) {
            for (int o = 32; o >= 1; o >>= 1) { lo[d] = fminf(lo[d], __shfl_xor(lo[d], o, 64)); hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], o, 64)); }
            if (lane == 0) { wl[wv][d] = lo[d]; wh[wv][d] = hi[d]; }
        }
        __syncthreads();
        if (tid < 3) {
            const int d = tid;
            float l = wl[0][d], h = wh[0][d];
            for (int w = 1; w < ICP_WAVES; ++w) { l = fminf(l, wl[w][d]); h = fmaxf(h, wh[w][d]); }
            const float c = (l + h) / 2.0f, sz = h - l;
            s_lo[d] = c - half_scale * sz; s_hi[d] = c + half_scale * sz;
        }
        __syncthreads();
    } else {
        // point-to-point mode: no mask; the bins need the extent of the cluster's own target segment
        const int tb = toff[k], te = toff[k + 1];
        double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int j = tb + tid; j < te; j += ICP_NT)
            for (int d = 0; d < 3; ++d) { const double v = frame[3 * (size_t)j + d]; lo[d] = fmin(lo[d], v); hi[d] = fmax(hi[d], v); }
        __shared__ double dl[ICP_WAVES][3], dh[ICP_WAVES][3];
        for (int d = 0; d < 3; ++d) {
            lo[d] = wave_min_f64(lo[d]); hi[d] = wave_max_f64(hi[d]);
            if (lane == 0) { dl[wv][d] = lo[d]; dh[wv][d] = hi[d]; }
        }
        __syncthreads();
        if (tid < 3) {
            const int d = tid;
            double l = dl[0][d], h = dh[0][d];
            for (int w = 1; w < ICP_WAVES; ++w) { l = fmin(l, dl[w][d]); h = fmax(h, dh[w][d]); }
            s_lo[d] = (float)l; s_hi[d] = (float)h;          // only the binning reads them in this mode
        }
        __syncthreads();
    }
    if (tid == 0) {
        // binning axis: the longest edge of the box.  Any choice is correct; this one prunes best.
        int ax = 0;
        float ext = s_hi[0] - s_lo[0];
        for (int d = 1; d < 3; ++d) if (s_hi[d] - s_lo[d] > ext) { ext = s_hi[d] - s_lo[d]; ax = d; }
        s_axis = ax; s_x0 = (double)s_lo[ax];
        s_inv = ext > 0.f && ext < INFINITY ? (double)ICP_NSLAB / (double)ext : 0.0;
    }
    if (tid < ICP_NSLAB) { tstart[tid] = 0; tfill[tid] = 0; }          // tstart doubles as the slab counters of the first pass
    __syncthreads();
    const int axis = s_axis;
    const double x0 = s_x0, inv_w = s_inv;
    const double blo0 = (double)s_lo[0], blo1 = (double)s_lo[1], blo2 = (double)s_lo[2];
    const double bhi0 = (double)s_hi[0], bhi1 = (double)s_hi[1], bhi2 = (double)s_hi[2];
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* sT = (double*)smem;                       // [lds_cap + ICP_PAD][3] masked target coordinates
    int* sI = (int*)(sT + 3 * (size_t)(lds_cap + ICP_PAD));   // [lds_cap + ICP_PAD] their frame indices (padding: see the fast path)
    double* sS = (double*)(sI + lds_cap + ICP_PAD);   // [ICP_SRC_LDS][3] moving source points
    int* sN = (int*)(sS + 3 * ICP_SRC_LDS);           // [ICP_SRC_LDS] matched target position / frame index, -1 = none
    // ---- 2. the candidate targets: frame points strictly inside the box (masked mode) or the cluster's own segment ----
    const int cb = toff ? toff[k] : 0, cend = toff ? toff[k + 1] : (ns > 0 ? nf : 0);
    auto candidate = [&](int j, double& x, double& y, double& zc) -> bool {
        if (j >= cend) return false;
        x = frame[3 * (size_t)j]; y = frame[3 * (size_t)j + 1]; zc = frame[3 * (size_t)j + 2];
        return toff ? true : (x > blo0 && x < bhi0 && y > blo1 && y < bhi1 && zc > blo2 && zc < bhi2);
    };
    for (int base = cb; base < cend; base += ICP_NT) {                 // pass 1: how many per slab
        double x, y, zc;
        if (candidate(base + tid, x, y, zc)) atomicAdd(&tstart[icp_slab(axis == 0 ? x : (axis == 1 ? y : zc), x0, inv_w)], 1);
    }
    __syncthreads();
    if (wv == 0) {                                    // exclusive prefix over the 64 slabs
        const int c = tstart[lane];
        int inc = c;
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
        __builtin_amdgcn_wave_barrier();
        tstart[lane] = inc - c;
        if (lane == 63) tstart[ICP_NSLAB] = inc;
    }
    __syncthreads();
    const int nt = tstart[ICP_NSLAB];
    const bool fast = nt <= lds_cap && ns <= ICP_SRC_LDS;             // block-uniform
    bool in_lds = nt <= lds_cap;
    const bool src_lds = ns <= ICP_SRC_LDS;
    if (fast) {
        for (int base = cb; base < cend; base += ICP_NT) {             // pass 2: into the slabs (any order inside one)
            const int j = base + tid;
            double x, y, zc;
            if (candidate(j, x, y, zc)) {
                const int sl = icp_slab(axis == 0 ? x : (axis == 1 ? y : zc), x0, inv_w);
                const int p = tstart[sl] + atomicAdd(&tfill[sl], 1);
                sT[3 * p] = x; sT[3 * p + 1] = y; sT[3 * p + 2] = zc; sI[p] = j;
            }
        }
        // padding: the last lanes of a split range read up to 66 entries past it; far-away points that never win
        // (each farther than the one before: equal distances would look like ties to the scan)
        if (tid < ICP_PAD) { sT[3 * (nt + tid)] = 1e150 * (double)(1 + tid); sT[3 * (nt + tid) + 1] = 1e150; sT[3 * (nt + tid) + 2] = 1e150; sI[nt + tid] = 0x7fffffff; }
    } else {
        // fallback: ordered compaction (ascending frame index) into the workspace list, and into LDS while it fits
        int run = 0;
        for (int base = cb; base < cend; base += ICP_NT) {
            const int j = base + tid;
            double x, y, zc;
            const bool in = candidate(j, x, y, zc);
            const unsigned long long m = __ballot(in);
            __syncthreads();                          // s_wofs of the previous round has been read
            if (lane == 0) s_wofs[wv] = __popcll(m);
            __syncthreads();
            int before = run, total = 0;
            for (int w = 0; w < ICP_WAVES; ++w) { const int c = s_wofs[w]; before += w < wv ? c : 0; total += c; }
            if (in) {
                const int slot = before + __popcll(m & ((1ull << lane) - 1ull));
                tidx[slot] = j;
                if (slot < lds_cap) { sT[3 * slot] = x; sT[3 * slot + 1] = y; sT[3 * slot + 2] = zc; sI[slot] = j; }
            }
            run += total;
        }
    }
    double* S = src_lds ? sS : (double*)(wz + o_srcw) + 3 * (size_t)b;      // flat pointer: LDS or workspace
    int* nn = src_lds ? sN : (int*)(wz + o_nn) + b;

    // ---- 3. sources into the world frame with the initial pose (fast path: binned like the targets, stable) ----
    if (tid < 16) { T[tid] = Min[16 * k + tid]; Vp[tid] = (tid % 5 == 0) ? 1.0 : 0.0; }
    for (int i = tid; i < 2 * ICP_WAVES * ICP_NSLAB; i += ICP_NT) (&wcnt[0][0])[i] = 0;
    __syncthreads();                                  // T, sT/sI and (fallback) tidx visible to the block
    if (fast) {
        double pw[2][3];
        int psl[2], prk[2];
#pragma unroll
        for (int rnd = 0; rnd < 2; ++rnd) {
            const int c = wv + ICP_WAVES * rnd, i = 64 * c + lane;
            const bool valid = i < ns;
            int sl = 0;
            if (valid) {
                const double* p = local + 3 * (size_t)(b + i);
                const double p0 = p[0], p1 = p[1], p2 = p[2];
                for (int a = 0; a < 3; ++a) pw[rnd][a] = fma(T[4 * a + 2], p2, fma(T[4 * a + 1], p1, T[4 * a] * p0)) + T[4 * a + 3];
                sl = icp_slab(axis == 0 ? pw[rnd][0] : (axis == 1 ? pw[rnd][1] : pw[rnd][2]), x0, inv_w);
            }
            // rank among the chunk's lower lanes of the same slab; the chunk's count per slab
            unsigned long long rem = __ballot(valid);
            int rank = 0;
            while (rem) {
                const int lead = __ffsll((long long)rem) - 1;
                const int lsl = __builtin_amdgcn_readlane(sl, lead);
                const unsigned long long m = __ballot(valid && sl == lsl);
                if (valid && sl == lsl) rank = __popcll(m & ((1ull << lane) - 1ull));
                if (lane == lead) wcnt[c][lsl] = __popcll(m);
                rem &= ~m;
            }
            psl[rnd] = sl; prk[rnd] = rank;
        }
        __syncthreads();
        if (wv == 0) {                                // slab-major, chunk-minor exclusive prefix
            int run = 0;
            for (int c = 0; c < 2 * ICP_WAVES; ++c) { const int t = wcnt[c][lane]; wcnt[c][lane] = run; run += t; }
            int inc = run;
            for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
            s_sbase[lane] = inc - run;
        }
        __syncthreads();
#pragma unroll
        for (int rnd = 0; rnd < 2; ++rnd) {
            const int c = wv + ICP_WAVES * rnd, i = 64 * c + lane;
            if (i < ns) {
                const int pos = s_sbase[psl[rnd]] + wcnt[c][psl[rnd]] + prk[rnd];
                S[3 * pos] = pw[rnd][0]; S[3 * pos + 1] = pw[rnd][1]; S[3 * pos + 2] = pw[rnd][2];
                nn[pos] = -1;
            }
        }
    } else if (row) {
        for (int i = tid; i < ns; i += ICP_ROWS) {
            const double* p = local + 3 * (size_t)(b + i);
            const double p0 = p[0], p1 = p[1], p2 = p[2];
            for (int a = 0; a < 3; ++a) S[3 * i + a] = fma(T[4 * a + 2], p2, fma(T[4 * a + 1], p1, T[4 * a] * p0)) + T[4 * a + 3];
        }
    }
    const double th2 = th * th;
    double fit = 0, rmse = 0;
    int it = 0;
    // matched target of a source point (position in LDS, or frame index when the targets did not fit)
    auto tgt = [&](int m, int a) -> double { return in_lds ? sT[3 * m + a] : frame[3 * (size_t)m + a]; };

    // reference point of the per-iteration moments (see the loop below)
    double shc[3] = {0.5 * (blo0 + bhi0), 0.5 * (blo1 + bhi1), 0.5 * (blo2 + bhi2)};
    // ---- fast path: slab-pruned exact nearest target; leaves the reduced [count, sum d2, sum src - shc (3), sum tgt - shc (3),
    //      sum (src - shc)(tgt - shc)^T (9)] in cm ----
    // G = 2^lg lanes share one source point and split the scanned range between them, so a wave owns only 64 / G consecutive
    // sorted sources: the range a wave must scan is the union of its sources' intervals, and it shrinks with the number of
    // sources in the wave until the intervals' own width (previous distance + slab granularity) dominates.  Measured on the
    // reference's frames (ns ~ 200, ~600 targets): 64 sources per wave scan ~190 targets each, 16 sources ~80, split four ways.
    const int lg = ns >= 128 ? 2 : (ns >= 64 ? 3 : (ns >= 32 ? 4 : (ns >= 16 ? 5 : 6)));     // small clusters: more lanes per source, all waves busy
    const int G = 1 << lg, spw = 64 >> lg;
    const int units = (ns + spw - 1) >> (6 - lg);                     // wave-sized work items
    const int nround = (units + ICP_WAVES - 1) / ICP_WAVES;
    const int wvs = __builtin_amdgcn_readfirstlane(wv);               // scalar copy: the scan bounds below must stay wave-uniform
                                                                      // for the compiler (a divergent range costs a branch per target)
    const int lg_g = lane & (G - 1), lg_s = lane >> lg;               // lane -> (share of the range, source within the wave)
    const int tsv = tstart[lane];
    auto correspond_fast = [&](double (&cm)[17], bool have_prev) {
        double cq[5] = {0, 0, 0, 0, 0};                // this lane's share of the moments (by lane & 3 within its source)
        __syncthreads();                              // the moves of S are visible
        ICP_STAMP(5);
        for (int rnd = 0; rnd < nround; ++rnd) {
            const int unit = rnd * ICP_WAVES + wvs;
            const bool active = unit < units;                         // wave-uniform
            const int i = unit * spw + lg_s;
            const bool live = active && i < ns;
            double s0 = 0, s1 = 0, s2 = 0, lo = INFINITY, hi = -INFINITY;
            if (live) {
                s0 = S[3 * i]; s1 = S[3 * i + 1]; s2 = S[3 * i + 2];
                const int pm = have_prev ? nn[i] : -1;
                lo = -INFINITY; hi = INFINITY;
                if (pm >= 0) {
                    // no target can be nearer than the previous match is now: the same expression the scan evaluates
                    const double dx = s0 - sT[3 * pm], dy = s1 - sT[3 * pm + 1], dz = s2 - sT[3 * pm + 2];
                    const double sx = axis == 0 ? s0 : (axis == 1 ? s1 : s2);
                    // (x rsqrt(x): a few ulp from sqrt(x), far inside the 1e-12 allowance)
                    const double d2p = (dx * dx + dy * dy) + dz * dz;
                    const double r = (d2p > 1e-280 ? d2p * fast_rsqrt(d2p) : 1e-140) * (1.0 + 1e-12) + (4e-16 * fabs(sx) + 1e-300);
                    lo = sx - r; hi = sx + r;
                }
            }
            const double wlo = wave_min_fast(lo), whi = wave_max_fast(hi);
            // slab -> first target: lane l of tsv holds tstart[l] (one v_readlane each instead of two dependent LDS reads)
            const int sa = __builtin_amdgcn_readfirstlane(icp_slab(wlo, x0, inv_w)), sb = __builtin_amdgcn_readfirstlane(icp_slab(whi, x0, inv_w)) + 1;
            const int r0 = active ? __builtin_amdgcn_readlane(tsv, sa) : 0;
            const int r1 = active ? (sb >= ICP_NSLAB ? nt : __builtin_amdgcn_readlane(tsv, sb & (ICP_NSLAB - 1))) : 0;
            const int per = (r1 - r0 + G - 1) >> lg;                  // targets per lane; the last lanes run into the following
                                                                      // targets or the padding behind the list: both harmless
            const int base = r0 + lg_g * per;
#ifdef CREG_STAMPS
            if (lane == 0) atomicAdd(&g_icp_stamps[stamp_b * 16 + 7], (unsigned long long)per);
            unsigned long long st2 = clock64();
            if (tid == 0) g_icp_stamps[stamp_b * 16 + 8] += st2 - stamp_t;
#endif
            // first minimum in scan order (strict <) and last one (<=): they differ exactly when a second candidate as near
            // as the best was seen -- then the wave rescans with the frame-index tie-break.  5 VALU ops per target on top
            // of the 8 of the distance: two compares, one v_min_f64, two selects.  (The one-compare-per-trip form of k_icp_nn
            // measured 1 % slower here: two workgroups of 8 waves per CU hide the chain.)
            double best = INFINITY; int bm = -1, bl = -1;
            const double* tp = sT + 3 * base;
            // whole trips of four: the last one may run up to three entries past the lane's share -- the next lane's targets
            // (scanned twice: harmless) or the padding behind the list
            for (int st = 0; st < per; st += 4) {     // unconditional body: the 12 LDS reads of a trip go out together
                double d2[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const double dx = s0 - tp[3 * (st + u)], dy = s1 - tp[3 * (st + u) + 1], dz = s2 - tp[3 * (st + u) + 2];
                    d2[u] = (dx * dx + dy * dy) + dz * dz;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    bm = d2[u] < best ? st + u : bm;
                    bl = d2[u] <= best ? st + u : bl;
                    best = vmin_f64(best, d2[u]);
                }
            }
            const unsigned long long tie = __ballot(bm != bl);
            int bj = 0x7fffffff;
            if (tie) {                                // equidistant candidates somewhere in the wave: the lowest frame index wins
                best = INFINITY; bm = -1;
                for (int st = 0; st < ((per + 3) & ~3); ++st) {
                    const double dx = s0 - tp[3 * st], dy = s1 - tp[3 * st + 1], dz = s2 - tp[3 * st + 2];
                    const double d2 = (dx * dx + dy * dy) + dz * dz;
                    const int j = sI[base + st];
                    if (d2 < best || (d2 == best && j < bj)) { best = d2; bm = st; bj = j; }
                }
            } else if (bm >= 0) bj = sI[base + bm];
            bm = bm >= 0 ? base + bm : -1;
#ifdef CREG_STAMPS
            { const unsigned long long now2 = clock64(); if (tid == 0) { g_icp_stamps[stamp_b * 16 + 9] += now2 - st2; g_icp_stamps[stamp_b * 16 + 10] += tie ? 1 : 0; } st2 = now2; }
#endif
            // the G lanes of a source: (distance, frame index) lexicographic minimum
            // (lanes of a source are adjacent: the first two steps are quad permutes on the DPP path, no LDS-crossbar trip)
            {
                auto take = [&](double od, int om, int oj) { if (od < best || (od == best && oj < bj)) { best = od; bm = om; bj = oj; } };
#define CREG_QUAD_XCHG(ctrl)                                                                                                       \
                take(__hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(best), ctrl, 0xF, 0xF, false),                  \
                                      __builtin_amdgcn_update_dpp(0, __double2loint(best), ctrl, 0xF, 0xF, false)),                  \
                     __builtin_amdgcn_update_dpp(0, bm, ctrl, 0xF, 0xF, false), __builtin_amdgcn_update_dpp(0, bj, ctrl, 0xF, 0xF, false))
                CREG_QUAD_XCHG(0xB1);                 // quad_perm [1,0,3,2]: lane ^ 1
                CREG_QUAD_XCHG(0x4E);                 // quad_perm [2,3,0,1]: lane ^ 2   (G >= 4 always)
#undef CREG_QUAD_XCHG
                for (int o = 4; o < G; o <<= 1) take(__shfl_xor(best, o, 64), __shfl_xor(bm, o, 64), __shfl_xor(bj, o, 64));
            }
            // every lane of the source knows the winner; lane 0 records it, lanes 0..3 share the 17 moments between them
            // (5 + 5 + 5 + 2: a quarter of the wave reductions below for each)
            const bool ok = live && bm >= 0 && best < th2;          // strict: open3d KDTreeFlann::SearchHybrid keeps d^2 < r^2
            if (live && lg_g == 0) nn[i] = ok ? bm : -1;
            if (ok && lg_g < 4) {
                const double sv[3] = {s0 - shc[0], s1 - shc[1], s2 - shc[2]};
                const double dv[3] = {sT[3 * bm] - shc[0], sT[3 * bm + 1] - shc[1], sT[3 * bm + 2] - shc[2]};
                if (lg_g == 0) { cq[0] += 1.0; cq[1] += best; cq[2] += sv[0]; cq[3] += sv[1]; cq[4] += sv[2]; }
                else if (lg_g == 1) { cq[0] += dv[0]; cq[1] += dv[1]; cq[2] += dv[2]; cq[3] = fma(sv[0], dv[0], cq[3]); cq[4] = fma(sv[0], dv[1], cq[4]); }
                else if (lg_g == 2) { cq[0] = fma(sv[0], dv[2], cq[0]); cq[1] = fma(sv[1], dv[0], cq[1]); cq[2] = fma(sv[1], dv[1], cq[2]);
                                      cq[3] = fma(sv[1], dv[2], cq[3]); cq[4] = fma(sv[2], dv[0], cq[4]); }
                else { cq[0] = fma(sv[2], dv[1], cq[0]); cq[1] = fma(sv[2], dv[2], cq[1]); }
            }
        }
        ICP_STAMP(1);
        // sums over the lanes of equal (lane & 3): two row rotations on the DPP path, two cross-row exchanges; fixed association
        const bool mine = wvs < units;
        if (mine) {
#pragma unroll
            for (int a = 0; a < 5; ++a) {
                double v = cq[a];
#define CREG_ROR_ADD(ctrl) v += __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), ctrl, 0xF, 0xF, false), \
                                                 __builtin_amdgcn_update_dpp(0, __double2loint(v), ctrl, 0xF, 0xF, false))
                CREG_ROR_ADD(0x124);                  // row_ror:4
                CREG_ROR_ADD(0x128);                  // row_ror:8 -> every lane holds its residue's sum over its row of 16
#undef CREG_ROR_ADD
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                cq[a] = v;
            }
        }
        __syncthreads();                              // previous readers of sc are done
        if (lane < 4) {
#pragma unroll
            for (int a = 0; a < 5; ++a) sc[wvs * 20 + lane * 5 + a] = mine ? cq[a] : 0.0;
        }
        __syncthreads();
        if (tid < 20) { double r = 0.0; for (int w = 0; w < ICP_WAVES; ++w) r += sc[w * 20 + tid]; sc[ICP_WAVES * 20 + tid] = r; }
        __syncthreads();
        // slot (residue, a) -> moment: residue 0: count, sum d2, s0..2; 1: d0..2, C00, C01; 2: C02, C10, C11, C12, C20; 3: C21, C22
        const double* so = sc + ICP_WAVES * 20;
        cm[0] = so[0]; cm[1] = so[1]; cm[2] = so[2]; cm[3] = so[3]; cm[4] = so[4];
        cm[5] = so[5]; cm[6] = so[6]; cm[7] = so[7]; cm[8] = so[8]; cm[9] = so[9];
        cm[10] = so[10]; cm[11] = so[11]; cm[12] = so[12]; cm[13] = so[13]; cm[14] = so[14];
        cm[15] = so[15]; cm[16] = so[16];
        ICP_STAMP(2);
    };

    // ---- fallback: unbinned lists (row threads own the sources, the target list is split over ICP_PARTS threads each) ----
    const int part_s = __builtin_amdgcn_readfirstlane(tid / ICP_ROWS), pi = tid % ICP_ROWS;
    const int t0 = (int)((long long)nt * part_s / ICP_PARTS), t1 = (int)((long long)nt * (part_s + 1) / ICP_PARTS);
    auto correspond_slow = [&](double (&cm)[17]) {
        for (int a = 0; a < 17; ++a) cm[a] = 0;
        __syncthreads();                              // the row threads' moves of S are visible
        for (int r0 = 0; r0 < ns; r0 += ICP_ROWS) {
            const int i = r0 + pi;
            double best = INFINITY; int bm = -1;
            if (i < ns) {
                const double s0 = S[3 * i], s1 = S[3 * i + 1], s2 = S[3 * i + 2];
                if (in_lds) {
#pragma unroll 4
                    for (int t = t0; t < t1; ++t) {
                        const double dx = s0 - sT[3 * t], dy = s1 - sT[3 * t + 1], dz = s2 - sT[3 * t + 2];
                        const double d2 = (dx * dx + dy * dy) + dz * dz;
                        if (d2 < best) { best = d2; bm = t; }
                    }
                } else {
                    for (int t = t0; t < t1; ++t) {
                        const int j = tidx[t];
                        const double dx = s0 - frame[3 * (size_t)j], dy = s1 - frame[3 * (size_t)j + 1], dz = s2 - frame[3 * (size_t)j + 2];
                        const double d2 = (dx * dx + dy * dy) + dz * dz;
                        if (d2 < best) { best = d2; bm = j; }
                    }
                }
            }
            sB[part_s * ICP_ROWS + pi] = best; sM[part_s * ICP_ROWS + pi] = bm;
            __syncthreads();
            if (row && i < ns) {                      // parts in ascending target order, strict '<': first minimum
                best = sB[pi]; bm = sM[pi];
#pragma unroll
                for (int q = 1; q < ICP_PARTS; ++q) { const double d = sB[q * ICP_ROWS + pi]; if (d < best) { best = d; bm = sM[q * ICP_ROWS + pi]; } }
                const bool ok = bm >= 0 && best < th2;
                nn[i] = ok ? bm : -1;
                if (ok) {
                    double sv[3], dv[3];
                    for (int a = 0; a < 3; ++a) { sv[a] = S[3 * i + a] - shc[a]; dv[a] = tgt(bm, a) - shc[a]; }
                    cm[0] += 1.0; cm[1] += best;
                    for (int a = 0; a < 3; ++a) { cm[2 + a] += sv[a]; cm[5 + a] += dv[a]; }
                    for (int a = 0; a < 3; ++a) for (int c = 0; c < 3; ++c) cm[8 + 3 * a + c] = fma(sv[a], dv[c], cm[8 + 3 * a + c]);
                }
            }
            if (r0 + ICP_ROWS < ns) __syncthreads();  // sB / sM are rewritten by the next round
        }
        bsum_n<17>(cm, sc);
    };
    auto correspond = [&](double (&cm)[17], bool have_prev) {
        if (fast) correspond_fast(cm, have_prev); else correspond_slow(cm);
        fit = ns > 0 ? cm[0] / (double)ns : 0.0;
        rmse = cm[0] > 0 ? sqrt(cm[1] / cm[0]) : 0.0;
    };
    ICP_STAMP(0);
    // One pass per iteration: a matched pair adds its count, squared distance, coordinates and their outer product, all
    // taken relative to the point shc -- the centroid of the previous iteration's matched targets (the box centre at the
    // start).  Horn's update puts the matched sources' centroid exactly there, so shc is within the iteration's small
    // motion of both centroids of the NEXT correspondences and  C = sum (s - shc)(d - shc)^T - n (ms - shc)(md - shc)^T
    // loses nothing to cancellation; the separate centred pass over the correspondences (and its block reduction) is gone.
    double cm[17];
    correspond(cm, false);
    for (it = 1; it <= max_iter; ++it) {
        // best rigid update from the current correspondences (their count and moments came with them)
        const double ncorr = cm[0];
        double ms[3], md[3], msr[3], mdr[3];          // centroids, absolute and relative to shc
        for (int a = 0; a < 3; ++a) {
            msr[a] = ncorr > 0 ? cm[2 + a] / ncorr : 0.0; mdr[a] = ncorr > 0 ? cm[5 + a] / ncorr : 0.0;
            ms[a] = shc[a] + msr[a]; md[a] = shc[a] + mdr[a];
        }
        double C[9];                                  // C[a][c] = sum (src-ms)_a (dst-md)_c
        for (int a = 0; a < 3; ++a) for (int c = 0; c < 3; ++c) C[3 * a + c] = fma(-ncorr * msr[a], mdr[c], cm[8 + 3 * a + c]);
        if (ncorr > 0) for (int a = 0; a < 3; ++a) shc[a] = md[a];
        ICP_STAMP(3);
        if (tid == 0) {
            for (int i = 0; i < 16; ++i) U[i] = (i % 5 == 0) ? 1.0 : 0.0;
            if (ncorr > 0) {
                const double Sxx = C[0], Sxy = C[1], Sxz = C[2], Syx = C[3], Syy = C[4], Syz = C[5], Szx = C[6], Szy = C[7], Szz = C[8];
                double N[4][4] = {{Sxx + Syy + Szz, Syz - Szy, Szx - Sxz, Sxy - Syx},
                                  {Syz - Szy, Sxx - Syy - Szz, Sxy + Syx, Szx + Sxz},
                                  {Szx - Sxz, Sxy + Syx, -Sxx + Syy - Szz, Syz + Szy},
                                  {Sxy - Syx, Szx + Sxz, Syz + Szy, -Sxx - Syy + Szz}};
                double q[4], R[9];
                horn_max_eigvec(N, q, Vp);
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
        ICP_STAMP(4);
#ifdef CREG_STAMPS
        if (tid == 0) g_icp_stamps[stamp_b * 16 + 6] += 1;
#endif
        for (int i = tid; i < ns; i += ICP_NT) {
            const double p0 = S[3 * i], p1 = S[3 * i + 1], p2 = S[3 * i + 2];
            for (int a = 0; a < 3; ++a) S[3 * i + a] = fma(U[4 * a + 2], p2, fma(U[4 * a + 1], p1, U[4 * a] * p0)) + U[4 * a + 3];
        }
        const double pf = fit, pr = rmse;
        correspond(cm, true);
        if (fabs(pf - fit) < 1e-6 && fabs(pr - rmse) < 1e-6) break;
    }
    // ---- 4. outputs: icp matrix (optionally with the old translation), cluster moved by it ----
    __syncthreads();
    if (tid == 0) {
        if (keep_t) { T[3] = Min[16 * k + 3]; T[7] = Min[16 * k + 7]; T[11] = Min[16 * k + 11]; }
        for (int i = 0; i < 16; ++i) P.Mout[z][16 * k + i] = T[i];
        P.n_iter_out[z][k] = it > max_iter ? max_iter : it;
    }
    __syncthreads();
    double* world_out = P.world_out[z];
    for (int i = tid; i < ns; i += ICP_NT) {
        const double* p = local + 3 * (size_t)(b + i);
        for (int a = 0; a < 3; ++a)
            world_out[3 * (size_t)(b + i) + a] = fma(T[4 * a + 2], p[2], fma(T[4 * a + 1], p[1], T[4 * a] * p[0])) + T[4 * a + 3];
    }
}


// =================================================================================================================
// Large regime (clusters of more than ICP_SRC_LDS points, or frames whose masked targets do not fit a CU's LDS: the
// BASELINE configs[4] shape has 2048-point clusters against ~3000 masked targets each): the same ICP as k_masked_icp's
// fast path -- binned targets and sources, exact search pruned by the previous match, one-pass shifted moments --
// iteration by iteration over MANY workgroups instead of one workgroup per cluster.  The bins are a 16 x 16 grid over the
// box's two longest edges (a cluster is a surface patch: slabs along one axis pruned 2.5x at this size, the grid ~12x).
//   k_icp_mask    (cluster)           box; frame indices inside it, grouped by grid cell (count pass + scatter pass), or in
//                                     ascending order for creg_aabb_mask_f64
//   k_icp_init    (cluster)           sources into the world frame with the initial pose, stable-sorted by cell
//   k_icp_pool    (cluster, slices)   the masked targets' coordinates, in cell order, into a pool (coalesced staging)
//   k_icp_nn      (64-source chunk)   applies the pending rigid update to its sources; nearest masked target of each: a wave =
//                                     16 consecutive sorted sources x 4 lanes, the lanes of a source walking the grid rows of
//                                     the cell rectangle the wave's (x - r, x + r) squares touch (r = distance to the
//                                     previous match); the chunk's moments; and, in the cluster's LAST chunk to finish, the
//                                     fit (one wave): chunk moments in chunk order -> fitness / RMSE, open3d's convergence
//                                     test against the previous ones, else Horn's closed form: new pose, pending update
//   k_icp_compact (one workgroup)     between batches: the chunks of the clusters still iterating (the next batch's grid)
//   k_icp_finish  (cluster)           outputs
// The host enqueues the search in batches of 16 and reads "clusters / chunks still running" between batches.
typedef float nn_f2 __attribute__((ext_vector_type(2)));          // (v_pk_* float32 pairs: the screen of k_icp_nn)
struct IcpLarge {                          // per problem
    const double* local; const float* world; const int* off; const int* woff; const double* frame; const double* Min;
    const int* toff;                           // point-to-point mode: cluster c's targets are frame[toff[c] .. toff[c + 1]), no mask; null = masked mode
    double* Mout; double* world_out; int* n_iter_out;
    double* srcw; int* tidx; int* tcount; float* box; int* nn; double* part; double* state; int* running;
    int* chunk0;                               // [k + 1] first source chunk of every cluster (k_icp_nn's block -> cluster map)
    int* tst;                                  // [k][ICP_NCELL + 1] first target of every cell, row-major in (a, b), g x g used (binned mode)
    int* chunk_cl;                             // [chunks] cluster of every chunk (k_icp_nn's block -> cluster map)
    double* prevt;                             // [n][3] coordinates of every source's current match (the next search's bound)
    double* tcx; double* tcy; double* tcz;     // pool of the clusters' masked target coordinates in cell order (coalesced staging)
    float4* tcf;                               // the same entries as (float32 coordinates relative to the cluster's box centre, frame index): what the screen stages
    unsigned* tlmax;                           // [k] bits of the largest |coordinate - box centre| among a cluster's masked targets (float32, rounded up)
    int* tbase;                                // [k] a cluster's first pool entry, -1: did not fit (its targets are gathered through tidx)
    int* arrive;                               // [k] chunks of the cluster that have finished the running search
    int* live;                                 // [chunks] chunks of the clusters still iterating (k_icp_compact), read when `use_live`
    int use_live;
    int pool_cap;
    int screen;                                // k_icp_nn: float32 screen in front of the fp64 scan (0: measurement / identity knob)
};
constexpr int ICP_CH = 64;                 // sources per k_icp_nn workgroup (4 waves of 16)
#ifndef ICP_LG
#define ICP_LG 2                           // log2 of the lanes that share one source point in k_icp_nn (measured at configs[4], 12 frames:
                                           // 4 lanes 38.5 ms per frame, 8 lanes (-DICP_LG=3) 43.6)
#endif
constexpr int ICP_G = 1 << ICP_LG;         // lane groups of a wave = lanes per source
constexpr int ICP_SPW = 64 >> ICP_LG;      // sources per wave
constexpr int ICP_NNW = ICP_CH / ICP_SPW;  // waves of a k_icp_nn workgroup
#ifndef ICP_SB_N
#define ICP_SB_N 64
#endif
constexpr int ICP_SB = ICP_SB_N;           // targets a lane group stages per batch (128 measured no better)
constexpr int ICP_ST = 48;                 // doubles of per-cluster state: T[16] U[16] prev_fit prev_rmse done n_updates | a: x0 inv_w axis | shc[3] | b: x0 inv_w axis | g
constexpr int ICP_NM = 17;                 // moments per chunk: count, sum d2, sum (s - shc), sum (d - shc), sum (s - shc)(d - shc)^T
constexpr int ICP_GMAX = 64;               // the grid over the box's longest (a) and second longest (b) edge has g x g cells, g per
constexpr int ICP_NCELL = ICP_GMAX * ICP_GMAX;      // cluster by its size: ~8 sources (~12 targets) per cell, 8 <= g <= 64
__device__ __forceinline__ int icp_grid_dim(int ns) {
    int g = (int)ceil(sqrt((double)ns / 8.0));
    return g < 8 ? 8 : (g > ICP_GMAX ? ICP_GMAX : g);
}
__device__ __forceinline__ int icp_bin(double x, double x0, double inv_w, int nbin) { return (int)fmin(fmax((x - x0) * inv_w, 0.0), (double)(nbin - 1)); }

// binned = 0: ascending compaction into tidx / tcount (creg_aabb_mask_f64, the G9 golden).
// binned = 1: tidx holds the same indices grouped by slab (any order inside one), tst the slab starts, state the binning.
__global__ __launch_bounds__(1024) void k_icp_mask(IcpLarge P, int nf, float half_scale, int from_pose, int binned) {
    __shared__ float wl[16][3], wh[16][3];
    __shared__ float s_lo[3], s_hi[3];
    __shared__ int s_wofs[16];
    __shared__ int cnt[ICP_NCELL + 1], fill[ICP_NCELL];
    const int k = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int b = P.off[k], e = P.off[k + 1];
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    const int* toff = P.toff;                  // block-uniform
    if (toff) {                                // point-to-point mode: no mask; the cells need the extent of the cluster's own target segment
        for (int j = toff[k] + tid; j < toff[k + 1]; j += 1024)           // (any box is correct: the binning clamps and is monotone)
            for (int d = 0; d < 3; ++d) { const float v = (float)P.frame[3 * (size_t)j + d]; lo[d] = fminf(lo[d], v); hi[d] = fmaxf(hi[d], v); }
    } else if (!from_pose) {
        const int* woff = P.woff ? P.woff : P.off;
        for (int i = woff[k] + tid; i < woff[k + 1]; i += 1024)
            for (int d = 0; d < 3; ++d) { const float v = P.world[3 * (size_t)i + d]; lo[d] = fminf(lo[d], v); hi[d] = fmaxf(hi[d], v); }
    } else {                                   // boxes of float32(M) . float32(local), creg_cluster_transform_f32's arithmetic
        float Tf[12];
        for (int q = 0; q < 12; ++q) Tf[q] = (float)P.Min[16 * k + q];
        for (int i = b + tid; i < e; i += 1024) {
            const float p0 = (float)P.local[3 * (size_t)i], p1 = (float)P.local[3 * (size_t)i + 1], p2 = (float)P.local[3 * (size_t)i + 2];
            for (int d = 0; d < 3; ++d) {
                const float v = fmaf(p2, Tf[4 * d + 2], fmaf(p1, Tf[4 * d + 1], p0 * Tf[4 * d])) + Tf[4 * d + 3];
                lo[d] = fminf(lo[d], v); hi[d] = fmaxf(hi[d], v);
            }
        }
    }
    for (int d = 0; d < 3; ++d) {
        for (int o = 32; o >= 1; o >>= 1) { lo[d] = fminf(lo[d], __shfl_xor(lo[d], o, 64)); hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], o, 64)); }
        if (lane == 0) { wl[wv][d] = lo[d]; wh[wv][d] = hi[d]; }
    }
    for (int q = tid; q <= ICP_NCELL; q += 1024) cnt[q] = 0;
    for (int q = tid; q < ICP_NCELL; q += 1024) fill[q] = 0;
    __syncthreads();
    if (tid < 3) {
        const int d = tid;
        float l = wl[0][d], h = wh[0][d];
        for (int w = 1; w < 16; ++w) { l = fminf(l, wl[w][d]); h = fmaxf(h, wh[w][d]); }
        if (toff) {
            //  point-to-point mode: l / h are float32 ROUNDINGS of the fp64 targets, which can lie half an ulp outside them -- the
            // float32 screen of k_icp_nn takes its length bound from this box ("the box holds the targets"): one ulp outwards makes that true;
            // an EMPTY target segment (+inf / -inf) would give a NaN centre and an infinite bound: a point box at the origin instead
            if (!(l <= h)) { l = 0.f; h = 0.f; }
            else { l = nextafterf(l, -INFINITY); h = nextafterf(h, INFINITY); }
        }
        const float c = (l + h) / 2.0f, sz = h - l;
        s_lo[d] = toff ? l : c - half_scale * sz; s_hi[d] = toff ? h : c + half_scale * sz;
        if (P.box) { P.box[6 * k + d] = s_lo[d]; P.box[6 * k + 3 + d] = s_hi[d]; }
    }
    __syncthreads();
    const double blo0 = (double)s_lo[0], blo1 = (double)s_lo[1], blo2 = (double)s_lo[2];
    const double bhi0 = (double)s_hi[0], bhi1 = (double)s_hi[1], bhi2 = (double)s_hi[2];
    int* tidx = P.tidx + (size_t)k * nf;
    if (binned) {
        // cells over the two longest edges of the box (a surface patch is two-dimensional: the third edge prunes nothing)
        float ext[3] = {s_hi[0] - s_lo[0], s_hi[1] - s_lo[1], s_hi[2] - s_lo[2]};
        int axa = 0;
        for (int d = 1; d < 3; ++d) if (ext[d] > ext[axa]) axa = d;
        int axb = axa == 0 ? 1 : 0;
        for (int d = 0; d < 3; ++d) if (d != axa && ext[d] > ext[axb]) axb = d;
        const int gd = icp_grid_dim(e - b), ncell = gd * gd;
        const double x0a = (double)s_lo[axa], inv_a = ext[axa] > 0.f && ext[axa] < INFINITY ? (double)gd / (double)ext[axa] : 0.0;
        const double x0b = (double)s_lo[axb], inv_b = ext[axb] > 0.f && ext[axb] < INFINITY ? (double)gd / (double)ext[axb] : 0.0;
        const int j0 = toff ? toff[k] : 0, nfe = e > b ? (toff ? toff[k + 1] : nf) : j0;
        __shared__ int wsum[16];
        for (int pass = 0; pass < 2; ++pass) {
            for (int j = j0 + tid; j < nfe; j += 1024) {
                const double p[3] = {P.frame[3 * (size_t)j], P.frame[3 * (size_t)j + 1], P.frame[3 * (size_t)j + 2]};
                if (toff || (p[0] > blo0 && p[0] < bhi0 && p[1] > blo1 && p[1] < bhi1 && p[2] > blo2 && p[2] < bhi2)) {
                    const double ca = axa == 0 ? p[0] : (axa == 1 ? p[1] : p[2]), cb = axb == 0 ? p[0] : (axb == 1 ? p[1] : p[2]);
                    const int cell = icp_bin(ca, x0a, inv_a, gd) * gd + icp_bin(cb, x0b, inv_b, gd);
                    if (pass == 0) atomicAdd(&cnt[cell], 1);
                    else tidx[cnt[cell] + atomicAdd(&fill[cell], 1)] = j;
                }
            }
            __syncthreads();
            if (pass == 0) {                          // exclusive prefix over the cells: four per thread, wave scans, wave totals
                int c4[4], run4 = 0;
                for (int q = 0; q < 4; ++q) { c4[q] = 4 * tid + q < ncell ? cnt[4 * tid + q] : 0; run4 += c4[q]; }
                int inc = run4;
                for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
                if (lane == 63) wsum[wv] = inc;
                __syncthreads();
                int acc = inc - run4, total = 0;
                for (int w = 0; w < 16; ++w) { const int t = wsum[w]; acc += w < wv ? t : 0; total += t; }
                for (int q = 0; q < 4; ++q) { if (4 * tid + q < ncell) cnt[4 * tid + q] = acc; acc += c4[q]; }
                if (tid == 0) cnt[ncell] = total;
                __syncthreads();
            }
        }
        for (int q = tid; q <= ncell; q += 1024) P.tst[(size_t)k * (ICP_NCELL + 1) + q] = cnt[q];
        if (tid == 0) {
            P.tcount[k] = cnt[ncell];
            double* st = P.state + ICP_ST * k;
            st[36] = x0a; st[37] = inv_a; st[38] = (double)axa;
            st[39] = 0.5 * (blo0 + bhi0); st[40] = 0.5 * (blo1 + bhi1); st[41] = 0.5 * (blo2 + bhi2);
            st[42] = x0b; st[43] = inv_b; st[44] = (double)axb; st[45] = (double)gd;
        }
        return;
    }
    int run = 0;
    for (int base = 0; base < nf; base += 1024) {
        const int j = base + tid;
        bool in = false;
        if (j < nf && e > b) {
            const double x = P.frame[3 * (size_t)j], y = P.frame[3 * (size_t)j + 1], z = P.frame[3 * (size_t)j + 2];
            in = x > blo0 && x < bhi0 && y > blo1 && y < bhi1 && z > blo2 && z < bhi2;
        }
        const unsigned long long m = __ballot(in);
        __syncthreads();
        if (lane == 0) s_wofs[wv] = __popcll(m);
        __syncthreads();
        int before = run, total = 0;
        for (int w = 0; w < 16; ++w) { const int c = s_wofs[w]; before += w < wv ? c : 0; total += c; }
        if (in) tidx[before + __popcll(m & ((1ull << lane) - 1ull))] = j;
        run += total;
    }
    if (tid == 0) P.tcount[k] = run;
}

// one workgroup per cluster: sources with the initial pose, stable-sorted by grid cell into srcw (k_icp_mask ran before).
// Two stable counting sorts of <= 64 bins each -- by column into the (still unused) match-coordinate buffer, then by row
// into srcw: least-significant-digit radix, so the final order is row-major by cell with the original order inside a cell.
__global__ __launch_bounds__(1024) void k_icp_init(IcpLarge P, int k_total) {
    __shared__ int base[ICP_GMAX], run[ICP_GMAX];
    __shared__ int wcnt[16][ICP_GMAX];
    const int k = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int b = P.off[k], e = P.off[k + 1], ns = e - b;
    double* st = P.state + ICP_ST * k;
    const double x0a = st[36], inv_a = st[37], x0b = st[42], inv_b = st[43];
    const int axa = (int)st[38], axb = (int)st[44], gd = (int)st[45];
    double T[12];
    for (int q = 0; q < 12; ++q) T[q] = P.Min[16 * k + q];
    for (int pass = 0; pass < 2; ++pass) {
        // pass 0: local -> world, key = column (b axis), into prevt; pass 1: prevt -> srcw, key = row (a axis)
        const double* src = pass == 0 ? P.local : P.prevt;
        double* dst = pass == 0 ? P.prevt : P.srcw;
        auto point = [&](int i, double (&w)[3]) -> int {
            const double p0 = src[3 * (size_t)(b + i)], p1 = src[3 * (size_t)(b + i) + 1], p2 = src[3 * (size_t)(b + i) + 2];
            if (pass == 0) { for (int a = 0; a < 3; ++a) w[a] = fma(T[4 * a + 2], p2, fma(T[4 * a + 1], p1, T[4 * a] * p0)) + T[4 * a + 3]; }
            else { w[0] = p0; w[1] = p1; w[2] = p2; }
            const int ax = pass == 0 ? axb : axa;
            const double cc = ax == 0 ? w[0] : (ax == 1 ? w[1] : w[2]);
            return pass == 0 ? icp_bin(cc, x0b, inv_b, gd) : icp_bin(cc, x0a, inv_a, gd);
        };
        __syncthreads();                              // the previous pass has written its output
        if (tid < ICP_GMAX) { base[tid] = 0; run[tid] = 0; }
        __syncthreads();
        for (int i = tid; i < ns; i += 1024) {
            double w[3];
            atomicAdd(&base[point(i, w)], 1);
        }
        __syncthreads();
        if (wv == 0) {
            const int c = base[lane];
            int inc = c;
            for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
            base[lane] = inc - c;
        }
        for (int r0 = 0; r0 < ns; r0 += 1024) {
            __syncthreads();                          // base / run of the previous round are final; wcnt may be rewritten
            for (int q = tid; q < 16 * ICP_GMAX; q += 1024) (&wcnt[0][0])[q] = 0;
            __syncthreads();
            const int i = r0 + tid;
            const bool valid = i < ns;
            double w[3] = {0, 0, 0};
            int sl = 0;
            if (valid) sl = point(i, w);
            unsigned long long rem = __ballot(valid);
            int rank = 0;
            while (rem) {                             // rank among the wave's lower lanes of the same bin; the wave's count per bin
                const int lead = __ffsll((long long)rem) - 1;
                const int lsl = __builtin_amdgcn_readlane(sl, lead);
                const unsigned long long m = __ballot(valid && sl == lsl);
                if (valid && sl == lsl) rank = __popcll(m & ((1ull << lane) - 1ull));
                if (lane == lead) wcnt[wv][lsl] = __popcll(m);
                rem &= ~m;
            }
            __syncthreads();
            if (tid < ICP_GMAX) {                     // per bin: exclusive prefix over the waves, then the round's total
                int acc = 0;
                for (int w2 = 0; w2 < 16; ++w2) { const int t = wcnt[w2][tid]; wcnt[w2][tid] = acc; acc += t; }
                const int r = run[tid];
                run[tid] = r + acc;
                for (int w2 = 0; w2 < 16; ++w2) wcnt[w2][tid] += r;
            }
            __syncthreads();
            if (valid) {
                const size_t pos = (size_t)b + base[sl] + wcnt[wv][sl] + rank;
                dst[3 * pos] = w[0]; dst[3 * pos + 1] = w[1]; dst[3 * pos + 2] = w[2];
                if (pass == 1) P.nn[pos] = -1;
            }
        }
    }
    if (tid < 16) { st[tid] = P.Min[16 * k + tid]; st[16 + tid] = (tid % 5 == 0) ? 1.0 : 0.0; }      // pose; pending update = identity
    if (tid == 0) {
        st[32] = 0.0; st[33] = 0.0; st[34] = 0.0; st[35] = 0.0;       // prev fitness, prev rmse, done, updates applied
        P.arrive[k] = 0;
        // a cluster without points has no chunk to run its fit: what two fits would leave (identity update, then converged)
        if (ns == 0) { st[34] = 1.0; st[35] = 1.0; }
        if (k == 0) {
            int live_clusters = 0;
            for (int j = 0; j < k_total; ++j) live_clusters += P.off[j + 1] > P.off[j];
            *P.running = live_clusters;
            int c = 0;                                                // chunks of ICP_CH sources, never across clusters
            for (int j = 0; j < k_total; ++j) { P.chunk0[j] = c; c += (P.off[j + 1] - P.off[j] + ICP_CH - 1) / ICP_CH; }
            P.chunk0[k_total] = c;
            int pb = 0;                                               // pool offsets of the clusters' cell-sorted target coordinates
            for (int j = 0; j < k_total; ++j) {
                const int ntj = P.tcount[j];
                if (pb + ntj <= P.pool_cap) { P.tbase[j] = pb; pb += ntj; } else P.tbase[j] = -1;
                P.tlmax[j] = 0u;
            }
        }
    }
    // the cluster of each of this cluster's chunks (the same count, recomputed here: no wait for thread 0 of cluster 0)
    {
        int c0 = 0;
        for (int j = tid; j < k; j += 1024) c0 += (P.off[j + 1] - P.off[j] + ICP_CH - 1) / ICP_CH;
        __shared__ int s_part[16];
        c0 = wave_sum(c0);
        __syncthreads();
        if (lane == 0) s_part[wv] = c0;
        __syncthreads();
        int first = 0;
        for (int w2 = 0; w2 < 16; ++w2) first += s_part[w2];
        const int mine = (ns + ICP_CH - 1) / ICP_CH;
        for (int q = tid; q < mine; q += 1024) P.chunk_cl[first + q] = k;
    }
}

// grid (cluster, slices): the cluster's masked targets, in cell order, into the coordinate pool (one gather per frame
// instead of one per search: k_icp_nn stages them with coalesced loads)
__global__ __launch_bounds__(256) void k_icp_pool(IcpLarge P, int nf) {
    const int c = blockIdx.x, tb = P.tbase[c];
    if (tb < 0) return;
    const int nt = P.tcount[c];
    const int* tidx = P.tidx + (size_t)c * nf;
    const float* bx6 = P.box + 6 * c;
    const double oc0 = 0.5 * ((double)bx6[0] + (double)bx6[3]), oc1 = 0.5 * ((double)bx6[1] + (double)bx6[4]), oc2 = 0.5 * ((double)bx6[2] + (double)bx6[5]);
    float lm = 0.f;
    for (int t = blockIdx.y * 256 + threadIdx.x; t < nt; t += gridDim.y * 256) {
        const int j = tidx[t];
        const double x = P.frame[3 * (size_t)j], y = P.frame[3 * (size_t)j + 1], z = P.frame[3 * (size_t)j + 2];
        P.tcx[tb + t] = x; P.tcy[tb + t] = y; P.tcz[tb + t] = z;
        const float fx = (float)(x - oc0), fy = (float)(y - oc1), fz = (float)(z - oc2);
        P.tcf[tb + t] = make_float4(fx, fy, fz, __int_as_float(j));
        lm = fmaxf(lm, fmaxf(fabsf(fx), fmaxf(fabsf(fy), fabsf(fz))));
    }
    lm = -wave_min_fast(-lm);
    if ((threadIdx.x & 63) == 0 && lm > 0.f) atomicMax(P.tlmax + c, __float_as_uint(lm));      // (non-negative floats order like their bits)
}

// Before the first iteration of a frame no source has a previous match, and a search without one scans its cluster's whole grid
// (one launch of 350-1100 us at configs[4]).  This gives every source SOME target to start from: the nearest one of its own grid
// cell, or of the nearest ring of cells (up to 3) that holds any.  Any target of the list is a valid start -- k_icp_nn's result is the
// lexicographic minimum over a square that contains the nearest target, whatever it starts from -- and a near one makes that square small.
__global__ __launch_bounds__(256) void k_icp_seed(IcpLarge P, int nf) {
    const int c = blockIdx.x;
    const double* st = P.state + ICP_ST * c;
    const double x0a = st[36], inv_a = st[37], x0b = st[42], inv_b = st[43];
    const int axa = (int)st[38], axb = (int)st[44], gd = (int)st[45];
    const int* tst = P.tst + (size_t)c * (ICP_NCELL + 1);
    const int* tidx = P.tidx + (size_t)c * nf;
    const int tb = P.tbase[c];
    for (int i = P.off[c] + blockIdx.y * 256 + threadIdx.x; i < P.off[c + 1]; i += gridDim.y * 256) {
        const double s[3] = {P.srcw[3 * (size_t)i], P.srcw[3 * (size_t)i + 1], P.srcw[3 * (size_t)i + 2]};
        const int ia = icp_bin(s[axa], x0a, inv_a, gd), ib = icp_bin(s[axb], x0b, inv_b, gd);
        // (round 4: the NEAREST target of the first ring that holds any, not the first one listed -- a cell is a column along the
        //  box's shortest edge, and its first entry can be the cluster's whole thickness away: the first search of a frame scanned
        //  1.1 ms worth of cells around such seeds)
        int t = -1;
        double td = INFINITY;
        for (int e = 0; e <= 3 && t < 0; ++e)
            for (int r = max(ia - e, 0); r <= min(ia + e, gd - 1); ++r) {
                const int q0 = tst[r * gd + max(ib - e, 0)], q1 = tst[r * gd + min(ib + e, gd - 1) + 1];
                for (int q = q0; q < q1; ++q) {
                    double x, y, z;
                    if (tb >= 0) { x = P.tcx[tb + q]; y = P.tcy[tb + q]; z = P.tcz[tb + q]; }
                    else { const size_t j = (size_t)tidx[q]; x = P.frame[3 * j]; y = P.frame[3 * j + 1]; z = P.frame[3 * j + 2]; }
                    const double dx = s[0] - x, dy = s[1] - y, dz = s[2] - z;
                    const double d = (dx * dx + dy * dy) + dz * dz;
                    if (d < td) { td = d; t = q; }
                }
            }
        if (t >= 0) {
            const int j = tidx[t];
            P.nn[i] = j;
            if (tb >= 0) { P.prevt[3 * (size_t)i] = P.tcx[tb + t]; P.prevt[3 * (size_t)i + 1] = P.tcy[tb + t]; P.prevt[3 * (size_t)i + 2] = P.tcz[tb + t]; }
            else { P.prevt[3 * (size_t)i] = P.frame[3 * (size_t)j]; P.prevt[3 * (size_t)i + 1] = P.frame[3 * (size_t)j + 1]; P.prevt[3 * (size_t)i + 2] = P.frame[3 * (size_t)j + 2]; }
        }
    }
}

// One workgroup: the chunks of the clusters that have not converged, in order, into P.live; their number behind the
// "clusters still running" word the host reads between batches (the next batch of searches launches only those chunks:
// in the tail of a frame a few clusters iterate on, and the thousands of workgroups that would only find "done" and leave
// delayed the live ones by 10-40 us per search).
__global__ __launch_bounds__(1024) void k_icp_compact(IcpLarge P, int k_total) {
    __shared__ int wsum[16];
    __shared__ int s_run;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) s_run = 0;
    __syncthreads();
    for (int c0 = 0; c0 < k_total; c0 += 1024) {
        const int c = c0 + tid;
        const bool lv = c < k_total && P.state[ICP_ST * c + 34] == 0.0;
        const int nch = lv ? P.chunk0[c + 1] - P.chunk0[c] : 0;
        int inc = nch;
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
        if (lane == 63) wsum[wv] = inc;
        __syncthreads();
        int base = s_run + inc - nch, total = 0;
        for (int w = 0; w < 16; ++w) { const int t = wsum[w]; base += w < wv ? t : 0; total += t; }
        for (int q = 0; q < nch; ++q) P.live[base + q] = P.chunk0[c] + q;
        __syncthreads();
        if (tid == 0) s_run += total;
        __syncthreads();
    }
    if (tid == 0) P.running[1] = s_run;
}

__device__ __forceinline__ double ld_agent_f64(const double* p) {
    return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_agent_f64(double* p, double v) {
    __hip_atomic_store((unsigned long long*)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One wave, run by the LAST chunk of cluster k to finish its search (k_icp_nn): moments of the chunks in chunk order ->
// fitness / RMSE, open3d's convergence test against the previous ones, else Horn's closed form: new pose, pending update.
// The other chunks' moments were stored with agent-scope stores and are read with agent-scope loads (no fences:
// MI355X_MICROARCH.md, "sc1 stores and loads both sides"); everything else the wave reads was written by earlier launches.
__device__ void icp_fit_cluster(const IcpLarge& P, int k, int max_iter, int lane) {
    double* st = P.state + ICP_ST * k;
    const int ns = P.off[k + 1] - P.off[k];
    double v = 0.0;
    if (lane < ICP_NM) {
        const int c0 = P.chunk0[k], c1 = P.chunk0[k + 1];
        for (int ch = c0; ch < c1; ch += 8) {                         // chunk order; a group's loads are issued together
            double pv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) pv[u] = ld_agent_f64(P.part + (size_t)min(ch + u, c1 - 1) * ICP_NM + lane);
#pragma unroll
            for (int u = 0; u < 8; ++u) if (ch + u < c1) v += pv[u];
        }
    }
    double cm[ICP_NM];
#pragma unroll
    for (int a = 0; a < ICP_NM; ++a) cm[a] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), a), __builtin_amdgcn_readlane(__double2loint(v), a));
    if (lane != 0) return;
    // fitness / inlier RMSE of the correspondences the last k_icp_nn produced (registration_icp's GetRegistrationResult)
    const double ncorr = cm[0];
    const double fit = ns > 0 ? cm[0] / (double)ns : 0.0, rmse = cm[0] > 0 ? sqrt(cm[1] / cm[0]) : 0.0;
    const double pf = st[32], pr = st[33];
    const int updates = (int)st[35];
    const bool converged = updates >= 1 && fabs(pf - fit) < 1e-6 && fabs(pr - rmse) < 1e-6;
    if (converged || updates >= max_iter) { st[34] = 1.0; atomicSub(P.running, 1); return; }
    double U[16], T[16], Vp[16];
    for (int i = 0; i < 16; ++i) { U[i] = (i % 5 == 0) ? 1.0 : 0.0; Vp[i] = U[i]; T[i] = st[i]; }
    if (ncorr > 0) {
        const double shc[3] = {st[39], st[40], st[41]};
        double ms[3], md[3], msr[3], mdr[3];
        for (int a = 0; a < 3; ++a) { msr[a] = cm[2 + a] / ncorr; mdr[a] = cm[5 + a] / ncorr; ms[a] = shc[a] + msr[a]; md[a] = shc[a] + mdr[a]; }
        double C[9];
        for (int a = 0; a < 3; ++a) for (int c = 0; c < 3; ++c) C[3 * a + c] = fma(-ncorr * msr[a], mdr[c], cm[8 + 3 * a + c]);
        const double Sxx = C[0], Sxy = C[1], Sxz = C[2], Syx = C[3], Syy = C[4], Syz = C[5], Szx = C[6], Szy = C[7], Szz = C[8];
        double N[4][4] = {{Sxx + Syy + Szz, Syz - Szy, Szx - Sxz, Sxy - Syx},
                          {Syz - Szy, Sxx - Syy - Szz, Sxy + Syx, Szx + Sxz},
                          {Szx - Sxz, Sxy + Syx, -Sxx + Syy - Szz, Syz + Szy},
                          {Sxy - Syx, Szx + Sxz, Syz + Szy, -Sxx - Syy + Szz}};
        double q[4], R[9];
        horn_max_eigvec(N, q, Vp);
        quat_to_matrix(q, R);
        for (int a = 0; a < 3; ++a) {
            U[4 * a] = R[3 * a]; U[4 * a + 1] = R[3 * a + 1]; U[4 * a + 2] = R[3 * a + 2];
            U[4 * a + 3] = md[a] - (R[3 * a] * ms[0] + R[3 * a + 1] * ms[1] + R[3 * a + 2] * ms[2]);
        }
        st[39] = md[0]; st[40] = md[1]; st[41] = md[2];              // the matched sources' centroid lands here
    }
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) {
        double s = 0;
        for (int m = 0; m < 4; ++m) s = fma(U[4 * r + m], T[4 * m + c], s);
        st[4 * r + c] = s;
    }
    for (int i = 0; i < 16; ++i) st[16 + i] = U[i];                   // k_icp_nn applies it to the sources
    st[32] = fit; st[33] = rmse; st[35] = (double)(updates + 1);
}

// Block = one chunk of ICP_CH = 64 sorted sources of ONE cluster (chunk0 maps blocks to clusters; blocks past the last
// chunk exit), 256 threads: the tail of a frame's ICP is a few clusters iterating on, and small chunks put their work on
// many CUs (1024-thread chunks of 256 sources: 59 us per search whatever the number of live clusters).  A wave owns 16 consecutive sorted sources -- about two cells' worth -- and four lanes share one
// source: lane group g scans grid row ra0 + g of the cell rectangle [ra0, ra1] x [cb0, cb1] that the wave's sources'
// (x - r, x + r) squares touch, r = distance to the previous match.  A row of the rectangle is one contiguous run of the
// cell-sorted target list; the runs are staged 16 targets per row at a time in the wave's own LDS slice (no block barrier
// in the scan), entries past a run's end as far-away points.
#ifdef CREG_ICP_BLK
// debug build only (tests/measure/icp_tail_blocks.py): per workgroup of the launches with at most 4 clusters iterating -- [0] start
// [1] end of the search (10 ns wall ticks) [2] entries per lane group scanned by wave 0 [3] rows << 16 | columns of wave 0's
// rectangle [4] cluster [5] launch grid
__device__ unsigned long long g_icp_blk[6][8192];
#endif
#ifndef ICP_NN_WPS
#define ICP_NN_WPS 3                       // minimum waves per SIMD the register allocation of k_icp_nn is held to
#endif
// Always-on work counters of the search (round 6: the roofline of the configs[4] frame names THIS kernel): [0] waves that searched,
// [1] float32 screen trips (a trip = 8 staged targets x the wave's 64 lanes = 512 pair evaluations), [2] trips that went on to the fp64
// evaluation, [3] live sources (source-iterations), [4] tie rescans.  Wave-uniform tallies in SGPRs, one atomic per counter and wave.
// 256 shards (blockIdx & 255): ~2 800 waves of a launch adding to ONE word serialise in the L2 -- 88 atomics per microsecond and word, which
// doubled the launch (50 -> 116 us, measured) when the tallies were first added unsharded.
constexpr int ICP_CT_SHARDS = 256;
__device__ unsigned long long g_icp_nn_ct[ICP_CT_SHARDS][8];
__global__ __launch_bounds__(64 * ICP_NNW, ICP_NN_WPS) void k_icp_nn(IcpLarge P, int n, int k_total, int nf, double th2, int max_iter) {
    constexpr int SB = ICP_SB, SR = ICP_SB + 2;                       // staged targets per lane group and batch; slice stride
                                                                      // (+2: the four groups' equal slots fall into different LDS banks)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* tx = (double*)smem;                                       // [waves][4 rows][SR] x | y | z, then the frame indices
    double* ty = tx + ICP_NNW * ICP_G * SR; double* tz = ty + ICP_NNW * ICP_G * SR;
    int* tj = (int*)(tz + ICP_NNW * ICP_G * SR);
    float* txf = (float*)(tj + ICP_NNW * ICP_G * SR);                  // the same targets relative to the wave's origin, float32 (the screen)
    float* tyf = txf + ICP_NNW * ICP_G * SR; float* tzf = tyf + ICP_NNW * ICP_G * SR;
    __shared__ double sc[ICP_NNW * ICP_NM];
    const int tid = threadIdx.x, lane = tid & 63;
    // (the host sizes the grid from the live-chunk count it knew one batch of launches ago -- an upper bound, the list only shrinks --
    //  so workgroups past the CURRENT count leave: what lies behind it in `live` are leftovers of longer lists, possibly duplicates)
    if (P.use_live && (int)blockIdx.x >= P.running[1]) return;
    const int blk = P.use_live ? P.live[blockIdx.x] : (int)blockIdx.x;     // the chunk this workgroup takes
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef CREG_STAMPS
    const bool tailonly = *P.running <= 4;                           // stamps of the tail of the frame only: few clusters still iterating
    if (tid == 0 && tailonly) atomicMin(&g_icp_wmin[0], wall_clock64());
    unsigned long long nst = clock64();
#define NN_STAMP(slot) do { const unsigned long long now_ = clock64(); if (tid == 0 && tailonly) atomicAdd(&g_icp_stamps[slot], now_ - nst); nst = now_; } while (0)
#else
#define NN_STAMP(slot) do { } while (0)
#endif
#ifdef CREG_ICP_BLK
    const bool blk_rec = *P.running <= 4 && blockIdx.x < 8192;
    unsigned long long blk_per = 0;
    const unsigned long long blk_t0 = wall_clock64();
#endif
    if (blk >= P.chunk0[k_total]) return;
    const int c = P.chunk_cl[blk];
    const double* st = P.state + ICP_ST * c;
    if (st[34] != 0.0) return;                                        // converged cluster: nothing moves any more
    const int g = lane / ICP_SPW, l16 = lane % ICP_SPW;               // lane group (the grid rows of the rectangle it scans), source in the wave
#ifdef CREG_STAMPS
    if (tid == 0 && tailonly) atomicMin(&g_icp_wmin[1], wall_clock64());
#endif
    const int i = P.off[c] + (blk - P.chunk0[c]) * ICP_CH + wv * ICP_SPW + l16;
    const bool live = i < P.off[c + 1];
    const double x0a = st[36], inv_a = st[37], x0b = st[42], inv_b = st[43];
    const int axa = (int)st[38], axb = (int)st[44], gd = (int)st[45];
    const double shc0 = st[39], shc1 = st[40], shc2 = st[41];
    double s0 = 0, s1 = 0, s2 = 0, alo = INFINITY, ahi = -INFINITY, blo = INFINITY, bhi = -INFINITY;
    double seed = 1e299;                                              // 1e299: below the staged padding's 3e300, above any real squared distance
    double pd2 = 0, pq0 = 0, pq1 = 0, pq2 = 0; int pmv = -2;          // the previous match: its squared distance now, its coordinates, its frame index
    if (live) {
        // the rigid update the last fit left pending (identity before the first one); the four lanes of a source agree,
        // lane group 0 stores
        const double p0 = P.srcw[3 * (size_t)i], p1 = P.srcw[3 * (size_t)i + 1], p2 = P.srcw[3 * (size_t)i + 2];
        const int pm = P.nn[i];                                       // -1 before the first search (k_icp_init)
        const double q0 = P.prevt[3 * (size_t)i], q1 = P.prevt[3 * (size_t)i + 1], q2 = P.prevt[3 * (size_t)i + 2];
        s0 = fma(st[16 + 2], p2, fma(st[16 + 1], p1, st[16] * p0)) + st[16 + 3];
        s1 = fma(st[16 + 6], p2, fma(st[16 + 5], p1, st[16 + 4] * p0)) + st[16 + 7];
        s2 = fma(st[16 + 10], p2, fma(st[16 + 9], p1, st[16 + 8] * p0)) + st[16 + 11];
        alo = -INFINITY; ahi = INFINITY; blo = -INFINITY; bhi = INFINITY;
        if (pm >= 0) {
            const double dx = s0 - q0, dy = s1 - q1, dz = s2 - q2;
            const double sa = axa == 0 ? s0 : (axa == 1 ? s1 : s2), sb = axb == 0 ? s0 : (axb == 1 ? s1 : s2);
            const double d2p = (dx * dx + dy * dy) + dz * dz;
            const double r0 = (d2p > 1e-280 ? d2p * fast_rsqrt(d2p) : 1e-140) * (1.0 + 1e-12) + 1e-300;
            const double ra = r0 + 4e-16 * fabs(sa), rb = r0 + 4e-16 * fabs(sb);
            alo = sa - ra; ahi = sa + ra; blo = sb - rb; bhi = sb + rb;
            // the previous match lies inside the rectangle and will be scanned with exactly this value (same operands, same
            // association): starting `best` a hair above it changes no result -- the match itself still passes the strict `<` --
            // and lets the float32 screen below reject almost every other target from the first trip on
            if (P.screen) seed = d2p * (1.0 + 0x1p-40) + 1e-300;
            pd2 = d2p; pq0 = q0; pq1 = q1; pq2 = q2; pmv = pm;
        }
    }
    NN_STAMP(8);
    if (live && g == 0) { P.srcw[3 * (size_t)i] = s0; P.srcw[3 * (size_t)i + 1] = s1; P.srcw[3 * (size_t)i + 2] = s2; }
    const bool any = __ballot(live) != 0;
    const int ra0 = __builtin_amdgcn_readfirstlane(icp_bin(wave_min_fast(alo), x0a, inv_a, gd));
    const int ra1 = __builtin_amdgcn_readfirstlane(icp_bin(wave_max_fast(ahi), x0a, inv_a, gd));
    const int cb0 = __builtin_amdgcn_readfirstlane(icp_bin(wave_min_fast(blo), x0b, inv_b, gd));
    const int cb1 = __builtin_amdgcn_readfirstlane(icp_bin(wave_max_fast(bhi), x0b, inv_b, gd));
    const int* tst = P.tst + (size_t)c * (ICP_NCELL + 1);
    const int* tidx = P.tidx + (size_t)c * nf;
    const int nrows_all = any ? ra1 - ra0 + 1 : 0;
#ifdef CREG_STAMPS
    const int nrows = nrows_all;
    if (lane == 0 && any && tailonly) { atomicAdd(&g_icp_stamps[1], 1ull); atomicAdd(&g_icp_stamps[3], (unsigned long long)P.tcount[c]); atomicAdd(&g_icp_stamps[5], (unsigned long long)nrows); atomicAdd(&g_icp_stamps[6], (unsigned long long)(cb1 - cb0 + 1)); }
    const unsigned long long stt = clock64();
#endif
    double* px = tx + (wv * ICP_G + g) * SR; double* py = ty + (wv * ICP_G + g) * SR; double* pz = tz + (wv * ICP_G + g) * SR;
    int* pj = tj + (wv * ICP_G + g) * SR;
    float* pxf = txf + (wv * ICP_G + g) * SR; float* pyf = tyf + (wv * ICP_G + g) * SR; float* pzf = tzf + (wv * ICP_G + g) * SR;
    // ---- float32 screen ----------------------------------------------------------------------------------------------
    // Coordinates relative to a wave origin o (the first live source), rounded to float32; d_f = the float32 squared distance.
    // With u = 2^-24, L >= every |coordinate - o| (the mask box holds the targets, the sources are measured) and D the
    // fp64 squared distance:  d_f <= D + 8 u (L sqrt(D) + D) + 13 u^2 L^2   (input rounding 2 u L per difference, three
    // roundings of the sum).  So D <= best implies d_f <= thr = best + 2^-20 (L sqrt(best) + best) + 2^-43 L^2 (twice the
    // bound), and a trip whose smallest d_f exceeds thr (rounded up to float32) in EVERY lane cannot change `best`, the slot
    // or the tie flag: it is skipped; every other trip runs the fp64 code below unchanged.  The screen only ever skips work.
    const int flive = __builtin_ctzll(__ballot(live) | (1ull << 63));
    const double o0 = __shfl(s0, flive, 64), o1 = __shfl(s1, flive, 64), o2 = __shfl(s2, flive, 64);
    const float sf0 = (float)(s0 - o0), sf1 = (float)(s1 - o1), sf2 = (float)(s2 - o2);
    double Lb = 0.0;
    {
        const float* bx6 = P.box + 6 * c;
        const double od[3] = {o0, o1, o2};
        for (int d = 0; d < 3; ++d) Lb = fmax(Lb, fmax(fabs((double)bx6[d] - od[d]), fabs((double)bx6[3 + d] - od[d])));
        const double ls = live ? fmax(fabs(s0 - o0), fmax(fabs(s1 - o1), fabs(s2 - o2))) : 0.0;
        Lb = fmax(Lb, wave_max_fast(ls)) * (1.0 + 0x1p-20) + 1e-30;   // (box corners are float32: slack for their rounding)
    }
    auto screen_thr = [&](double b) -> float {
        if (!(b < 1e290)) return INFINITY;
        const double rt = (double)sqrtf((float)b) * (1.0 + 0x1p-20);
        return (float)((b + 0x1p-20 * (Lb * rt + b) + 0x1p-43 * Lb * Lb) * (1.0 + 0x1p-22));
    };
    const int tb = P.tbase[c];                                        // >= 0: coordinates in the pool, in the order of tidx
    // Lane group g walks the runs of grid rows rb + g, + 4, + 8, + 12 of a band of <= 16 rows as ONE sequence of L entries
    // (entry p -> run and offset by three compares), so the scan is a flat loop over batches of SB entries and the four groups
    // carry nearly equal loads whatever the number of rows.  A rectangle of more than 16 rows (the first search of a large
    // cluster: the whole grid) takes several bands.
    int rs0 = 0, rs1 = 0, rs2 = 0, rs3 = 0, c1 = 0, c2 = 0, c3 = 0, L = 0, per = 0;
    auto band = [&](int rb) {
        const int nrows = min(16, ra1 - rb + 1);
        int q0v = 0, q1v = 0;                         // lane l < rows: the run of grid row rb + l
        if (lane < nrows) { q0v = tst[(rb + lane) * gd + cb0]; q1v = tst[(rb + lane) * gd + cb1 + 1]; }
        const int r0 = g, r1 = g + ICP_G, r2 = g + 2 * ICP_G, r3 = g + 3 * ICP_G;   // this group's rows of the band (those < 16)
        const int a0 = __shfl(q0v, r0 & 63, 64), e0 = __shfl(q1v, r0 & 63, 64), a1 = __shfl(q0v, r1 & 63, 64), e1 = __shfl(q1v, r1 & 63, 64);
        const int a2 = __shfl(q0v, r2 & 63, 64), e2 = __shfl(q1v, r2 & 63, 64), a3 = __shfl(q0v, r3 & 63, 64), e3 = __shfl(q1v, r3 & 63, 64);
        rs0 = a0; rs1 = a1; rs2 = a2; rs3 = a3;
        c1 = r0 < nrows ? e0 - a0 : 0;
        c2 = c1 + (r1 < nrows ? e1 - a1 : 0);
        c3 = c2 + (r2 < nrows ? e2 - a2 : 0);
        L = c3 + (r3 < nrows ? e3 - a3 : 0);
        int pr = L;
        for (int o = ICP_SPW; o < 64; o <<= 1) pr = max(pr, __shfl_xor(pr, o, 64));
        per = __builtin_amdgcn_readfirstlane(pr);                     // the longest of the four sequences
#ifdef CREG_ICP_BLK
        blk_per += per;
#endif
#ifdef CREG_STAMPS
        if (lane == 0 && tailonly) { atomicAdd(&g_icp_stamps[0], (unsigned long long)per); atomicAdd(&g_icp_stamps[4], 1ull); }
#endif
    };
    constexpr int EPL = SB / ICP_SPW;                                 // staged entries per lane and batch
    int ct_f32 = 0, ct_f64 = 0, ct_tie = 0;                           // (wave-uniform: the trips' branches are ballots)
    double best = seed; int bslot = -1, bj = 0x7fffffff;
    double bx = 0, by = 0, bz = 0;                                    // the best target's coordinates
    bool tief = false;
    NN_STAMP(9);
    if (P.screen && tb >= 0) {
        // ---- pool path: only what the screen needs is staged ---------------------------------------------------------------
        // The pool holds every masked target of the cluster as (float32 coordinates relative to the box centre oc, frame index):
        // one 16-byte load per entry, 16 bytes of LDS, and TWO batches of loads in flight (16 VGPRs a batch instead of 28).
        // The fp64 coordinates stay in the pool: the rare trip that passes the screen reads its eight entries from there (L2).
        // Same bound as above with o = oc and L from the cluster's own extent (k_icp_pool) and the wave's sources.
        const float* bxc = P.box + 6 * c;
        const double oc0 = 0.5 * ((double)bxc[0] + (double)bxc[3]), oc1 = 0.5 * ((double)bxc[1] + (double)bxc[4]), oc2 = 0.5 * ((double)bxc[2] + (double)bxc[5]);
        const float cf0 = (float)(s0 - oc0), cf1 = (float)(s1 - oc1), cf2 = (float)(s2 - oc2);
        const double lsrc = live ? fmax(fabs(s0 - oc0), fmax(fabs(s1 - oc1), fabs(s2 - oc2))) : 0.0;
        const double Lc = fmax((double)__uint_as_float(P.tlmax[c]), wave_max_fast(lsrc)) * (1.0 + 0x1p-20) + 1e-30;
        auto thr_of = [&](double b) -> float {
            if (!(b < 1e290)) return INFINITY;
            const double rt = (double)sqrtf((float)b) * (1.0 + 0x1p-20);
            return (float)((b + 0x1p-20 * (Lc * rt + b) + 0x1p-43 * Lc * Lc) * (1.0 + 0x1p-22));
        };
        // The previous match is a target of this very list, and its squared distance from the moved source is pd2 exactly as the scan
        // would compute it: the search STARTS from it (best = pd2, its index and coordinates) and leaves that one entry out (the
        // screen would let it through in every scan: a trip's fp64 operands come from L2, ~1.5 us).  Any other target at exactly
        // pd2 passes the screen and raises the tie flag as before; the rescan of pass 1 starts from nothing.
        if (pmv >= 0) { best = pd2; bj = pmv; bx = pq0; by = pq1; bz = pq2; bslot = 0; }
        float thr = thr_of(best);
        const float4* tcf = P.tcf + tb;
        const double* gx = P.tcx + tb; const double* gy = P.tcy + tb; const double* gz = P.tcz + tb;
        auto seqpos = [&](int pq) { return pq < c1 ? rs0 + pq : (pq < c2 ? rs1 + (pq - c1) : (pq < c3 ? rs2 + (pq - c2) : rs3 + (pq - c3))); };
        float4 v2[2][EPL];
        auto fetchp = [&](auto SET, int t0) {
#pragma unroll
            for (int u = 0; u < EPL; ++u) {
                const int pq = t0 + ICP_SPW * u + l16;
                const bool in = pq < L;
                const float4 v = tcf[in ? seqpos(pq) : 0];
                v2[decltype(SET)::value][u] = in ? v : make_float4(1e30f, 1e30f, 1e30f, __int_as_float(0x7fffffff));
            }
        };
        auto publishp = [&](auto SET) {
#pragma unroll
            for (int u = 0; u < EPL; ++u) {
                const int e = ICP_SPW * u + l16;
                const float4 v = v2[decltype(SET)::value][u];
                pxf[e] = v.x; pyf[e] = v.y; pzf[e] = v.z; pj[e] = __float_as_int(v.w);
            }
        };
        for (int pass = 0; pass < 2; ++pass) {                        // pass 1 only after a tie was seen: frame-index tie-break
          for (int rb = ra0; rb < ra0 + nrows_all; rb += 16) {
            band(rb);
            using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
            if (per > 0) fetchp(S0{}, 0);
            if (per > SB) fetchp(S1{}, SB);
            auto scan_batch = [&](int t0) {
                const int cnt = min(SB, per - t0);
                int bm = -1;
                if (pass == 0) {
                    for (int t = 0; t < cnt; t += 8) {
                        float mf = INFINITY, ef[8];
#pragma unroll
                        for (int u = 0; u < 8; u += 2) {
                            const nn_f2 X = *(const nn_f2*)(pxf + t + u), Y = *(const nn_f2*)(pyf + t + u), Z = *(const nn_f2*)(pzf + t + u);
                            const nn_f2 fx = nn_f2{cf0, cf0} - X, fy = nn_f2{cf1, cf1} - Y, fz = nn_f2{cf2, cf2} - Z;
                            const nn_f2 e = __builtin_elementwise_fma(fz, fz, __builtin_elementwise_fma(fy, fy, fx * fx));
                            ef[u] = e.x; ef[u + 1] = e.y;
                            mf = fminf(mf, fminf(e.x, e.y));
                        }
                        ++ct_f32;
                        if (!__ballot(mf <= thr)) continue;
                        // Round 6: only the entries that PASSED the screen are evaluated in fp64 -- a lane's candidates of a trip (usually one), two
                        // at a time, their pool coordinates requested together and kept: ONE dependent L2 round trip per passing trip where there were
                        // three (operands of entries 0-3, of 4-7, then the winner's coordinates again).  An entry the screen rejects has D > best, so it
                        // can neither win nor tie: the minimum, the tie flag and the slot are what the eight-entry evaluation gave.
                        unsigned cm = 0;                                  // bit u: entry u passed this lane's screen and is not its previous match
#pragma unroll
                        for (int u = 0; u < 8; ++u) cm |= (ef[u] <= thr && pj[t + u] != pmv) ? 1u << u : 0u;
                        if (!__ballot(cm != 0)) continue;                 // only previous matches came through
                        ++ct_f64;
                        bool any_lt = false;
                        while (__ballot(cm != 0)) {                        // four candidates per round: one round unless a lane has more
                            int uu[4]; double X[4], Y[4], Z[4], dd[4];
#pragma unroll
                            for (int c4 = 0; c4 < 4; ++c4) {
                                uu[c4] = cm ? __builtin_ctz(cm) : -1; cm &= cm - 1;
                                const int pq = t0 + t + max(uu[c4], 0);
                                const bool in = uu[c4] >= 0 && pq < L;
                                const int tp = in ? seqpos(pq) : 0;
                                X[c4] = gx[tp]; Y[c4] = gy[tp]; Z[c4] = gz[tp];
                                if (!in) uu[c4] = -1;
                            }
#pragma unroll
                            for (int c4 = 0; c4 < 4; ++c4) {
                                const double dx = s0 - X[c4], dy = s1 - Y[c4], dz = s2 - Z[c4];
                                dd[c4] = uu[c4] >= 0 ? (dx * dx + dy * dy) + dz * dz : 1e300;
                            }
                            // the earliest of the smallest, by selects (an index into the register arrays would send them to scratch)
                            const bool s01 = dd[1] < dd[0], s23 = dd[3] < dd[2];
                            const double m01 = s01 ? dd[1] : dd[0], m23 = s23 ? dd[3] : dd[2];
                            const int u01 = s01 ? uu[1] : uu[0], u23 = s23 ? uu[3] : uu[2];
                            const double x01 = s01 ? X[1] : X[0], y01 = s01 ? Y[1] : Y[0], z01 = s01 ? Z[1] : Z[0];
                            const double x23 = s23 ? X[3] : X[2], y23 = s23 ? Y[3] : Y[2], z23 = s23 ? Z[3] : Z[2];
                            const bool sh = m23 < m01;
                            const double m = sh ? m23 : m01;
                            const int eq = (dd[0] == m) + (dd[1] == m) + (dd[2] == m) + (dd[3] == m);
                            const bool lt = m < best;
                            tief |= m == best || (lt && eq > 1);
                            bm = lt ? t + (sh ? u23 : u01) : bm;
                            bx = lt ? (sh ? x23 : x01) : bx; by = lt ? (sh ? y23 : y01) : by; bz = lt ? (sh ? z23 : z01) : bz;
                            best = lt ? m : best;
                            any_lt |= lt;
                        }
                        if (__ballot(any_lt)) thr = thr_of(best);
                    }
                    if (bm >= 0) { bj = pj[bm]; bslot = bm; }
                } else {
                    for (int t = 0; t < cnt; ++t) {
                        const int pq = t0 + t;
                        const bool in = pq < L;
                        const int tp = in ? seqpos(pq) : 0;
                        const double x = in ? gx[tp] : 1e150, y = in ? gy[tp] : 1e150, z = in ? gz[tp] : 1e150;
                        const double dx = s0 - x, dy = s1 - y, dz = s2 - z;
                        const double d2 = (dx * dx + dy * dy) + dz * dz;
                        const int j = pj[t];
                        if (d2 < best || (d2 == best && j < bj)) { best = d2; bj = j; bx = x; by = y; bz = z; bslot = t; }
                    }
                }
                __builtin_amdgcn_wave_barrier();
            };
            for (int t0 = 0; t0 < per; t0 += 2 * SB) {
                publishp(S0{});
                __builtin_amdgcn_wave_barrier();
                if (t0 + 2 * SB < per) fetchp(S0{}, t0 + 2 * SB);
                scan_batch(t0);
                if (t0 + SB < per) {
                    publishp(S1{});
                    __builtin_amdgcn_wave_barrier();
                    if (t0 + 3 * SB < per) fetchp(S1{}, t0 + 3 * SB);
                    scan_batch(t0 + SB);
                }
            }
          }
          if (pass == 0) {
              if (!__ballot(tief)) break;
              ++ct_tie;
              best = 1e299; bslot = -1; bj = 0x7fffffff;
          }
        }
    } else {
    float thr_f = screen_thr(best);
    // staging registers of one batch: the loads of batch b + 1 are in flight while batch b is scanned from LDS
    int jv[EPL];
    double cx[EPL], cy[EPL], cz[EPL];
    auto fetch = [&](int t0) {
#pragma unroll
        for (int u = 0; u < EPL; ++u) {
            const int pq = t0 + ICP_SPW * u + l16;                         // entry of the group's sequence
            const int t = pq < c1 ? rs0 + pq : (pq < c2 ? rs1 + (pq - c1) : (pq < c3 ? rs2 + (pq - c2) : rs3 + (pq - c3)));
            const bool in = pq < L;
            const int j = in ? tidx[t] : -1;
            jv[u] = j;
            if (tb >= 0) {                            // block-uniform: consecutive lanes read consecutive pool entries
                const int tt = in ? t : 0;
                cx[u] = P.tcx[tb + tt]; cy[u] = P.tcy[tb + tt]; cz[u] = P.tcz[tb + tt];
            }
        }
        if (tb < 0) {                                 // the pool was full: gather through the frame indices
#pragma unroll
            for (int u = 0; u < EPL; ++u) {
                const size_t jj = (size_t)max(jv[u], 0);
                cx[u] = P.frame[3 * jj]; cy[u] = P.frame[3 * jj + 1]; cz[u] = P.frame[3 * jj + 2];
            }
        }
    };
    auto publish = [&]() {                            // registers -> the lane group's LDS slice; entries past the sequence as far-away points
#pragma unroll
        for (int u = 0; u < EPL; ++u) {
            const int e = ICP_SPW * u + l16;
            const bool in = jv[u] >= 0;
            px[e] = in ? cx[u] : 1e150; py[e] = in ? cy[u] : 1e150; pz[e] = in ? cz[u] : 1e150; pj[e] = in ? jv[u] : 0x7fffffff;
            pxf[e] = in ? (float)(cx[u] - o0) : 1e30f; pyf[e] = in ? (float)(cy[u] - o1) : 1e30f; pzf[e] = in ? (float)(cz[u] - o2) : 1e30f;
        }
    };
    for (int pass = 0; pass < 2; ++pass) {                            // pass 1 only after a tie was seen: frame-index tie-break
      for (int rb = ra0; rb < ra0 + nrows_all; rb += 16) {
        band(rb);
        if (per > 0) fetch(0);
        for (int t0 = 0; t0 < per; t0 += SB) {
            publish();
            __builtin_amdgcn_wave_barrier();
            if (t0 + SB < per) fetch(t0 + SB);
            NN_STAMP(10);
            const int cnt = min(SB, per - t0);                        // uniform; rounded up to a multiple of 8 (padding is harmless)
            int bm = -1;                                              // slot of this batch's improvement
            if (pass == 0) {
                // Eight entries per step, their distances independent of one another and of `best`; one comparison of the
                // step's minimum with `best`.  (One entry at a time -- compare, select, min against the running best -- is a
                // chain of dependent fp64 operations: 75 ns per entry when a wave has its SIMD to itself, which is the tail
                // of a frame, tests/measure/icp_tail_blocks.py; 60 ns this way.)  bm = the first slot of the batch's minimum
                // if it beats `best` strictly, as before; the tie flag is raised whenever the minimum is attained twice --
                // inside a step, or again in a later step or batch -- which is all the rescan needs.
                for (int t = 0; t < cnt; t += 8) {
                    if (P.screen) {
                        float mf = INFINITY;
#pragma unroll
                        for (int u = 0; u < 8; u += 2) {
                            const nn_f2 X = *(const nn_f2*)(pxf + t + u), Y = *(const nn_f2*)(pyf + t + u), Z = *(const nn_f2*)(pzf + t + u);
                            const nn_f2 fx = nn_f2{sf0, sf0} - X, fy = nn_f2{sf1, sf1} - Y, fz = nn_f2{sf2, sf2} - Z;
                            const nn_f2 e = __builtin_elementwise_fma(fz, fz, __builtin_elementwise_fma(fy, fy, fx * fx));
                            mf = fminf(mf, fminf(e.x, e.y));
                        }
                        ++ct_f32;
                        if (!__ballot(mf <= thr_f)) continue;             // no lane has a target that could matter in this trip
                    }
                    ++ct_f64;
                    double d[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const double dx = s0 - px[t + u], dy = s1 - py[t + u], dz = s2 - pz[t + u];
                        d[u] = (dx * dx + dy * dy) + dz * dz;
                    }
                    const double m = vmin_f64(vmin_f64(vmin_f64(d[0], d[1]), vmin_f64(d[2], d[3])), vmin_f64(vmin_f64(d[4], d[5]), vmin_f64(d[6], d[7])));
                    int first = 7, eq = 0;
#pragma unroll
                    for (int u = 7; u >= 0; --u) { const bool e = d[u] == m; first = e ? u : first; eq += e ? 1 : 0; }
                    const bool lt = m < best;
                    tief |= m == best || (lt && eq > 1);
                    bm = lt ? t + first : bm;
                    best = lt ? m : best;
                    if (P.screen && __ballot(lt)) thr_f = screen_thr(best);
                }
            } else {
                for (int t = 0; t < cnt; ++t) {
                    const double dx = s0 - px[t], dy = s1 - py[t], dz = s2 - pz[t];
                    const double d2 = (dx * dx + dy * dy) + dz * dz;
                    const int j = pj[t];
                    if (d2 < best || (d2 == best && j < bj)) { best = d2; bm = t; bj = j; }
                }
            }
            if (bm >= 0) { bx = px[bm]; by = py[bm]; bz = pz[bm]; bj = pj[bm]; bslot = bm; }
            NN_STAMP(11);
            __builtin_amdgcn_wave_barrier();
        }
      }
        if (pass == 0) {
            if (!__ballot(tief)) break;
            ++ct_tie;
            best = 1e299; bslot = -1; bj = 0x7fffffff;
        }
    }
    }
    if (any) {
        const int nlive = __popcll(__ballot(live && g == 0));
        if (lane == 0) {
            unsigned long long* ct = g_icp_nn_ct[blockIdx.x & (ICP_CT_SHARDS - 1)];
            atomicAdd(ct + 0, 1ull); atomicAdd(ct + 3, (unsigned long long)nlive);
            if (ct_f32) atomicAdd(ct + 1, (unsigned long long)ct_f32);
            if (ct_f64) atomicAdd(ct + 2, (unsigned long long)ct_f64);
            if (ct_tie) atomicAdd(ct + 4, (unsigned long long)ct_tie);
        }
    }
#ifdef CREG_ICP_BLK
    if (tid == 0 && blk_rec) { g_icp_blk[0][blockIdx.x] = blk_t0; g_icp_blk[5][blockIdx.x] = gridDim.x;
                               g_icp_blk[1][blockIdx.x] = wall_clock64(); g_icp_blk[2][blockIdx.x] = blk_per;
                               g_icp_blk[3][blockIdx.x] = ((unsigned long long)nrows_all << 16) | (unsigned)(cb1 - cb0 + 1); g_icp_blk[4][blockIdx.x] = c; }
#endif
    if (bslot < 0) { best = INFINITY; bj = 0x7fffffff; }              // a lane group whose rows were all empty has no candidate
#ifdef CREG_STAMPS
    if (lane == 0 && any && tailonly) { atomicAdd(&g_icp_stamps[7], clock64() - stt); if (tief) atomicAdd(&g_icp_stamps[2], 1ull); }
#endif
    // the four lane groups of a source: (distance, frame index) lexicographic minimum, the winner's coordinates along
    for (int o = ICP_SPW; o < 64; o <<= 1) {
        const double od = __shfl_xor(best, o, 64); const int oj = __shfl_xor(bj, o, 64);
        const double ox = __shfl_xor(bx, o, 64), oy = __shfl_xor(by, o, 64), oz = __shfl_xor(bz, o, 64);
        if (od < best || (od == best && oj < bj)) { best = od; bj = oj; bx = ox; by = oy; bz = oz; }
    }
    NN_STAMP(12);
#ifdef CREG_ICP_RECT_STATS
    {   // measurement build (tests/measure/icp_rect_stats.py): the entries of the cell rectangle this wave SCANNED (from the distances to the
        // previous matches) against those of the rectangle its FINAL nearest distances would have needed -- what a centre-out scan with a
        // shrinking radius could save at most
        double fal = INFINITY, fah = -INFINITY, fbl = INFINITY, fbh = -INFINITY;
        if (live) {
            const double sa = axa == 0 ? s0 : (axa == 1 ? s1 : s2), sb = axb == 0 ? s0 : (axb == 1 ? s1 : s2);
            if (bj != 0x7fffffff) { const double r1 = sqrt(best) * (1.0 + 1e-12) + 1e-300; fal = sa - r1; fah = sa + r1; fbl = sb - r1; fbh = sb + r1; }
            else { fal = -INFINITY; fah = INFINITY; fbl = -INFINITY; fbh = INFINITY; }
        }
        const int fa0 = __builtin_amdgcn_readfirstlane(icp_bin(wave_min_fast(fal), x0a, inv_a, gd)), fa1 = __builtin_amdgcn_readfirstlane(icp_bin(wave_max_fast(fah), x0a, inv_a, gd));
        const int fb0 = __builtin_amdgcn_readfirstlane(icp_bin(wave_min_fast(fbl), x0b, inv_b, gd)), fb1 = __builtin_amdgcn_readfirstlane(icp_bin(wave_max_fast(fbh), x0b, inv_b, gd));
        int e1 = 0, e0 = 0;
        if (any) {
            if (lane <= fa1 - fa0) e1 = tst[(fa0 + lane) * gd + fb1 + 1] - tst[(fa0 + lane) * gd + fb0];
            if (lane <= ra1 - ra0) e0 = tst[(ra0 + lane) * gd + cb1 + 1] - tst[(ra0 + lane) * gd + cb0];
            for (int o = 32; o >= 1; o >>= 1) { e1 += __shfl_xor(e1, o, 64); e0 += __shfl_xor(e0, o, 64); }
            if (lane == 0) {
                unsigned long long* ct = g_icp_nn_ct[blockIdx.x & (ICP_CT_SHARDS - 1)];
                atomicAdd(ct + 5, (unsigned long long)e1); atomicAdd(ct + 6, (unsigned long long)e0);
                atomicAdd(ct + 7, (unsigned long long)((ra1 - ra0 + 1) * (cb1 - cb0 + 1)) | ((unsigned long long)((fa1 - fa0 + 1) * (fb1 - fb0 + 1)) << 32));
            }
        }
    }
#endif
    double cm[ICP_NM];
    for (int a = 0; a < ICP_NM; ++a) cm[a] = 0;
    if (live && g == 0) {
        const bool ok = bj != 0x7fffffff && best < th2;
        P.nn[i] = ok ? bj : -1;
        if (ok) {
            P.prevt[3 * (size_t)i] = bx; P.prevt[3 * (size_t)i + 1] = by; P.prevt[3 * (size_t)i + 2] = bz;
            const double sv[3] = {s0 - shc0, s1 - shc1, s2 - shc2};
            const double dv[3] = {bx - shc0, by - shc1, bz - shc2};
            cm[0] = 1.0; cm[1] = best;
            for (int a = 0; a < 3; ++a) { cm[2 + a] = sv[a]; cm[5 + a] = dv[a]; }
            for (int a = 0; a < 3; ++a) for (int q = 0; q < 3; ++q) cm[8 + 3 * a + q] = sv[a] * dv[q];
        }
    }
    // the chunk's moments: DPP wave sums, then 0 + w0 + w1 + w2 + w3 (fixed association)
#pragma unroll
    for (int a = 0; a < ICP_NM; ++a) cm[a] = wave_sum_fast(cm[a]);
    if (lane == 0) {
#pragma unroll
        for (int a = 0; a < ICP_NM; ++a) sc[wv * ICP_NM + a] = cm[a];
    }
    NN_STAMP(13);
    __syncthreads();
    NN_STAMP(14);
    if (tid < ICP_NM) {
        double r = 0.0;
        for (int w = 0; w < ICP_NNW; ++w) r += sc[w * ICP_NM + tid];
        st_agent_f64(P.part + (size_t)blk * ICP_NM + tid, r);
    }
    // the cluster's last chunk to get here runs the fit.  A bare s_barrier does not wait for the st_agent_f64 stores
    // above: the storing wave drains its vmcnt explicitly before the barrier, so the partial moments are acknowledged by
    // the memory side before thread 0 bumps the arrival counter (P.part is reused by every iteration launch: without the
    // drain the last chunk could read the previous iteration's moments).
    __shared__ int s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const int nch = P.chunk0[c + 1] - P.chunk0[c];
        const int last = atomicAdd(&P.arrive[c], 1) == nch - 1;
        if (last) P.arrive[c] = 0;                                    // for the next launch
        s_last = last;
    }
    __syncthreads();
#ifdef CREG_STAMPS
    if (tid == 0 && tailonly) atomicMax(&g_icp_wmin[2], wall_clock64());
#endif
    if (s_last && wv == 0) icp_fit_cluster(P, c, max_iter, lane);
#ifdef CREG_STAMPS
    if (s_last && tid == 0 && tailonly) atomicMax(&g_icp_wmin[3], wall_clock64());
#endif
#ifdef CREG_STAMPS
    if (tid == 0 && tailonly) atomicAdd(&g_icp_stamps[15], 1ull);
#endif
}

__global__ __launch_bounds__(256) void k_icp_finish(IcpLarge P, int keep_t) {
    __shared__ double T[16];
    const int k = blockIdx.x, tid = threadIdx.x;
    const double* st = P.state + ICP_ST * k;
    if (tid < 16) T[tid] = st[tid];
    __syncthreads();
    if (tid == 0 && keep_t) { T[3] = P.Min[16 * k + 3]; T[7] = P.Min[16 * k + 7]; T[11] = P.Min[16 * k + 11]; }
    __syncthreads();
    if (tid < 16) P.Mout[16 * k + tid] = T[tid];
    if (tid == 0) P.n_iter_out[k] = (int)st[35];
    const int b = P.off[k], e = P.off[k + 1];
    for (int i = b + tid; i < e; i += 256) {
        const double* p = P.local + 3 * (size_t)i;
        for (int a = 0; a < 3; ++a)
            P.world_out[3 * (size_t)i + a] = fma(T[4 * a + 2], p[2], fma(T[4 * a + 1], p[1], T[4 * a] * p[0])) + T[4 * a + 3];
    }
}

struct IcpLargeLayout { size_t srcw, tidx, tcount, box, nn, part, state, running, chunk0, tst, chunk_cl, prevt, tcx, tcy, tcz, tbase, arrive, live, tcf, tlmax, total; int pool_cap; };
static IcpLargeLayout icp_large_layout(int64_t n, int64_t nf, int k) {
    IcpLargeLayout L; size_t o = 0;
    auto take = [&](size_t b) { size_t r = o; o = align_up(o + b, 256); return r; };
    L.srcw = take(sizeof(double) * 3 * n); L.tidx = take(sizeof(int) * (size_t)k * nf); L.tcount = take(sizeof(int) * k);
    L.box = take(sizeof(float) * 6 * k); L.nn = take(sizeof(int) * n);
    L.part = take(sizeof(double) * ICP_NM * (size_t)(n / ICP_CH + k + 1));
    L.state = take(sizeof(double) * ICP_ST * k); L.running = take(sizeof(int) * 4); L.chunk0 = take(sizeof(int) * (k + 1));
    L.tst = take(sizeof(int) * (size_t)k * (ICP_NCELL + 1));
    L.chunk_cl = take(sizeof(int) * (size_t)(n / ICP_CH + k + 1)); L.prevt = take(sizeof(double) * 3 * n);
    // coordinate pool: room for 4 nf masked targets over all clusters (they overlap by the box scale: ~1.5 nf in practice);
    // clusters beyond it fall back to gathering
    const int64_t cap = 4 * nf < (int64_t)k * nf ? 4 * nf : (int64_t)k * nf;
    L.pool_cap = (int)(cap < (1ll << 30) ? cap : (1ll << 30));
    L.tcx = take(sizeof(double) * (size_t)L.pool_cap); L.tcy = take(sizeof(double) * (size_t)L.pool_cap); L.tcz = take(sizeof(double) * (size_t)L.pool_cap);
    L.tbase = take(sizeof(int) * k); L.arrive = take(sizeof(int) * k); L.live = take(sizeof(int) * (size_t)(n / ICP_CH + k + 1));
    L.tcf = take(sizeof(float4) * (size_t)L.pool_cap); L.tlmax = take(sizeof(unsigned) * k); L.total = o;
    return L;
}
// the regime switch (host-side sizes only): average cluster above the LDS source budget, or a frame too large for the
// LDS target table to matter
static bool icp_large_regime(int64_t n, int64_t nf, int k) { return n / (k > 0 ? k : 1) > ICP_SRC_LDS || nf > 65536; }

}  // namespace creg
using namespace creg;

static size_t icp_ws_one(int64_t n, int64_t nf, int k) {
    const size_t a = icp_layout(n, nf, k).total, b = icp_large_layout(n, nf, k).total;
    return a > b ? a : b;
}

extern "C" size_t creg_icp_workspace_bytes(int64_t n, int64_t nf, int32_t k) {
    if (n < 1 || nf < 1 || k < 1) return 0;
    return icp_ws_one(n, nf, k);
}

extern "C" size_t creg_icp_batch_workspace_bytes(int64_t n, int64_t nf, int32_t k, int32_t batch) {
    if (n < 1 || nf < 1 || k < 1 || batch < 1) return 0;
    return icp_ws_one(n, nf, k) * (size_t)batch;
}

// One problem through the multi-launch path.  The host follows the device one batch of iterations behind (no stream synchronisation;
// it returns when the convergence it has read back says so, with the last launches and k_icp_finish still in the queue).
// creg_icp_nn_counters (measurement hook): while `g_icp_nn_timing` is set every k_icp_nn launch is bracketed by two HIP events on the
// launch stream and the call ends with a stream synchronisation that adds their elapsed times to the two host tallies.
static bool g_icp_nn_timing = false;
static double g_icp_nn_us = 0.0;
static long long g_icp_nn_launches = 0;
static int icp_large_run(const creg_icp_problem& q, int64_t n, int32_t k, int64_t nf, double scale, double th,
                         int32_t max_iteration, int32_t keep_translation, char* ws, hipStream_t s) {
    const IcpLargeLayout L = icp_large_layout(n, nf, k);
    IcpLarge P;
    P.local = q.local; P.world = q.world; P.off = q.seg_offsets; P.woff = q.world_offsets; P.frame = q.frame; P.Min = q.M;
    P.Mout = q.M_out; P.world_out = q.world_out; P.n_iter_out = q.n_iter_out;
    P.srcw = (double*)(ws + L.srcw); P.tidx = (int*)(ws + L.tidx); P.tcount = (int*)(ws + L.tcount); P.box = (float*)(ws + L.box);
    P.nn = (int*)(ws + L.nn); P.part = (double*)(ws + L.part); P.state = (double*)(ws + L.state); P.running = (int*)(ws + L.running);
    P.chunk0 = (int*)(ws + L.chunk0); P.tst = (int*)(ws + L.tst); P.chunk_cl = (int*)(ws + L.chunk_cl); P.prevt = (double*)(ws + L.prevt);
    P.tcx = (double*)(ws + L.tcx); P.tcy = (double*)(ws + L.tcy); P.tcz = (double*)(ws + L.tcz); P.tbase = (int*)(ws + L.tbase); P.arrive = (int*)(ws + L.arrive); P.live = (int*)(ws + L.live); P.use_live = 0; P.pool_cap = L.pool_cap;
    P.tcf = (float4*)(ws + L.tcf); P.tlmax = (unsigned*)(ws + L.tlmax);
    P.toff = q.tgt_offsets;                          // point-to-point mode (round 5): the cluster's own target segment, no mask
    const int nn_smem = ICP_NNW * ICP_G * (ICP_SB + 2) * 40;  // k_icp_nn: per wave 4 lane groups x ICP_SB (+2: bank offset) staged targets (x, y, z fp64, frame index, x, y, z float32)
    { static int scr = -1; if (scr < 0) { const char* e = getenv("CREG_ICP_SCREEN"); scr = e ? atoi(e) != 0 : 1; } P.screen = scr; }
    CREG_HIP(hipFuncSetAttribute((const void*)k_icp_nn, hipFuncAttributeMaxDynamicSharedMemorySize, nn_smem));
    hipLaunchKernelGGL(k_icp_mask, dim3(k), dim3(1024), 0, s, P, (int)nf, (float)(0.5 * scale), q.world ? 0 : 1, 1);
    hipLaunchKernelGGL(k_icp_init, dim3(k), dim3(1024), 0, s, P, k);
    hipLaunchKernelGGL(k_icp_pool, dim3(k, 8), dim3(256), 0, s, P, (int)nf);
    if (P.screen) hipLaunchKernelGGL(k_icp_seed, dim3(k, 8), dim3(256), 0, s, P, (int)nf);
    const int nblk = cdiv(n, ICP_CH) + k;                // an upper bound of sum_c ceil(ns_c / ICP_CH); the surplus blocks exit
    // every launch is one search + (last chunk of each cluster) one fit = one convergence test and, unless converged, one
    // update; max_iteration updates need one launch more.  After every batch of 16 launches the live chunks are compacted on the
    // device and the two counters (clusters still iterating, their chunks) come back through pinned memory -- but the host no
    // longer waits for a batch before it enqueues the next one (round 3 synchronised the stream every 16 iterations: the queue ran
    // dry for a host round trip each time): it reads the counters of batch b - 1 after batch b is in the queue, so the device
    // always has a batch ahead; the price is at most one surplus batch of launches whose workgroups all leave at once.
    static thread_local int* h_run = nullptr;                        // pinned: ICP_LAG_RING x {clusters running, live chunks}
    static thread_local hipEvent_t h_last = nullptr;                 // behind the LAST copy into h_run of this thread's previous call
    constexpr int ICP_LAG_RING = 4;
    // (portable: the ring is used from whichever device is current.  A call returns with its last copy still in flight -- the next one
    //  on this thread, possibly on another stream or device, must not reuse the slots before that copy has landed: a late copy would
    //  overwrite the new call's counters, typically with "0 clusters running", and end its iteration loop early)
    if (!h_run) CREG_HIP(hipHostMalloc((void**)&h_run, sizeof(int) * 2 * ICP_LAG_RING, hipHostMallocPortable));
    if (h_last) { (void)hipEventSynchronize(h_last); (void)hipEventDestroy(h_last); h_last = nullptr; }
    hipEvent_t ev[ICP_LAG_RING];
    for (auto& e : ev) CREG_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    std::vector<hipEvent_t> tev;                                      // (timing mode only)
    int grid = nblk, rc_loop = CREG_OK;                              // live chunks as of the last batch whose counters were read
    bool converged = false;
    int64_t done = 0;
    for (int b = 0; !converged && done <= (int64_t)max_iteration; ++b) {
        const int batch = 16;
        for (int q = 0; q < batch; ++q, ++done) {
            if (g_icp_nn_timing) { hipEvent_t e; if (hipEventCreate(&e) == hipSuccess) { (void)hipEventRecord(e, s); tev.push_back(e); } }
            hipLaunchKernelGGL(k_icp_nn, dim3(grid), dim3(64 * ICP_NNW), nn_smem, s, P, (int)n, k, (int)nf, th * th, max_iteration);
            if (g_icp_nn_timing) { hipEvent_t e; if (hipEventCreate(&e) == hipSuccess) { (void)hipEventRecord(e, s); tev.push_back(e); } }
#ifdef CREG_STAMPS
            hipLaunchKernelGGL(k_icp_wall_fold, dim3(1), dim3(1), 0, s);
#endif
        }
        hipLaunchKernelGGL(k_icp_compact, dim3(1), dim3(1024), 0, s, P, k);
        if (hipGetLastError() != hipSuccess) { rc_loop = CREG_EHIP; break; }
        if (hipMemcpyAsync(h_run + 2 * (b % ICP_LAG_RING), P.running, 2 * sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipEventRecord(ev[b % ICP_LAG_RING], s) != hipSuccess) { rc_loop = CREG_EHIP; break; }
        P.use_live = 1;
        if (b >= 1) {                                                // the counters of the batch BEFORE the one just enqueued
            if (hipEventSynchronize(ev[(b - 1) % ICP_LAG_RING]) != hipSuccess) { rc_loop = CREG_EHIP; break; }
            const int* r = h_run + 2 * ((b - 1) % ICP_LAG_RING);
            converged = r[0] <= 0;
            grid = r[1] > 0 ? r[1] : 1;
        }
    }
    if (rc_loop == CREG_OK && !converged) {                          // the last batch's counters (max_iteration reached, or one batch only)
        // nothing to decide any more: k_icp_finish reads the device state
    }
    for (auto& e : ev) (void)hipEventDestroy(e);
    if (hipEventCreateWithFlags(&h_last, hipEventDisableTiming) == hipSuccess) {
        if (hipEventRecord(h_last, s) != hipSuccess) { (void)hipEventDestroy(h_last); h_last = nullptr; (void)hipStreamSynchronize(s); }
    } else { h_last = nullptr; (void)hipStreamSynchronize(s); }
    if (rc_loop != CREG_OK) { set_error("creg_masked_icp: HIP error in the iteration loop: %s", hipGetErrorString(hipGetLastError())); return rc_loop; }
    hipLaunchKernelGGL(k_icp_finish, dim3(k), dim3(256), 0, s, P, keep_translation);
    CREG_LAUNCH_CHECK();
    if (!tev.empty()) {
        (void)hipStreamSynchronize(s);
        for (size_t i = 0; i + 1 < tev.size(); i += 2) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, tev[i], tev[i + 1]) == hipSuccess) { g_icp_nn_us += 1e3 * ms; ++g_icp_nn_launches; }
        }
        for (auto e : tev) (void)hipEventDestroy(e);
    }
    return CREG_OK;
}

#ifdef CREG_ICP_RECT_STATS
extern "C" int creg_debug_icp_rect(double* out4) {      // entries of the final-distance rectangles, of the scanned ones, cells scanned, cells needed (sums over waves)
    static unsigned long long c[creg::ICP_CT_SHARDS][8];
    CREG_HIP(hipMemcpyFromSymbol(c, HIP_SYMBOL(creg::g_icp_nn_ct), sizeof(c)));
    out4[0] = out4[1] = out4[2] = out4[3] = 0.0;
    for (int sh = 0; sh < creg::ICP_CT_SHARDS; ++sh) { out4[0] += (double)c[sh][5]; out4[1] += (double)c[sh][6]; out4[2] += (double)(c[sh][7] & 0xffffffffull); out4[3] += (double)(c[sh][7] >> 32); }
    return CREG_OK;
}
#endif
extern "C" int creg_icp_nn_counters(double* out8, int32_t reset, int32_t timing) {
    if (out8) {
        static unsigned long long c[creg::ICP_CT_SHARDS][8];
        CREG_HIP(hipMemcpyFromSymbol(c, HIP_SYMBOL(creg::g_icp_nn_ct), sizeof(c)));
        for (int i = 0; i < 5; ++i) { double t = 0.0; for (int sh = 0; sh < creg::ICP_CT_SHARDS; ++sh) t += (double)c[sh][i]; out8[i] = t; }
        out8[5] = g_icp_nn_us; out8[6] = (double)g_icp_nn_launches; out8[7] = 0.0;
    }
    if (reset) {
        static const unsigned long long z[creg::ICP_CT_SHARDS][8] = {};
        CREG_HIP(hipMemcpyToSymbol(HIP_SYMBOL(creg::g_icp_nn_ct), z, sizeof(z)));
        g_icp_nn_us = 0.0; g_icp_nn_launches = 0;
    }
    if (timing >= 0) g_icp_nn_timing = timing != 0;
    return CREG_OK;
}

static int icp_launch(const creg_icp_problem* pr, int batch, int64_t n, int32_t k, int64_t nf, double scale, double th,
                      int32_t max_iteration, int32_t keep_translation, void* workspace, size_t workspace_bytes,
                      hipStream_t s, const char* who) {
    CREG_REQUIRE(pr && workspace, "%s: null pointer", who);
    CREG_REQUIRE(batch >= 1 && batch <= ICP_BATCH_MAX, "%s: batch must be in 1..%d", who, ICP_BATCH_MAX);
    CREG_REQUIRE(k >= 1 && nf >= 1 && nf < (1ll << 31) && max_iteration >= 1, "%s: bad size", who);
    CREG_REQUIRE(n >= 1 && n < (1ll << 31), "%s: no source points", who);
    const IcpLayout L = icp_layout(n, nf, k);
    const size_t one = icp_ws_one(n, nf, k);
    CREG_REQUIRE(workspace_bytes >= one * (size_t)batch, "%s: workspace too small (%zu < %zu)", who, workspace_bytes,
                 one * (size_t)batch);
    // (CREG_ICP_P2P_ONE_WORKGROUP=1: point-to-point problems stay in the one-workgroup kernel whatever their size, as before round 5 --
    //  a measurement knob for tests/measure/bench_icp_p2p_regimes.py)
    const char* p2p_small = getenv("CREG_ICP_P2P_ONE_WORKGROUP");
    if (icp_large_regime(n, nf, k) && !(pr[0].tgt_offsets && p2p_small && p2p_small[0] == '1')) {
        // clusters / frames beyond what one CU's LDS holds: many workgroups per iteration instead of one per cluster
        for (int i = 0; i < batch; ++i) {
            const creg_icp_problem& q = pr[i];
            CREG_REQUIRE(q.local && q.seg_offsets && q.frame && q.M && q.M_out && q.world_out && q.n_iter_out,
                         "%s: null pointer in problem %d", who, i);
            const int rc = icp_large_run(q, n, k, nf, scale, th, max_iteration, keep_translation, (char*)workspace + one * (size_t)i, s);
            if (rc) return rc;
        }
        return CREG_OK;
    }
    IcpBatch B;
    for (int i = 0; i < batch; ++i) {
        const creg_icp_problem& q = pr[i];
        CREG_REQUIRE(q.local && q.seg_offsets && q.frame && q.M && q.M_out && q.world_out && q.n_iter_out,
                     "%s: null pointer in problem %d", who, i);
        B.local[i] = q.local; B.world[i] = q.world; B.off[i] = q.seg_offsets; B.frame[i] = q.frame; B.Min[i] = q.M;
        B.toff[i] = q.tgt_offsets; B.woff[i] = q.world_offsets;
        B.Mout[i] = q.M_out; B.world_out[i] = q.world_out; B.n_iter_out[i] = q.n_iter_out;
    }
    const int lds_cap = (int)(nf < 4096 ? (nf + 1) & ~1ll : 4096);  // masked targets kept in LDS (28 B each, <= 112 KB); even: the doubles behind the int table stay aligned
    const int smem = (lds_cap + ICP_PAD) * 28 + ICP_SRC_LDS * 28;   // padding behind the target list
    // per device, not per process: set on every call (a cached flag would leave a second GPU at the 64 KB default)
    CREG_HIP(hipFuncSetAttribute((const void*)k_masked_icp, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (4096 + ICP_PAD) * 28 + ICP_SRC_LDS * 28));
    hipLaunchKernelGGL(k_masked_icp, dim3(k, batch), dim3(ICP_NT), smem, s, B, (int)nf, (float)(0.5 * scale), th,
                       max_iteration, keep_translation, (char*)workspace, one, L.srcw, L.tidx, L.nn, lds_cap);
    CREG_LAUNCH_CHECK();
    return CREG_OK;
}

extern "C" int creg_masked_icp_f64(const double* local, const float* world, const int32_t* world_offsets, int64_t n,
                                   const int32_t* seg_offsets, int32_t k, const double* frame, int64_t nf, const double* M, double scale, double th,
                                   int32_t max_iteration, int32_t keep_translation, double* M_out, double* world_out,
                                   int32_t* n_iter_out, void* workspace, size_t workspace_bytes, creg_stream_t stream) {
    const creg_icp_problem p{local, world, seg_offsets, frame, M, M_out, world_out, n_iter_out, nullptr, world_offsets};
    return icp_launch(&p, 1, n, k, nf, scale, th, max_iteration, keep_translation, workspace, workspace_bytes,
                      (hipStream_t)stream, "creg_masked_icp_f64");
}

extern "C" int creg_masked_icp_batch_f64(const creg_icp_problem* problems, int32_t batch, int64_t n, int32_t k, int64_t nf,
                                         double scale, double th, int32_t max_iteration, int32_t keep_translation,
                                         void* workspace, size_t workspace_bytes, creg_stream_t stream) {
    return icp_launch(problems, batch, n, k, nf, scale, th, max_iteration, keep_translation,
                      workspace, workspace_bytes, (hipStream_t)stream, "creg_masked_icp_batch_f64");
}

extern "C" int creg_aabb_mask_f64(const float* world, const int32_t* world_offsets, int32_t k, const double* frame, int64_t nf,
                                  double scale, int32_t* mask_idx, int32_t* mask_count, float* boxes, creg_stream_t stream) {
    CREG_REQUIRE(world && world_offsets && frame && mask_idx && mask_count && k >= 1 && nf >= 1 && nf < (1ll << 31),
                 "creg_aabb_mask_f64: bad argument");
    IcpLarge P{};
    P.world = world; P.off = world_offsets; P.woff = world_offsets; P.frame = frame; P.toff = nullptr;
    P.tidx = mask_idx; P.tcount = mask_count; P.box = boxes;
    hipLaunchKernelGGL(k_icp_mask, dim3(k), dim3(1024), 0, (hipStream_t)stream, P, (int)nf, (float)(0.5 * scale), 0, 0);
    CREG_LAUNCH_CHECK();
    return CREG_OK;
}

// ------------------------------------------------------------------------------------------ closed-form fit alone
// creg_kabsch_f64: the (weighted) least-squares rigid fit of paired points, per segment -- the fit every ICP iteration of K4
// runs on its correspondences (open3d TransformationEstimationPointToPoint = Eigen::umeyama without scaling), callable on its
// own: Horn's unit quaternion of the centred cross-covariance (horn_max_eigvec: Newton + adjugate, Jacobi fallback), which is
// the proper rotation Umeyama's reflection fix selects.  One workgroup per segment; two passes (weighted centroids, then the
// centred covariance: no cancellation), every reduction in a fixed order (DPP wave sums, waves in sequence).
constexpr int KB_NT = 256;
__device__ __forceinline__ double kb_block_sum(double v, double* sh, int tid) {
    v = wave_sum_fast(v);
    __syncthreads();                                   // previous readers of sh are done
    if ((tid & 63) == 0) sh[tid >> 6] = v;
    __syncthreads();
    double r = sh[0];
#pragma unroll
    for (int w = 1; w < KB_NT / 64; ++w) r += sh[w];
    return r;
}
__global__ __launch_bounds__(KB_NT) void k_kabsch(const double* __restrict__ src, const double* __restrict__ dst,
                                                  const double* __restrict__ wgt, const int* __restrict__ off, double* __restrict__ T_out) {
    __shared__ double sh[KB_NT / 64];
    const int k = blockIdx.x, tid = threadIdx.x, b = off[k], e = off[k + 1];
    double sw = 0, s[3] = {0, 0, 0}, d[3] = {0, 0, 0};
    for (int i = b + tid; i < e; i += KB_NT) {
        const double w = wgt ? wgt[i] : 1.0;
        sw += w;
#pragma unroll
        for (int a = 0; a < 3; ++a) { s[a] = fma(w, src[3 * (size_t)i + a], s[a]); d[a] = fma(w, dst[3 * (size_t)i + a], d[a]); }
    }
    sw = kb_block_sum(sw, sh, tid);
    double ms[3], md[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) { s[a] = kb_block_sum(s[a], sh, tid); d[a] = kb_block_sum(d[a], sh, tid); }
    const bool any = sw > 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) { ms[a] = any ? s[a] / sw : 0.0; md[a] = any ? d[a] / sw : 0.0; }
    double C[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};                       // C[a][c] = sum w (s - ms)_a (d - md)_c
    for (int i = b + tid; i < e; i += KB_NT) {
        const double w = wgt ? wgt[i] : 1.0;
        double sc[3], dc[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) { sc[a] = src[3 * (size_t)i + a] - ms[a]; dc[a] = dst[3 * (size_t)i + a] - md[a]; }
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int c = 0; c < 3; ++c) C[3 * a + c] = fma(w * sc[a], dc[c], C[3 * a + c]);
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) C[i] = kb_block_sum(C[i], sh, tid);
    if (tid != 0) return;
    double T[16];
    for (int i = 0; i < 16; ++i) T[i] = (i % 5 == 0) ? 1.0 : 0.0;
    if (any) {
        const double Sxx = C[0], Sxy = C[1], Sxz = C[2], Syx = C[3], Syy = C[4], Syz = C[5], Szx = C[6], Szy = C[7], Szz = C[8];
        double N[4][4] = {{Sxx + Syy + Szz, Syz - Szy, Szx - Sxz, Sxy - Syx},
                          {Syz - Szy, Sxx - Syy - Szz, Sxy + Syx, Szx + Sxz},
                          {Szx - Sxz, Sxy + Syx, -Sxx + Syy - Szz, Syz + Szy},
                          {Sxy - Syx, Szx + Sxz, Syz + Szy, -Sxx - Syy + Szz}};
        double q[4], R[9], Vp[16];
        for (int i = 0; i < 16; ++i) Vp[i] = (i % 5 == 0) ? 1.0 : 0.0;
        horn_max_eigvec(N, q, Vp);
        quat_to_matrix(q, R);
        for (int a = 0; a < 3; ++a) {
            T[4 * a] = R[3 * a]; T[4 * a + 1] = R[3 * a + 1]; T[4 * a + 2] = R[3 * a + 2];
            T[4 * a + 3] = md[a] - (R[3 * a] * ms[0] + R[3 * a + 1] * ms[1] + R[3 * a + 2] * ms[2]);
        }
    }
    for (int i = 0; i < 16; ++i) T_out[16 * (size_t)k + i] = T[i];
}

extern "C" int creg_kabsch_f64(const double* src, const double* dst, const double* weights, int64_t n, const int32_t* offsets,
                               int32_t k, double* T_out, creg_stream_t stream) {
    CREG_REQUIRE(src && dst && offsets && T_out, "creg_kabsch_f64: null pointer");
    CREG_REQUIRE(k >= 1 && n >= 0 && n < (1ll << 31), "creg_kabsch_f64: bad size");
    hipLaunchKernelGGL(k_kabsch, dim3(k), dim3(KB_NT), 0, (hipStream_t)stream, src, dst, weights, offsets, T_out);
    CREG_LAUNCH_CHECK();
    return CREG_OK;
}

extern "C" int creg_icp_p2p_f64(const double* src, int64_t n_src, const int32_t* src_offsets, const double* tgt,
                                int64_t n_tgt, const int32_t* tgt_offsets, int32_t k, const double* init, double th,
                                int32_t max_iteration, double* T_out, double* src_out, int32_t* n_iter_out,
                                void* workspace, size_t workspace_bytes, creg_stream_t stream) {
    CREG_REQUIRE(tgt_offsets, "creg_icp_p2p_f64: null pointer");
    const creg_icp_problem p{src, nullptr, src_offsets, tgt, init, T_out, src_out, n_iter_out, tgt_offsets, nullptr};
    return icp_launch(&p, 1, n_src, k, n_tgt, 1.0, th, max_iteration, 0, workspace, workspace_bytes, (hipStream_t)stream,
                      "creg_icp_p2p_f64");
}

#ifdef CREG_ICP_BLK
extern "C" int creg_debug_icp_blk(unsigned long long* out) {
    CREG_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(creg::g_icp_blk), sizeof(unsigned long long) * 6 * 8192));
    return CREG_OK;
}
#endif
#ifdef CREG_STAMPS
extern "C" int creg_debug_icp_wall(unsigned long long* out8, int reset) {
    if (out8) CREG_HIP(hipMemcpyFromSymbol(out8, HIP_SYMBOL(creg::g_icp_wall), sizeof(unsigned long long) * 8));
    if (reset) { unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}; CREG_HIP(hipMemcpyToSymbol(HIP_SYMBOL(creg::g_icp_wall), z, sizeof(z)));
                 unsigned long long m[4] = {~0ull, ~0ull, 0ull, 0ull}; CREG_HIP(hipMemcpyToSymbol(HIP_SYMBOL(creg::g_icp_wmin), m, sizeof(m))); }
    return CREG_OK;
}
extern "C" int creg_debug_icp_stamps(unsigned long long* out8, int reset) {
    if (out8) CREG_HIP(hipMemcpyFromSymbol(out8, HIP_SYMBOL(creg::g_icp_stamps), sizeof(unsigned long long) * 512 * 16));
    if (reset) { static unsigned long long z[512 * 16]; CREG_HIP(hipMemcpyToSymbol(HIP_SYMBOL(creg::g_icp_stamps), z, sizeof(z))); }
    return CREG_OK;
}
#endif
