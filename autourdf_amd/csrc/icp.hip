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

constexpr int ICP_NT = 512;                           // threads per workgroup (16 waves)
constexpr int ICP_ROWS = 256;                          // threads that own source points ("row" threads)
constexpr int ICP_PARTS = ICP_NT / ICP_ROWS;           // the target list is split this many ways per source point
constexpr int ICP_BATCH_MAX = 16;
constexpr int ICP_SRC_LDS = 1024;                      // source points of a cluster kept in LDS (28 B each)

// Sum N values held by the row threads (waves 0..3) over the block, result in every thread.
// Fixed association: DPP wave sum per wave, then 0 + w0 + w1 + w2 + w3.
template <int N>
__device__ __forceinline__ void bsum_n(double (&v)[N], double* sc /* [4][N] */) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (wv < ICP_ROWS / 64) {
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = wave_sum_fast(v[i]);
    }
    __syncthreads();                                   // previous readers of sc are done
    if (lane == 0 && wv < ICP_ROWS / 64) {
#pragma unroll
        for (int i = 0; i < N; ++i) sc[wv * N + i] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < N; ++i) { double r = 0.0; for (int w = 0; w < ICP_ROWS / 64; ++w) r += sc[w * N + i]; v[i] = r; }
}

#ifdef CREG_STAMPS
// debug build only (CREG_EXTRA_FLAGS=-DCREG_STAMPS, tests/measure/icp_stamps.py): shader-clock cycles of the phases of
// workgroup (0, 0), accumulated over its iterations: [0] mask+setup [1] NN scan [2] combine + fitness [3] sums [4] Horn
// on lane 0 [5] move [6] iterations
__device__ unsigned long long g_icp_stamps[512 * 8];      // [workgroup (x + gridDim.x * y) % 512][slot]
#define ICP_STAMP(slot) do { if (stamp_on && threadIdx.x == 0) { const unsigned long long now_ = clock64(); g_icp_stamps[stamp_b * 8 + slot] += now_ - stamp_t; stamp_t = now_; } } while (0)
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

// grid (k, batch): one 1024-thread workgroup per cluster per problem.  Masked targets (<= lds_cap) and the
// cluster's moving source points (<= ICP_SRC_LDS) live in LDS for the whole loop; larger ones fall back to
// the workspace in global memory through the same (flat) pointers.  Threads 0..255 own the source points
// (sums, moves); the nearest-target search of a point is split over 4 threads (quarters of the target list,
// ascending, combined with strict '<' so the first minimum wins exactly as a sequential scan).
__global__ __launch_bounds__(ICP_NT) void k_masked_icp(IcpBatch P, int nf, float half_scale, double th, int max_iter,
                                                       int keep_t, char* __restrict__ ws, size_t ws_stride, size_t o_srcw,
                                                       size_t o_tidx, size_t o_nn, int lds_cap) {
    __shared__ double sc[4 * 9];
    __shared__ float s_lo[3], s_hi[3];
    __shared__ int s_wofs[ICP_NT / 64];
    __shared__ double T[16], U[16], Vp[16];
    __shared__ double sB[ICP_PARTS][ICP_ROWS];         // per part: best squared distance of the round's points
    __shared__ int sM[ICP_PARTS][ICP_ROWS];            //           and its target
    const int z = blockIdx.y;
#ifdef CREG_STAMPS
    const bool stamp_on = true;
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
    {
        __shared__ float wl[ICP_NT / 64][3], wh[ICP_NT / 64][3];
        for (int d = 0; d < 3; ++d) {
            for (int o = 32; o >= 1; o >>= 1) { lo[d] = fminf(lo[d], __shfl_xor(lo[d], o, 64)); hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], o, 64)); }
            if (lane == 0) { wl[wv][d] = lo[d]; wh[wv][d] = hi[d]; }
        }
        __syncthreads();
        if (tid < 3) {
            const int d = tid;
            float l = wl[0][d], h = wh[0][d];
            for (int w = 1; w < ICP_NT / 64; ++w) { l = fminf(l, wl[w][d]); h = fmaxf(h, wh[w][d]); }
            const float c = (l + h) / 2.0f, sz = h - l;
            s_lo[d] = c - half_scale * sz; s_hi[d] = c + half_scale * sz;
        }
        __syncthreads();
    }
    }
    // ---- ordered compaction of the frame points strictly inside the box ----
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* sT = (double*)smem;                       // [lds_cap][3] masked target coordinates
    int* sI = (int*)(sT + 3 * (size_t)lds_cap);       // [lds_cap]    their frame indices
    double* sS = (double*)(sI + lds_cap);             // [ICP_SRC_LDS][3] moving source points
    int* sN = (int*)(sS + 3 * ICP_SRC_LDS);           // [ICP_SRC_LDS] matched target slot / frame index, -1 = none
    int run = 0;                                      // points kept so far (block-uniform)
    if (toff) {                                       // point-to-point mode: the cluster's own target segment, unmasked
        const int tb = toff[k];
        run = toff[k + 1] - tb;
        for (int t = tid; t < run; t += ICP_NT) {
            const int j = tb + t;
            tidx[t] = j;
            if (t < lds_cap) { sT[3 * t] = frame[3 * (size_t)j]; sT[3 * t + 1] = frame[3 * (size_t)j + 1]; sT[3 * t + 2] = frame[3 * (size_t)j + 2]; sI[t] = j; }
        }
    }
    const double blo0 = (double)s_lo[0], blo1 = (double)s_lo[1], blo2 = (double)s_lo[2];
    const double bhi0 = (double)s_hi[0], bhi1 = (double)s_hi[1], bhi2 = (double)s_hi[2];
    for (int base = 0; base < (toff ? 0 : nf); base += ICP_NT) {
        const int j = base + tid;
        bool in = false;
        double x = 0, y = 0, zc = 0;
        if (j < nf && ns > 0) {
            x = frame[3 * (size_t)j]; y = frame[3 * (size_t)j + 1]; zc = frame[3 * (size_t)j + 2];
            in = x > blo0 && x < bhi0 && y > blo1 && y < bhi1 && zc > blo2 && zc < bhi2;
        }
        const unsigned long long m = __ballot(in);
        __syncthreads();                              // s_wofs of the previous round has been read
        if (lane == 0) s_wofs[wv] = __popcll(m);
        __syncthreads();
        int before = run, total = 0;
        for (int w = 0; w < ICP_NT / 64; ++w) { const int c = s_wofs[w]; before += w < wv ? c : 0; total += c; }
        if (in) {
            const int slot = before + __popcll(m & ((1ull << lane) - 1ull));
            tidx[slot] = j;
            if (slot < lds_cap) { sT[3 * slot] = x; sT[3 * slot + 1] = y; sT[3 * slot + 2] = zc; sI[slot] = j; }
        }
        run += total;
    }
    const int nt = run;
    const bool in_lds = nt <= lds_cap;
    const bool src_lds = ns <= ICP_SRC_LDS;
    double* S = src_lds ? sS : (double*)(wz + o_srcw) + 3 * (size_t)b;      // flat pointer: LDS or workspace
    int* nn = src_lds ? sN : (int*)(wz + o_nn) + b;

    // ---- 2. ICP ----
    if (tid < 16) { T[tid] = Min[16 * k + tid]; Vp[tid] = (tid % 5 == 0) ? 1.0 : 0.0; }
    __syncthreads();                                  // T, sT/sI and (not in_lds) tidx visible to the block
    if (row)
        for (int i = tid; i < ns; i += ICP_ROWS) {
            const double* p = local + 3 * (size_t)(b + i);
            const double p0 = p[0], p1 = p[1], p2 = p[2];
            for (int a = 0; a < 3; ++a) S[3 * i + a] = fma(T[4 * a + 2], p2, fma(T[4 * a + 1], p1, T[4 * a] * p0)) + T[4 * a + 3];
        }
    const double th2 = th * th;
    double fit = 0, rmse = 0;
    int it = 0;
    const int part = __builtin_amdgcn_readfirstlane(tid / ICP_ROWS), pi = tid % ICP_ROWS;
    const int t0 = (int)((long long)nt * part / ICP_PARTS), t1 = (int)((long long)nt * (part + 1) / ICP_PARTS);
    // matched target of a source point (slot in LDS, or frame index when the targets did not fit)
    auto tgt = [&](int m, int a) -> double { return in_lds ? sT[3 * m + a] : frame[3 * (size_t)m + a]; };
    auto correspond = [&](double& fitness, double& rm) {
        double ce[2] = {0, 0};
        __syncthreads();                              // the row threads' moves of S are visible
        ICP_STAMP(5);
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
            sB[part][pi] = best; sM[part][pi] = bm;
            __syncthreads();
            ICP_STAMP(1);
            if (row && i < ns) {                      // quarters in ascending target order, strict '<': first minimum
                best = sB[0][pi]; bm = sM[0][pi];
#pragma unroll
                for (int q = 1; q < ICP_PARTS; ++q) { const double d = sB[q][pi]; if (d < best) { best = d; bm = sM[q][pi]; } }
                if (bm >= 0 && best <= th2) { nn[i] = bm; ce[0] += 1.0; ce[1] += best; } else nn[i] = -1;
            }
            if (r0 + ICP_ROWS < ns) __syncthreads();  // sB / sM are rewritten by the next round
        }
        bsum_n<2>(ce, sc);
        ICP_STAMP(2);
        fitness = ns > 0 ? ce[0] / (double)ns : 0.0;
        rm = ce[0] > 0 ? sqrt(ce[1] / ce[0]) : 0.0;
        return ce[0];
    };
    ICP_STAMP(0);
    double ncorr = correspond(fit, rmse);
    for (it = 1; it <= max_iter; ++it) {
        // best rigid update from the current correspondences
        double mm[6] = {0, 0, 0, 0, 0, 0};            // sums of matched source / target coordinates
        if (row)
            for (int i = tid; i < ns; i += ICP_ROWS) {
                const int m = nn[i];
                if (m < 0) continue;
                for (int a = 0; a < 3; ++a) { mm[a] += S[3 * i + a]; mm[3 + a] += tgt(m, a); }
            }
        bsum_n<6>(mm, sc);
        if (ncorr > 0) for (int a = 0; a < 6; ++a) mm[a] /= ncorr;
        const double* ms = mm; const double* md = mm + 3;
        double C[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};            // C[a][c] = sum (src-ms)_a (dst-md)_c
        if (row)
            for (int i = tid; i < ns; i += ICP_ROWS) {
                const int m = nn[i];
                if (m < 0) continue;
                double sv[3], dv[3];
                for (int a = 0; a < 3; ++a) { sv[a] = S[3 * i + a] - ms[a]; dv[a] = tgt(m, a) - md[a]; }
                for (int a = 0; a < 3; ++a) for (int c = 0; c < 3; ++c) C[3 * a + c] = fma(sv[a], dv[c], C[3 * a + c]);
            }
        bsum_n<9>(C, sc);
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
                sym4_max_eigvec(N, q, Vp);
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
        if (stamp_on && tid == 0) g_icp_stamps[stamp_b * 8 + 6] += 1;
#endif
        if (row)
            for (int i = tid; i < ns; i += ICP_ROWS) {
                const double p0 = S[3 * i], p1 = S[3 * i + 1], p2 = S[3 * i + 2];
                for (int a = 0; a < 3; ++a) S[3 * i + a] = fma(U[4 * a + 2], p2, fma(U[4 * a + 1], p1, U[4 * a] * p0)) + U[4 * a + 3];
            }
        const double pf = fit, pr = rmse;
        ncorr = correspond(fit, rmse);
        if (fabs(pf - fit) < 1e-6 && fabs(pr - rmse) < 1e-6) break;
    }
    // ---- 3. outputs: icp matrix (optionally with the old translation), cluster moved by it ----
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
// BASELINE configs[4] shape has 2048-point clusters against ~3000 masked targets each): the same ICP as k_masked_icp,
// iteration by iteration over MANY workgroups instead of one workgroup per cluster running out of global memory:
//   k_icp_mask   (cluster)            box, ordered compaction of the frame indices inside it
//   k_icp_init   (cluster)            sources into the world frame with the initial pose
//   k_icp_nn     (256-source chunk)   nearest masked target of every source point: tiles of the cluster's targets staged in
//                                     LDS, sequential-equivalent first minimum (strict '<' in ascending target order)
//   k_icp_fit    (cluster)            fitness / RMSE of the new correspondences, open3d's convergence test against the
//                                     previous ones, else Horn's closed form from them, pose and source update
//   k_icp_finish (cluster)            outputs
// The host enqueues [fit, nn] in batches of 16 and reads one "clusters still running" word between batches.
struct IcpLarge {                          // per problem
    const double* local; const float* world; const int* off; const int* woff; const double* frame; const double* Min;
    double* Mout; double* world_out; int* n_iter_out;
    double* srcw; int* tidx; int* tcount; float* box; int* nn; double* d2; double* state; int* running;
    int* chunk0;                               // [k + 1] first source chunk of every cluster (k_icp_nn's block -> cluster map)
};
constexpr int ICP_CH = 256;                // sources per k_icp_nn workgroup
constexpr int ICP_ST = 48;                 // doubles of per-cluster state: T[16] Vp[16] prev_fit prev_rmse done n_updates ...

__global__ __launch_bounds__(1024) void k_icp_mask(IcpLarge P, int nf, float half_scale, int from_pose) {
    __shared__ float wl[16][3], wh[16][3];
    __shared__ float s_lo[3], s_hi[3];
    __shared__ int s_wofs[16];
    const int k = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int b = P.off[k], e = P.off[k + 1];
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    if (!from_pose) {
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
    __syncthreads();
    if (tid < 3) {
        const int d = tid;
        float l = wl[0][d], h = wh[0][d];
        for (int w = 1; w < 16; ++w) { l = fminf(l, wl[w][d]); h = fmaxf(h, wh[w][d]); }
        const float c = (l + h) / 2.0f, sz = h - l;
        s_lo[d] = c - half_scale * sz; s_hi[d] = c + half_scale * sz;
        if (P.box) { P.box[6 * k + d] = s_lo[d]; P.box[6 * k + 3 + d] = s_hi[d]; }
    }
    __syncthreads();
    const double blo0 = (double)s_lo[0], blo1 = (double)s_lo[1], blo2 = (double)s_lo[2];
    const double bhi0 = (double)s_hi[0], bhi1 = (double)s_hi[1], bhi2 = (double)s_hi[2];
    int* tidx = P.tidx + (size_t)k * nf;
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

__global__ __launch_bounds__(256) void k_icp_init(IcpLarge P, int k_total) {
    const int k = blockIdx.x, tid = threadIdx.x;
    const int b = P.off[k], e = P.off[k + 1];
    double T[12];
    for (int q = 0; q < 12; ++q) T[q] = P.Min[16 * k + q];
    for (int i = b + tid; i < e; i += 256) {
        const double p0 = P.local[3 * (size_t)i], p1 = P.local[3 * (size_t)i + 1], p2 = P.local[3 * (size_t)i + 2];
        for (int a = 0; a < 3; ++a) P.srcw[3 * (size_t)i + a] = fma(T[4 * a + 2], p2, fma(T[4 * a + 1], p1, T[4 * a] * p0)) + T[4 * a + 3];
    }
    if (tid < 16) { P.state[ICP_ST * k + tid] = P.Min[16 * k + tid]; P.state[ICP_ST * k + 16 + tid] = (tid % 5 == 0) ? 1.0 : 0.0; }
    if (tid == 0) {
        double* st = P.state + ICP_ST * k;
        st[32] = 0.0; st[33] = 0.0; st[34] = 0.0; st[35] = 0.0;       // prev fitness, prev rmse, done, updates applied
        if (k == 0) {
            *P.running = k_total;
            int c = 0;                                                // chunks of ICP_CH sources, never across clusters
            for (int j = 0; j < k_total; ++j) { P.chunk0[j] = c; c += (P.off[j + 1] - P.off[j] + ICP_CH - 1) / ICP_CH; }
            P.chunk0[k_total] = c;
        }
    }
}

// Block = one chunk of ICP_CH = 256 sources of ONE cluster (chunk0 maps blocks to clusters; blocks past the last chunk
// exit), one source per thread: every thread of the block scans the same target list, staged tile by tile in LDS.
// (Measured at N = 262144, K = 128: blocks over the concatenated sources that straddle two clusters 51 ms per frame's ICP,
//  cluster-aligned chunks 32 ms, two sources per thread -- half the LDS broadcast reads per pair, half the waves -- 41 ms.)
__global__ __launch_bounds__(256) void k_icp_nn(IcpLarge P, int n, int k_total, int nf, double th2) {
    __shared__ double tx[256], ty[256], tz[256];
    const int tid = threadIdx.x, blk = blockIdx.x;
    if (blk >= P.chunk0[k_total]) return;
    int lo = 0, hi = k_total;                                         // cluster c with chunk0[c] <= blk < chunk0[c + 1]
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (P.chunk0[mid] <= blk) lo = mid; else hi = mid; }
    const int c = lo;
    if (P.state[ICP_ST * c + 34] != 0.0) return;                      // converged cluster: its correspondences stay
    const int i = P.off[c] + (blk - P.chunk0[c]) * ICP_CH + tid;
    const bool live = i < P.off[c + 1];
    double s0 = 0, s1 = 0, s2 = 0;
    if (live) { s0 = P.srcw[3 * (size_t)i]; s1 = P.srcw[3 * (size_t)i + 1]; s2 = P.srcw[3 * (size_t)i + 2]; }
    double best = INFINITY; int bm = -1;
    const int nt = P.tcount[c];
    const int* tidx = P.tidx + (size_t)c * nf;
    for (int t0 = 0; t0 < nt; t0 += 256) {
        __syncthreads();
        if (t0 + tid < nt) {
            const int j = tidx[t0 + tid];
            tx[tid] = P.frame[3 * (size_t)j]; ty[tid] = P.frame[3 * (size_t)j + 1]; tz[tid] = P.frame[3 * (size_t)j + 2];
        }
        __syncthreads();
        const int cnt = min(256, nt - t0);
#pragma unroll 8
        for (int t = 0; t < cnt; ++t) {
            const double dx = s0 - tx[t], dy = s1 - ty[t], dz = s2 - tz[t];
            const double d = (dx * dx + dy * dy) + dz * dz;
            if (d < best) { best = d; bm = t0 + t; }                  // the slot; its frame index is looked up once at the end
        }
    }
    if (live) {
        const bool ok = bm >= 0 && best <= th2;
        P.nn[i] = ok ? tidx[bm] : -1;
        P.d2[i] = ok ? best : 0.0;
    }
}

template <int N>
__device__ __forceinline__ void bsum512(double (&v)[N], double* sc /* [8][N] */) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = wave_sum_fast(v[i]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < N; ++i) sc[wv * N + i] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < N; ++i) { double r = 0.0; for (int w = 0; w < 8; ++w) r += sc[w * N + i]; v[i] = r; }
}

__global__ __launch_bounds__(512) void k_icp_fit(IcpLarge P, int max_iter) {
    __shared__ double sc[8 * 9];
    __shared__ double T[16], U[16], Vp[16];
    const int k = blockIdx.x, tid = threadIdx.x;
    double* st = P.state + ICP_ST * k;
    if (st[34] != 0.0) return;                                        // done
    const int b = P.off[k], e = P.off[k + 1], ns = e - b;
    // fitness / inlier RMSE of the correspondences the last k_icp_nn produced (registration_icp's GetRegistrationResult)
    double ce[2] = {0, 0};
    for (int i = b + tid; i < e; i += 512) if (P.nn[i] >= 0) { ce[0] += 1.0; ce[1] += P.d2[i]; }
    bsum512<2>(ce, sc);
    const double ncorr = ce[0];
    const double fit = ns > 0 ? ce[0] / (double)ns : 0.0, rmse = ce[0] > 0 ? sqrt(ce[1] / ce[0]) : 0.0;
    const double pf = st[32], pr = st[33];
    const int updates = (int)st[35];
    const bool converged = updates >= 1 && fabs(pf - fit) < 1e-6 && fabs(pr - rmse) < 1e-6;
    if (converged || updates >= max_iter) {
        __syncthreads();
        if (tid == 0) { st[34] = 1.0; atomicSub(P.running, 1); }
        return;
    }
    if (tid < 16) { T[tid] = st[tid]; Vp[tid] = st[16 + tid]; }
    double mm[6] = {0, 0, 0, 0, 0, 0};
    for (int i = b + tid; i < e; i += 512) {
        const int m = P.nn[i];
        if (m < 0) continue;
        for (int a = 0; a < 3; ++a) { mm[a] += P.srcw[3 * (size_t)i + a]; mm[3 + a] += P.frame[3 * (size_t)m + a]; }
    }
    bsum512<6>(mm, sc);
    if (ncorr > 0) for (int a = 0; a < 6; ++a) mm[a] /= ncorr;
    const double* ms = mm; const double* md = mm + 3;
    double C[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = b + tid; i < e; i += 512) {
        const int m = P.nn[i];
        if (m < 0) continue;
        double sv[3], dv[3];
        for (int a = 0; a < 3; ++a) { sv[a] = P.srcw[3 * (size_t)i + a] - ms[a]; dv[a] = P.frame[3 * (size_t)m + a] - md[a]; }
        for (int a = 0; a < 3; ++a) for (int c = 0; c < 3; ++c) C[3 * a + c] = fma(sv[a], dv[c], C[3 * a + c]);
    }
    bsum512<9>(C, sc);
    if (tid == 0) {
        for (int i = 0; i < 16; ++i) U[i] = (i % 5 == 0) ? 1.0 : 0.0;
        if (ncorr > 0) {
            const double Sxx = C[0], Sxy = C[1], Sxz = C[2], Syx = C[3], Syy = C[4], Syz = C[5], Szx = C[6], Szy = C[7], Szz = C[8];
            double N[4][4] = {{Sxx + Syy + Szz, Syz - Szy, Szx - Sxz, Sxy - Syx},
                              {Syz - Szy, Sxx - Syy - Szz, Sxy + Syx, Szx + Sxz},
                              {Szx - Sxz, Sxy + Syx, -Sxx + Syy - Szz, Syz + Szy},
                              {Sxy - Syx, Szx + Sxz, Syz + Szy, -Sxx - Syy + Szz}};
            double q[4], R[9];
            sym4_max_eigvec(N, q, Vp);
            quat_to_matrix(q, R);
            for (int a = 0; a < 3; ++a) {
                U[4 * a] = R[3 * a]; U[4 * a + 1] = R[3 * a + 1]; U[4 * a + 2] = R[3 * a + 2];
                U[4 * a + 3] = md[a] - (R[3 * a] * ms[0] + R[3 * a + 1] * ms[1] + R[3 * a + 2] * ms[2]);
            }
        }
        for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) {
            double s = 0;
            for (int m = 0; m < 4; ++m) s = fma(U[4 * r + m], T[4 * m + c], s);
            st[4 * r + c] = s;
        }
        for (int i = 0; i < 16; ++i) st[16 + i] = Vp[i];
        st[32] = fit; st[33] = rmse; st[35] = (double)(updates + 1);
    }
    __syncthreads();
    for (int i = b + tid; i < e; i += 512) {
        const double p0 = P.srcw[3 * (size_t)i], p1 = P.srcw[3 * (size_t)i + 1], p2 = P.srcw[3 * (size_t)i + 2];
        for (int a = 0; a < 3; ++a) P.srcw[3 * (size_t)i + a] = fma(U[4 * a + 2], p2, fma(U[4 * a + 1], p1, U[4 * a] * p0)) + U[4 * a + 3];
    }
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

struct IcpLargeLayout { size_t srcw, tidx, tcount, box, nn, d2, state, running, chunk0, total; };
static IcpLargeLayout icp_large_layout(int64_t n, int64_t nf, int k) {
    IcpLargeLayout L; size_t o = 0;
    auto take = [&](size_t b) { size_t r = o; o = align_up(o + b, 256); return r; };
    L.srcw = take(sizeof(double) * 3 * n); L.tidx = take(sizeof(int) * (size_t)k * nf); L.tcount = take(sizeof(int) * k);
    L.box = take(sizeof(float) * 6 * k); L.nn = take(sizeof(int) * n); L.d2 = take(sizeof(double) * n);
    L.state = take(sizeof(double) * ICP_ST * k); L.running = take(sizeof(int) * 4); L.chunk0 = take(sizeof(int) * (k + 1)); L.total = o;
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

// One problem through the multi-launch path.  Synchronises the stream between batches of iterations.
static int icp_large_run(const creg_icp_problem& q, int64_t n, int32_t k, int64_t nf, double scale, double th,
                         int32_t max_iteration, int32_t keep_translation, char* ws, hipStream_t s) {
    const IcpLargeLayout L = icp_large_layout(n, nf, k);
    IcpLarge P;
    P.local = q.local; P.world = q.world; P.off = q.seg_offsets; P.woff = q.world_offsets; P.frame = q.frame; P.Min = q.M;
    P.Mout = q.M_out; P.world_out = q.world_out; P.n_iter_out = q.n_iter_out;
    P.srcw = (double*)(ws + L.srcw); P.tidx = (int*)(ws + L.tidx); P.tcount = (int*)(ws + L.tcount); P.box = (float*)(ws + L.box);
    P.nn = (int*)(ws + L.nn); P.d2 = (double*)(ws + L.d2); P.state = (double*)(ws + L.state); P.running = (int*)(ws + L.running); P.chunk0 = (int*)(ws + L.chunk0);
    if (q.tgt_offsets) {
        set_error("creg_masked_icp: point-to-point mode (tgt_offsets) is not available in the large-cluster regime");
        return CREG_EINVAL;
    }
    hipLaunchKernelGGL(k_icp_mask, dim3(k), dim3(1024), 0, s, P, (int)nf, (float)(0.5 * scale), q.world ? 0 : 1);
    hipLaunchKernelGGL(k_icp_init, dim3(k), dim3(256), 0, s, P, k);
    const int nblk = cdiv(n, ICP_CH) + k;                // an upper bound of sum_c ceil(ns_c / ICP_CH); the surplus blocks exit
    hipLaunchKernelGGL(k_icp_nn, dim3(nblk), dim3(256), 0, s, P, (int)n, k, (int)nf, th * th);
    CREG_LAUNCH_CHECK();
    int running = k;
    // every k_icp_fit call is one convergence test + (unless converged) one update; max_iteration updates need one call more
    for (int64_t done = 0; running > 0 && done <= (int64_t)max_iteration; ) {
        const int batch = 16;
        for (int b = 0; b < batch; ++b, ++done) {
            hipLaunchKernelGGL(k_icp_fit, dim3(k), dim3(512), 0, s, P, max_iteration);
            hipLaunchKernelGGL(k_icp_nn, dim3(nblk), dim3(256), 0, s, P, (int)n, k, (int)nf, th * th);
        }
        CREG_LAUNCH_CHECK();
        CREG_HIP(hipMemcpyAsync(&running, P.running, sizeof(int), hipMemcpyDeviceToHost, s));
        CREG_HIP(hipStreamSynchronize(s));
    }
    hipLaunchKernelGGL(k_icp_finish, dim3(k), dim3(256), 0, s, P, keep_translation);
    CREG_LAUNCH_CHECK();
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
    if (icp_large_regime(n, nf, k) && !pr[0].tgt_offsets) {
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
    const int lds_cap = (int)(nf < 4096 ? nf : 4096);             // masked targets kept in LDS (28 B each, <= 112 KB)
    const int smem = lds_cap * 28 + ICP_SRC_LDS * 28;
    // per device, not per process: set on every call (a cached flag would leave a second GPU at the 64 KB default)
    CREG_HIP(hipFuncSetAttribute((const void*)k_masked_icp, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 4096 * 28 + ICP_SRC_LDS * 28));
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
    P.world = world; P.off = world_offsets; P.woff = world_offsets; P.frame = frame;
    P.tidx = mask_idx; P.tcount = mask_count; P.box = boxes;
    hipLaunchKernelGGL(k_icp_mask, dim3(k), dim3(1024), 0, (hipStream_t)stream, P, (int)nf, (float)(0.5 * scale), 0);
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

#ifdef CREG_STAMPS
extern "C" int creg_debug_icp_stamps(unsigned long long* out8, int reset) {
    if (out8) CREG_HIP(hipMemcpyFromSymbol(out8, HIP_SYMBOL(creg::g_icp_stamps), sizeof(unsigned long long) * 512 * 8));
    if (reset) { static unsigned long long z[512 * 8]; CREG_HIP(hipMemcpyToSymbol(HIP_SYMBOL(creg::g_icp_stamps), z, sizeof(z))); }
    return CREG_OK;
}
#endif
