// kmeans.hip -- K2: Lloyd k-means in fp64 for 3-D points, sklearn.cluster.k_means(init=array,
// n_init=1) semantics (reference mlp_reg.py:204, cluster_icp.py:67), plus the stable grouping /
// change-of-frame step that follows it (mlp_reg.py:208-217).
//
// Per Lloyd iteration ONE launch (bit-reproducible):
//   k_assign[_mfma]  labels = first argmin_k fma(x2,b2,fma(x1,b1,fma(x0,b0,|c|^2))),  b = -2c; the per-cluster sums of
//                    the M-step in the same pass, as EXACT integers: a point's coordinates enter as int64 fixed point
//                    (scale 2^s chosen per call so that n points cannot overflow; a grid of range * 2^-43 at n = 262144,
//                    2^-49 at n = 4096: below the error of any float64 summation order).  Integer sums are exact, hence
//                    order-independent -- atomics without losing determinism -- and INCREMENTAL: the accumulators persist
//                    over the iterations and only a point whose label changed moves its coordinates from the old
//                    cluster's sum to the new one's (no drift: integers).  After the first few Lloyd iterations a handful of
//                    points change, so the M-step costs next to nothing; per-workgroup tables in LDS, non-zero entries
//                    flushed to the global accumulators with integer atomics.
//                    The LAST workgroup to arrive (agent-scope atomics on both sides of the hand-off, every wave drains
//                    vmcnt(0) before the arrival counter: km_arrive_last) runs the M-step tail in the same launch: sums -> centres (x 1/w), relocation of empty clusters, centre shift, convergence (strict
//                    label equality first, then tol), next b / |c|^2 rows.
// The host enqueues iterations in batches of 32 and reads the `done` word between batches; launches of
// iterations after convergence return immediately.
// VALU form of creg_kmeans_lloyd_f64 (round 3): the E-step runs over a spatially sorted copy of the frame and evaluates, per
// workgroup and per wave, only the centres that can be nearest inside the bounding box of its points (k_km_assign_pruned:
// labels identical to the full sweep), and a persistent kernel (k_km_persist) iterates inside ONE launch with the points and
// labels in registers -- 38 -> 17 us per Lloyd iteration at N = 262144, K = 128.  The matrix-core form keeps the full sweep
// in the caller's order, one launch per iteration; both give identical labels, centres, inertia and iteration counts.
#include <atomic>
#include <cstdlib>
#include "creg_common.h"
#include "creg_dev.h"

namespace creg {

struct KmFlags {
    int changed;      // points whose label differs from the previous iteration (this iteration)
    int done;         // convergence reached; later iterations are no-ops
    int strict;       // converged by label equality (no final E-step needed)
    int n_iter;       // iterations executed (i + 1 of the breaking iteration)
    int cur;          // which centre buffer holds the current centres
    int arrive;       // workgroups of the running E-step launch that have flushed their sums (last one runs the M-step tail)
    int reloc;        // an empty cluster was found: the NEXT launch runs the M-step tail (with the relocation) instead of an E-step
    int pad[1];
    int arrive8[8];   // first-level arrival counters (workgroup index mod 8)
    double mean[3];
    double tol;
    double shift_tot;
    double inertia;
    double fix_scale, fix_inv;   // fixed-point scale 2^s of the M-step accumulators and its inverse
    unsigned long long gen;      // persistent Lloyd kernel: (M-step tails completed << 8) | 1 when the kernel is to exit
    int abort;                   // persistent Lloyd kernel: a workgroup gave up waiting (the host starts the call over without that kernel)
    int pad2;
    double amax;                 // largest |centred coordinate| of the frame
    double guard;                // pruned E-step: slack on squared distances, far above the rounding of the fma chain (see k_km_assign_pruned)
};

// scale 2^s with n * range * 2^s < 2^62: the int64 sums of n fixed-point coordinates cannot overflow
__device__ __forceinline__ double km_fix_scale(double range, int n) {
    if (!(range > 0.0)) return 1.0;
    int e;
    frexp(range, &e);                                    // range < 2^e
    const int nb = 32 - __clz(n);                        // n < 2^nb
    return ldexp(1.0, 62 - nb - e);
}
__device__ __forceinline__ unsigned long long km_fix(double x, double scale) { return (unsigned long long)__double2ll_rn(x * scale); }

typedef double double4v __attribute__((ext_vector_type(4)));

// B rows: (-2cx, -2cy, -2cz, |c|^2)
__device__ __forceinline__ void make_b(const double* c, double* b) {
    b[0] = -2.0 * c[0]; b[1] = -2.0 * c[1]; b[2] = -2.0 * c[2];
    b[3] = fma(c[2], c[2], fma(c[1], c[1], c[0] * c[0]));
}

__global__ __launch_bounds__(1024) void k_km_stats(const double* __restrict__ X, int n, double tol_rel,
                                                   KmFlags* __restrict__ f) {
    __shared__ double sc[16];
    __shared__ double mean[3];
    // the three coordinate sums of a pass run side by side (each keeps its own order: 1024 strided partials + tree)
    {
        double s0 = 0, s1 = 0, s2 = 0;
        for (int i = threadIdx.x; i < n; i += 1024) { s0 += X[3 * (size_t)i]; s1 += X[3 * (size_t)i + 1]; s2 += X[3 * (size_t)i + 2]; }
        s0 = block_sum<double, 1024>(s0, sc); s1 = block_sum<double, 1024>(s1, sc); s2 = block_sum<double, 1024>(s2, sc);
        if (threadIdx.x == 0) { mean[0] = s0 / (double)n; mean[1] = s1 / (double)n; mean[2] = s2 / (double)n; }
    }
    __syncthreads();
    double var = 0, amax = 0;
    {
        const double m0 = mean[0], m1 = mean[1], m2 = mean[2];
        double s0 = 0, s1 = 0, s2 = 0;
        for (int i = threadIdx.x; i < n; i += 1024) {
            const double t0 = X[3 * (size_t)i] - m0, t1 = X[3 * (size_t)i + 1] - m1, t2 = X[3 * (size_t)i + 2] - m2;
            s0 = fma(t0, t0, s0); s1 = fma(t1, t1, s1); s2 = fma(t2, t2, s2);
            amax = fmax(amax, fmax(fabs(t0), fmax(fabs(t1), fabs(t2))));
        }
        s0 = block_sum<double, 1024>(s0, sc); s1 = block_sum<double, 1024>(s1, sc); s2 = block_sum<double, 1024>(s2, sc);
        if (threadIdx.x == 0) { var += s0 / (double)n; var += s1 / (double)n; var += s2 / (double)n; }
    }
    __shared__ double s_amax[16];
    for (int off = 32; off >= 1; off >>= 1) amax = fmax(amax, __shfl_xor(amax, off, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_amax[threadIdx.x >> 6] = amax;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = 0;
        for (int w = 0; w < 16; ++w) r = fmax(r, s_amax[w]);
        f->fix_scale = km_fix_scale(r, n); f->fix_inv = 1.0 / f->fix_scale;
        f->amax = r; f->guard = 3e-10 * r * r;
        f->gen = 0ull; f->abort = 0;
        f->mean[0] = mean[0]; f->mean[1] = mean[1]; f->mean[2] = mean[2];
        f->tol = (var / 3.0) * tol_rel;
        f->changed = 0; f->done = 0; f->strict = 0; f->n_iter = 0; f->cur = 0; f->arrive = 0; f->reloc = 0;
        for (int i = 0; i < 8; ++i) f->arrive8[i] = 0;
        f->shift_tot = 0; f->inertia = 0;
    }
}

__global__ __launch_bounds__(256) void k_km_center(const double* __restrict__ X, int n,
                                                   const double* __restrict__ init, int k,
                                                   const KmFlags* __restrict__ f, double* __restrict__ Xc,
                                                   double* __restrict__ C, double* __restrict__ B,
                                                   int* __restrict__ labels_prev, const int* __restrict__ inv) {
    // inv (pruned E-step): the centred copy goes out in the spatially sorted order, point i to row inv[i]
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const size_t r = inv ? (size_t)inv[i] : (size_t)i;
#pragma unroll
        for (int d = 0; d < 3; ++d) Xc[3 * r + d] = X[3 * (size_t)i + d] - f->mean[d];
        labels_prev[i] = -1;
    }
    if (i < k) {
        double c[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) { c[d] = init[3 * i + d] - f->mean[d]; C[3 * i + d] = c[d]; }
        make_b(c, B + 4 * i);
    }
}

// ---- spatial order for the pruned E-step ------------------------------------------------------------------------
// The points of a frame never move during a k_means() call, only the centres do.  Sorted once by the Morton code of a
// 2^b-per-axis grid (counting sort: count, scan, scatter), every run of 256 PT consecutive points has a small bounding box,
// and an E-step workgroup only evaluates the centres that can be nearest to SOME point of its box.  The order inside a grid
// cell is whatever the scatter's atomics produce; nothing observable depends on it (a point's label depends on the point
// alone, the M-step sums are exact integers, the relocation and the inertia keep the caller's order through `inv`).
__device__ __forceinline__ unsigned km_spread3(unsigned v) {   // 10 bits -> bits 0, 3, 6, ...
    v = (v | (v << 16)) & 0x030000FFu; v = (v | (v << 8)) & 0x0300F00Fu; v = (v | (v << 4)) & 0x030C30C3u; v = (v | (v << 2)) & 0x09249249u;
    return v;
}
__global__ __launch_bounds__(256) void k_km_cell_count(const double* __restrict__ X, int n, const KmFlags* __restrict__ f, int bits,
                                                       int* __restrict__ key, int* __restrict__ cnt) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double r = f->amax, sc = r > 0.0 ? (double)(1 << bits) / (2.0 * r) : 0.0;
    unsigned q[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const double t = ((X[3 * (size_t)i + d] - f->mean[d]) + r) * sc;
        q[d] = (unsigned)min(max((int)t, 0), (1 << bits) - 1);
    }
    const int c = (int)(km_spread3(q[0]) | (km_spread3(q[1]) << 1) | (km_spread3(q[2]) << 2));
    key[i] = c;
    atomicAdd(&cnt[c], 1);
}
// counts -> first rows, in place.  The cells fall into 64 aligned Morton ranges (boxes a quarter of the frame's extent per
// axis), one workgroup each; a range starts on a multiple of `align` rows, the rows skipped stay dummies (perm = -1, coordinates
// NaN).  A workgroup's run of `align` consecutive rows therefore never straddles two of the ranges: without this, the handful of
// runs that span a big jump of the Morton order get boxes as large as the frame, keep all k centres, and the whole iteration
// waits ~8 us for their full sweeps.  Two launches: the ranges' totals, then every range scans its own cells behind the aligned
// totals of the ranges before it (coalesced; one workgroup walking all 2^18 cells took 385 us).
__global__ __launch_bounds__(256) void k_km_cell_totals(const int* __restrict__ cnt, int ncell, int* __restrict__ gtot) {
    __shared__ int sc[4];
    const int per = ncell / 64, c0 = blockIdx.x * per;           // ncell = 8^b >= 512: a multiple of 64
    int s = 0;
    for (int c = threadIdx.x; c < per; c += 256) s += cnt[c0 + c];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sc[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) gtot[blockIdx.x] = sc[0] + sc[1] + sc[2] + sc[3];
}
__global__ __launch_bounds__(256) void k_km_cell_scan(int* __restrict__ cnt, int ncell, int align, const int* __restrict__ gtot) {
    __shared__ int wsum[4];
    __shared__ int s_run;
    const int per = ncell / 64, c0 = blockIdx.x * per, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) {
        int base = 0;
        for (int g = 0; g < (int)blockIdx.x; ++g) base += (gtot[g] + align - 1) / align * align;
        s_run = base;
    }
    __syncthreads();
    for (int t0 = 0; t0 < per; t0 += 256) {                      // 256 cells per step: wave scans, then the waves' offsets
        const int c = t0 + threadIdx.x;
        const int v = c < per ? cnt[c0 + c] : 0;
        int inc = v;
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
        if (lane == 63) wsum[wv] = inc;
        __syncthreads();
        int base = s_run + inc - v;
        for (int w = 0; w < wv; ++w) base += wsum[w];
        if (c < per) cnt[c0 + c] = base;
        __syncthreads();
        if (threadIdx.x == 0) s_run += wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void k_km_cell_scatter(int n, const int* __restrict__ key, int* __restrict__ cursor,
                                                         int* __restrict__ perm, int* __restrict__ inv) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int r = atomicAdd(&cursor[key[i]], 1);
    perm[r] = i; inv[i] = r;
}
// bounding box (min xyz, max xyz) of every run of `run` sorted points
__global__ __launch_bounds__(256) void k_km_boxes(const double* __restrict__ Xs, int n, int run, double* __restrict__ box) {
    __shared__ double sm[4][6];
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = blockIdx.x * run + threadIdx.x; i < min(n, (blockIdx.x + 1) * run); i += 256)
#pragma unroll
        for (int d = 0; d < 3; ++d) { const double v = Xs[3 * (size_t)i + d]; lo[d] = fmin(lo[d], v); hi[d] = fmax(hi[d], v); }
    for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
        for (int d = 0; d < 3; ++d) { lo[d] = fmin(lo[d], __shfl_xor(lo[d], off, 64)); hi[d] = fmax(hi[d], __shfl_xor(hi[d], off, 64)); }
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int d = 0; d < 3; ++d) { sm[threadIdx.x >> 6][d] = lo[d]; sm[threadIdx.x >> 6][3 + d] = hi[d]; }
    __syncthreads();
    if (threadIdx.x < 6) {
        double v = sm[0][threadIdx.x];
        for (int w = 1; w < 4; ++w) v = threadIdx.x < 3 ? fmin(v, sm[w][threadIdx.x]) : fmax(v, sm[w][threadIdx.x]);
        box[6 * (size_t)blockIdx.x + threadIdx.x] = v;
    }
}
__global__ __launch_bounds__(256) void k_km_unsort(const int* __restrict__ sorted, const int* __restrict__ perm, int n, int* __restrict__ out) {
    const int r = blockIdx.x * 256 + threadIdx.x;                // n rows, dummies among them
    if (r < n && perm[r] >= 0) out[perm[r]] = sorted[r];
}

__device__ __forceinline__ unsigned long long ld_agent(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int ld_agent(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_agent(const double* p) {
    return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_agent(double* p, double v) {
    __hip_atomic_store((unsigned long long*)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct KmTailSh { double fv[16]; int fi[16]; int nempty, argmax, last, bi; double bv; };

// (value descending, index ascending) maximum over the block; the result is returned to every thread.  Two barriers.
template <int NT>
__device__ __forceinline__ void km_block_argmax(double& bv, int& bi, KmTailSh& S) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int off = 32; off >= 1; off >>= 1) {
        const double ov = __shfl_xor(bv, off, 64); const int oi = __shfl_xor(bi, off, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    __syncthreads();                                             // previous readers of S.bv / S.bi are done
    if (lane == 0) { S.fv[wv] = bv; S.fi[wv] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 0; w < NT / 64; ++w) if (S.fv[w] > bv || (S.fv[w] == bv && S.fi[w] < bi)) { bv = S.fv[w]; bi = S.fi[w]; }
        S.bv = bv; S.bi = bi;
    }
    __syncthreads();
    bv = S.bv; bi = S.bi;
}

// Every workgroup of a launch calls this once after its global atomics / agent-scope stores; true in the last workgroup to
// arrive.  ORDERING: a bare s_barrier does not wait for a wave's outstanding memory operations on gfx950 (hipcc emits no
// s_waitcnt vmcnt(0) in front of it when the wave only has stores / non-returning atomics in flight), so EVERY wave drains
// its own vmcnt explicitly before the barrier -- only then are all of the workgroup's sums and flags acknowledged by the
// memory side when thread 0 bumps the counter (cdna_hip_programming.md, Guideline 16 R1: "EVERY storing wave" drains).
// Two levels -- 8 shard counters, then one -- so that a few hundred workgroups finishing together do not queue on one
// word (MI355X_MICROARCH.md: ~12 ns per arrival on one counter).
__device__ __forceinline__ bool km_arrive_last(KmFlags* __restrict__ f, KmTailSh& S) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const int sh = blockIdx.x & 7, in_shard = ((int)gridDim.x - sh + 7) >> 3, shards = min(8, (int)gridDim.x);
        int last = 0;
        if (atomicAdd(&f->arrive8[sh], 1) == in_shard - 1) last = atomicAdd(&f->arrive, 1) == shards - 1;
        S.last = last;
    }
    __syncthreads();
    return S.last != 0;
}
__device__ __forceinline__ void km_arrive_reset(KmFlags* __restrict__ f) {
    for (int i = 0; i < 8; ++i) f->arrive8[i] = 0;
    f->arrive = 0;
}

// M-step tail of one Lloyd iteration, run by ONE workgroup of NT threads (the last one of its launch to arrive):
// sums -> centres, _relocate_empty_clusters_dense, centre shift, convergence flags, next (-2c, |c|^2) rows.
// `acc` are the exact integer sums of the CURRENT labels (they persist: the E-step moves a point between sums when
// its label changes); `shift` is >= 4 k doubles of LDS scratch.
// Everything read here that another workgroup wrote in THIS launch goes through agent-scope atomics on both sides (the
// sums, the changed-label count; on the relocation path the per-point distances and the per-segment maxima), so neither side
// needs a fence (MI355X_MICROARCH.md: "8-B agent atomics both sides").
// relocate = false (the tail of an E-step launch): an empty cluster defers the tail -- the relocation needs the distance of
// every point to its centre, i.e. the labels the other workgroups have just stored with plain stores.  `reloc` is set and
// the NEXT launch does it (km_reloc_pass in every workgroup, then this function with relocate = true in the last one).
template <int NT>
__device__ void km_mstep_tail(const double* __restrict__ X, int n, const int* __restrict__ labels, int k,
                              const unsigned long long* __restrict__ acc, double* __restrict__ C2, double* B,
                              double* __restrict__ Cw, double* far_d, double* segv, int* segi, int segsz, int nseg,
                              KmFlags* __restrict__ f, KmTailSh& S, double* __restrict__ shift, bool relocate,
                              const int* __restrict__ inv = nullptr) {
    // inv (pruned E-step): X and labels are stored in the spatially sorted order, point i in row inv[i]; the relocation keeps
    // working in the caller's point order (far_d, the segments and the tie-breaks of the farthest-point choice index points by i)
    const int tid = threadIdx.x;
    const int cur = f->cur;
    const double finv = f->fix_inv, tol = f->tol;
    const double* Cold = C2 + (size_t)cur * 3 * k;
    double* Cnew = C2 + (size_t)(cur ^ 1) * 3 * k;
    double* cpark = shift + k;                                  // [k][3] (the scratch holds 4 k doubles)
    if (tid == 0) S.nempty = 0;
    __syncthreads();
    for (int j = tid; j < k; j += NT) {
        double a[4];
        for (int d = 0; d < 3; ++d) a[d] = (double)(long long)ld_agent(acc + 4 * j + d) * finv;
        a[3] = (double)(long long)ld_agent(acc + 4 * j + 3);
        if (a[3] == 0.0) atomicAdd(&S.nempty, 1);                  // integer count: order independent
        if (relocate) for (int d = 0; d < 4; ++d) Cw[4 * j + d] = a[d];
        else {
            // common path: straight to the centre, parked in LDS until the barrier says "no empty cluster" (an empty
            // cluster defers the whole tail, nothing is kept)
            double s2 = 0;
            const double alpha = 1.0 / a[3];
            for (int d = 0; d < 3; ++d) { const double c = a[d] * alpha; cpark[3 * j + d] = c; const double t = c - Cold[3 * j + d]; s2 += t * t; }
            const double sh = sqrt(s2);
            shift[j] = sh * sh;
        }
    }
    __syncthreads();
    if (!relocate) {
        if (S.nempty > 0) {                                      // block-uniform; rare: defer
            if (tid == 0) { f->reloc = 1; km_arrive_reset(f); }
            return;
        }
        for (int j = tid; j < k; j += NT) {
            const double c[3] = {cpark[3 * j], cpark[3 * j + 1], cpark[3 * j + 2]};
            for (int d = 0; d < 3; ++d) Cnew[3 * j + d] = c[d];
            make_b(c, B + 4 * j);
        }
    } else {
        if (S.nempty > 0) {
            // _relocate_empty_clusters_dense: the farthest points (descending distance, ties to the lower index) seed the
            // empty clusters and leave their old ones.  Two levels: the maxima of the nseg segments km_reloc_pass left,
            // then a rescan of the winner's segment only.
            double dv = -1; int di = 0x7fffffff;
            for (int sg = tid; sg < nseg; sg += NT) { const double v = ld_agent(segv + sg); const int i = ld_agent(segi + sg); if (v > dv || (v == dv && i < di)) { dv = v; di = i; } }
            km_block_argmax<NT>(dv, di, S);
            if (dv > 0) {                                        // the largest distance: all zero -> nothing moves (sklearn)
                for (int j = 0; j < k; ++j) {
                    if (Cw[4 * j + 3] != 0.0) continue;          // uniform: Cw only changes under barriers
                    double bv = -1; int bi = 0x7fffffff;
                    for (int sg = tid; sg < nseg; sg += NT) { const double v = ld_agent(segv + sg); const int i = ld_agent(segi + sg); if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; } }
                    km_block_argmax<NT>(bv, bi, S);
                    const int sb = bi / segsz;
                    if (tid == 0) {
                        st_agent(far_d + bi, -2.0);
                        const size_t rb = inv ? (size_t)inv[bi] : (size_t)bi;
                        const int old = labels[rb];
                        for (int d = 0; d < 3; ++d) { Cw[4 * old + d] -= X[3 * rb + d]; Cw[4 * j + d] = X[3 * rb + d]; }
                        Cw[4 * j + 3] = 1.0; Cw[4 * old + 3] -= 1.0;
                    }
                    __syncthreads();
                    double rv = -3; int ri = 0x7fffffff;          // the segment's new maximum (taken points hold -2)
                    for (int i = sb * segsz + tid; i < min(n, (sb + 1) * segsz); i += NT) { const double v = ld_agent(far_d + i); if (v > rv || (v == rv && i < ri)) { rv = v; ri = i; } }
                    km_block_argmax<NT>(rv, ri, S);
                    if (tid == 0) { st_agent(segv + sb, rv); st_agent(segi + sb, ri); }
                    __syncthreads();
                }
            }
            // first cluster of maximal weight (weight desc, index asc): the centre a still-empty cluster takes
            if (tid == 0) { int am = 0; for (int j = 1; j < k; ++j) if (Cw[4 * j + 3] > Cw[4 * am + 3]) am = j; S.argmax = am; }
            __syncthreads();
        }
        for (int j = tid; j < k; j += NT) {
            const double w = Cw[4 * j + 3];
            double c[3];
            if (w > 0) { const double alpha = 1.0 / w; for (int d = 0; d < 3; ++d) c[d] = Cw[4 * j + d] * alpha; }
            else { const double wa = Cw[4 * S.argmax + 3]; const double alpha = 1.0 / wa;
                   for (int d = 0; d < 3; ++d) c[d] = Cw[4 * S.argmax + d] * alpha; }
            double s2 = 0;
            for (int d = 0; d < 3; ++d) { Cnew[3 * j + d] = c[d]; const double t = c[d] - Cold[3 * j + d]; s2 += t * t; }
            const double sh = sqrt(s2);
            shift[j] = sh * sh;
            make_b(c, B + 4 * j);
        }
        __syncthreads();
    }
    if (tid == 0) {
        const int changed = ld_agent(&f->changed);
        double tot = 0;
        for (int j0 = 0; j0 < k; j0 += 8) {                      // fixed order (cluster index); a group's loads are issued together
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = shift[min(j0 + u, k - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u) if (j0 + u < k) tot += v[u];
        }
        f->shift_tot = tot;
        f->cur = cur ^ 1;
        f->n_iter += 1;
        if (changed == 0) { f->strict = 1; f->done = 1; }
        else if (tot <= tol) { f->done = 1; }
        f->changed = 0;
        f->reloc = 0;
        km_arrive_reset(f);
    }
}

// a point whose label changed moves from the old cluster's sums to the new one's (prev < 0: first iteration, add only)
__device__ __forceinline__ void km_move(unsigned long long* sA, int lab, int pv, double x0, double x1, double x2, double fscale) {
    const unsigned long long v0 = km_fix(x0, fscale), v1 = km_fix(x1, fscale), v2 = km_fix(x2, fscale);
    atomicAdd(&sA[4 * lab], v0); atomicAdd(&sA[4 * lab + 1], v1); atomicAdd(&sA[4 * lab + 2], v2); atomicAdd(&sA[4 * lab + 3], 1ull);
    if (pv >= 0) {
        atomicAdd(&sA[4 * pv], 0ull - v0); atomicAdd(&sA[4 * pv + 1], 0ull - v1); atomicAdd(&sA[4 * pv + 2], 0ull - v2);
        atomicAdd(&sA[4 * pv + 3], ~0ull);
    }
}

#ifndef CREG_STAMPS
#define KM_PHASE(p_) do { } while (0)
#define KM_BLK(p_) do { } while (0)
#define KM_PP(p_) do { } while (0)
#endif
// Lloyd launches only: the M-step tail's operands (B = the rows `B` points at, writable), the relocation scratch (far: n
// doubles, segv / segi: one entry per workgroup) and the label buffers.  A launch works on iteration t = f->n_iter: it
// writes lab[t & 1] and compares with lab[(t - 1) & 1] (prev0 = "no label" at t = 0).
#ifdef CREG_STAMPS
// debug build only: 10 ns wall-clock ticks, per E-step launch relative to its first block's start: [0] launches [1] sum of mean-ish block start [2] sum of
// last block end (arrival) [3] sum of tail end [4] blocks; scratch: g_km_w = {launch start (min), last arrival (max), tail end}
__device__ unsigned long long g_km_stamps[8];
__device__ unsigned long long g_km_w[4];
__device__ unsigned long long g_km_pp[16];                      // persistent kernel: sums over workgroups and iterations of the time per phase (10 ns ticks), [15] = workgroup-iterations
__device__ unsigned long long g_km_it[4][1024];                 // persistent kernel, iteration 150: per workgroup rows landed | arrived | tail done | saw gen
__device__ unsigned long long g_km_blk[4][1024];                // per workgroup of the LAST launch: start, operands landed, E-step end, after arrival
__device__ unsigned long long g_km_ph[8], g_km_phs[8];          // pruned E-step: latest block past phase p (max), summed over launches
#define KM_PP(p_) do { if (threadIdx.x == 0) { const unsigned long long now_ = wall_clock64(); pp[p_] += now_ - pp_t; pp_t = now_; \
                       if (t == 150 && blockIdx.x < 1024 && ((p_) == 0 || (p_) == 3 || (p_) == 4 || (p_) == 5)) g_km_it[(p_) == 0 ? 0 : (p_) - 2][blockIdx.x] = now_; } } while (0)
#define KM_BLK(p_) do { if (threadIdx.x == 0 && blockIdx.x < 1024) g_km_blk[p_][blockIdx.x] = wall_clock64(); } while (0)
#define KM_PHASE(p_) do { if (threadIdx.x == 0 && lloyd && blockIdx.x == 300) g_km_ph[p_] = wall_clock64(); } while (0)   // one workgroup's timeline (atomics from all of them would serialise)
__global__ void k_km_fold() {
    if (g_km_w[2] != 0ull) { g_km_stamps[0] += 1; g_km_stamps[2] += g_km_w[1] - g_km_w[0]; g_km_stamps[3] += g_km_w[2] - g_km_w[0]; g_km_stamps[5] += g_km_w[3] - g_km_w[0]; }
    for (int p = 0; p < 8; ++p) { if (g_km_w[2] != 0ull && g_km_ph[p] != 0ull) g_km_phs[p] += g_km_ph[p] - g_km_w[0]; g_km_ph[p] = 0ull; }
    g_km_w[0] = ~0ull; g_km_w[1] = 0ull; g_km_w[2] = 0ull; g_km_w[3] = 0ull;
}
#endif
struct KmTail { double* B; double* C2; double* Cw; double* far_d; double* segv; int* segi; int* lab[2]; const int* prev0; int max_iter; const int* inv;
                double* ring; unsigned long long* genrep; unsigned long long* slots;
                int nrow; const int* perm; int spin_limit; int rw; };                  // pruned form: rows of the sorted copy (dummies: perm < 0)
// persistent Lloyd kernel: centre rows of the iterations of ONE launch at distinct addresses (see k_km_persist), and copies of `gen`
constexpr int KMP_RING = 320, KMP_GENREP = 8, KMP_GENREP_STRIDE = 512;       // (stride in 8-byte words: 4 KB apart)
__host__ __device__ __forceinline__ size_t kmp_ring_stride(int k) { return ((size_t)4 * k + 15) & ~(size_t)15; }   // doubles, 128-byte multiple

// The workgroup's table of sums goes to the global accumulators; the last workgroup of the launch to have done so runs
// the M-step tail.
template <int NT>
__device__ __forceinline__ void km_flush_and_tail(const unsigned long long* sA, const double* __restrict__ X, int n,
                                                  int k, unsigned long long* __restrict__ acc,
                                                  KmFlags* __restrict__ f, const KmTail& T, double* lds_scratch) {
    __shared__ KmTailSh S;
    __syncthreads();
    for (int i = threadIdx.x; i < 4 * k; i += NT) { const unsigned long long v = sA[i]; if (v) atomicAdd(&acc[i], v); }
    const bool last_wg = km_arrive_last(f, S);
    KM_BLK(3);
    if (!last_wg) return;
#ifdef CREG_STAMPS
    if (threadIdx.x == 0) g_km_w[1] = wall_clock64();
#endif
    km_mstep_tail<NT>(X, n, nullptr, k, acc, T.C2, T.B, T.Cw, T.far_d, T.segv, T.segi, 0, 0, f, S, lds_scratch, false);
#ifdef CREG_STAMPS
    __syncthreads();
    if (threadIdx.x == 0) g_km_w[2] = wall_clock64();
#endif
}

// Entry of a Lloyd launch: done / budget spent -> nothing; a deferred M-step tail pending -> this launch does the
// relocation instead of an E-step: every workgroup computes its segment's point-to-centre distances and their maximum
// (the labels of the previous launch are visible now), the last one to arrive runs the tail.  Returns true when the
// launch should run its E-step, with the iteration's label buffers selected.
template <int NT>
__device__ __forceinline__ bool km_lloyd_entry(const double* __restrict__ X, int n, int k, unsigned long long* __restrict__ acc,
                                               KmFlags* __restrict__ f, const KmTail& T, double* lds_scratch,
                                               int*& labels, const int*& prev) {
    if (f->done) return false;
    const int t = f->n_iter;
    if (t >= T.max_iter) return false;
    labels = (t & 1) ? T.lab[1] : T.lab[0];
    prev = t == 0 ? T.prev0 : ((t & 1) ? T.lab[0] : T.lab[1]);
    if (!f->reloc) return true;
    __shared__ KmTailSh S;
    const int nseg = gridDim.x, segsz = (n + nseg - 1) / nseg, sg = blockIdx.x;
    const double* Cold = T.C2 + (size_t)f->cur * 3 * k;
    double bv = -1; int bi = 0x7fffffff;
    for (int i = sg * segsz + threadIdx.x; i < min(n, (sg + 1) * segsz); i += NT) {
        const size_t r = T.inv ? (size_t)T.inv[i] : (size_t)i;
        const double* c = Cold + 3 * labels[r];
        const double a = X[3 * r] - c[0], b = X[3 * r + 1] - c[1], e = X[3 * r + 2] - c[2];
        const double d = (a * a + b * b) + e * e;
        st_agent(T.far_d + i, d);
        if (d > bv) { bv = d; bi = i; }                          // ascending i: the first maximum stays
    }
    km_block_argmax<NT>(bv, bi, S);
    if (threadIdx.x == 0) { st_agent(T.segv + sg, bv); st_agent(T.segi + sg, bi); }
    if (km_arrive_last(f, S))
        km_mstep_tail<NT>(X, n, labels, k, acc, T.C2, T.B, T.Cw, T.far_d, T.segv, T.segi, segsz, nseg, f, S, lds_scratch, true, T.inv);
    return false;
}

// E-step, VALU form: PT points per thread (the centre rows are read from LDS once for all of them), 256 threads.
template <int PT>
__global__ __launch_bounds__(256) void k_km_assign(const double* __restrict__ X, int n,
                                                   const double* B, int k,
                                                   int* labels_in, const int* prev_in,
                                                   KmFlags* __restrict__ f, int raw, unsigned long long* __restrict__ acc, KmTail T) {
    // raw: `B` holds the centres (k,3) and every workgroup derives its (-2c, |c|^2) rows itself -- the standalone
    // entry point then needs no device scratch (the library never allocates)
    // acc (Lloyd only): global [k][4] int64 sums of the M-step (fixed-point x, y, z and the count) of the labels in `prev`
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef CREG_STAMPS
    if (threadIdx.x == 0 && f) atomicMin(&g_km_w[0], wall_clock64());
#endif
    double* sB = (double*)smem;
    unsigned long long* sA = (unsigned long long*)(sB + 4 * k);
    int* labels = labels_in;
    const int* prev = prev_in;
    if (f && !km_lloyd_entry<256>(X, n, k, acc, f, T, sB, labels, prev)) return;      // Lloyd launch: buffers of iteration f->n_iter
    if (raw) { for (int j = threadIdx.x; j < k; j += 256) make_b(B + 3 * j, sB + 4 * j); }
    else for (int i = threadIdx.x; i < 4 * k; i += 256) sB[i] = B[i];
    if (acc) for (int i = threadIdx.x; i < 4 * k; i += 256) sA[i] = 0ull;
    const double fscale = acc ? f->fix_scale : 0.0;
    __syncthreads();
    const int i0 = blockIdx.x * (256 * PT) + threadIdx.x;
    double x[PT][3], best[PT];
    int lab[PT];
#pragma unroll
    for (int q = 0; q < PT; ++q) {
        const int i = min(i0 + 256 * q, n - 1);
        x[q][0] = X[3 * (size_t)i]; x[q][1] = X[3 * (size_t)i + 1]; x[q][2] = X[3 * (size_t)i + 2];
        best[q] = fma(x[q][2], sB[2], fma(x[q][1], sB[1], fma(x[q][0], sB[0], sB[3])));
        lab[q] = 0;
    }
    for (int j = 1; j < k; ++j) {
        const double b0 = sB[4 * j], b1 = sB[4 * j + 1], b2 = sB[4 * j + 2], b3 = sB[4 * j + 3];
#pragma unroll
        for (int q = 0; q < PT; ++q) {
            const double d = fma(x[q][2], b2, fma(x[q][1], b1, fma(x[q][0], b0, b3)));
            if (d < best[q]) { best[q] = d; lab[q] = j; }
        }
    }
    int diff = 0;
#pragma unroll
    for (int q = 0; q < PT; ++q) {
        const int i = i0 + 256 * q;
        if (i < n) {
            labels[i] = lab[q];
            if (prev) {
                const int pv = prev[i];
                if (pv != lab[q]) { ++diff; if (acc) km_move(sA, lab[q], pv, x[q][0], x[q][1], x[q][2], fscale); }
            }
        }
    }
#ifdef CREG_STAMPS
    if (threadIdx.x == 0 && f) atomicMax(&g_km_w[3], wall_clock64());      // last block's end of the E-step proper
#endif
    if (prev && f) {
        diff = wave_sum(diff);
        if ((threadIdx.x & 63) == 0 && diff) atomicAdd(&f->changed, diff);
    }
    if (acc) km_flush_and_tail<256>(sA, X, n, k, acc, f, T, sB);
}

// bounding box of a wave's points (dummy rows are NaN: fmin / fmax pass them over); wave-uniform result
template <int PT>
__device__ __forceinline__ void km_wave_box(const double (&x)[PT][3], double (&wb)[6]) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        double lo = INFINITY, hi = -INFINITY;
#pragma unroll
        for (int q = 0; q < PT; ++q) { lo = fmin(lo, x[q][d]); hi = fmax(hi, x[q][d]); }
        for (int off = 32; off >= 1; off >>= 1) { lo = fmin(lo, __shfl_xor(lo, off, 64)); hi = fmax(hi, __shfl_xor(hi, off, 64)); }
        wb[d] = lo; wb[3 + d] = hi;
    }
}

// The pruned sweep of one workgroup (256 threads, PT points per thread in x, centre rows in sB, box = min xyz | max xyz): see
// k_km_assign_pruned.  Called by every thread (barriers inside); lab = the sweep's first argmin per point.
template <int PT>
__device__ __forceinline__ void km_pruned_sweep(const double (&x)[PT][3], const double* sB, double* slb, int* cand, int k,
                                                const double (&bx)[6], double guard, double* s_mub, int* s_cnt, int (&lab)[PT],
                                                const double (&wb)[6], int* nc_out = nullptr) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const bool has_points = bx[0] <= bx[3];                      // a run of dummy rows only has the empty box (inf, -inf): no survivors
    double mub = INFINITY;
    for (int j = tid; j < k; j += 256) {
        const double c0 = -0.5 * sB[4 * j], c1 = -0.5 * sB[4 * j + 1], c2 = -0.5 * sB[4 * j + 2];   // b = -2c exactly
        const double a0 = bx[0] - c0, e0 = c0 - bx[3], a1 = bx[1] - c1, e1 = c1 - bx[4], a2 = bx[2] - c2, e2 = c2 - bx[5];
        const double n0 = fmax(fmax(a0, e0), 0.0), n1 = fmax(fmax(a1, e1), 0.0), n2 = fmax(fmax(a2, e2), 0.0);
        const double f0 = fmax(-a0, -e0), f1 = fmax(-a1, -e1), f2 = fmax(-a2, -e2);
        slb[j] = fma(n2, n2, fma(n1, n1, n0 * n0));
        mub = fmin(mub, fma(f2, f2, fma(f1, f1, f0 * f0)));
    }
    for (int off = 32; off >= 1; off >>= 1) mub = fmin(mub, __shfl_xor(mub, off, 64));
    if (lane == 0) s_mub[wv] = mub;
    __syncthreads();
    const double thr = fmin(fmin(s_mub[0], s_mub[1]), fmin(s_mub[2], s_mub[3])) * (1.0 + 1e-9) + guard;
    int nc = 0;
    for (int j0 = 0; j0 < k; j0 += 256) {                        // ordered compaction, 256 centres per pass
        const int j = j0 + tid;
        const bool keep = has_points && j < k && slb[j] <= thr;
        const unsigned long long m = __ballot(keep);
        if (lane == 0) s_cnt[wv] = __popcll(m);
        __syncthreads();
        int off = nc + __popcll(m & ((1ull << lane) - 1ull));
        for (int w = 0; w < wv; ++w) off += s_cnt[w];
        if (keep) cand[off] = j;
        nc += s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        __syncthreads();
    }
    // Second level, per wave, when the workgroup's list is long (its run spans a sparse corner of the frame): the same test
    // against the box of the wave's own 64 PT points, over the listed centres only, into the wave's own list (cand + k (1 + wave)).
    // The point's nearest centre is in both lists, each being a superset of the centres that can be nearest inside its box.
    const int* mylist = cand;
    if (nc > 12) {
        int* wl = cand + k * (1 + wv);
        const bool whas = wb[0] <= wb[3];
        double wmub = INFINITY;
        for (int c0 = 0; c0 < nc; c0 += 64) {
            const int j = cand[min(c0 + lane, nc - 1)];
            const double c0x = -0.5 * sB[4 * j], c1x = -0.5 * sB[4 * j + 1], c2x = -0.5 * sB[4 * j + 2];
            const double f0 = fmax(c0x - wb[0], wb[3] - c0x), f1 = fmax(c1x - wb[1], wb[4] - c1x), f2 = fmax(c2x - wb[2], wb[5] - c2x);
            wmub = fmin(wmub, fma(f2, f2, fma(f1, f1, f0 * f0)));
        }
        for (int off = 32; off >= 1; off >>= 1) wmub = fmin(wmub, __shfl_xor(wmub, off, 64));
        const double wthr = wmub * (1.0 + 1e-9) + guard;
        int wn = 0;
        for (int c0 = 0; c0 < nc; c0 += 64) {
            const int c = c0 + lane;
            const int j = cand[min(c, nc - 1)];
            const double c0x = -0.5 * sB[4 * j], c1x = -0.5 * sB[4 * j + 1], c2x = -0.5 * sB[4 * j + 2];
            const double n0 = fmax(fmax(wb[0] - c0x, c0x - wb[3]), 0.0), n1 = fmax(fmax(wb[1] - c1x, c1x - wb[4]), 0.0), n2 = fmax(fmax(wb[2] - c2x, c2x - wb[5]), 0.0);
            const bool keep = whas && c < nc && fma(n2, n2, fma(n1, n1, n0 * n0)) <= wthr;
            const unsigned long long m = __ballot(keep);
            if (keep) wl[wn + __popcll(m & ((1ull << lane) - 1ull))] = j;
            wn += __popcll(m);
        }
        __builtin_amdgcn_wave_barrier();
        mylist = wl; nc = wn;
    }
    if (nc_out) *nc_out = nc;
    double best[PT];
#pragma unroll
    for (int q = 0; q < PT; ++q) { best[q] = INFINITY; lab[q] = 0; }
    // four survivors per step: their indices, then their rows, are in flight together (one survivor per step is a chain of two
    // dependent LDS reads -- a workgroup whose box keeps 50-128 centres then takes 5-10 us and the whole iteration waits for it);
    // the last step repeats the last survivor, which cannot pass the strict `<` twice
    for (int c = 0; c < nc; c += 4) {
        int j[4];
        double b[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) j[u] = mylist[min(c + u, nc - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u) { b[u][0] = sB[4 * j[u]]; b[u][1] = sB[4 * j[u] + 1]; b[u][2] = sB[4 * j[u] + 2]; b[u][3] = sB[4 * j[u] + 3]; }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int q = 0; q < PT; ++q) {
                const double d = fma(x[q][2], b[u][2], fma(x[q][1], b[u][1], fma(x[q][0], b[u][0], b[u][3])));
                if (d < best[q]) { best[q] = d; lab[q] = j[u]; }
            }
    }
}

// E-step, pruned form (Lloyd launches and the final E-step of creg_kmeans_lloyd_f64's VALU path): X holds the centred frame in
// the spatial order above, `box` the bounding box of every workgroup's 256 PT points.  With lb_j / ub_j the smallest / largest
// squared distance from the box to centre j, no point of the box can have centre j nearest unless lb_j <= min_j' ub_j'; the
// survivors (a handful of the 128 centres of a configs[4] frame) are listed in ascending j and evaluated with the SAME fma
// chain and the same strict `<` as the full sweep, so the label is the sweep's first argmin: a pruned centre is farther from
// every point of the box than the survivor that realises min ub by more than `guard` = 3e-10 amax^2, while the chain's value
// differs from the true |x - c|^2 - |x|^2 by a few ulp of amax^2 (~1e-15 amax^2) -- it can neither win nor tie.
template <int PT>
__global__ __launch_bounds__(256) void k_km_assign_pruned(const double* __restrict__ X, int n, const double* B, int k,
                                                          const double* __restrict__ box, int* labels_in,
                                                          KmFlags* __restrict__ f, int lloyd, unsigned long long* __restrict__ acc, KmTail T) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ double s_mub[4];
    __shared__ int s_cnt[4];
    double* sB = (double*)smem;
    unsigned long long* sA = (unsigned long long*)(sB + 4 * k);
    double* slb = (double*)(sA + 4 * k);
    int* cand = (int*)(slb + k);
    int* labels = labels_in;
    const int* prev = nullptr;
    // every kernel argument into registers NOW, in one batch of scalar loads (the compiler otherwise fetches each field of the
    // argument block where it is first used -- a separate round trip to the argument buffer every time)
    asm volatile("" :: "s"(X), "s"(n), "s"(B), "s"(k), "s"(box), "s"(labels_in), "s"(f), "s"(lloyd), "s"(acc));
    asm volatile("" :: "s"(T.B), "s"(T.C2), "s"(T.Cw), "s"(T.far_d), "s"(T.segv), "s"(T.segi), "s"(T.lab[0]), "s"(T.lab[1]), "s"(T.prev0), "s"(T.max_iter), "s"(T.inv));
#ifdef CREG_STAMPS
    if (threadIdx.x == 0 && lloyd) atomicMin(&g_km_w[0], wall_clock64());
#endif
    KM_BLK(0);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // Every operand of the launch is requested before anything is waited for -- the points, the centre rows, the box, and the
    // previous labels from BOTH label buffers (which one holds them depends on the iteration count, which arrives with the
    // flags): one memory round trip for the flags and the operands together instead of two or three in a row.
    // a wave owns T.rw <= 64 PT consecutive rows (lane l: rows l, 64 + l, ...; the last slot is partly masked)
    const int rw = T.rw, i0 = (blockIdx.x * 4 + wv) * rw + lane, nrow = T.nrow;
    double x[PT][3];
    int lab[PT], pv[PT], pa[PT], pb[PT], own[PT];
#pragma unroll
    for (int q = 0; q < PT; ++q) {
        const int i = min(i0 + 64 * q, nrow - 1);
        x[q][0] = X[3 * (size_t)i]; x[q][1] = X[3 * (size_t)i + 1]; x[q][2] = X[3 * (size_t)i + 2];
        own[q] = 64 * q + lane < rw ? T.perm[i] : -1;              // < 0: a dummy row, or a slot beyond the wave's rows
        if (64 * q + lane >= rw) { x[q][0] = x[q][1] = x[q][2] = __builtin_nan(""); }   // (kept out of the wave's box like a dummy)
        pa[q] = lloyd ? T.lab[0][i] : -1; pb[q] = lloyd ? T.lab[1][i] : -1;      // (first iteration: whatever the workspace holds, unused)
    }
    double br[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) br[u] = tid + 256 * u < 4 * k ? B[tid + 256 * u] : 0.0;       // k <= 128: the whole table
    const double fscale = lloyd ? f->fix_scale : 0.0, guard = f->guard;
    const double* bx = box + 6 * (size_t)blockIdx.x;
    const double bx6[6] = {bx[0], bx[1], bx[2], bx[3], bx[4], bx[5]};
    if (lloyd && !km_lloyd_entry<256>(X, n, k, acc, f, T, sB, labels, prev)) return;
    KM_PHASE(0);
#pragma unroll
    for (int q = 0; q < PT; ++q) pv[q] = prev == T.lab[0] ? pa[q] : prev == T.lab[1] ? pb[q] : -1;
    if (lloyd) for (int i = tid; i < 4 * k; i += 256) sA[i] = 0ull;
#pragma unroll
    for (int u = 0; u < 2; ++u) if (tid + 256 * u < 4 * k) sB[tid + 256 * u] = br[u];
    for (int i = tid + 512; i < 4 * k; i += 256) sB[i] = B[i];
    __syncthreads();
    KM_PHASE(1); KM_BLK(1);
    double wb[6];
    km_wave_box<PT>(x, wb);
    km_pruned_sweep<PT>(x, sB, slb, cand, k, bx6, guard, s_mub, s_cnt, lab, wb);
    KM_PHASE(3);
    int diff = 0;
#pragma unroll
    for (int q = 0; q < PT; ++q) {
        const int i = i0 + 64 * q;
        if (64 * q + lane < rw) {
            labels[i] = lab[q];
            if (lloyd && own[q] >= 0 && pv[q] != lab[q]) { ++diff; km_move(sA, lab[q], pv[q], x[q][0], x[q][1], x[q][2], fscale); }
        }
    }
#ifdef CREG_STAMPS
    if (threadIdx.x == 0 && lloyd) atomicMax(&g_km_w[3], wall_clock64());
#endif
    KM_BLK(2);
    if (lloyd) {
        diff = wave_sum(diff);
        if (lane == 0 && diff) atomicAdd(&f->changed, diff);
        km_flush_and_tail<256>(sA, X, n, k, acc, f, T, sB);
    }
}

// ---- persistent Lloyd kernel -------------------------------------------------------------------------------------
// A launch boundary between two Lloyd iterations costs far more than the pruned E-step itself (measured at the configs[4]
// shape, tests/measure/km_stamps.py: the workgroups finish their sweep 2-4 us after they start, yet an iteration took 26 us --
// every launch begins by re-fetching its operands through caches the boundary has just written back and invalidated, and
// ends with two more such round trips).  Here ONE launch runs iterations until convergence / max_iter / an empty cluster:
// a workgroup keeps its points and their labels in registers for the whole call; per iteration it fetches the 4 k centre
// words, runs the pruned sweep, moves the changed points between the exact integer sums, arrives; the last workgroup to arrive
// runs the M-step tail and publishes `gen`; the others wait for it.  What crosses workgroups inside the launch goes through
// agent-scope accesses on both sides, every wave draining its stores before the hand-off (km_arrive_last; the tail before
// `gen`).  The state in memory between launches (flags, centres, sums, the two label buffers) is exactly the multi-launch
// path's, so the two kinds of launch are interchangeable steps of one state machine: an empty cluster makes this kernel exit
// with `reloc` set, the next ordinary launch relocates, and the host starts the kernel again.
// All workgroups must be resident at once (the host checks the grid against the occupancy the runtime reports); the wait is
// bounded all the same -- a workgroup that polls for ~2^20 round trips (~1 s: other streams kept the device so busy that the
// grid never was resident together) sets `abort` and leaves; the host then discards the attempt and runs the whole call
// again with one launch per iteration (same results).
__device__ __forceinline__ void st_agent(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// common path of km_mstep_tail (same arithmetic, same order) with agent-scope loads and write-through stores; returns through
// `gen`: bit 0 = leave the kernel (converged, max_iter reached, or an empty cluster awaits the relocation launch)
template <int NT>
__device__ void km_tail_persist(int k, const unsigned long long* __restrict__ acc, double* __restrict__ C2, double* __restrict__ B,
                                KmFlags* __restrict__ f, KmTailSh& S, double* __restrict__ shift, int max_iter,
                                double* __restrict__ Bnext, bool ring_full, unsigned long long* __restrict__ genrep,
                                int cur, int it_done, unsigned long long gen_next, double finv, double tol, int changed) {
    // cur / it_done / gen_next: every workgroup counts the launch's tails itself, so the tail reads nothing but the sums and
    // the changed-label count (ONE round trip), and its centre and flag stores share one drain before `gen` goes out
    const int tid = threadIdx.x;
    const double* Cold = C2 + (size_t)cur * 3 * k;
    double* Cnew = C2 + (size_t)(cur ^ 1) * 3 * k;
    double* cpark = shift + k;
    // changed: some label of the grid changed in this iteration (gathered from the arrival slots); f->changed stays 0 between
    // iterations, as the multi-launch tail leaves it
    if (tid == 0) S.nempty = 0;
    __syncthreads();
    for (int j = tid; j < k; j += NT) {
        double a[4], co[3];
        for (int d = 0; d < 3; ++d) a[d] = (double)(long long)ld_agent(acc + 4 * j + d);
        a[3] = (double)(long long)ld_agent(acc + 4 * j + 3);
        for (int d = 0; d < 3; ++d) co[d] = ld_agent(Cold + 3 * j + d);
        for (int d = 0; d < 3; ++d) a[d] *= finv;
        if (a[3] == 0.0) atomicAdd(&S.nempty, 1);
        double s2 = 0;
        const double alpha = 1.0 / a[3];
        for (int d = 0; d < 3; ++d) { const double c = a[d] * alpha; cpark[3 * j + d] = c; const double t = c - co[d]; s2 += t * t; }
        const double sh = sqrt(s2);
        shift[j] = sh * sh;
    }
    __syncthreads();
    const bool empty = S.nempty > 0;
    if (!empty)
        for (int j = tid; j < k; j += NT) {
            const double c[3] = {cpark[3 * j], cpark[3 * j + 1], cpark[3 * j + 2]};
            double b[4];
            make_b(c, b);
            for (int d = 0; d < 3; ++d) st_agent(Cnew + 3 * j + d, c[d]);
            for (int d = 0; d < 4; ++d) { st_agent(B + 4 * j + d, b[d]); st_agent(Bnext + 4 * j + d, b[d]); }
        }
    int leave = 1;
    if (tid == 0) {
        if (empty) { st_agent(&f->reloc, 1); st_agent(&f->changed, changed ? 1 : 0); }   // deferred: the relocation launch finishes this iteration (and tests `changed`)
        else {
            double tot = 0;
            for (int j0 = 0; j0 < k; j0 += 8) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = shift[min(j0 + u, k - 1)];
#pragma unroll
                for (int u = 0; u < 8; ++u) if (j0 + u < k) tot += v[u];
            }
            st_agent(&f->shift_tot, tot);
            st_agent(&f->cur, cur ^ 1);
            st_agent(&f->n_iter, it_done);
            int done = 0;
            if (changed == 0) { st_agent(&f->strict, 1); done = 1; }
            else if (tot <= tol) done = 1;
            if (done) st_agent(&f->done, 1);
            leave = done || it_done >= max_iter || ring_full;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                             // every wave's centre and flag stores are acknowledged
    if (tid == 0) {
        const unsigned long long g = (gen_next << 8) | (unsigned long long)leave;
        for (int r = 0; r < KMP_GENREP; ++r) st_agent(genrep + (size_t)r * KMP_GENREP_STRIDE, g);
        st_agent(&f->gen, g);
    }
}

template <int PT>
__global__ __launch_bounds__(256, PT == 2 ? 3 : PT == 4 ? 2 : 1) void k_km_persist(const double* __restrict__ X, int n, double* B, int k,
                                                    const double* __restrict__ box, KmFlags* __restrict__ f,
                                                    unsigned long long* __restrict__ acc, KmTail T) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ double s_mub[4];
    __shared__ int s_cnt[4];
    __shared__ KmTailSh S;
    __shared__ unsigned long long s_gen;
    double* sB = (double*)smem;
    unsigned long long* sA = (unsigned long long*)(sB + 4 * k);
    double* slb = (double*)(sA + 4 * k);
    int* cand = (int*)(slb + k);
    const int tid = threadIdx.x, lane = tid & 63;
    asm volatile("" :: "s"(X), "s"(n), "s"(B), "s"(k), "s"(box), "s"(f), "s"(acc), "s"(T.C2), "s"(T.lab[0]), "s"(T.lab[1]), "s"(T.max_iter));
    const int rw = T.rw, i0 = (blockIdx.x * 4 + (tid >> 6)) * rw + lane, nrow = T.nrow;   // a wave owns T.rw <= 64 PT consecutive rows
    double x[PT][3];
    int lab[PT], pv[PT], pa[PT], pb[PT];
    bool own[PT];
#pragma unroll
    for (int q = 0; q < PT; ++q) {
        const int i = min(i0 + 64 * q, nrow - 1);
        x[q][0] = X[3 * (size_t)i]; x[q][1] = X[3 * (size_t)i + 1]; x[q][2] = X[3 * (size_t)i + 2];
        own[q] = 64 * q + lane < rw && T.perm[i] >= 0;             // one of the wave's rows and not a dummy
        if (64 * q + lane >= rw) { x[q][0] = x[q][1] = x[q][2] = __builtin_nan(""); }   // (kept out of the wave's box like a dummy)
        pa[q] = T.lab[0][i]; pb[q] = T.lab[1][i];
    }
    const double* bx = box + 6 * (size_t)blockIdx.x;
    const double bx6[6] = {bx[0], bx[1], bx[2], bx[3], bx[4], bx[5]};
    const double fscale = f->fix_scale, guard = f->guard, finv = f->fix_inv, tol = f->tol;
    const int cur0 = f->cur;
    if (f->done || f->reloc) return;                             // (launch-uniform: nothing in this launch has written them yet)
    double wb[6];
    km_wave_box<PT>(x, wb);                                      // once per launch: the points do not move
    int t = f->n_iter;
    if (t >= T.max_iter) return;
    unsigned long long want = (ld_agent(&f->gen) >> 8) + 1ull;     // no tail can run before this workgroup has arrived once
#pragma unroll
    for (int q = 0; q < PT; ++q) pv[q] = t == 0 ? -1 : ((t - 1) & 1) ? pb[q] : pa[q];
    // Centre rows: iteration `my` of this launch reads slot my of a ring nobody has read in this launch (slot 0 = B itself,
    // written before the launch), so ordinary cached loads are safe -- no stale copy can sit in an L2 -- and the 32 workgroups
    // of an XCD share one fetch.  (Agent-scope loads of ONE 4 KB table by 256 workgroups queue on the few memory channels that
    // hold it: 32 us per iteration measured, slower than a launch per iteration.)
    const size_t rstride = kmp_ring_stride(k);
    const unsigned long long* gen_mine = T.genrep + (size_t)(blockIdx.x % KMP_GENREP) * KMP_GENREP_STRIDE;
#ifdef CREG_STAMPS
    unsigned long long pp[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pp_t = wall_clock64(), pp_n = 0;
#endif
    for (int my = 0;; ++my) {
        const double* Bt = my == 0 ? B : T.ring + (size_t)my * rstride;
        for (int i = tid; i < 4 * k; i += 256) { sB[i] = Bt[i]; sA[i] = 0ull; }
        __syncthreads();
        KM_PP(0);
        int nc_dbg = 0;
        km_pruned_sweep<PT>(x, sB, slb, cand, k, bx6, guard, s_mub, s_cnt, lab, wb, &nc_dbg);
        KM_PP(1);
        int diff = 0;
#pragma unroll
        for (int q = 0; q < PT; ++q)
            if (own[q] && pv[q] != lab[q]) { ++diff; km_move(sA, lab[q], pv[q], x[q][0], x[q][1], x[q][2], fscale); }
#pragma unroll
        for (int q = 0; q < PT; ++q) pv[q] = lab[q];
        // Arrival without a shared word.  Hundreds of read-modify-writes (or write-through stores) of ONE word queue at the memory
        // side -- ~12 ns each: the per-wave changed-label count alone made the workgroups that had changes arrive 10 us after the
        // others (tests/measure/km_persist_phases.py).  Every workgroup instead drains its sum atomics and writes its OWN 8-byte
        // slot, tagged with the iteration's generation: (gen << 32) | "one of my labels changed".  Workgroup 0 runs every tail of
        // the launch: its 256 threads watch the slots (two each per look, one round trip) until all carry the generation.
        const int blk_changed = __syncthreads_or(diff != 0);
#ifdef CREG_STAMPS
        if (tid == 0 && t == 150 && blockIdx.x > 0 && blockIdx.x < 1024) g_km_it[2][blockIdx.x] = ((unsigned long long)nc_dbg << 32) | (unsigned)wave_sum(diff);
#endif
        for (int i = tid; i < 4 * k; i += 256) { const unsigned long long v = sA[i]; if (v) atomicAdd(&acc[i], v); }
        KM_PP(2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const bool last_wg = blockIdx.x == 0;
        int any_changed = blk_changed;
        if (!last_wg) { if (tid == 0) st_agent(T.slots + blockIdx.x, (want << 32) | (unsigned long long)(blk_changed != 0)); }
        else {
            int spins = 0, ok;
            bool gave_up = false;
            do {
                int mine = 1;
                for (int i = tid; i < (int)gridDim.x; i += 256)
                    if (i > 0) { const unsigned long long v = ld_agent(T.slots + i); mine &= (v >> 32) == want; any_changed |= mine ? (int)(v & 1ull) : 0; }
                ok = __syncthreads_and(mine);
                if (!ok) {
                    // block-uniform decision (`spins` is): ONE thread reads the abort word and the barrier spreads it -- per-thread reads
                    // could disagree and leave part of the workgroup in the barrier above while the rest had left the loop
                    int ab = ++spins >= T.spin_limit;
                    if ((spins & 63) == 0) ab |= __syncthreads_or(tid == 0 ? ld_agent(&f->abort) : 0);
                    if (ab) { gave_up = true; break; }
                }
            } while (!ok);
            if (gave_up) {
                // nothing of this attempt is used (the host sees `abort` and starts the call over): tell the others to leave NOW
                // -- generation `want` with the leave bit -- instead of letting each of them spin to its own limit, and return
                // without running a tail on partial sums
                if (tid == 0) {
                    st_agent(&f->abort, 1);
                    for (int r = 0; r < KMP_GENREP; ++r) st_agent(T.genrep + (size_t)r * KMP_GENREP_STRIDE, (want << 8) | 1ull);
                }
                return;
            }
            any_changed = __syncthreads_or(any_changed);
        }
        KM_PP(3);
        if (last_wg)
            km_tail_persist<256>(k, acc, T.C2, B, f, S, sB, T.max_iter, T.ring + (size_t)(my + 1) * rstride, my + 2 >= KMP_RING, T.genrep,
                                 cur0 ^ (my & 1), t + 1, want, finv, tol, any_changed);
#ifdef CREG_STAMPS
        if (last_wg) KM_PP(4);
        ++pp_n;
#endif
        if (tid == 0) {
            unsigned long long g = 0ull;
            int spins = 0;
            for (;;) {
                g = ld_agent(gen_mine);                          // one of 8 copies, a poll every ~0.3 us: the tail's own traffic is not queued behind the waiters
                if ((g >> 8) >= want) break;
                __builtin_amdgcn_s_sleep(10);                        // (8 or 32 copies, a poll every 0.3 or 1 us: no measurable difference)
                if (++spins >= T.spin_limit || ((spins & 63) == 0 && ld_agent(&f->abort))) { st_agent(&f->abort, 1); g = ~0ull; break; }
            }
            s_gen = g;
        }
        __syncthreads();
        KM_PP(5);
        const unsigned long long g = s_gen;
        if (g == ~0ull) return;                                  // gave up: the host reports it, nothing of this call is used
        if (g & 1ull) break;
        ++want; ++t;
    }
    // (s_gen is rewritten only after the next arrival's barriers, which every thread reaches after reading it)
#ifdef CREG_STAMPS
    if (tid == 0) { for (int p = 0; p < 6; ++p) atomicAdd(&g_km_pp[p], pp[p]); atomicAdd(&g_km_pp[15], pp_n);
                    if (blockIdx.x == 0) for (int p = 0; p < 6; ++p) g_km_pp[8 + p] = pp[p]; }
#endif
    int* out = (t & 1) ? T.lab[1] : T.lab[0];                    // the labels of iteration t, where the multi-launch path keeps them
#pragma unroll
    for (int q = 0; q < PT; ++q) if (64 * q + lane < rw) out[i0 + 64 * q] = lab[q];
}

// E-step, matrix-core form: v_mfma_f64_16x16x4_f64 evaluates a 16-centre x 16-point tile of |c|^2 - 2 x.c as
// D = C + A.B with A = the centres' (-2c0, -2c1, -2c2, 0) rows, B = the points' (x0; x1; x2; 0) columns and
// C = |c|^2 per ROW; the k-ordered fma chain of the instruction is the chain the VALU form spells out, so both
// forms give bit-identical distances and labels.
// Layout (gfx950 f64 MFMA): A[i=l&15][kk=l>>4], B[kk=l>>4][j=l&15], D[i=(l>>4)+4r][j=l&15]: a lane owns ONE point
// (column l&15) and sees 4 centres of every tile in ascending order, so its running first minimum needs no
// cross-lane work inside the sweep; the four lane groups of a point are combined once per point tile (two steps).
// A wave keeps up to 8 centre tiles (k <= 128) in registers and walks many point tiles (v1 reloaded the centres
// for every 16 points and reduced 4 rows x 16 lanes per tile: 78 us at n = 2^20, k = 128; this form: 53 us, the VALU
// form 50 us -- on CDNA4 one fp64 16x16x4 MFMA occupies the matrix pipe for 64 cycles, i.e. the fp64 matrix peak equals
// the fp64 vector peak, a quarter of every K=4 slot is padding and the argmin stays on the VALU, so the contraction
// only draws level here; kept selectable because the north star asks for it).
template <int NT>   // centre tiles held in registers (k <= 16 NT); 0: any k, centre operands re-read from LDS per tile
__global__ __launch_bounds__(256) void k_km_assign_mfma(const double* __restrict__ X, int n,
                                                        const double* B, int k,
                                                        int* labels_in, const int* prev_in,
                                                        KmFlags* __restrict__ f, int raw, unsigned long long* __restrict__ acc, KmTail T) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* sB = (double*)smem;                                  // [kpad][4], rows past k: (0, 0, 0, +inf)
    int* labels = labels_in;
    const int* prev = prev_in;
    if (f && !km_lloyd_entry<256>(X, n, k, acc, f, T, sB, labels, prev)) return;
    const int kpad = (k + 15) & ~15, ktiles = kpad / 16;
    unsigned long long* sA = (unsigned long long*)(sB + 5 * (size_t)kpad);       // [k][4] M-step sums (behind sB and sC)
    if (acc) for (int i = threadIdx.x; i < 4 * k; i += 256) sA[i] = 0ull;
    const double fscale = acc ? f->fix_scale : 0.0;
    // C operands as the lanes read them: sC[tile][lane group g][r] = |c|^2 of centre 16 tile + g + 4 r, 32 contiguous
    // bytes per (tile, group) -> two ds_read_b128 per MFMA instead of eight register copies of a resident table
    double* sC = sB + 4 * (size_t)kpad;                          // [ktiles][4][4]
    if (raw) {                                                   // `B` = centres (k,3): build the rows here (see k_km_assign)
        for (int j = threadIdx.x; j < kpad; j += 256) {
            if (j < k) make_b(B + 3 * j, sB + 4 * j);
            else { sB[4 * j] = 0.0; sB[4 * j + 1] = 0.0; sB[4 * j + 2] = 0.0; sB[4 * j + 3] = INFINITY; }
        }
        __syncthreads();
    } else
        for (int i = threadIdx.x; i < 4 * kpad; i += 256) sB[i] = (i >> 2) < k ? B[i] : ((i & 3) == 3 ? INFINITY : 0.0);
    for (int i = threadIdx.x; i < 16 * ktiles; i += 256) {
        const int t = i >> 4, gg = (i >> 2) & 3, r = i & 3, j = 16 * t + gg + 4 * r;
        sC[i] = j >= k ? INFINITY : raw ? sB[4 * j + 3] : B[4 * j + 3];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, g = lane >> 4, col = lane & 15;
    constexpr int NR = NT > 0 ? NT : 1;
    double at[NR];
    if (NT > 0) {
#pragma unroll
        for (int t = 0; t < NR; ++t) at[t] = g < 3 ? sB[4 * (16 * min(t, ktiles - 1) + col) + g] : 0.0;
    }
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4, ntile = (n + 15) >> 4;
    int diff = 0;
    double bnext = 0.0;
    if (wid < ntile) bnext = g < 3 ? X[3 * (size_t)min(wid * 16 + col, n - 1) + g] : 0.0;
    for (int tile = wid; tile < ntile; tile += nw) {
        const double b = bnext;
        if (tile + nw < ntile) bnext = g < 3 ? X[3 * (size_t)min((tile + nw) * 16 + col, n - 1) + g] : 0.0;
        double best = INFINITY;
        int lab = 0x7fffffff;
        int opq = 4 * g;                                         // opaque per iteration: keeps the C-operand reads inside the
        asm volatile("" : "+v"(opq));                            // loop (hoisted, they become 80 AGPRs + accvgpr moves)
        // the lane's running first minimum; `lab` holds the CODE 4 tile + r of the winner (a compile-time constant per
        // element in the register-resident form, so the select takes an inline constant) and becomes the centre index
        // 16 tile + g + 4 r after the sweep
        auto track = [&](const double4v& c, int t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {                        // centres 16 t + g + 4 r: ascending, strict < keeps the first
                const bool lt = c[r] < best;
                best = lt ? c[r] : best;
                lab = lt ? 4 * t + r : lab;
            }
        };
        if (NT > 0) {
#pragma unroll
            for (int t0 = 0; t0 < NR; t0 += 2) {                 // two independent MFMAs in flight, then their tracking
                double4v c[2];
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    if (t0 + u < NR) {
                        const int tt = min(t0 + u, ktiles - 1);  // tiles past k repeat the last one: never smaller, never first
                        c[u] = *(const double4v*)(sC + 16 * tt + opq);
                        c[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(at[t0 + u], b, c[u], 0, 0, 0);
                    }
#pragma unroll
                for (int u = 0; u < 2; ++u) if (t0 + u < min(NR, ktiles)) track(c[u], t0 + u);
            }
        } else {
            for (int t = 0; t < ktiles; ++t) {
                const double a = g < 3 ? sB[4 * (16 * t + col) + g] : 0.0;
                double4v c = *(const double4v*)(sC + 16 * t + opq);
                c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
                track(c, t);
            }
        }
        lab = lab == 0x7fffffff ? lab : 16 * (lab >> 2) + g + 4 * (lab & 3);
#pragma unroll
        for (int off = 16; off <= 32; off <<= 1) {               // the point's four lane groups: (value, index) lexicographic
            const double ov = __shfl_xor(best, off, 64);
            const int oi = __shfl_xor(lab, off, 64);
            const bool take = (ov < best) || (ov == best && oi < lab);
            best = take ? ov : best; lab = take ? oi : lab;
        }
        const int i = tile * 16 + col;
        if (i < n) {
            if (g == 0) labels[i] = lab;
            if (prev) {
                // the point's lane groups hold x, y, z (b) and nothing: when its label changed, group g moves coordinate g
                // (or the count) from the old cluster's sum to the new one's -- one LDS atomic pair each
                const int pv = prev[i];
                if (pv != lab) {
                    if (g == 0) ++diff;
                    if (acc) {
                        const unsigned long long v = g < 3 ? km_fix(b, fscale) : 1ull;
                        atomicAdd(&sA[4 * lab + g], v);
                        if (pv >= 0) atomicAdd(&sA[4 * pv + g], 0ull - v);
                    }
                }
            }
        }
    }
    if (prev && f) {
        diff = wave_sum(diff);
        if (lane == 0 && diff) atomicAdd(&f->changed, diff);
    }
    if (acc) km_flush_and_tail<256>(sA, X, n, k, acc, f, T, sB);
}

__global__ __launch_bounds__(1024) void k_km_finish(const double* __restrict__ X, int n,
                                                    const int* __restrict__ labels, int k,
                                                    const double* __restrict__ C2, KmFlags* __restrict__ f,
                                                    double* __restrict__ centers, double* __restrict__ inertia,
                                                    int* __restrict__ n_iter, int raw) {
    // raw: X is the caller's frame and is centred here (the same subtraction k_km_center stores: identical values) -- the
    // pruned path keeps its centred copy in another order, and the inertia sum keeps the caller's point order
    __shared__ double sc[16];
    const double* C = C2 + (size_t)f->cur * 3 * k;
    const double m0 = raw ? f->mean[0] : 0.0, m1 = raw ? f->mean[1] : 0.0, m2 = raw ? f->mean[2] : 0.0;
    double s = 0;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const double* c = C + 3 * labels[i];
        const double a = (X[3 * (size_t)i] - m0) - c[0], b = (X[3 * (size_t)i + 1] - m1) - c[1], e = (X[3 * (size_t)i + 2] - m2) - c[2];
        s += (a * a + b * b) + e * e;
    }
    s = block_sum<double, 1024>(s, sc);
    if (threadIdx.x == 0) { inertia[0] = s; n_iter[0] = f->n_iter; }
    for (int j = threadIdx.x; j < 3 * k; j += 1024) centers[j] = C[j] + f->mean[j % 3];
}

// ---- one-launch Lloyd for small clouds, one workgroup per problem ----------------------------------
// For n <= 5120 (the 4096 / 5000-point frames of the reference) the centred frame (24 B/pt) and two
// 16-bit label arrays fit one CU's LDS, so a whole k_means() -- mean/tol, centring, up to max_iter Lloyd iterations with
// the strict-convergence / tol test, final E-step, inertia -- runs as ONE workgroup with no host
// involvement, and B independent problems (the sequences of a batch) are B workgroups of one launch.
// Every sum is formed in exactly the order of the multi-launch path above (k_km_stats: 1024 strided
// partials + tree; k_km_accumulate: 256 strided partials per cluster + tree; k_km_finalize; k_km_finish),
// so both paths give bit-identical labels, centres, inertia and iteration counts (tested).
constexpr int KMS_MAXB = 16;
struct KmBatch {
    const double* X[KMS_MAXB]; const double* init[KMS_MAXB];
    double* centers[KMS_MAXB]; int* labels[KMS_MAXB]; double* inertia[KMS_MAXB]; int* n_iter[KMS_MAXB];
};

__global__ __launch_bounds__(1024) void k_km_small(KmBatch A, int n, int k, int max_iter, double tol_rel,
                                                   char* __restrict__ ws, size_t ws_stride, int c_in_lds, int x_in_lds) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // x_in_lds: the centred frame lives in LDS (n <= 5120); else (n <= 16384: franka-sized frames) every use re-reads the
    // frame from global memory -- it stays in this XCD's L2 for the whole launch -- and subtracts the mean on the fly (the
    // same subtraction, the same bits), so those frames' k_means() is one asynchronous launch too
    double* Xc = (double*)smem;                               // [n][3] centred points (x_in_lds)
    const int nx = x_in_lds ? n : 0;
    __shared__ double sc[16];
    __shared__ double s_mean[3], s_tol;
    __shared__ unsigned long long accI[4 * 128];              // M-step sums: fixed-point x, y, z and the count per cluster (k <= 128)
    __shared__ double s_fscale, s_finv;
    __shared__ int s_changed, s_done, s_strict, s_it, s_nempty, s_argmax;
    __shared__ double s_dmax, s_fv[16];
    __shared__ int s_fi[16];
    const int z = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const double* X = A.X[z];
    char* w = ws + (size_t)z * ws_stride;                     // per-problem global scratch: C2 (6k), Cw (4k), far (n)
    double* far_d = (double*)w + 14 * k;
    // LDS: Xc [3n] | (-2c, |c|^2) rows [4k] | labels ping-pong as 16-bit values [2n] (k <= 128 here).  The
    // E-step reads every centre row and the accumulation sweeps read every label ceil(k/4) times per
    // iteration; from global memory each of those reads was a dependent round trip.
    double* Bm = Xc + 3 * (size_t)nx;
    unsigned short* lab[2] = {(unsigned short*)(Bm + 4 * k), (unsigned short*)(Bm + 4 * k) + n};
    // centre ping-pong C2 [2][k][3] (+ spare) and per-cluster sums Cw [k][4]: in LDS when they fit next to the
    // frame (every Lloyd iteration reads and writes them several times between barriers -- from global memory
    // each of those was a dependent ~1 us round trip inside the workgroup), else in the global scratch
    double* C2 = c_in_lds ? (double*)(smem + (((size_t)(3 * nx + 4 * k) * 8 + 4 * (size_t)n + 7) & ~(size_t)7)) : (double*)w;
    double* Cw = C2 + 10 * k;
    // ---- mean / tol (k_km_stats: the same sums in the same order) ----
    double var = 0;
    {
        double s0 = 0, s1 = 0, s2 = 0;
        for (int i = tid; i < n; i += 1024) { s0 += X[3 * (size_t)i]; s1 += X[3 * (size_t)i + 1]; s2 += X[3 * (size_t)i + 2]; }
        s0 = block_sum<double, 1024>(s0, sc); s1 = block_sum<double, 1024>(s1, sc); s2 = block_sum<double, 1024>(s2, sc);
        if (tid == 0) { s_mean[0] = s0 / (double)n; s_mean[1] = s1 / (double)n; s_mean[2] = s2 / (double)n; }
    }
    __syncthreads();
    double amax = 0;
    {
        const double m0 = s_mean[0], m1 = s_mean[1], m2 = s_mean[2];
        double s0 = 0, s1 = 0, s2 = 0;
        for (int i = tid; i < n; i += 1024) {
            const double t0 = X[3 * (size_t)i] - m0, t1 = X[3 * (size_t)i + 1] - m1, t2 = X[3 * (size_t)i + 2] - m2;
            s0 = fma(t0, t0, s0); s1 = fma(t1, t1, s1); s2 = fma(t2, t2, s2);
            amax = fmax(amax, fmax(fabs(t0), fmax(fabs(t1), fabs(t2))));
        }
        s0 = block_sum<double, 1024>(s0, sc); s1 = block_sum<double, 1024>(s1, sc); s2 = block_sum<double, 1024>(s2, sc);
        if (tid == 0) { var += s0 / (double)n; var += s1 / (double)n; var += s2 / (double)n; }
    }
    for (int off = 32; off >= 1; off >>= 1) amax = fmax(amax, __shfl_xor(amax, off, 64));
    __syncthreads();
    if (lane == 0) s_fv[wv] = amax;
    if (tid < 4 * k) accI[tid] = 0ull;
    __syncthreads();
    if (tid == 0) {
        double r = 0;
        for (int q = 0; q < 16; ++q) r = fmax(r, s_fv[q]);
        s_fscale = km_fix_scale(r, n); s_finv = 1.0 / s_fscale;
        s_tol = (var / 3.0) * tol_rel; s_done = 0; s_strict = 0; s_it = 0; s_changed = 0; s_nempty = 0;
    }
    // ---- centre (k_km_center) ----
    if (x_in_lds) for (int i = tid; i < 3 * n; i += 1024) Xc[i] = X[i] - s_mean[i % 3];
    const double mean0 = s_mean[0], mean1 = s_mean[1], mean2 = s_mean[2];
    auto xc = [&](int i, int d) -> double { return x_in_lds ? Xc[3 * i + d] : X[3 * (size_t)i + d] - (d == 0 ? mean0 : d == 1 ? mean1 : mean2); };
    for (int i = tid; i < n; i += 1024) lab[1][i] = 0xFFFF;    // iteration 0 compares against lab[1] = "no label"
    if (tid < k) {
        double c[3];
        for (int d = 0; d < 3; ++d) { c[d] = A.init[z][3 * tid + d] - s_mean[d]; C2[3 * tid + d] = c[d]; }
        make_b(c, Bm + 4 * tid);
    }
    __threadfence_block();
    __syncthreads();
    int cur = 0;                                             // centre buffer holding the current centres
    __shared__ double s_sh[128];
    for (int it = 0; it < max_iter; ++it) {
        unsigned short* lcur = lab[it & 1];
        const unsigned short* lprev = lab[(it + 1) & 1];
        // ---- E-step (k_km_assign): four points per thread share every centre row read; a point whose label changed
        //      moves from the old cluster's exact integer sums to the new one's (accI persists over the iterations)
        int diff = 0;
        for (int i0 = 0; i0 < n; i0 += 4096) {
            double x[4][3], best[4];
            int lb[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = min(i0 + 1024 * q + tid, n - 1);
                x[q][0] = xc(i, 0); x[q][1] = xc(i, 1); x[q][2] = xc(i, 2);
                best[q] = fma(x[q][2], Bm[2], fma(x[q][1], Bm[1], fma(x[q][0], Bm[0], Bm[3])));
                lb[q] = 0;
            }
            for (int j = 1; j < k; ++j) {
                const double b0 = Bm[4 * j], b1 = Bm[4 * j + 1], b2 = Bm[4 * j + 2], b3 = Bm[4 * j + 3];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const double d = fma(x[q][2], b2, fma(x[q][1], b1, fma(x[q][0], b0, b3)));
                    if (d < best[q]) { best[q] = d; lb[q] = j; }
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = i0 + 1024 * q + tid;
                if (i < n) {
                    const int pv = lprev[i];
                    lcur[i] = (unsigned short)lb[q];
                    if (pv != lb[q]) { ++diff; km_move(accI, lb[q], pv == 0xFFFF ? -1 : pv, x[q][0], x[q][1], x[q][2], s_fscale); }
                }
            }
        }
        if (diff) atomicAdd(&s_changed, diff);
        __threadfence_block();
        __syncthreads();
        // ---- M-step tail (km_mstep_tail): thread j owns cluster j (k <= 128) ----
        const double* Cold = C2 + (size_t)cur * 3 * k;
        double* Cnew = C2 + (size_t)(cur ^ 1) * 3 * k;
        double a4[4] = {0, 0, 0, 1};
        if (tid < k) {
            for (int d = 0; d < 3; ++d) a4[d] = (double)(long long)accI[4 * tid + d] * s_finv;
            a4[3] = (double)(long long)accI[4 * tid + 3];
            for (int d = 0; d < 4; ++d) Cw[4 * tid + d] = a4[d];
            if (a4[3] == 0.0) atomicAdd(&s_nempty, 1);
        }
        __threadfence_block();
        __syncthreads();
        if (s_nempty > 0) {                                    // block-uniform; rare
            double dmax = 0;
            for (int i = tid; i < n; i += 1024) {
                const double* c = Cold + 3 * lcur[i];
                const double a = xc(i, 0) - c[0], b = xc(i, 1) - c[1], e = xc(i, 2) - c[2];
                const double d = (a * a + b * b) + e * e;
                far_d[i] = d;
                dmax = fmax(dmax, d);
            }
            for (int off = 32; off >= 1; off >>= 1) dmax = fmax(dmax, __shfl_xor(dmax, off, 64));
            __syncthreads();
            if (lane == 0) sc[wv] = dmax;
            __syncthreads();
            if (tid == 0) { double m = 0; for (int i = 0; i < 16; ++i) m = fmax(m, sc[i]); s_dmax = m; }
            __syncthreads();
            if (s_dmax > 0) {
                for (int j = 0; j < k; ++j) {
                    if (Cw[4 * j + 3] != 0.0) continue;
                    double bv = -1; int bi = 0x7fffffff;
                    for (int i = tid; i < n; i += 1024) if (far_d[i] > bv) { bv = far_d[i]; bi = i; }
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
                        for (int d = 0; d < 3; ++d) { Cw[4 * old + d] -= xc(bi, d); Cw[4 * j + d] = xc(bi, d); }
                        Cw[4 * j + 3] = 1.0; Cw[4 * old + 3] -= 1.0;
                    }
                    __threadfence_block();
                    __syncthreads();
                }
            }
            if (tid == 0) { int am = 0; for (int j = 1; j < k; ++j) if (Cw[4 * j + 3] > Cw[4 * am + 3]) am = j; s_argmax = am; }
            __threadfence_block();
            __syncthreads();
            if (tid < k) for (int d = 0; d < 4; ++d) a4[d] = Cw[4 * tid + d];
        }
        if (tid < k) {
            const int j = tid;
            double c[3];
            if (a4[3] > 0) { const double alpha = 1.0 / a4[3]; for (int d = 0; d < 3; ++d) c[d] = a4[d] * alpha; }
            else { const double alpha = 1.0 / Cw[4 * s_argmax + 3]; for (int d = 0; d < 3; ++d) c[d] = Cw[4 * s_argmax + d] * alpha; }
            double s = 0;
            for (int d = 0; d < 3; ++d) { Cnew[3 * j + d] = c[d]; const double t = c[d] - Cold[3 * j + d]; s += t * t; }
            const double sh = sqrt(s);
            s_sh[j] = sh * sh;
            make_b(c, Bm + 4 * j);
        }
        __threadfence_block();
        __syncthreads();
        if (tid == 0) {
            double tot = 0;
            for (int j0 = 0; j0 < k; j0 += 8) {                 // fixed order (cluster index); loads of a group issued together
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = s_sh[min(j0 + u, 127)];
#pragma unroll
                for (int u = 0; u < 8; ++u) if (j0 + u < k) tot += v[u];
            }
            s_it = it + 1;
            if (s_changed == 0) { s_strict = 1; s_done = 1; }
            else if (tot <= s_tol) s_done = 1;
            s_changed = 0; s_nempty = 0;
        }
        cur ^= 1;
        __syncthreads();
        if (s_done) break;
    }
    // ---- final E-step when not strictly converged, labels into the caller's buffer ----
    unsigned short* last = lab[(s_it - 1) & 1];
    if (!s_strict) {
        for (int i = tid; i < n; i += 1024) {
            const double x0 = xc(i, 0), x1 = xc(i, 1), x2 = xc(i, 2);
            double best = fma(x2, Bm[2], fma(x1, Bm[1], fma(x0, Bm[0], Bm[3])));
            int lb = 0;
            for (int j = 1; j < k; ++j) {
                const double d = fma(x2, Bm[4 * j + 2], fma(x1, Bm[4 * j + 1], fma(x0, Bm[4 * j], Bm[4 * j + 3])));
                if (d < best) { best = d; lb = j; }
            }
            last[i] = (unsigned short)lb;
        }
        __threadfence_block();
        __syncthreads();
    }
    for (int i = tid; i < n; i += 1024) A.labels[z][i] = (int)last[i];
    // ---- inertia, un-centred centres (k_km_finish) ----
    const double* C = C2 + (size_t)cur * 3 * k;
    double s = 0;
    for (int i = tid; i < n; i += 1024) {
        const double* c = C + 3 * last[i];
        const double a = xc(i, 0) - c[0], b = xc(i, 1) - c[1], e = xc(i, 2) - c[2];
        s += (a * a + b * b) + e * e;
    }
    s = block_sum<double, 1024>(s, sc);
    if (tid == 0) { A.inertia[z][0] = s; A.n_iter[z][0] = s_it; }
    for (int j = tid; j < 3 * k; j += 1024) A.centers[z][j] = C[j] + s_mean[j % 3];
}

static size_t kms_stride(int64_t n, int k) { return align_up(sizeof(double) * (14 * (size_t)k + n), 256); }

// ---- grouping by label + inverse-pose change of frame ------------------------------------------
// Every index is a compile-time constant (round 5): the row exchange of the partial pivoting is a conditional swap of the pivot row
// with each later row in turn -- as a run-time row index it put the 4 x 8 matrix into scratch memory (272 bytes per lane, 93 scratch
// instructions in k_group_scatter / k_group_scatter_big).  Same operations in the same order: bit-identical.
__device__ __forceinline__ void inv4x4(const double* M, double* I) {      // Gauss-Jordan, partial pivoting
    double a[4][8];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) { a[r][c] = M[4 * r + c]; a[r][4 + c] = (r == c) ? 1.0 : 0.0; }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        int p = c;                                        // first row of the largest |entry| in column c, rows c .. 3
        double best = fabs(a[c][c]);
#pragma unroll
        for (int r = c + 1; r < 4; ++r) { const double v = fabs(a[r][c]); if (v > best) { best = v; p = r; } }
#pragma unroll
        for (int r = c + 1; r < 4; ++r) {
            const bool sw = p == r;
#pragma unroll
            for (int q = 0; q < 8; ++q) { const double t = a[c][q], u = a[r][q]; a[c][q] = sw ? u : t; a[r][q] = sw ? t : u; }
        }
        const double inv = 1.0 / a[c][c];
#pragma unroll
        for (int q = 0; q < 8; ++q) a[c][q] *= inv;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (r == c) continue;
            const double fct = a[r][c];
#pragma unroll
            for (int q = 0; q < 8; ++q) a[r][q] = fma(-fct, a[c][q], a[r][q]);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) I[4 * r + c] = a[r][4 + c];
}

// grid.y = problem of a batch (the per-frame pointers come from the table)
constexpr int GRP_MAXB = 16;
struct GroupBatch { const double* X[GRP_MAXB]; const int* labels[GRP_MAXB]; const double* M[GRP_MAXB]; double* out[GRP_MAXB]; int* off[GRP_MAXB]; };

__global__ __launch_bounds__(1024) void k_group_offsets(GroupBatch G, int n, int k) {
    const int* __restrict__ labels = G.labels[blockIdx.y];
    int* __restrict__ off = G.off[blockIdx.y];
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* cnt = (int*)smem;
    for (int j = threadIdx.x; j <= k; j += 1024) cnt[j] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 1024) atomicAdd(&cnt[labels[i]], 1);
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int j = 0; j < k; ++j) { const int c = cnt[j]; off[j] = run; run += c; }
        off[k] = run;
    }
}

// one wave per cluster walks the labels in order: ballot + prefix popcount gives the stable slot
__global__ __launch_bounds__(64) void k_group_scatter(GroupBatch G, int n, int m_is_inverse) {
    const double* __restrict__ X = G.X[blockIdx.y];
    const int* __restrict__ labels = G.labels[blockIdx.y];
    const int* __restrict__ off = G.off[blockIdx.y];
    const double* __restrict__ M = G.M[blockIdx.y];
    double* __restrict__ out = G.out[blockIdx.y];
    const int j = blockIdx.x, lane = threadIdx.x;
    __shared__ double I[16];                              // inv(M_j), by one lane (no device scratch)
    if (m_is_inverse) { if (lane < 16) I[lane] = M[16 * j + lane]; }     // the caller inverted the pose (np.linalg.inv on the host)
    else if (lane == 0) inv4x4(M + 16 * j, I);
    __syncthreads();
    int pos = off[j];
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        const bool mine = (i < n) && (labels[i] == j);
        const unsigned long long m = __ballot(mine);
        if (mine) {
            const int slot = pos + __popcll(m & ((1ull << lane) - 1ull));
            const double p0 = X[3 * (size_t)i], p1 = X[3 * (size_t)i + 1], p2 = X[3 * (size_t)i + 2];
#pragma unroll
            for (int a = 0; a < 3; ++a)
                out[3 * (size_t)slot + a] = fma(I[4 * a + 2], p2, fma(I[4 * a + 1], p1, I[4 * a] * p0)) + I[4 * a + 3];
        }
        pos += __popcll(m);
    }
}

// ---- the same grouping for large frames (n > 16384): counts over many workgroups, one 1024-thread workgroup per cluster
// for the ordered compaction (the one-wave-per-cluster walk above took 1.85 ms at n = 262144, k = 128) ----
__global__ __launch_bounds__(1024) void k_group_count(const int* __restrict__ labels, int n, int k, int* __restrict__ off) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* cnt = (int*)smem;
    for (int j = threadIdx.x; j < k; j += 1024) cnt[j] = 0;
    __syncthreads();
    const int i0 = blockIdx.x * 4096 + threadIdx.x;
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int i = i0 + 1024 * q; if (i < n) atomicAdd(&cnt[labels[i]], 1); }
    __syncthreads();
    for (int j = threadIdx.x; j < k; j += 1024) { const int c = cnt[j]; if (c) atomicAdd(&off[j + 1], c); }      // integers: order independent
}

// off[0] = 0, off[j + 1] = count of cluster j  ->  off[j + 1] = sum of the counts up to j  (k <= 4096: four per thread)
__global__ __launch_bounds__(1024) void k_group_excl(int* __restrict__ off, int k) {
    __shared__ int wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int c[4], run = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int j = 4 * tid + q; c[q] = j < k ? off[j + 1] : 0; run += c[q]; }
    int inc = run;
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    int base = inc - run;
    for (int w = 0; w < wv; ++w) base += wsum[w];
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int j = 4 * tid + q; base += c[q]; if (j < k) off[j + 1] = base; }
    if (tid == 0) off[0] = 0;
}

__global__ __launch_bounds__(1024) void k_group_scatter_big(const double* __restrict__ X, int n, const int* __restrict__ labels,
                                                            const int* __restrict__ off, const double* __restrict__ M,
                                                            double* __restrict__ out, int m_is_inverse) {
    const int j = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    __shared__ double I[16];
    __shared__ int wtot[2][16];
    if (m_is_inverse) { if (tid < 16) I[tid] = M[16 * j + tid]; }
    else if (tid == 0) inv4x4(M + 16 * j, I);
    __syncthreads();
    int pos = off[j];
    for (int base = 0, r = 0; base < n; base += 4096, ++r) {
        const int i0 = base + 4 * tid;                        // four consecutive points per thread: slots stay in index order
        bool fl[4];
        int c = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) { fl[q] = (i0 + q < n) && labels[min(i0 + q, n - 1)] == j; c += fl[q]; }
        int inc = c;
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
        if (lane == 63) wtot[r & 1][wv] = inc;
        __syncthreads();                                      // one barrier per round: the table alternates
        int before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) { const int v = wtot[r & 1][w]; before += w < wv ? v : 0; total += v; }
        int slot = pos + before + inc - c;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (fl[q]) {
                const size_t i = (size_t)(i0 + q);
                const double p0 = X[3 * i], p1 = X[3 * i + 1], p2 = X[3 * i + 2];
#pragma unroll
                for (int a = 0; a < 3; ++a)
                    out[3 * (size_t)slot + a] = fma(I[4 * a + 2], p2, fma(I[4 * a + 1], p1, I[4 * a] * p0)) + I[4 * a + 3];
                ++slot;
            }
        pos += total;
    }
}

static int seg_count(int64_t n) { int s = (int)((n + 16383) / 16384); return s < 1 ? 1 : (s > 64 ? 64 : s); }

// pruned E-step: points per workgroup and grid bits per axis of the spatial order
// Geometry of the pruned E-step: G workgroups of 4 waves, each wave RW consecutive rows of the sorted copy (PT = ceil(RW / 64) row
// slots per lane), rows = 4 G RW >= n + 64 (RW - 1) -- every one of the 64 Morton ranges starts on a WAVE boundary (the per-wave
// filter is the level that must not straddle two ranges; 3 % dummy rows at n = 262144 instead of 12 % with workgroup-aligned runs).
// G is a multiple of 256 above 256, so that every CU holds the same number of workgroups of the persistent kernel: two slots per
// lane while that gives at most 768 workgroups (three per CU), then four (512), then eight.  (Measured at n = 262144: 768 x 4 x 88
// rows 16.0 us per Lloyd iteration, 576 x 4 x 128 -- 64 CUs with three workgroups -- 16.2: the late arrivals are the waves whose
// box lies in the sparse end of the Morton order and keeps 40-55 centres, not the CUs with one workgroup more.)
struct KmGeom { int G, RW, PT; };
static KmGeom km_geometry(int64_t n) {
    static int forced = -1;                                      // measurement knob CREG_KMP_PT (2 | 4 | 8)
    if (forced < 0) { const char* e = getenv("CREG_KMP_PT"); forced = e ? atoi(e) : 0; }
    for (int pt = 2; pt <= 8; pt *= 2) {
        if ((forced == 2 || forced == 4 || forced == 8) && pt != forced) continue;
        int64_t g = ((n - 64 + 64 * pt - 1) / (64 * pt) + 64 + 3) / 4;                          // 4 g - 64 waves of 64 pt rows cover n - 64
        if (g < 32) g = 32;
        if (g > 256) g = (g + 255) / 256 * 256;
        const int64_t cap = pt == 2 ? 768 : 512;
        if (g <= cap || pt == 8 || pt == forced) {
            const int64_t rw = (n - 64 + (4 * g - 64) - 1) / (4 * g - 64);
            return KmGeom{(int)g, (int)(rw < 1 ? 1 : rw), pt};
        }
    }
    return KmGeom{32, 1, 2};
}
static int km_pruned_pt(int64_t n) { return km_geometry(n).PT; }
static int km_spin_limit() {                                     // polls before a waiting workgroup gives up (test knob CREG_KM_SPIN_LIMIT: 1 forces the fallback)
    static int v = -1;
    if (v < 0) { const char* e = getenv("CREG_KM_SPIN_LIMIT"); v = e && atoi(e) > 0 ? atoi(e) : (1 << 20); }
    return v;
}
static bool km_persist_enabled() {
    static int v = -1;                                           // measurement knob: CREG_KM_PERSIST=0 keeps one launch per iteration
    if (v < 0) { const char* e = getenv("CREG_KM_PERSIST"); v = e ? atoi(e) != 0 : 1; }
    return v != 0;
}
static int km_cell_bits(int64_t n) { return n >= 131072 ? 6 : n >= 16384 ? 5 : n >= 2048 ? 4 : 3; }   // ~8 points per cell of a surface
static bool km_pruned_enabled() {
    static int v = -1;                                           // measurement knob (tests/measure): CREG_KM_PRUNE=0 keeps the full sweep
    if (v < 0) { const char* e = getenv("CREG_KM_PRUNE"); v = e ? atoi(e) != 0 : 1; }
    return v != 0;
}

struct KmLayout { size_t xc, c2, b, cw, part, far, segv, segi, prev, lab2, flags, lab3, perm, inv, key, cell, box, ring, genrep, slots, total; };
// rows of the sorted copy: the points plus the dummies that align the 64 Morton ranges to the waves' runs
static int64_t km_rows(int64_t n) { const KmGeom g = km_geometry(n); return (int64_t)4 * g.G * g.RW; }
static KmLayout km_layout(int64_t n, int k) {
    KmLayout L; size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o = align_up(o + bytes, 256); return r; };
    const int64_t nr = km_rows(n);                                // rows of the sorted copy: the points plus the dummies that align the ranges
    L.xc = take(sizeof(double) * 3 * nr); L.c2 = take(sizeof(double) * 6 * k); L.b = take(sizeof(double) * 4 * k);
    L.cw = take(sizeof(double) * 4 * k); L.part = take(sizeof(unsigned long long) * 4 * k);      // part: the int64 accumulators
    L.far = take(sizeof(double) * n);
    // relocation pass: one entry per workgroup of the E-step launch (at most n / 64: the matrix-core form; the pruned form's
    // grid covers the dummy rows too)
    const size_t nseg = (size_t)((n + 63) / 64) + (size_t)(nr / 512) + 1025;
    L.segv = take(sizeof(double) * nseg); L.segi = take(sizeof(int) * nseg);
    L.prev = take(sizeof(int) * n); L.lab2 = take(sizeof(int) * nr);
    L.flags = take(sizeof(KmFlags));
    L.lab3 = take(sizeof(int) * nr); L.perm = take(sizeof(int) * nr); L.inv = take(sizeof(int) * n); L.key = take(sizeof(int) * n);
    L.cell = take(sizeof(int) * (((size_t)1 << (3 * km_cell_bits(n))) + 64));   // cell counts / first rows, then the 64 ranges' totals
    L.box = take(sizeof(double) * 6 * (size_t)((nr + 511) / 512 + 1024));   // one per workgroup of the pruned E-step
    L.ring = take(sizeof(double) * kmp_ring_stride(k) * KMP_RING);
    L.genrep = take(sizeof(unsigned long long) * KMP_GENREP_STRIDE * KMP_GENREP);
    L.slots = take(sizeof(unsigned long long) * 1024);          // (the persistent grid is at most 512 workgroups; genrep and slots are zeroed together)
    L.total = o;
    return L;
}

static int km_points_per_thread(int n) {
    static int forced = -1;                                      // measurement knob (tests/measure): CREG_KM_PT = 1 | 2 | 4
    if (forced < 0) { const char* e = getenv("CREG_KM_PT"); forced = e ? atoi(e) : 0; }
    if (forced == 1 || forced == 2 || forced == 4) return forced;
    return n >= 32768 ? 2 : 1;                                  // measured at n = 262144, k = 128: 13.5 / 11.7 / 15.0 us for 1 / 2 / 4
}

static int launch_assign(const double* X, int n, const double* B, int k, int* labels, const int* prev,
                         KmFlags* f, int use_mfma, hipStream_t s, int raw = 0, unsigned long long* acc = nullptr,
                         KmTail T = KmTail{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, {nullptr, nullptr}, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0, 0}) {
    constexpr int LDS_MAX = 128 * 1024;
    if (use_mfma) {
        const int ntile = cdiv(n, 16);
        int blocks = cdiv(ntile, 4);
        if (blocks > 2048) blocks = 2048;                        // a wave then walks several point tiles with its centres in registers
        const size_t smem = sizeof(double) * 5 * ((k + 15) & ~15) + sizeof(unsigned long long) * 4 * k;   // centre rows + the C-operand table + M-step sums
#define CREG_KM_MFMA(NT_)                                                                                                        \
        do {                                                                                                                      \
            if (smem > 48 * 1024 && hipFuncSetAttribute((const void*)k_km_assign_mfma<NT_>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX) != hipSuccess) return 1; \
            hipLaunchKernelGGL(k_km_assign_mfma<NT_>, dim3(blocks), dim3(256), smem, s, X, n, B, k, labels, prev, f, raw, acc, T);  \
        } while (0)
        if (k <= 16) CREG_KM_MFMA(1);
        else if (k <= 32) CREG_KM_MFMA(2);
        else if (k <= 64) CREG_KM_MFMA(4);
        else if (k <= 128) CREG_KM_MFMA(8);
        else CREG_KM_MFMA(0);
#undef CREG_KM_MFMA
    } else {
        const size_t smem = sizeof(double) * 8 * k;
        const int pt = km_points_per_thread(n);
#define CREG_KM_VALU(PT_)                                                                                                        \
        do {                                                                                                                      \
            if (smem > 48 * 1024 && hipFuncSetAttribute((const void*)k_km_assign<PT_>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX) != hipSuccess) return 1; \
            hipLaunchKernelGGL(k_km_assign<PT_>, dim3(cdiv(n, 256 * PT_)), dim3(256), smem, s, X, n, B, k, labels, prev, f, raw, acc, T); \
        } while (0)
        if (pt == 4) CREG_KM_VALU(4);
        else if (pt == 2) CREG_KM_VALU(2);
        else CREG_KM_VALU(1);
#undef CREG_KM_VALU
    }
    return 0;
}

// the persistent kernel's grid must be resident all at once: 0 = not possible here (the caller keeps one launch per iteration)
template <int PT>
static int persist_fits(int blocks, size_t smem) {
    if (smem > 48 * 1024 && hipFuncSetAttribute((const void*)k_km_persist<PT>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess) return 0;
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)k_km_persist<PT>, 256, smem) != hipSuccess) return 0;
    return blocks <= (per_cu > 4 ? 4 : per_cu) * cus && blocks <= 1024;   // at most four per CU are counted on; one arrival slot each
}
static int launch_persist(const double* Xs, int n, double* B, int k, const double* box, KmFlags* f, unsigned long long* acc,
                          const KmTail& T, int pt, hipStream_t s, bool probe_only) {
    const size_t smem = sizeof(double) * 9 * k + sizeof(int) * 5 * k;   // centre rows, M-step sums, lower bounds, the workgroup's and the four waves' survivor lists
    const int blocks = T.nrow / (4 * T.rw);
#define CREG_KM_PERSIST(PT_)                                                                                                     \
    do {                                                                                                                          \
        if (!persist_fits<PT_>(blocks, smem)) return 1;                                                                           \
        if (!probe_only) hipLaunchKernelGGL(k_km_persist<PT_>, dim3(blocks), dim3(256), smem, s, Xs, n, B, k, box, f, acc, T);     \
    } while (0)
    if (pt == 8) CREG_KM_PERSIST(8);
    else if (pt == 4) CREG_KM_PERSIST(4);
    else CREG_KM_PERSIST(2);
#undef CREG_KM_PERSIST
    return 0;
}

static int launch_assign_pruned(const double* Xs, int n, const double* B, int k, const double* box, int* labels, KmFlags* f,
                                int lloyd, unsigned long long* acc, const KmTail& T, hipStream_t s) {
    const size_t smem = sizeof(double) * 9 * k + sizeof(int) * 5 * k;   // centre rows, M-step sums, lower bounds, the workgroup's and the four waves' survivor lists
#define CREG_KM_PRUNED(PT_)                                                                                                      \
    do {                                                                                                                          \
        if (smem > 48 * 1024 && hipFuncSetAttribute((const void*)k_km_assign_pruned<PT_>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess) return 1; \
        hipLaunchKernelGGL(k_km_assign_pruned<PT_>, dim3(T.nrow / (4 * T.rw)), dim3(256), smem, s, Xs, n, B, k, box, labels, f, lloyd, acc, T); \
    } while (0)
    const int pt = km_pruned_pt(n);
    if (pt == 8) CREG_KM_PRUNED(8);
    else if (pt == 4) CREG_KM_PRUNED(4);
    else CREG_KM_PRUNED(2);
#undef CREG_KM_PRUNED
    return 0;
}

}  // namespace creg
using namespace creg;

extern "C" size_t creg_kmeans_workspace_bytes(int64_t n, int32_t k) {
    if (n < 1 || k < 1) return 0;
    return km_layout(n, k).total;
}

constexpr int KM_RETRY_WITHOUT_PERSIST = 1000;
constexpr int KM_PERSIST_BACKOFF = 64;       // calls that skip the persistent kernel after an abandoned attempt
static int km_lloyd_run(const double* X, int64_t n, const double* init, int32_t k,
                        int32_t max_iter, double tol_rel, int32_t use_mfma, double* centers,
                        int32_t* labels, double* inertia, int32_t* n_iter, void* workspace,
                        size_t workspace_bytes, creg_stream_t stream, bool allow_persist) {
    CREG_REQUIRE(X && init && centers && labels && inertia && n_iter && workspace, "creg_kmeans_lloyd_f64: null pointer");
    CREG_REQUIRE(n >= 1 && n < (1ll << 31) && k >= 1 && k <= 1024 && max_iter >= 1,
                 "creg_kmeans_lloyd_f64: need 1 <= n < 2^31, 1 <= k <= 1024, max_iter >= 1");
    const KmLayout L = km_layout(n, k);
    CREG_REQUIRE(workspace_bytes >= L.total, "creg_kmeans_lloyd_f64: workspace too small (%zu < %zu)", workspace_bytes, L.total);
    hipStream_t s = (hipStream_t)stream;
    char* w = (char*)workspace;
    double* Xc = (double*)(w + L.xc); double* C2 = (double*)(w + L.c2); double* B = (double*)(w + L.b);
    double* Cw = (double*)(w + L.cw); unsigned long long* acc = (unsigned long long*)(w + L.part); double* far_d = (double*)(w + L.far);
    // VALU form: the E-step runs over a spatially sorted copy of the frame and prunes the centres per workgroup (labels in
    // the sorted order in two workspace buffers, put back in the caller's order at the end); matrix-core form: full sweep
    const bool pruned = !use_mfma && km_pruned_enabled() && n < (1ll << 30);
    int* lab[2] = {pruned ? (int*)(w + L.lab3) : labels, (int*)(w + L.lab2)};
    int* prev0 = (int*)(w + L.prev);
    int* perm = (int*)(w + L.perm); int* inv = (int*)(w + L.inv); double* box = (double*)(w + L.box);
    KmFlags* f = (KmFlags*)(w + L.flags);
    const KmGeom geo = km_geometry(n);
    CREG_REQUIRE(!pruned || (km_rows(n) >= n + 64 * (int64_t)(geo.RW - 1) && geo.RW <= 64 * geo.PT && km_rows(n) < (1ll << 31)),
                 "creg_kmeans_lloyd_f64: internal: geometry of the sorted copy (%d workgroups x 4 x %d rows)", geo.G, geo.RW);
    const int ni = (int)n, nrow = pruned ? (int)km_rows(n) : (int)n;
    CREG_HIP(hipMemsetAsync(acc, 0, sizeof(unsigned long long) * 4 * k, s));

    hipLaunchKernelGGL(k_km_stats, dim3(1), dim3(1024), 0, s, X, ni, tol_rel, f);
    if (pruned) {
        const int bits = km_cell_bits(n), ncell = 1 << (3 * bits);
        int* key = (int*)(w + L.key); int* cell = (int*)(w + L.cell); int* gtot = cell + ncell;
        CREG_HIP(hipMemsetAsync(cell, 0, sizeof(int) * ncell, s));
        hipLaunchKernelGGL(k_km_cell_count, dim3(cdiv(n, 256)), dim3(256), 0, s, X, ni, f, bits, key, cell);
        hipLaunchKernelGGL(k_km_cell_totals, dim3(64), dim3(256), 0, s, cell, ncell, gtot);
        hipLaunchKernelGGL(k_km_cell_scan, dim3(64), dim3(256), 0, s, cell, ncell, geo.RW, gtot);
        CREG_HIP(hipMemsetAsync(Xc, 0xFF, sizeof(double) * 3 * (size_t)nrow, s));       // dummy rows: NaN coordinates (ignored by the boxes' fmin / fmax) ...
        CREG_HIP(hipMemsetAsync(perm, 0xFF, sizeof(int) * (size_t)nrow, s));            // ... and perm = -1
        hipLaunchKernelGGL(k_km_cell_scatter, dim3(cdiv(n, 256)), dim3(256), 0, s, ni, key, cell, perm, inv);
    }
    hipLaunchKernelGGL(k_km_center, dim3(cdiv(n > k ? n : k, 256)), dim3(256), 0, s, X, ni, init, k, f, Xc, C2, B, prev0, pruned ? inv : nullptr);
    if (pruned) hipLaunchKernelGGL(k_km_boxes, dim3(geo.G), dim3(256), 0, s, Xc, nrow, 4 * geo.RW, box);
    CREG_LAUNCH_CHECK();
    // labels ping-pong between the caller's buffer and lab2 so "previous labels" needs no copy;
    // iteration `it` writes lab[it & 1] and compares with the buffer written by it - 1.
    int done = 0, n_done = 0;
    KmFlags host;
    const KmTail T{B, C2, Cw, far_d, (double*)(w + L.segv), (int*)(w + L.segi), {lab[0], lab[1]}, prev0, max_iter, pruned ? inv : nullptr,
                   (double*)(w + L.ring), (unsigned long long*)(w + L.genrep), (unsigned long long*)(w + L.slots), nrow, perm, km_spin_limit(), geo.RW};
    if (pruned) CREG_HIP(hipMemsetAsync(w + L.genrep, 0, L.slots + sizeof(unsigned long long) * 1024 - L.genrep, s));
    // Pruned form: ONE ordinary launch (an E-step, or the relocation a persistent launch left pending), then the persistent
    // kernel, which iterates until convergence / max_iter / the next empty cluster; one host round trip per such pair.
    const bool persist = allow_persist && pruned && km_persist_enabled() && launch_persist(Xc, ni, B, k, box, f, acc, T, km_pruned_pt(n), s, true) == 0;
    for (int round = 0; persist && n_done < max_iter && !done; ++round) {
        // every pair completes at least the ordinary launch's step (an iteration, or a pending relocation)
        CREG_REQUIRE(round <= 2 * max_iter + 2, "creg_kmeans_lloyd_f64: the Lloyd state machine makes no progress");
        CREG_REQUIRE(launch_assign_pruned(Xc, ni, B, k, box, nullptr, f, 1, acc, T, s) == 0, "creg_kmeans_lloyd_f64: cannot raise the dynamic LDS limit of the E-step");
        CREG_REQUIRE(launch_persist(Xc, ni, B, k, box, f, acc, T, km_pruned_pt(n), s, false) == 0, "creg_kmeans_lloyd_f64: the persistent Lloyd kernel does not fit");
        CREG_LAUNCH_CHECK();
        CREG_HIP(hipMemcpyAsync(&host, f, sizeof(KmFlags), hipMemcpyDeviceToHost, s));
        CREG_HIP(hipStreamSynchronize(s));
        // a workgroup waited ~1 s for the others (the device was kept busy by other streams' kernels, so the grid was not
        // resident together): nothing of this attempt is used, the call starts over with one launch per iteration
        if (host.abort) return KM_RETRY_WITHOUT_PERSIST;
        done = host.done; n_done = host.n_iter;
    }
    while (n_done < max_iter && !done) {
        // 32 launches per host round trip.  A launch runs the E-step of iteration f->n_iter with the exact incremental sums, and
        // its last workgroup the M-step tail; launches after convergence (or after max_iter iterations) return at once, and a
        // launch that follows the discovery of an empty cluster runs the deferred tail instead (see km_lloyd_entry).
        for (int b = 0; b < 32; ++b)
        {
            CREG_REQUIRE((pruned ? launch_assign_pruned(Xc, ni, B, k, box, nullptr, f, 1, acc, T, s)
                                 : launch_assign(Xc, ni, B, k, nullptr, nullptr, f, use_mfma, s, 0, acc, T)) == 0,
                         "creg_kmeans_lloyd_f64: cannot raise the dynamic LDS limit of the E-step");
#ifdef CREG_STAMPS
            hipLaunchKernelGGL(k_km_fold, dim3(1), dim3(1), 0, s);
#endif
        }
        CREG_LAUNCH_CHECK();
        CREG_HIP(hipMemcpyAsync(&host, f, sizeof(KmFlags), hipMemcpyDeviceToHost, s));
        CREG_HIP(hipStreamSynchronize(s));
        done = host.done; n_done = host.n_iter;
    }
    // which buffer holds the labels of the last executed iteration
    int* last = lab[(host.n_iter - 1) & 1];
    if (!host.strict) {      // rerun the E-step so labels match the final centres (_kmeans.py:736-748)
        CREG_REQUIRE((pruned ? launch_assign_pruned(Xc, ni, B, k, box, last, f, 0, nullptr, T, s)
                             : launch_assign(Xc, ni, B, k, last, nullptr, nullptr, use_mfma, s)) == 0, "creg_kmeans_lloyd_f64: E-step launch failed");
    }
    if (pruned) hipLaunchKernelGGL(k_km_unsort, dim3(cdiv(nrow, 256)), dim3(256), 0, s, last, perm, nrow, labels);
    else if (last != labels) CREG_HIP(hipMemcpyAsync(labels, last, sizeof(int) * n, hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(k_km_finish, dim3(1), dim3(1024), 0, s, pruned ? X : Xc, ni, labels, k, C2, f, centers, inertia, n_iter, pruned ? 1 : 0);
    CREG_LAUNCH_CHECK();
    return CREG_OK;
}

extern "C" int creg_kmeans_lloyd_f64(const double* X, int64_t n, const double* init, int32_t k,
                                     int32_t max_iter, double tol_rel, int32_t use_mfma, double* centers,
                                     int32_t* labels, double* inertia, int32_t* n_iter, void* workspace,
                                     size_t workspace_bytes, creg_stream_t stream) {
    // A persistent attempt that had to be abandoned (the device was kept busy by other streams, so the grid was not resident
    // together) costs a spin of ~1 s before it is noticed: after one, the next KM_PERSIST_BACKOFF calls of this process go
    // straight to one launch per iteration before the persistent kernel is tried again.
    static std::atomic<int> backoff{0};
    const bool try_persist = backoff.load(std::memory_order_relaxed) <= 0;
    if (!try_persist) backoff.fetch_sub(1, std::memory_order_relaxed);
    int rc = km_lloyd_run(X, n, init, k, max_iter, tol_rel, use_mfma, centers, labels, inertia, n_iter, workspace, workspace_bytes, stream, try_persist);
    if (rc == KM_RETRY_WITHOUT_PERSIST) {
        backoff.store(KM_PERSIST_BACKOFF, std::memory_order_relaxed);
        rc = km_lloyd_run(X, n, init, k, max_iter, tol_rel, use_mfma, centers, labels, inertia, n_iter, workspace, workspace_bytes, stream, false);
    }
    return rc;
}

extern "C" int creg_kmeans_assign_f64(const double* X, int64_t n, const double* C, int32_t k, int32_t use_mfma,
                                      int32_t* labels, creg_stream_t stream) {
    CREG_REQUIRE(X && C && labels && n >= 1 && n < (1ll << 31) && k >= 1 && k <= 1024, "creg_kmeans_assign_f64: bad argument");
    // no device scratch: the kernels derive the (-2c, |c|^2) rows from the centres in their prologue
    CREG_REQUIRE(launch_assign(X, (int)n, C, k, labels, nullptr, nullptr, use_mfma, (hipStream_t)stream, 1) == 0, "creg_kmeans_assign_f64: E-step launch failed");
    CREG_LAUNCH_CHECK();
    return CREG_OK;
}

extern "C" int creg_group_to_local_f64(const double* X, int64_t n, const int32_t* labels, int32_t k,
                                       const double* M, int32_t m_is_inverse, double* out_local,
                                       int32_t* seg_offsets, creg_stream_t stream) {
    CREG_REQUIRE(X && labels && M && out_local && seg_offsets && n >= 1 && n < (1ll << 31) && k >= 1 && k <= 4096,
                 "creg_group_to_local_f64: bad argument");
    hipStream_t s = (hipStream_t)stream;
    GroupBatch G;
    G.X[0] = X; G.labels[0] = labels; G.M[0] = M; G.out[0] = out_local; G.off[0] = seg_offsets;
    if (n > 16384) {                                           // large frames: many-workgroup count, workgroup-per-cluster compaction
        CREG_HIP(hipMemsetAsync(seg_offsets, 0, sizeof(int) * ((size_t)k + 1), s));
        hipLaunchKernelGGL(k_group_count, dim3(cdiv(n, 4096)), dim3(1024), sizeof(int) * k, s, labels, (int)n, k, seg_offsets);
        hipLaunchKernelGGL(k_group_excl, dim3(1), dim3(1024), 0, s, seg_offsets, k);
        hipLaunchKernelGGL(k_group_scatter_big, dim3(k), dim3(1024), 0, s, X, (int)n, labels, seg_offsets, M, out_local, m_is_inverse);
        CREG_LAUNCH_CHECK();
        return CREG_OK;
    }
    hipLaunchKernelGGL(k_group_offsets, dim3(1, 1), dim3(1024), sizeof(int) * (k + 1), s, G, (int)n, k);
    hipLaunchKernelGGL(k_group_scatter, dim3(k, 1), dim3(64), 0, s, G, (int)n, m_is_inverse);
    CREG_LAUNCH_CHECK();
    return CREG_OK;
}

extern "C" int creg_group_to_local_batch_f64(const double* const* X, int64_t n, const int32_t* const* labels, int32_t k,
                                             const double* const* M, int32_t m_is_inverse, int32_t batch,
                                             double* const* out_local, int32_t* const* seg_offsets, creg_stream_t stream) {
    CREG_REQUIRE(X && labels && M && out_local && seg_offsets && n >= 1 && n < (1ll << 31) && k >= 1 && k <= 4096,
                 "creg_group_to_local_batch_f64: bad argument");
    CREG_REQUIRE(batch >= 1 && batch <= GRP_MAXB, "creg_group_to_local_batch_f64: batch must be in 1..%d", GRP_MAXB);
    GroupBatch G;
    for (int b = 0; b < batch; ++b) {
        CREG_REQUIRE(X[b] && labels[b] && M[b] && out_local[b] && seg_offsets[b], "creg_group_to_local_batch_f64: null pointer in problem %d", b);
        G.X[b] = X[b]; G.labels[b] = labels[b]; G.M[b] = M[b]; G.out[b] = out_local[b]; G.off[b] = seg_offsets[b];
    }
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_group_offsets, dim3(1, batch), dim3(1024), sizeof(int) * (k + 1), s, G, (int)n, k);
    hipLaunchKernelGGL(k_group_scatter, dim3(k, batch), dim3(64), 0, s, G, (int)n, m_is_inverse);
    CREG_LAUNCH_CHECK();
    return CREG_OK;
}

extern "C" size_t creg_kmeans_batch_workspace_bytes(int64_t n, int32_t k, int32_t batch) {
    if (n < 1 || k < 1 || batch < 1) return 0;
    return kms_stride(n, k) * (size_t)batch;
}

extern "C" int creg_kmeans_lloyd_batch_f64(const double* const* X, int64_t n, const double* const* init, int32_t k,
                                           int32_t batch, int32_t max_iter, double tol_rel, double* const* centers,
                                           int32_t* const* labels, double* const* inertia, int32_t* const* n_iter,
                                           void* workspace, size_t workspace_bytes, creg_stream_t stream) {
    CREG_REQUIRE(X && init && centers && labels && inertia && n_iter && workspace, "creg_kmeans_lloyd_batch_f64: null pointer");
    CREG_REQUIRE(batch >= 1 && batch <= KMS_MAXB, "creg_kmeans_lloyd_batch_f64: batch must be in [1, %d]", KMS_MAXB);
    CREG_REQUIRE(n >= 1 && n <= 16384 && k >= 1 && k <= 128 && max_iter >= 1,
                 "creg_kmeans_lloyd_batch_f64: needs n <= 16384 and k <= 128 (centres and labels resident in LDS; the frame too up to 5120 points); use creg_kmeans_lloyd_f64 otherwise");
    CREG_REQUIRE(workspace_bytes >= kms_stride(n, k) * (size_t)batch, "creg_kmeans_lloyd_batch_f64: workspace too small");
    KmBatch A;
    for (int b = 0; b < batch; ++b) {
        CREG_REQUIRE(X[b] && init[b] && centers[b] && labels[b] && inertia[b] && n_iter[b], "creg_kmeans_lloyd_batch_f64: null pointer in problem %d", b);
        A.X[b] = X[b]; A.init[b] = init[b]; A.centers[b] = centers[b]; A.labels[b] = labels[b]; A.inertia[b] = inertia[b]; A.n_iter[b] = n_iter[b];
    }
    const int x_in_lds = n <= 5120;
    int smem = (int)(sizeof(double) * ((x_in_lds ? 3 * n : 0) + 4 * k) + 2 * sizeof(unsigned short) * n);
    const int with_c = ((smem + 7) & ~7) + (int)sizeof(double) * 14 * k;
    const int c_in_lds = with_c <= 5120 * 28 + 128 * 32;           // the limit requested from the runtime below
    if (c_in_lds) smem = with_c;
    // per device, not per process: set on every call (a cached flag would leave a second GPU at the 64 KB default)
    CREG_HIP(hipFuncSetAttribute((const void*)k_km_small, hipFuncAttributeMaxDynamicSharedMemorySize, 5120 * 28 + 128 * 32));
    hipLaunchKernelGGL(k_km_small, dim3(batch), dim3(1024), smem, (hipStream_t)stream, A, (int)n, k, max_iter, tol_rel,
                       (char*)workspace, kms_stride(n, k), c_in_lds, x_in_lds);
    CREG_LAUNCH_CHECK();
    return CREG_OK;
}

#ifdef CREG_STAMPS
extern "C" int creg_debug_km_stamps(unsigned long long* out8, int reset) {
    if (out8) CREG_HIP(hipMemcpyFromSymbol(out8, HIP_SYMBOL(creg::g_km_stamps), sizeof(unsigned long long) * 8));
    if (out8 && reset == 3) { CREG_HIP(hipMemcpyFromSymbol(out8, HIP_SYMBOL(creg::g_km_blk), sizeof(unsigned long long) * 4096)); return CREG_OK; }
    if (out8 && reset == 5) { CREG_HIP(hipMemcpyFromSymbol(out8, HIP_SYMBOL(creg::g_km_it), sizeof(unsigned long long) * 4096)); return CREG_OK; }
    if (out8 && reset == 4) { CREG_HIP(hipMemcpyFromSymbol(out8, HIP_SYMBOL(creg::g_km_pp), sizeof(unsigned long long) * 16)); return CREG_OK; }
    if (out8 && reset == 2) { CREG_HIP(hipMemcpyFromSymbol(out8, HIP_SYMBOL(creg::g_km_phs), sizeof(unsigned long long) * 8)); return CREG_OK; }
    if (reset) { unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}; CREG_HIP(hipMemcpyToSymbol(HIP_SYMBOL(creg::g_km_stamps), z, sizeof(z)));
                 { unsigned long long z16[16] = {0}; CREG_HIP(hipMemcpyToSymbol(HIP_SYMBOL(creg::g_km_pp), z16, sizeof(z16))); }
                 CREG_HIP(hipMemcpyToSymbol(HIP_SYMBOL(creg::g_km_phs), z, sizeof(z))); CREG_HIP(hipMemcpyToSymbol(HIP_SYMBOL(creg::g_km_ph), z, sizeof(z)));
                 unsigned long long m[4] = {~0ull, 0ull, 0ull, 0ull}; CREG_HIP(hipMemcpyToSymbol(HIP_SYMBOL(creg::g_km_w), m, sizeof(m))); }
    return CREG_OK;
}
#endif
