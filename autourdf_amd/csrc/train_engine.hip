// train_engine.hip -- A1: the whole `train` loop of the reference (PointCloud/mlp_reg.py:17-152)
// as a device-resident plan.  One epoch = FIVE small launches, no host round trip (round 3: the forward of the
// hidden layers no longer has a launch of its own -- every parameter row is read once per epoch, by the kernel
// that updates it, which also computes the NEXT epoch's activation of that row from the registers it holds):
//
//   k_head    output layer(s) + residual + pose assembly + calculate_pc (mlp_reg.py:62-94,155-170),
//             one workgroup per pose row / cluster
//   k_nn_l1   L1 nearest neighbour both ways (chamfer_distance, mlp_reg.py:96); its epilogue emits
//             the loss partials and the sign scatter of the y->x term (integer atomics: exact)
//   k_gradc   loss, best tracking (mlp_reg.py:102-111), ReduceLROnPlateau, Adam scalars, early stop;
//             per-cluster reduction to dL/dR, dL/dt, backward through the pose head and the output
//             layer(s): its pose row of dL/d(hidden pre-activation)
//   k_bwd2    backward through the hidden layer(s) to the encoder activation, COMPLETE per column block
//             (a workgroup owns 16 hidden units of the encoder and all H2 rows of their W2 columns), then
//             those encoder rows' weight gradients + Adam and the NEXT epoch's encoder activation
//   k_dw      hidden / output rows: weight gradients fused with the Adam update (no gradient buffer), and
//             -- for the hidden rows -- the NEXT epoch's hidden activation from the just-updated weights
//   (k_l1, k_l2: the first epoch's activations, once per train)
//
// Everything is fp32 like the reference; every reduction has a fixed order (bit-reproducible
// run to run).  State that the reference keeps in Python (min_loss, count, scheduler, lr) lives in
// a double-buffered device struct indexed by epoch parity.
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>
#include "creg_common.h"
#include "nn_l1.h"

#ifndef CREG_L2_IN_BD_DEFAULT
#define CREG_L2_IN_BD_DEFAULT false        // (until measured faster)
#endif

namespace creg {

struct Dims {
    int rot, K, KP, IN, H, HA, HB, H2, OA, OB, NP, NT, epochs;      // KP: K rounded up to the k_bd D role's contraction chunk (rows K..KP-1 of x1 / h2 / g_h2 / g_out stay 0)
    float slope;
    // flat parameter offsets
    int oW1, ob1, oW2, ob2, oW3A, ob3A, oW3B, ob3B, NPAR;
    int nbx, nby;  // NN blocks per direction (= loss partial counts)
    int nyb;       // 64-point blocks of the sorted target frame (0: exhaustive search only)
    int npb;       // upper bound of the blocks of the predicted cloud, clusters padded (0: that direction exhaustive)
    int ppl;       // points per lane and block visit: blocks hold 64 * ppl points
    int nbt, nbp;  // boxes per lane of the search over the target frame / the predicted cloud (their box tables hold 64 nbt / 64 nbp)
    int rows;      // the search takes sixteen queries per wave in slot order (nn_l1_rows, round 5): loss partials per 16-slot group
};

constexpr int PS_MAXB = 512;                   // most blocks of the predicted cloud (clusters padded): eight boxes per lane in the search

struct Hyper {      // uploaded per run
    float lr, factor;
    int patience, stop;
};

struct TrainState {
    double lr, sched_best;
    int sched_bad, count, stopped, step;
    float min_loss;
    int epochs_run, best_epoch;
    float step_size, bc2_sqrt;   // Adam scalars for the update of the epoch that produced this state
    float last_loss;
    float next_bc2s;             // bias corrections of step `step + 1` (k_prep's tables), fetched by the epoch that produced
    double next_bc1;             //   this state: the next epoch reads them WITH the state instead of in a round trip after it
};

struct Ws {         // device pointers into the caller's workspace
    float *P, *P1, *AM, *AV;     // parameters double-buffered by optimizer-step parity: epoch e reads P[e & 1] (P / P1) and writes the
                                 //   other one, so no kernel of an epoch can see a row another workgroup has already updated
    float *pose_in, *enc, *x1[2], *h2[2], *head_save, *m2, *gm2;      // x1 / h2: double-buffered by epoch parity
    float4 *pts4, *y4, *pred4, *ys4, *psl4, *ps4;
    float *ybox, *pbox;
    int* sb;
    int* sgn_x;
    int4* cnt4;
    float *lossp_x, *lossp_y;
    float *g_out, *g_h2;
    TrainState* state;
    double* bc1;        // [epochs + 1] Adam bias corrections by step, filled once per train by k_prep:
    float* bc2s;        //   1 - 0.9^t and sqrt(1 - 0.999^t) (two double pow() off the per-epoch critical path)
    float *best_m, *best_pred, *loss_hist, *lr_hist, *result;
    int* off;
    Hyper* hyper;
    unsigned long long* ysum;   // [4] per chunk of the target frame: fingerprint of the points its k-d leaves were built from (k_sort_y)
    int* sync;          // fused backward launch (k_gbd): [0] arrivals of the gradient role's blocks (k_head zeroes it every epoch),
                        //   [32 .. 35] (its own 128-byte line) the record the consumers need of the advanced state: stopped, step_size, bc2_sqrt
};

// Problem b of a batch lives in its own copy of the workspace layout, `bytes` = b * stride further on:
// every member of Ws is a pointer, so shifting the struct is shifting each of them.
static_assert(sizeof(Ws) % sizeof(char*) == 0, "Ws must hold pointers only");
__host__ __device__ __forceinline__ Ws ws_shift(Ws W, size_t bytes) {
    char** p = reinterpret_cast<char**>(&W);
#pragma unroll
    for (size_t i = 0; i < sizeof(Ws) / sizeof(char*); ++i) p[i] += bytes;
    return W;
}

// pose representations of train() (mlp_reg.py:64-90; creg_train_shape.rot)
constexpr int ROT_Q = 0, ROT_DQ = 1, ROT_6D = 2, ROT_RPY = 3;
constexpr int POSE_IN = 12;            // floats per pose row of the model input (q: 7, dq: 8, 6d: 9, rpy: 6)
__device__ __forceinline__ float act_f(float v, float slope) { return v > 0.f ? v : v * slope; }
__device__ __forceinline__ float act_grad(float post, float slope) { return post > 0.f ? 1.f : slope; }

__device__ __forceinline__ int seg_of(const int* __restrict__ off, int k, int n) {
    int lo = 0, hi = k;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (off[mid] <= n) lo = mid; else hi = mid; }
    return lo;
}

// ------------------------------------------------------------------------------------------ prep
// per-problem inputs of a batch (grid.z): one launch stages all of them
constexpr int PREP_MAXB = 64;
struct PrepBatch { const float* m[PREP_MAXB]; const float* y[PREP_MAXB]; const float* pts[PREP_MAXB]; Hyper hy[PREP_MAXB]; };
__global__ __launch_bounds__(256) void k_prep(Dims D, Ws W0, size_t bstride, PrepBatch pb) {
    const Ws W = ws_shift(W0, blockIdx.z * bstride);
    const Hyper hy = pb.hy[blockIdx.z];
    const float* __restrict__ m = pb.m[blockIdx.z];
    const float* __restrict__ y = pb.y[blockIdx.z];
    const float* __restrict__ pts = pb.pts[blockIdx.z];
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int stride = gridDim.x * 256;
    for (int n = t; n < D.NP; n += stride) {
        const int c = seg_of(W.off, D.K, n);
        W.pts4[n] = make_float4(pts[3 * (size_t)n], pts[3 * (size_t)n + 1], pts[3 * (size_t)n + 2], __int_as_float(c));
    }
    for (int j = t; j < D.NT; j += stride)
        W.y4[j] = make_float4(y[3 * (size_t)j], y[3 * (size_t)j + 1], y[3 * (size_t)j + 2], 0.f);
    for (int i = t; i < D.NPAR; i += stride) { W.AM[i] = 0.f; W.AV[i] = 0.f; }
    for (int i = t; i < D.epochs; i += stride) { W.loss_hist[i] = NAN; W.lr_hist[i] = NAN; }
    // pose rows K .. KP-1 of the K-row matrices: zero for the whole train (k_bd's D role contracts over KP rows; nobody writes them)
    for (int i = t; i < (D.KP - D.K) * D.H; i += stride) { W.x1[0][D.K * D.H + i] = 0.f; W.x1[1][D.K * D.H + i] = 0.f; }
    for (int i = t; i < (D.KP - D.K) * D.H2; i += stride) { W.h2[0][D.K * D.H2 + i] = 0.f; W.h2[1][D.K * D.H2 + i] = 0.f; W.g_h2[D.K * D.H2 + i] = 0.f; }
    for (int i = t; i < (D.KP - D.K) * 16; i += stride) W.g_out[D.K * 16 + i] = 0.f;
    for (int i = t; i <= D.epochs; i += stride) { W.bc1[i] = 1.0 - pow(0.9, (double)i); W.bc2s[i] = (float)sqrt(1.0 - pow(0.999, (double)i)); }
    if (blockIdx.x == 0) {
        for (int r = threadIdx.x; r < D.K; r += 256) {
            const float* M = m + 16 * r;
            const float R[9] = {M[0], M[1], M[2], M[4], M[5], M[6], M[8], M[9], M[10]};
            const float tr[3] = {M[3], M[7], M[11]};
            float in[POSE_IN];
            for (int i = 0; i < POSE_IN; ++i) in[i] = 0.f;
            int nin;
            if (D.rot == ROT_Q) {                 // cat([t, matrix_to_quaternion(R)])  mlp_reg.py:65-66
                float q[4];
                matrix_to_quat(R, q);
                in[0] = tr[0]; in[1] = tr[1]; in[2] = tr[2]; in[3] = q[0]; in[4] = q[1]; in[5] = q[2]; in[6] = q[3];
                nin = 7;
            } else if (D.rot == ROT_DQ) {         // transform_to_dualquat  mlp_reg.py:80
                se3_to_dq(R, tr, in, FLT_EPSILON);
                nin = 8;
            } else if (D.rot == ROT_6D) {         // cat([t, matrix_to_rotation_6d(R)]): the first two rows  mlp_reg.py:87-88
                in[0] = tr[0]; in[1] = tr[1]; in[2] = tr[2];
                for (int i = 0; i < 6; ++i) in[3 + i] = R[i];
                nin = 9;
            } else {                              // cat([t, matrix_to_euler_angles(R, "XYZ")])  mlp_reg.py:73-74
                in[0] = tr[0]; in[1] = tr[1]; in[2] = tr[2];
                matrix_to_euler_xyz(R, in + 3);
                nin = 6;
            }
            for (int i = 0; i < POSE_IN; ++i) W.pose_in[POSE_IN * r + i] = in[i];
            // [sin x, cos x, sin 2x, cos 2x, sin 4x, cos 4x, sin 8x, cos 8x]  model_utils.py:141-150
            float* e = W.enc + (size_t)r * D.IN;
            for (int f = 0; f < 4; ++f) {
                const float mul = (float)(1 << f);
                for (int i = 0; i < nin; ++i) {
                    e[(2 * f) * nin + i] = sinf(mul * in[i]);
                    e[(2 * f + 1) * nin + i] = cosf(mul * in[i]);
                }
            }
        }
        if (threadIdx.x == 0) {
            TrainState s;
            *W.hyper = hy;
            s.lr = (double)hy.lr; s.sched_best = INFINITY; s.sched_bad = 0; s.count = 0; s.stopped = 0;
            s.step = 0; s.min_loss = 1000.f; s.epochs_run = 0; s.best_epoch = -1; s.step_size = 0.f;
            s.bc2_sqrt = 1.f; s.last_loss = NAN;
            s.next_bc1 = 1.0 - 0.9; s.next_bc2s = (float)sqrt(1.0 - 0.999);         // step 1 (the tables are filled by this launch)
            W.state[0] = s; W.state[1] = s;
            for (int i = 0; i < 64; ++i) W.sync[i] = 0;          // [1]: arrivals of the backward launch's B role since the train began (k_bd with the next hidden activation inside)
        }
    }
}

// ------------------------------------------------------------------------------------------ layer 1 (epoch 0 only)
__global__ __launch_bounds__(256) void k_l1(Dims D, Ws W0, int par, size_t bstride) {
    const Ws W = ws_shift(W0, blockIdx.z * bstride);
    const int lane = threadIdx.x & 63;
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= D.H) return;
    const float w = lane < D.IN ? W.P[D.oW1 + (size_t)o * D.IN + lane] : 0.f;
    const float w2 = 64 + lane < D.IN ? W.P[D.oW1 + (size_t)o * D.IN + 64 + lane] : 0.f;      // ('6d': 72 features)
    const float b = W.P[D.ob1 + o];
    for (int r = 0; r < D.K; ++r) {
        float t = lane < D.IN ? w * W.enc[(size_t)r * D.IN + lane] : 0.f;
        if (64 + lane < D.IN) t = fmaf(w2, W.enc[(size_t)r * D.IN + 64 + lane], t);
        const float v = wave_sum_fast(t) + b;
        if (lane == 0) (par ? W.x1[1] : W.x1[0])[(size_t)r * D.H + o] = act_f(v, D.slope);      // (a runtime index into the shifted struct put all of it -- 312 bytes a lane -- into scratch)
    }
}

// ------------------------------------------------------------------------------------------ layer 2
// The hidden activation h2 = act(x1 . W2^T + b2), [K x H] . [H x H2]: once for epoch 0 and at the end of every epoch (the next
// epoch's activation from the freshly updated rows and the next encoder activation).  Round 4: on the matrix cores
// (v_mfma_f32_16x16x4_f32, exact float32: a k-ordered fmaf chain).  A workgroup owns 16 hidden units (one N-tile), its 8 waves split
// the H inputs (wave w: inputs [KW w, KW (w + 1)), KW = H / 8), M-tiles = 16 pose rows, two per pass.  Operands straight from
// global memory, 16 bytes per lane where the width allows (KW % 16 == 0; k order inside a wave o(g, i, kk) = 16 g + 4 kk + i as in
// k_bd's B role): B[k][j] = W2[u0 + j][o], A[i][k] = x1[r][o] (rows past K repeat row K - 1 into accumulator rows nobody reads).
// The eight partial tiles are summed in wave order through LDS, + bias, activation.  (Round 3: 1024-thread workgroups staged all of
// x1 in LDS -- LDS-DMA, a full wait, a barrier -- then every wave ran 20 dependent steps of two ds_read_b128, 8 FMAs and a DPP wave sum.)
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int L2_THREADS = 512;
constexpr int L2_RB = 32;             // pose rows per pass (two M-tiles)
constexpr int L2_SMEM = (L2_THREADS / 64) * L2_RB * 16 * 4;
template <int NC>
__device__ __forceinline__ void l2_body(const Dims& D, const Ws& W, int par, int blk) {
    constexpr int KW = 8 * NC, NS = KW / 4;
    constexpr bool V4 = KW % 16 == 0;
    constexpr int NL = V4 ? NS / 4 : NS;           // loads per operand tile and lane
    __shared__ __attribute__((aligned(16))) float red[(L2_THREADS / 64) * L2_RB * 16];
    const float* Pc = par ? W.P1 : W.P;
    const float* x1cur = par ? W.x1[1] : W.x1[0];     // (a runtime index into the shifted struct would go to scratch)
    float* h2out = par ? W.h2[1] : W.h2[0];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, lj = lane & 15, kk = lane >> 4;
    const int u0 = blk * 16, H = D.H;                // grid covers H2 exactly (H2 % 32 == 0)
    auto load_t = [&](const float* rowp, float* dst) {     // one operand tile of this lane: its wave's KW inputs of row `rowp`
        const float* p = rowp + wv * KW + (V4 ? 4 * kk : kk);
#pragma unroll
        for (int q = 0; q < NL; ++q) {
            if constexpr (V4) { const float4 v = *(const float4*)(p + 16 * q); dst[4 * q] = v.x; dst[4 * q + 1] = v.y; dst[4 * q + 2] = v.z; dst[4 * q + 3] = v.w; }
            else dst[q] = p[4 * q];
        }
    };
    float bw[NS], a0[NS], a1[NS];
    load_t(Pc + D.oW2 + (size_t)(u0 + lj) * H, bw);
    load_t(x1cur + (size_t)min(lj, D.K - 1) * H, a0);
    const bool two0 = D.K > 16;                        // workgroup-uniform
    if (two0) load_t(x1cur + (size_t)min(16 + lj, D.K - 1) * H, a1);
    const float bias = Pc[D.ob2 + u0 + (tid & 15)];
    __builtin_amdgcn_sched_barrier(0);
    const int stopped = W.state[par].stopped;          // requested LAST (hipcc waits for it where it is issued: see k_bd); the state
                                                       //   the epoch that ENDS with this launch has produced
    __builtin_amdgcn_sched_barrier(0);
    for (int r0 = 0; r0 < D.K; r0 += L2_RB) {          // K <= 32: one pass
        const int nr = min(L2_RB, D.K - r0);
        const bool two = nr > 16;
        if (r0) {
            load_t(x1cur + (size_t)min(r0 + lj, D.K - 1) * H, a0);
            if (two) load_t(x1cur + (size_t)min(r0 + 16 + lj, D.K - 1) * H, a1);
        }
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        if (two) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[s], bw[s], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], bw[s], acc1, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int s = 0; s < NS; ++s) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[s], bw[s], acc0, 0, 0, 0);
        }
        if (r0) __syncthreads();                       // the previous pass has read red
        {   // D[4 kk + v][lj] of tile t -> red[wave][16 t + 4 kk + v][lj]
            float* o = red + (wv * L2_RB + 4 * kk) * 16 + lj;
#pragma unroll
            for (int v = 0; v < 4; ++v) { o[v * 16] = acc0[v]; if (two) o[(16 + v) * 16] = acc1[v]; }
        }
        __syncthreads();
        if (stopped) return;                           // a stopped train keeps its activations (workgroup-uniform)
        if (tid < nr * 16) {                           // tid = (pose row of the pass) * 16 + unit
            float sum = red[tid];
#pragma unroll
            for (int w2 = 1; w2 < L2_THREADS / 64; ++w2) sum += red[w2 * L2_RB * 16 + tid];
            h2out[(size_t)(r0 + (tid >> 4)) * D.H2 + u0 + (tid & 15)] = act_f(sum + bias, D.slope);
        }
    }
}
template <int NC>
__global__ __launch_bounds__(L2_THREADS) void k_l2(Dims D, Ws W0, int par, size_t bstride) {
    // every kernel argument in the entry block, one wait (see k_bd)
    asm volatile("" :: "s"(W0.P), "s"(W0.P1), "s"(W0.x1[0]), "s"(W0.x1[1]), "s"(W0.h2[0]), "s"(W0.h2[1]), "s"(W0.state), "s"(D.K), "s"(D.H),
                 "s"(D.H2), "s"(D.oW2), "s"(D.ob2), "s"(D.slope), "s"(par), "s"(bstride));
    const Ws W = ws_shift(W0, blockIdx.z * bstride);
    l2_body<NC>(D, W, par, blockIdx.x);
}

// ------------------------------------------------------------------------------------------ head + transform
// Block r: output layer for pose row r (one wave per output unit), pose assembly, then the rigid
// transform of cluster r's points (clusters are stored back to back) -- calculate_pc needs no
// launch of its own and nothing is recomputed.
// X: nine output units ('6d': 3 + 6) -- wave 0 takes the ninth as a second dot product.
template <int NC, bool X = false>
__global__ __launch_bounds__(512) void k_head(Dims D, Ws W0, int par, size_t bstride) {
    // every kernel argument in the entry block, one wait (see k_bd)
    asm volatile("" :: "s"(W0.P), "s"(W0.P1), "s"(W0.h2[0]), "s"(W0.h2[1]), "s"(W0.pose_in), "s"(W0.head_save), "s"(W0.m2), "s"(W0.pts4), "s"(W0.pred4),
                 "s"(W0.psl4), "s"(W0.ps4), "s"(W0.pbox), "s"(W0.sb), "s"(W0.cnt4), "s"(W0.state), "s"(W0.off), "s"(W0.sync), "s"(D.rot), "s"(D.K), "s"(D.H2), "s"(D.HA),
                 "s"(D.HB), "s"(D.OA), "s"(D.OB), "s"(D.NP), "s"(D.npb), "s"(D.ppl), "s"(D.nbp), "s"(D.oW3A), "s"(D.ob3A), "s"(D.oW3B), "s"(D.ob3B), "s"(par),
                 "s"(bstride));
    const Ws W = ws_shift(W0, blockIdx.z * bstride);
    if (blockIdx.x == 0 && threadIdx.x == 0) W.sync[0] = 0;      // the fused backward launch of this epoch (k_gbd) counts its gradient blocks from zero
    const float* h2cur = par ? W.h2[1] : W.h2[0];
    const float* Pc = par ? W.P1 : W.P;
    __shared__ float outs[12];
    __shared__ float m2s[12];
    const int r = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int NO = D.OA + D.OB;
    // Request order.  Round 3 issued, in this order, the pose row, the cluster's bounds, its first block, the first point (which needs
    // the block index) and only then the operands of the dot products -- hipcc turned the wave-uniform ones into scalar loads and
    // waited for them one after the other: FOUR dependent memory round trips in front of the dot products' loads.  Now the dot
    // products' operands go out first (every wave: waves past the output units mirror the last one), then everything scalar.
    const int o = min(wave, (X ? 8 : NO) - 1);
    const float *w, *a; int n;
    if (o < D.OA) { w = Pc + D.oW3A + (size_t)o * D.HA; a = h2cur + (size_t)r * D.H2; n = D.HA; }
    else { w = Pc + D.oW3B + (size_t)(o - D.OA) * D.HB; a = h2cur + (size_t)r * D.H2 + D.HA; n = D.HB; }
    float wv[NC], av[NC];                          // n <= 64 NC: every load of the dot in flight at once
#pragma unroll
    for (int c = 0; c < NC; ++c) { const int i = min(c * 64 + lane, n - 1); wv[c] = w[i]; av[c] = a[i]; }
    const float bias = o < D.OA ? Pc[D.ob3A + o] : Pc[D.ob3B + o - D.OA];
    float wv9[NC], av9[NC], bias9 = 0.f;           // X: output unit 8 (branch B's last), every wave requests it, wave 0 uses it
    if constexpr (X) {
        const float* w9 = Pc + D.oW3B + (size_t)(8 - D.OA) * D.HB;
        const float* a9 = h2cur + (size_t)r * D.H2 + D.HA;
#pragma unroll
        for (int c = 0; c < NC; ++c) { const int i = min(c * 64 + lane, D.HB - 1); wv9[c] = w9[i]; av9[c] = a9[i]; }
        bias9 = Pc[D.ob3B + 8 - D.OA];
    }
    __builtin_amdgcn_sched_barrier(0);
    const int stopped = W.state[par].stopped;      // early stop (mlp_reg.py:107-111): the remaining epochs of a captured graph
                                                   // return at their first barrier
    float pin[POSE_IN];
#pragma unroll
    for (int i = 0; i < POSE_IN; ++i) pin[i] = W.pose_in[POSE_IN * r + i];
    const int b = W.off[r], e = W.off[r + 1];
    // block-sorted walk (D.npb): cluster r owns the slots of blocks [sb[r], sb[r + 1])
    const int BS = 64 * D.ppl;
    const int sb0 = W.sb[r], sb1 = W.sb[r + 1];    // (unconditional, so that both come in the batch above: behind `D.npb ? ... : 0` they were two more dependent round trips)
    const int s0 = D.npb ? BS * sb0 : 0, s1 = D.npb ? BS * sb1 : 0;
    const float4 p_first = D.npb ? W.psl4[min(s0 + (int)threadIdx.x, BS * D.npb - 1)] : W.pts4[min(b + (int)threadIdx.x, D.NP - 1)];
    {
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) s = fmaf(c * 64 + lane < n ? wv[c] : 0.f, av[c], s);
        s = wave_sum_fast(s) + bias;
        if (lane == 0 && wave < NO) outs[o] = s;
        if constexpr (X) {
            float s9 = 0.f;
#pragma unroll
            for (int c = 0; c < NC; ++c) s9 = fmaf(c * 64 + lane < D.HB ? wv9[c] : 0.f, av9[c], s9);
            s9 = wave_sum_fast(s9) + bias9;
            if (lane == 0 && wave == 0) outs[8] = s9;
        }
    }
    __syncthreads();
    if (stopped) return;
    if (threadIdx.x == 0) {
        const float* in = pin;
        float R[9], t[3], save[16];
        for (int i = 0; i < 16; ++i) save[i] = 0.f;
        if (D.rot == ROT_6D) {
            // xyz + orig[:, :3], r6d + orig[:, 3:]  model_utils.py:214 ; rotation_6d_to_matrix  mlp_reg.py:89
            for (int i = 0; i < 3; ++i) t[i] = outs[i] + in[i];
            float d6[6];
            for (int i = 0; i < 6; ++i) { d6[i] = outs[3 + i] + in[3 + i]; save[i] = d6[i]; }
            rot6d_to_matrix(d6, R);
        } else if (D.rot == ROT_RPY) {
            // xyz + orig[:, :3], tanh(.) + orig[:, 3:]  model_utils.py:237-242,274 ; euler_angles_to_matrix(r, "XYZ")  mlp_reg.py:75
            for (int i = 0; i < 3; ++i) t[i] = outs[i] + in[i];
            float e3[3];
            for (int i = 0; i < 3; ++i) { const float th = tanhf(outs[3 + i]); e3[i] = th + in[3 + i]; save[i] = e3[i]; save[4 + i] = th; }
            euler_xyz_to_matrix(e3, R);
        } else if (D.rot == ROT_Q) {
            // xyz + orig[:, :3] ; normalize(q + orig[:, 3:])   model_utils.py:159
            for (int i = 0; i < 3; ++i) t[i] = outs[i] + in[i];
            float v[4], n2 = 0.f;
            for (int i = 0; i < 4; ++i) { v[i] = outs[3 + i] + in[3 + i]; n2 = fmaf(v[i], v[i], n2); }
            const float nrm = sqrtf(n2), den = fmaxf(nrm, 1e-12f);
            float u[4];
            for (int i = 0; i < 4; ++i) u[i] = v[i] / den;
            quat_to_matrix(u, R);
            for (int i = 0; i < 4; ++i) save[i] = u[i];
            save[4] = nrm;
        } else {
            float dq[8];
            for (int i = 0; i < 8; ++i) dq[i] = outs[i] + in[i];              // x + orig  model_utils.py:99
            dq_to_se3(dq, R, t);
            for (int i = 0; i < 8; ++i) save[i] = dq[i];
        }
        for (int a = 0; a < 3; ++a) { m2s[4 * a] = R[3 * a]; m2s[4 * a + 1] = R[3 * a + 1]; m2s[4 * a + 2] = R[3 * a + 2]; m2s[4 * a + 3] = t[a]; }
        for (int i = 0; i < 12; ++i) W.m2[16 * r + i] = m2s[i];
        W.m2[16 * r + 12] = 0.f; W.m2[16 * r + 13] = 0.f; W.m2[16 * r + 14] = 0.f; W.m2[16 * r + 15] = 1.f;
        for (int i = 0; i < 16; ++i) W.head_save[16 * r + i] = save[i];
    }
    __syncthreads();
    if (D.npb) {
        // slot order: a wave covers 64 consecutive slots per trip, so a block's box is a wave reduction of the
        // coordinates just computed (exact: the search compares against these very values); with 256-point blocks
        // the four waves of a block combine through LDS
        __shared__ float wbox[8][6];
        for (int base = s0; base < s1; base += 512) {            // trip count uniform over the workgroup
            const int sl = base + threadIdx.x;
            const bool in = sl < s1;
            const float4 p = !in ? make_float4(0.f, 0.f, 0.f, __int_as_float(0x7fffffff)) : (base == s0 ? p_first : W.psl4[sl]);
            const int n = __float_as_int(p.w);
            const bool real = n != 0x7fffffff;
            float o[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) o[a] = fmaf(p.z, m2s[4 * a + 2], fmaf(p.y, m2s[4 * a + 1], p.x * m2s[4 * a])) + m2s[4 * a + 3];
            if (real) {
                W.pred4[n] = make_float4(o[0], o[1], o[2], 0.f);
                W.cnt4[n] = make_int4(0, 0, 0, 0);
            }
            if (in) W.ps4[sl] = real ? make_float4(o[0], o[1], o[2], p.w) : make_float4(INFINITY, INFINITY, INFINITY, p.w);
            float bx[6];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                bx[a] = wave_min_fast(real ? o[a] : INFINITY);
                bx[3 + a] = -wave_min_fast(real ? -o[a] : INFINITY);
            }
            if (D.ppl == 1) {
                if (lane == 0 && in) {
                    float* pb = W.pbox + (sl >> 6);            // six planes of 64 nbp boxes
#pragma unroll
                    for (int a = 0; a < 6; ++a) pb[a * 64 * D.nbp] = bx[a];
                }
            } else {                                             // 4 waves per block, 2 blocks per trip
                if (lane == 0) {
#pragma unroll
                    for (int a = 0; a < 6; ++a) wbox[wave][a] = bx[a];
                }
                __syncthreads();
                if (threadIdx.x < 12) {
                    const int hb = threadIdx.x / 6, a = threadIdx.x % 6;
                    if (base + 256 * hb < s1) {
                        float v = wbox[4 * hb][a];
                        for (int w = 1; w < 4; ++w) v = a < 3 ? fminf(v, wbox[4 * hb + w][a]) : fmaxf(v, wbox[4 * hb + w][a]);
                        W.pbox[a * 64 * D.nbp + (base >> 8) + hb] = v;
                    }
                }
                __syncthreads();
            }
        }
        return;
    }
    for (int n = b + threadIdx.x; n < e; n += 512) {
        const float4 p = (n == b + (int)threadIdx.x) ? p_first : W.pts4[n];
        float o[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) o[a] = fmaf(p.z, m2s[4 * a + 2], fmaf(p.y, m2s[4 * a + 1], p.x * m2s[4 * a])) + m2s[4 * a + 3];
        W.pred4[n] = make_float4(o[0], o[1], o[2], 0.f);
        W.cnt4[n] = make_int4(0, 0, 0, 0);
    }
}

// ------------------------------------------------------------------------------------------ target blocks
// Once per train: the target frame (it does not move during the 300 epochs) cut into blocks of BS = 64 * D.ppl
// points with an axis-aligned box each, for nn_l1_block_pruned.  The blocks are the leaves of a balanced k-d tree
// -- sort all slots on x and halve, sort each half on y and halve, ... (x,y,z,x,y,z: a 4 x 4 x 4 grid in rank
// space for 64 leaves) -- so the boxes of different blocks do not overlap and a query usually has to look into one
// to three of them (Morton-curve blocks, tried first, overlapped enough to need eight).  One workgroup per problem;
// every level is a bitonic sort of the slots in LDS restricted to segments of npow >> level, keys = (order-preserving
// bits of the coordinate, original index): unique, so the layout is deterministic.  Slots past n_tgt hold +inf keys
// and stay at the end through every level.  Any permutation gives the same search result -- ties are broken on the
// original index stored with the point -- so none of this is visible in the plan's outputs.
__device__ __forceinline__ unsigned ordered_bits(float c) {                // total order of the floats, below 0xFFFFFFFF
    unsigned u = (unsigned)__float_as_int(c);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return u == 0xFFFFFFFFu ? 0xFFFFFFFEu : u;
}
// ascending bitonic sort of key[0, npow) restricted to aligned segments of `seg` slots.
// A wave's 64 compare-exchanges of a stage with stride <= 64 stay inside its own 128 consecutive slots (and so do
// those of its later trips), so runs of such stages need no workgroup barrier -- the wave's DS instructions execute
// in order; only stages with a longer stride, and the hand-over to / from them, synchronise the workgroup: 35 barrier
// pairs instead of 308 stages with one for the six k-d levels of 4096 slots.
__device__ __forceinline__ void bitonic_segments(unsigned long long* key, int npow, int seg, int tid, int nthreads) {
    bool wide_before = true;                                   // the caller's writes count as a wide stage
    for (int size = 2; size <= seg; size <<= 1)
        for (int stride = size >> 1; stride >= 1; stride >>= 1) {
            const bool wide = stride > 64;
            if (wide || wide_before) __syncthreads();
            else { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); }
            for (int t = tid; t < (npow >> 1); t += nthreads) {
                const int i = ((t & ~(stride - 1)) << 1) | (t & (stride - 1)), j = i | stride;
                const unsigned long long a = key[i], b = key[j];
                const bool up = size == seg || (i & size) == 0;            // every segment ends ascending
                if ((a > b) == up) { key[i] = b; key[j] = a; }
            }
            wide_before = wide;
        }
    __syncthreads();
}
static int pow2_at_least(int n) { int p = 64; while (p < n) p <<= 1; return p; }

// Frames above YS_CHUNK = 16384 points (the keys of one workgroup's LDS) are cut in index order into chunks of that size,
// one workgroup (grid.x) and one set of 64 leaves of 256 points each: every chunk is a sample of the whole frame, so a
// query has to look into at least one block per chunk -- a few times the work of a single tree, still a small fraction of
// the exhaustive sweep (round 2 fell back to it above 16384 targets).
constexpr int YS_CHUNK = 16384;
// check != 0 (round 5): the caller says the frame is the one the leaves were built from (creg_train_args.y_unchanged) -- the
// workgroup first compares a position-dependent 64-bit fingerprint of its chunk of the staged frame with the one stored when the leaves
// were built and returns if they agree (a launch of a few microseconds instead of the sort: 145 us at 4096 points, 657 at 16384);
// a frame that differs -- a buffer reused for the next frame, two registrars sharing a plan -- is sorted again instead of being
// searched through stale leaves.
__global__ __launch_bounds__(1024) void k_sort_y(Dims D, Ws W0, size_t bstride, int npow, int check) {
    const Ws W = ws_shift(W0, blockIdx.z * bstride);
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    unsigned long long* key = (unsigned long long*)smem_raw;               // npow keys
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int BS = 64 * D.ppl;
    const int base = blockIdx.x * YS_CHUNK, nc = min(YS_CHUNK, D.NT - base);        // this chunk's targets [base, base + nc)
    const int blk0 = base / BS, nblk = (nc + BS - 1) / BS;                            // its blocks [blk0, blk0 + nblk)
    __shared__ unsigned long long s_fp;
    {
        if (tid == 0) s_fp = 0ull;
        __syncthreads();
        unsigned long long h = 0ull;
        for (int j = tid; j < nc; j += 1024) {
            const float4 p = W.y4[base + j];
            const unsigned long long a = (unsigned)__float_as_int(p.x), b = (unsigned)__float_as_int(p.y), c = (unsigned)__float_as_int(p.z);
            unsigned long long v = a ^ (b << 21) ^ (c << 42) ^ (c >> 22);
            v *= 0x9E3779B97F4A7C15ull;                                               // (mix, then weigh by the position: a permuted frame differs)
            h += (v ^ (v >> 29)) * (2ull * (unsigned long long)j + 1ull);
        }
        atomicAdd(&s_fp, h);                                                          // integer sum: order independent
        __syncthreads();
        const unsigned long long fp = s_fp ^ ((unsigned long long)nc << 48) ^ (unsigned long long)D.ppl ^ ((unsigned long long)D.rows << 8);
        if (check && W.ysum[blockIdx.x] == fp) return;                                // workgroup-uniform: the leaves are this frame's
        __syncthreads();
        if (tid == 0) W.ysum[blockIdx.x] = fp;
    }
    for (int j = tid; j < npow; j += 1024) key[j] = j < nc ? (unsigned long long)j : ~0ull;
    __syncthreads();
    int level = 0;
    // (D.rows: the sixteen-queries-per-wave search takes its QUERIES from these slots too, 16 consecutive ones per wave -- two more
    //  levels make every aligned group of 16 slots a k-d cell of its leaf: the fewer target blocks the sixteen need between them)
    for (int seg = npow; seg > (D.rows ? 16 : BS); seg >>= 1, ++level) {
        const int axis = level % 3;
        for (int j = tid; j < npow; j += 1024) {
            const unsigned long long k = key[j];
            if (k != ~0ull) {
                const int idx = (int)(k & 0xFFFFull);
                const float4 p = W.y4[base + idx];
                key[j] = ((unsigned long long)ordered_bits(axis == 0 ? p.x : axis == 1 ? p.y : p.z) << 32) | (unsigned)idx;
            }
        }
        __syncthreads();
        bitonic_segments(key, npow, seg, tid, 1024);
    }
    // sorted slots + per-block boxes: wave w handles the 64-slot groups w, w + 16, ...; a block is D.ppl groups
    float* boxes = (float*)(key + npow);                                   // [group][6] scratch behind the keys
    const int groups = nblk * D.ppl;
    for (int gq = wv; gq < groups; gq += 16) {
        const unsigned long long k = key[gq * 64 + lane];
        float4 p = make_float4(INFINITY, INFINITY, INFINITY, __int_as_float(0x7fffffff));
        float l[3] = {INFINITY, INFINITY, INFINITY}, h[3] = {-INFINITY, -INFINITY, -INFINITY};
        if (k != ~0ull) {
            const int j = base + (int)(k & 0xFFFFull);
            const float4 y = W.y4[j];
            p = make_float4(y.x, y.y, y.z, __int_as_float(j));
            l[0] = h[0] = y.x; l[1] = h[1] = y.y; l[2] = h[2] = y.z;
        }
        W.ys4[(size_t)blk0 * BS + gq * 64 + lane] = p;
#pragma unroll
        for (int c = 0; c < 3; ++c)
            for (int off = 32; off >= 1; off >>= 1) {
                l[c] = fminf(l[c], __shfl_xor(l[c], off, 64));
                h[c] = fmaxf(h[c], __shfl_xor(h[c], off, 64));
            }
        if (lane == 0) {
            float* o = boxes + 6 * gq;
            o[0] = l[0]; o[1] = l[1]; o[2] = l[2]; o[3] = h[0]; o[4] = h[1]; o[5] = h[2];
        }
    }
    __syncthreads();
    for (int b = tid; b < nblk; b += 1024) {
        float o[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
        for (int g = 0; g < D.ppl; ++g)
            for (int c = 0; c < 3; ++c) {
                o[c] = fminf(o[c], boxes[6 * (b * D.ppl + g) + c]);
                o[3 + c] = fmaxf(o[3 + c], boxes[6 * (b * D.ppl + g) + 3 + c]);
            }
        for (int c = 0; c < 6; ++c) W.ybox[c * 64 * D.nbt + blk0 + b] = o[c];     // six planes of 64 nbt boxes
    }
}

// The same for the predicted cloud, whose points move rigidly with their cluster every epoch: each cluster's
// LOCAL points are cut into k-d leaves of BS once per train (a cluster of n points owns ceil(n / BS) blocks, the
// last one padded), and k_head, which transforms cluster r's points anyway, walks them in slot order and reduces
// each block's box from the transformed coordinates it has just computed.  One workgroup per cluster; all segments
// of a level are split by ONE bitonic sort over the cluster's slots with key = (first block of the slot's segment,
// coordinate, index): segments hold exactly BS * blocks slots, so after the sort each one sits in its own slot range
// again, ordered on the level's axis, and is halved at a block boundary (padding sorts last inside its segment and
// so stays in the cluster's last block).  psl4[slot] = (local xyz, bits(point index)), padding = index INT_MAX;
// sb[c] = first block of cluster c, sb[K] = blocks in use.
__global__ __launch_bounds__(512) void k_sort_p(Dims D, Ws W0, size_t bstride) {
    const Ws W = ws_shift(W0, blockIdx.z * bstride);
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    unsigned long long* key = (unsigned long long*)smem_raw;
    __shared__ unsigned short sf[PS_MAXB], sm[PS_MAXB];                 // per block of this cluster: its segment's first block, block count
    __shared__ int s_more;
    const int tid = threadIdx.x, c = blockIdx.x;
    const int BS = 64 * D.ppl, bshift = D.ppl == 1 ? 6 : 8;
    int first = 0;                                                       // blocks of the clusters before this one
    for (int i = 0; i < c; ++i) first += (W.off[i + 1] - W.off[i] + BS - 1) >> bshift;
    const int p0 = W.off[c], n = W.off[c + 1] - p0, m = (n + BS - 1) >> bshift;
    if (tid == 0) {
        W.sb[c] = first;
        if (c == D.K - 1) W.sb[D.K] = first + m;
    }
    if (m == 0) return;
    const int ns = m * BS;
    int npow = 64;
    while (npow < ns) npow <<= 1;
    if (npow > YS_CHUNK) {
        // a cluster above 16384 slots does not fit one workgroup's LDS: its blocks keep the points' given order (any
        // permutation gives the same search result -- the boxes are reduced from whatever a block holds -- it only prunes less)
        for (int sl = tid; sl < ns; sl += 512) {
            float4 o = make_float4(0.f, 0.f, 0.f, __int_as_float(0x7fffffff));
            if (sl < n) { const float4 p = W.pts4[p0 + sl]; o = make_float4(p.x, p.y, p.z, __int_as_float(p0 + sl)); }
            W.psl4[(size_t)first * BS + sl] = o;
        }
        return;
    }
    for (int sl = tid; sl < npow; sl += 512) key[sl] = sl >= ns ? ~0ull : sl < n ? (unsigned long long)(p0 + sl) : 0xFFFFull;
    for (int b = tid; b < m; b += 512) { sf[b] = 0; sm[b] = (unsigned short)m; }
    __syncthreads();
    int levels = 0;
    for (int level = 0; level < 9; ++level) {
        if (tid == 0) s_more = 0;
        __syncthreads();
        for (int b = tid; b < m; b += 512) if (sm[b] > 1) s_more = 1;
        __syncthreads();
        if (!s_more) break;
        levels = level + 1;
        const int axis = level % 3;
        for (int sl = tid; sl < ns; sl += 512) {
            const unsigned idx = (unsigned)(key[sl] & 0xFFFFull);
            unsigned u = 0xFFFFFFFFu;
            if (idx != 0xFFFFu) {
                const float4 p = W.pts4[idx];
                u = ordered_bits(axis == 0 ? p.x : axis == 1 ? p.y : p.z);
            }
            key[sl] = ((unsigned long long)sf[sl >> bshift] << 48) | ((unsigned long long)u << 16) | idx;
        }
        __syncthreads();
        bitonic_segments(key, npow, npow, tid, 512);
        unsigned short nf = 0, nm = 0;
        const bool mine = tid < m;
        if (mine) {
            const int f = sf[tid], mm = sm[tid];
            nf = (unsigned short)f; nm = (unsigned short)mm;
            if (mm >= 2) {
                const int ml = mm >> 1;
                if (tid < f + ml) nm = (unsigned short)ml;
                else { nf = (unsigned short)(f + ml); nm = (unsigned short)(mm - ml); }
            }
        }
        __syncthreads();
        if (mine) { sf[tid] = nf; sm[tid] = nm; }
        __syncthreads();
    }
    if (D.rows)            // two more levels inside every 64-slot block: aligned groups of 16 slots are k-d cells (see k_sort_y)
        for (int sub = 0; sub < 2; ++sub) {
            const int axis = (levels + sub) % 3;
            __syncthreads();
            for (int sl = tid; sl < ns; sl += 512) {
                const unsigned idx = (unsigned)(key[sl] & 0xFFFFull);
                unsigned u = 0xFFFFFFFFu;                     // padding sorts last inside its segment
                if (idx != 0xFFFFu) {
                    const float4 p = W.pts4[idx];
                    u = ordered_bits(axis == 0 ? p.x : axis == 1 ? p.y : p.z);
                }
                key[sl] = ((unsigned long long)u << 16) | idx;
            }
            __syncthreads();
            bitonic_segments(key, npow, 64 >> sub, tid, 512);
        }
    for (int sl = tid; sl < ns; sl += 512) {
        const unsigned idx = (unsigned)(key[sl] & 0xFFFFull);
        float4 o = make_float4(0.f, 0.f, 0.f, __int_as_float(0x7fffffff));
        if (idx != 0xFFFFu) { const float4 p = W.pts4[idx]; o = make_float4(p.x, p.y, p.z, __int_as_float((int)idx)); }
        W.psl4[(size_t)first * BS + sl] = o;
    }
}

// ------------------------------------------------------------------------------------------ NN epilogue
// Runs inside the nearest-neighbour launch (results are final there): sign bits of the x->y term,
// integer sign scatter of the y->x term (exact, order independent), per-block loss partials.
struct EngineEpi {
    int* sgn_x; int4* cnt4; float* lossp_x; float* lossp_y;
    size_t bstride;
    const int* stopped;            // &state[epoch & 1].stopped of problem 0
    __device__ __forceinline__ void shift(int z) {
        const size_t b = (size_t)z * bstride;
        sgn_x = (int*)((char*)sgn_x + b); cnt4 = (int4*)((char*)cnt4 + b);
        lossp_x = (float*)((char*)lossp_x + b); lossp_y = (float*)((char*)lossp_y + b);
        stopped = (const int*)((const char*)stopped + b);
    }
    __device__ __forceinline__ void operator()(int dir, int q, int idx, float d, float qx, float qy, float qz,
                                               float tx, float ty, float tz, float& acc) const {
        if (dir == 0) {          // query pred[q], nearest y[idx]: knn(p1=x,p2=y) grad_p1 sign = (p1 > p2 ? +1 : -1)
            sgn_x[q] = (qx > tx ? 1 : 0) | (qy > ty ? 2 : 0) | (qz > tz ? 4 : 0);
        } else {                 // query y[q], nearest pred[idx]: knn(p1=y,p2=x) grad_p2 -= sign
            atomicAdd(&cnt4[idx].x, (qx > tx) ? -1 : 1);
            atomicAdd(&cnt4[idx].y, (qy > ty) ? -1 : 1);
            atomicAdd(&cnt4[idx].z, (qz > tz) ? -1 : 1);
        }
        acc += d;
    }
    __device__ __forceinline__ void finish(int dir, int blk, float s, float*) const { (dir == 0 ? lossp_x : lossp_y)[blk] = s; }
};

// The plan's nearest-neighbour launch: blocks of direction 0 search the block-sorted target frame for the predicted
// points (pruned, exact); direction 1 searches the predicted cloud for the target points -- over k_head's
// block-sorted copy of it (P1) or exhaustively.  Same epilogue, same per-block loss partials as k_nn_l1: the
// launches are interchangeable bit for bit.
template <bool P1, int PPL, int NBT, int NBP>
__global__ __launch_bounds__(NN_BLOCK) void k_nn_plan(const float* A, int na, const float* B, int nb, int blocksA,
                                                      int blocksB, EngineEpi epi, NnBlocks yb, NnBlocks pb, size_t zstride) {
    // grid.x = (blocksA + blocksB) * problems, direction 1 (the long blocks when it is exhaustive) of ALL problems
    // first: the dispatcher hands workgroups out in index order, so the long ones spread over the CUs before the
    // short ones fill in
    // every kernel argument in the entry block, one wait (see k_bd)
    asm volatile("" :: "s"(A), "s"(na), "s"(B), "s"(nb), "s"(blocksA), "s"(blocksB), "s"(epi.sgn_x), "s"(epi.cnt4), "s"(epi.lossp_x), "s"(epi.lossp_y),
                 "s"(epi.bstride), "s"(epi.stopped), "s"(yb.ts4), "s"(yb.tbox), "s"(yb.nblk), "s"(pb.ts4), "s"(pb.tbox), "s"(pb.nblk_dev), "s"(zstride));
    const int nz = gridDim.x / (blocksA + blocksB);
    int i = blockIdx.x, z, bx;
    if (i < blocksB * nz) { z = i / blocksB; bx = blocksA + (i - z * blocksB); }
    else { i -= blocksB * nz; z = i / blocksA; bx = i - z * blocksA; }
    const size_t zb = z * zstride;
    A = (const float*)((const char*)A + zb);
    B = (const float*)((const char*)B + zb);
    yb.ts4 = (const float4*)((const char*)yb.ts4 + zb);
    yb.tbox = (const float*)((const char*)yb.tbox + zb);
    epi.shift(z);
    if (bx < blocksA) nn_l1_block_pruned<NBT, PPL, EngineEpi, true, false>(A, na, 4, yb, 0, epi, bx, epi.stopped, B, 4);
    else if constexpr (P1) {
        pb.ts4 = (const float4*)((const char*)pb.ts4 + zb);
        pb.tbox = (const float*)((const char*)pb.tbox + zb);
        pb.nblk_dev = (const int*)((const char*)pb.nblk_dev + zb);
        nn_l1_block_pruned<NBP, PPL, EngineEpi, true, true>(B, nb, 4, pb, 1, epi, bx - blocksA, epi.stopped, A, 4);
    } else nn_l1_block<4, int, EngineEpi>(A, na, 4, B, nb, 4, nullptr, nullptr, nullptr, nullptr, blocksA, epi, bx);
}

// Round 5: the same launch with sixteen queries per wave (nn_l1_rows): both directions take their QUERIES in slot order of their own
// block-sorted copy (ps4 / ys4: neighbours in space share a wave), 64-point blocks on both sides.
template <int NBT, int NBP>
__global__ __launch_bounds__(NN_ROWS_BLOCK) void k_nn_rows(const float* A, const float* B, int blocksA, int blocksB, EngineEpi epi, NnBlocks yb, NnBlocks pb,
                                                      size_t zstride) {
    asm volatile("" :: "s"(A), "s"(B), "s"(blocksA), "s"(blocksB), "s"(epi.sgn_x), "s"(epi.cnt4), "s"(epi.lossp_x), "s"(epi.lossp_y),
                 "s"(epi.bstride), "s"(epi.stopped), "s"(yb.ts4), "s"(yb.tbox), "s"(yb.nblk), "s"(pb.ts4), "s"(pb.tbox), "s"(pb.nblk_dev), "s"(zstride));
    const int nz = gridDim.x / (blocksA + blocksB);
    int i = blockIdx.x, z, bx;
    if (i < blocksB * nz) { z = i / blocksB; bx = blocksA + (i - z * blocksB); }
    else { i -= blocksB * nz; z = i / blocksA; bx = i - z * blocksA; }
    const size_t zb = z * zstride;
    A = (const float*)((const char*)A + zb);
    B = (const float*)((const char*)B + zb);
    yb.ts4 = (const float4*)((const char*)yb.ts4 + zb);
    yb.tbox = (const float*)((const char*)yb.tbox + zb);
    pb.ts4 = (const float4*)((const char*)pb.ts4 + zb);
    pb.tbox = (const float*)((const char*)pb.tbox + zb);
    pb.nblk_dev = (const int*)((const char*)pb.nblk_dev + zb);
    epi.shift(z);
    if (bx < blocksA) nn_l1_rows<NBT, EngineEpi, false, true>(pb.ts4, 0, pb.nblk_dev, pb.nblk, yb, 0, epi, bx, epi.stopped, B, epi.lossp_x);
    else nn_l1_rows<NBP, EngineEpi, true, false>(yb.ts4, yb.nblk, nullptr, yb.nblk, pb, 1, epi, bx - blocksA, epi.stopped, A, epi.lossp_y);
}

// ------------------------------------------------------------------------------------------ control + cluster grads
// K blocks.  Every block derives the same loss and the same decisions from the same inputs; block 0
// advances the double-buffered state.  Then block k reduces cluster k's point gradients to
// [dL/dR | dL/dt] in a fixed order and thread 0 pulls them back through the pose head.
// bc1 / bc2_sqrt: the bias corrections of step S.step + 1 (k_prep's table)
__device__ __forceinline__ TrainState advance_state(const TrainState& S, float loss, const Hyper& hy, double bc1, float bc2_sqrt) {
    TrainState N = S;
    const bool improved = loss < S.min_loss;
    N.last_loss = loss;
    N.epochs_run = S.epochs_run + 1;
    // (selects of values, not conditional stores into N: hipcc turned those into ONE store through a computed address into a stack
    //  copy of the struct -- 16 bytes of scratch per lane and a scratch round trip in the middle of the kernel)
    const int count = improved ? 0 : S.count + 1;
    N.min_loss = improved ? loss : S.min_loss;
    N.best_epoch = improved ? S.epochs_run : S.best_epoch;
    N.count = count;
    N.stopped = (!improved && count > hy.stop) ? 1 : S.stopped;                      // mlp_reg.py:107-111
    if (!N.stopped) {
        // optimizer.step() of this epoch uses S.lr (torch.optim.Adam, betas (0.9, 0.999), eps 1e-8)
        N.step = S.step + 1;
        N.step_size = (float)(S.lr / bc1);
        N.bc2_sqrt = bc2_sqrt;
        // scheduler.step(loss): ReduceLROnPlateau(mode='min', threshold 1e-4 rel, cooldown 0, min_lr 0, eps 1e-8)
        const double cur = (double)loss;
        if (cur < S.sched_best * (1.0 - 1e-4)) { N.sched_best = cur; N.sched_bad = 0; }
        else N.sched_bad = S.sched_bad + 1;
        if (N.sched_bad > hy.patience) {
            const double nl = fmax(S.lr * (double)hy.factor, 0.0);
            if (S.lr - nl > 1e-8) N.lr = nl;
            N.sched_bad = 0;
        }
    }
    return N;
}

constexpr int GC_QMAX = 3;             // hidden units per thread of k_gradc's last phase: H2 <= 768 = 3 x 256
// FUSED (the gradient role of k_gbd, 256 live threads of a 512-thread workgroup): what the other roles of the SAME launch consume --
// g_out, g_h2, the three scalars of the advanced state -- goes out as 16-byte write-through (sc1) stores, and every block, also one
// that leaves early (a stopped train), drains its stores and counts itself in W.sync[0] on its way out.
__device__ __forceinline__ void gradc_arrive(const Ws& W) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this thread's write-through stores have been acknowledged ...
    __syncthreads();                                       // ... every thread's
    if (threadIdx.x == 0) __hip_atomic_fetch_add(W.sync, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool FUSED>
__device__ __forceinline__ void gradc_role(const Dims& D, const Ws& W, int epoch, int nbx, int nby, int k) {
    __shared__ float red[4][14];
    __shared__ float s_loss;
    __shared__ float s_go[16];
    __shared__ __attribute__((aligned(16))) float s_gh[FUSED ? 256 * GC_QMAX : 4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // ONE round trip for everything that does not depend on another load: the state (with the bias corrections of the
    // coming step), the cluster bounds, the hyper-parameters, the NN launch's loss partials, the pose row, and the
    // operands of the last phase (this thread's hidden units of pose row k and their output-layer weights) -- round 2
    // waited for the state first (the early exit of a stopped train) and then for the table entries it indexes
    const TrainState S = W.state[epoch & 1];
    const int handoff_err = W.sync[3];                  // an in-launch hand-off of an earlier epoch timed out (the opt-in fused launches only): see below
    const int b0 = W.off[k], e0 = W.off[k + 1];
    const Hyper hy = *W.hyper;
    float a = tid < nbx ? W.lossp_x[tid] : 0.f, b = tid < nby ? W.lossp_y[tid] : 0.f;
    const float m2v = W.m2[16 * k + (tid & 15)];
    float sv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) sv[i] = W.head_save[16 * k + i];
    const float* h2cur = (epoch & 1) ? W.h2[1] : W.h2[0];
    float w3v[GC_QMAX][8], h2v[GC_QMAX];
#pragma unroll
    for (int q = 0; q < GC_QMAX; ++q) {
        const int o = min(q * 256 + tid, D.H2 - 1);
        const bool isA = o < D.HA;
        const float* Pc = (epoch & 1) ? W.P1 : W.P;
        const float* wb = isA ? Pc + D.oW3A + o : Pc + D.oW3B + (o - D.HA);
        const int stride = isA ? D.HA : D.HB, nj = isA ? D.OA : D.OB;
#pragma unroll
        for (int j = 0; j < 8; ++j) w3v[q][j] = wb[(size_t)min(j, nj - 1) * stride];
        h2v[q] = h2cur[(size_t)k * D.H2 + o];
    }
    // hipcc sinks a load into the block that uses it -- here past the early exit below, i.e. BEHIND the state's round trip: a use it
    // cannot move keeps the whole batch above in one round trip with the state
    asm volatile("" :: "v"(a), "v"(b), "v"(m2v), "v"(sv[0]), "v"(sv[1]), "v"(sv[2]), "v"(sv[3]), "v"(sv[4]), "v"(sv[5]), "v"(sv[6]), "v"(sv[7]),
                 "v"(sv[8]), "v"(sv[9]), "v"(sv[10]), "v"(sv[11]), "v"(sv[12]), "v"(sv[13]), "v"(sv[14]), "v"(sv[15]), "v"(b0), "v"(e0), "v"(hy.lr));
#pragma unroll
    for (int q = 0; q < GC_QMAX; ++q)
        asm volatile("" :: "v"(h2v[q]), "v"(w3v[q][0]), "v"(w3v[q][1]), "v"(w3v[q][2]), "v"(w3v[q][3]), "v"(w3v[q][4]), "v"(w3v[q][5]), "v"(w3v[q][6]),
                     "v"(w3v[q][7]));
    if (S.stopped || handoff_err) {                     // (a failed hand-off stops the train and poisons what it returns: loud, not silently inconsistent)
        if (k == 0 && tid == 0) {
            TrainState E = S; E.stopped = 1;
            if (handoff_err) { E.min_loss = NAN; W.result[0] = NAN; }
            W.state[(epoch + 1) & 1] = E;
            if constexpr (FUSED) st4_wt((float*)W.sync, 32, make_float4(__int_as_float(1), 0.f, 0.f, 0.f));
        }
        if constexpr (FUSED) gradc_arrive(W);
        return;
    }
    // second round trip: this thread's first point of the cluster (the loss reduction runs in its shadow)
    const int n_first = min(b0 + tid, D.NP - 1);
    const float4 p_first = W.pts4[n_first];
    const int4 c_first = W.cnt4[n_first];
    const int s_first = W.sgn_x[n_first];
    const float4 pr_first = W.pred4[n_first];
    // ---- loss = sum_x / NP + sum_y / NT from the NN launch's per-block partials (fixed order)
    for (int i = tid + 256; i < nbx; i += 256) a += W.lossp_x[i];
    for (int i = tid + 256; i < nby; i += 256) b += W.lossp_y[i];
    a = wave_sum_fast(a); b = wave_sum_fast(b);
    if (lane == 0) { red[wv][0] = a; red[wv][1] = b; }
    __syncthreads();
    if (tid == 0) {
        const float sa = ((red[0][0] + red[1][0]) + red[2][0]) + red[3][0];
        const float sb = ((red[0][1] + red[1][1]) + red[2][1]) + red[3][1];
        s_loss = sa / (float)D.NP + sb / (float)D.NT;
    }
    __syncthreads();
    const float loss = s_loss;
    TrainState N = advance_state(S, loss, hy, S.next_bc1, S.next_bc2s);
    const bool improved = loss < S.min_loss;
    if (improved) {                                     // best_pcd / best_m  (mlp_reg.py:102-106)
        for (int n = b0 + tid; n < e0; n += 256) {
            const float4 p = (n == b0 + tid) ? pr_first : W.pred4[n];
            W.best_pred[3 * (size_t)n] = p.x; W.best_pred[3 * (size_t)n + 1] = p.y; W.best_pred[3 * (size_t)n + 2] = p.z;
        }
        if (tid < 16) W.best_m[16 * k + tid] = m2v;
    }
    if (k == 0 && tid == 0) {
        N.next_bc1 = W.bc1[min(N.step + 1, D.epochs)];          // for the next epoch (another launch): off this one's critical path
        N.next_bc2s = W.bc2s[min(N.step + 1, D.epochs)];
        W.state[(epoch + 1) & 1] = N;
        W.loss_hist[S.epochs_run] = loss;
        W.lr_hist[S.epochs_run] = (float)S.lr;
        W.result[0] = N.min_loss; W.result[1] = (float)N.epochs_run; W.result[2] = (float)N.lr; W.result[3] = (float)N.best_epoch;
        if constexpr (FUSED) st4_wt((float*)W.sync, 32, make_float4(__int_as_float(N.stopped), N.step_size, N.bc2_sqrt, 0.f));
    }
    if (N.stopped) {                                    // the reference breaks before backward()
        if constexpr (FUSED) gradc_arrive(W);
        return;
    }
    // ---- per-cluster reduction of the point gradients
    const float gx = 1.0f / (float)D.NP, gy = 1.0f / (float)D.NT;
    float acc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = 0.f;
    for (int n = b0 + tid; n < e0; n += 256) {
        const bool first = n == b0 + tid;
        const float4 p = first ? p_first : W.pts4[n];
        const int4 c = first ? c_first : W.cnt4[n];
        const int sb = first ? s_first : W.sgn_x[n];
        const float g[3] = {((sb & 1) ? gx : -gx) + gy * (float)c.x, ((sb & 2) ? gx : -gx) + gy * (float)c.y,
                            ((sb & 4) ? gx : -gx) + gy * (float)c.z};
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            acc[4 * q] = fmaf(g[q], p.x, acc[4 * q]); acc[4 * q + 1] = fmaf(g[q], p.y, acc[4 * q + 1]);
            acc[4 * q + 2] = fmaf(g[q], p.z, acc[4 * q + 2]); acc[4 * q + 3] += g[q];
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 12; ++i) { const float r = wave_sum_fast(acc[i]); if (lane == 0) red[wv][i] = r; }
    __syncthreads();
    if (tid == 0) {
        float G12[12];
        for (int i = 0; i < 12; ++i) G12[i] = ((red[0][i] + red[1][i]) + red[2][i]) + red[3][i];
        for (int i = 0; i < 16; ++i) W.gm2[16 * k + i] = i < 12 ? G12[i] : 0.f;
        const float G[9] = {G12[0], G12[1], G12[2], G12[4], G12[5], G12[6], G12[8], G12[9], G12[10]};
        const float gt[3] = {G12[3], G12[7], G12[11]};
        float go[16];                          // [0..2] branch A, [4..11] branch B
        for (int i = 0; i < 16; ++i) go[i] = 0.f;
        if (D.rot == ROT_6D) {
            go[0] = gt[0]; go[1] = gt[1]; go[2] = gt[2];
            rot6d_to_matrix_vjp(sv, G, go + 4);
        } else if (D.rot == ROT_RPY) {
            go[0] = gt[0]; go[1] = gt[1]; go[2] = gt[2];
            float ge[3];
            euler_xyz_to_matrix_vjp(sv, G, ge);
            for (int i = 0; i < 3; ++i) go[4 + i] = ge[i] * (1.f - sv[4 + i] * sv[4 + i]);      // through the Tanh that ends decoder_2
        } else if (D.rot == ROT_Q) {
            go[0] = gt[0]; go[1] = gt[1]; go[2] = gt[2];
            float gu[4];
            quat_to_matrix_vjp(sv, G, gu);
            const float nrm = sv[4];
            if (nrm > 1e-12f) {
                const float dot = sv[0] * gu[0] + sv[1] * gu[1] + sv[2] * gu[2] + sv[3] * gu[3];
#pragma unroll
                for (int i = 0; i < 4; ++i) go[4 + i] = (gu[i] - sv[i] * dot) / nrm;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) go[4 + i] = gu[i] / 1e-12f;
            }
        } else {
            float gdq[8];
            dq_to_se3_vjp(sv, G, gt, gdq);
            for (int i = 0; i < 8; ++i) go[4 + i] = gdq[i];
        }
        if constexpr (FUSED) {
            for (int i = 0; i < 12; ++i) s_go[i] = go[i];
            for (int i = 0; i < 3; ++i) st4_wt(W.g_out, 16 * k + 4 * i, make_float4(go[4 * i], go[4 * i + 1], go[4 * i + 2], go[4 * i + 3]));
        } else {
            for (int i = 0; i < 12; ++i) { W.g_out[16 * k + i] = go[i]; s_go[i] = go[i]; }
        }
    }
    __syncthreads();
    // ---- pose row k of dL/d(hidden pre-activation): g_h2[k][o] = act'(h2[k][o]) * sum_j g_out[k][j] W3[j][o]
    // (a row needs only its own g_out, so it is final here: k_bwd2 and k_dw read the finished matrix)
#pragma unroll
    for (int q = 0; q < GC_QMAX; ++q) {
        const int o = q * 256 + tid;
        if (o < D.H2) {
            const bool isA = o < D.HA;
            const int nj = isA ? D.OA : D.OB, gofs = isA ? 0 : 4;
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) if (j < nj) sum = fmaf(s_go[gofs + j], w3v[q][j], sum);
            const float gh = sum * act_grad(h2v[q], D.slope);
            if constexpr (FUSED) s_gh[o] = gh;           // the row leaves as 16-byte write-through stores below
            else W.g_h2[(size_t)k * D.H2 + o] = gh;
        }
    }
    if constexpr (FUSED) {
        __syncthreads();
        if (4 * tid < D.H2) st4_wt(W.g_h2, k * D.H2 + 4 * tid, *(const float4*)(s_gh + 4 * tid));
        gradc_arrive(W);
    }
}

__global__ __launch_bounds__(256) void k_gradc(Dims D, Ws W0, int epoch, int nbx, int nby, size_t bstride) {
    // every kernel argument in the entry block, one wait (see k_bd)
    asm volatile("" :: "s"(W0.P), "s"(W0.P1), "s"(W0.h2[0]), "s"(W0.h2[1]), "s"(W0.head_save), "s"(W0.m2), "s"(W0.gm2), "s"(W0.pts4), "s"(W0.pred4),
                 "s"(W0.sgn_x), "s"(W0.cnt4), "s"(W0.lossp_x), "s"(W0.lossp_y), "s"(W0.g_out), "s"(W0.g_h2), "s"(W0.state), "s"(W0.bc1), "s"(W0.bc2s),
                 "s"(W0.best_m), "s"(W0.best_pred), "s"(W0.loss_hist), "s"(W0.lr_hist), "s"(W0.result), "s"(W0.off), "s"(W0.hyper), "s"(W0.sync),
                 "s"(D.rot), "s"(D.K), "s"(D.H2), "s"(D.HA), "s"(D.HB), "s"(D.OA), "s"(D.OB), "s"(D.NP), "s"(D.NT), "s"(D.epochs), "s"(D.slope),
                 "s"(D.oW3A), "s"(D.oW3B), "s"(epoch), "s"(nbx), "s"(nby), "s"(bstride));
    const Ws W = ws_shift(W0, blockIdx.z * bstride);
    gradc_role<false>(D, W, epoch, nbx, nby, blockIdx.x);
}


// ------------------------------------------------------------------------------------------ Adam
__device__ __forceinline__ float adam_value(float p, float& mm, float& vv, float g, float step_size, float bc2_sqrt) {
    // torch single-tensor Adam: exp_avg.lerp_(g, 1-b1); exp_avg_sq.mul_(b2).addcmul_(g, g, 1-b2);
    // denom = sqrt(v)/sqrt(bc2) + eps; p.addcdiv_(exp_avg, denom, value=-step_size)
    const float b1w = (float)(1.0 - 0.9), b2 = 0.999f, b2w = (float)(1.0 - 0.999);
    mm = mm + b1w * (g - mm);
    vv = vv * b2 + (b2w * g) * g;
    const float denom = sqrtf(vv) / bc2_sqrt + 1e-8f;
    return p + (-step_size * mm) / denom;
}

// ------------------------------------------------------------------------------------------ backward to x1 + encoder update
// g_x1 = act'(x1) * (g_h2 . W2), COMPLETE per column block: a workgroup owns B2_CB = 16 hidden units of the encoder,
// i.e. 16 columns of W2 over all H2 rows, so nothing is left to reduce across workgroups (round 2 wrote 16 partial slabs
// of g_x1 -- 0.65 MB per problem -- that k_dw's encoder blocks gathered with 4-byte loads, every 128-byte line fetched by
// four workgroups on different XCDs: the 1.6x read amplification of that kernel).  The same workgroup then owns the 16
// encoder rows W1[c0 .. c0+16): weight gradients, Adam, and the NEXT epoch's encoder activation of its 16 units from the
// registers that hold the updated rows (the MLP input is the same every epoch: m.clone() of the same m, mlp_reg.py:62).
// Threads: 8 column pairs x (8 AW) row slices of W2; a thread keeps its OPS rows of its two columns in registers (all
// loads in flight at once, 64-byte row segments), g_h2 comes from LDS (LDS-DMA, RB pose rows per pass, broadcast reads);
// the slice partials of a (pose row, column) are summed in a fixed tree: the 8 slices of a wave (row_ror DPP, two lane
// permutes), then the waves in order through LDS.  The rows of a pass are straight-line code (rows past the end repeat the
// last one and are not stored): per-row branches kept hipcc from overlapping the rows' LDS reads and lane permutes.
#ifdef CREG_BD_STAMPS
// Measurement build (tests/measure/bd_stamps.py): where the two roles of k_bd spend a launch.  Thread 0 of every workgroup reads the
// 100 MHz wall clock at its phase boundaries; per launch slot (epoch & 255) the earliest start and the latest end of each role, per
// role the summed phase durations and workgroup counts.  Recorded only while the host has armed it (the back-to-back leg of
// creg_train_plan_profile).
struct BdStamps {
    unsigned long long first[256], endB[256], endD[256], startB[256], startD[256];
    unsigned long long phase[2][8], blocks[2];
    int armed;
};
__device__ BdStamps g_bd_st;
#define BD_T0 unsigned long long bd_t[8]; int bd_n = 0; if (threadIdx.x == 0) bd_t[bd_n++] = bd_entry;
#define BD_T  if (threadIdx.x == 0) bd_t[bd_n++] = wall_clock64();
#define BD_TEND(role, epoch) do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); if (threadIdx.x == 0) { bd_t[bd_n++] = wall_clock64(); \
    if (g_bd_st.armed) { const int sl_ = (epoch) & 255; atomicMin(&g_bd_st.first[sl_], bd_t[0]); \
        atomicMin(role ? &g_bd_st.startD[sl_] : &g_bd_st.startB[sl_], bd_t[0]); atomicMax(role ? &g_bd_st.endD[sl_] : &g_bd_st.endB[sl_], bd_t[bd_n - 1]); \
        for (int k_ = 1; k_ < bd_n; ++k_) atomicAdd(&g_bd_st.phase[role][k_ - 1], bd_t[k_] - bd_t[k_ - 1]); atomicAdd(&g_bd_st.blocks[role], 1ull); } } } while (0)
#else
#define BD_T0
#define BD_T
#define BD_TEND(role, epoch)
#endif
constexpr int BD_THREADS = 512;       // workgroups of the backward launch k_bd (both roles)
constexpr int B2_THREADS = BD_THREADS;
constexpr int B2_WAVES = B2_THREADS / 64;
constexpr int B2_CB = 16;             // columns of W2 (= hidden units of the encoder) per workgroup: ONE 16-wide MFMA tile
constexpr int B2_RB = 32;             // pose rows per pass: two 16-row MFMA tiles (K <= 32: one pass)
__host__ __device__ inline int b2_smem_floats(int K, int IN, int H2) {
    // the per-wave partial tiles [8][32][16], the features [K][IN], g_x1 [K][16] and the next-activation tile [K][16] of this block's
    // columns, the block's slab of W2 [H2][16]
    return B2_WAVES * B2_RB * B2_CB + K * IN + 2 * K * B2_CB + H2 * B2_CB;
}
// sum over the 16 lanes of a DPP row; every lane of the row receives it
__device__ __forceinline__ float row_sum16(float v) {
    CREG_DPP_STEP(v, 0xB1, 0xF);    // quad_perm [1,0,3,2]
    CREG_DPP_STEP(v, 0x4E, 0xF);    // quad_perm [2,3,0,1]
    CREG_DPP_STEP(v, 0x141, 0xF);   // row_half_mirror
    CREG_DPP_STEP(v, 0x140, 0xF);   // row_mirror
    return v;
}

// Round 4: the product g_h2 . W2 of a block is a [K x H2] . [H2 x 16] GEMM and runs on the matrix cores as
// v_mfma_f32_16x16x4_f32 tiles -- exact float32, bit for bit a k-ordered fmaf chain (cdna_hip_programming.md section 3), so the
// plan stays deterministic and bit-reproducible; only the association of the sum over the hidden rows differs from round 3's
// (8 lane slices x 8 waves then, 8 waves x one in-order chain of H2 / 8 rows now).  Tiles: M = 16 pose rows (K = 20: two tiles, the
// second one mostly padding -- rows past K repeat row K - 1 and land in accumulator rows nobody reads), N = the block's 16 columns,
// contraction over the hidden rows o: wave w owns rows [w KW, (w + 1) KW), KW / 4 k-steps.  MFMA operand layout: lane l holds
// A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15]; D[i = 4 (l >> 4) + v][j = l & 15] in accumulator register v.
//   B operand = W2[o][c0 + j]: the block's slab W2[:, c0 .. c0 + 16) goes to LDS by LDS-DMA, 16 bytes per lane (a wave-instruction
//     moves 16 rows of 64 bytes), each wave staging exactly the rows it multiplies -- so its own vmcnt wait is all the
//     synchronisation the slab needs -- and is read back one dword per lane and k-step.  (First build: one global dword load per
//     lane and k-step.  A CU's address path takes ~16 cycles per wave-instruction whatever its width, so those 24 quarter-width
//     loads per lane cost 3 000 cycles per workgroup: the block spent 2.5 us just ISSUING its loads.)
//   A operand = g_h2[r][o]: straight from global memory too (k_gradc wrote it a launch earlier) -- with the rows of a wave ordered
//     o(g, i, kk) = 16 g + 4 kk + i (k-step 4 g + i, k index kk: any bijection works as long as A and B agree) a lane's operands of
//     four consecutive k-steps are ONE 16-byte load g_h2[r][16 g + 4 kk .. + 3] (KW % 16 == 0: hidden 256 / 512); narrower shapes
//     load a dword per k-step with o(s, kk) = 4 s + kk.
// Nothing of the GEMM is staged in LDS any more (round 3 staged all of g_h2, 61 KB per workgroup, and waited for ALL of it before the
// first FMA): every load is in flight at once and the MFMAs start on the first operands that land.  The per-wave partial tiles are
// summed over the waves in wave order through LDS as before.  Then, unchanged: activation gradient, the 16 encoder rows' weight
// gradients + Adam, and the NEXT epoch's encoder activation of the block's 16 units from the registers that hold the updated rows.
// ---- fused backward launch (k_gbd): the consumers' side of the hand-off ----------------------------------------------------------
// A consumer role first requests everything that does NOT depend on this epoch's gradients -- its parameter rows and Adam moments,
// its slab of W2, the activations: 96-110 KB per workgroup, the part of k_bd that a CU's ~11 B/cycle intake makes slow -- then waits
// here until the K blocks of the gradient role have counted themselves in, and only then asks for the gradients and the three
// scalars of the advanced state (sc1 loads: the producers wrote them through with sc1 stores and nothing of this launch has touched
// those lines before, MI355X_MICROARCH.md "valid forms").  One polling lane per workgroup; the poll is bounded (a few hundred
// milliseconds): a launch whose producers never arrive -- which cannot happen while workgroups are dispatched in index order, the
// gradient role owning the lowest indices -- ends as a stopped train instead of hanging the queue.
struct GbdRecord { int stopped; float step_size, bc2_sqrt; int step; };
__device__ __forceinline__ bool gbd_wait(const Ws& W, int K) {
    __shared__ int s_ok;
    if (threadIdx.x == 0) {
        int it = 0, seen = __hip_atomic_load(W.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (seen < K && ++it < (1 << 19)) { __builtin_amdgcn_s_sleep(16); seen = __hip_atomic_load(W.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        s_ok = seen >= K;
        if (!s_ok) __hip_atomic_store(W.sync + 3, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      //  a hand-off that timed out is an ERROR of the train:
    }                                                                                                     //   k_gradc of the next epoch / k_params_home stop it and return NaN as its min_loss
    __syncthreads();
    return s_ok != 0;
}
__device__ __forceinline__ GbdRecord gbd_record(const Ws& W, bool ok) {
    const u32x4v v = __builtin_amdgcn_raw_buffer_load_b128(buf_rsrc(W.sync, 64 * 4), 32 * 4, 0, 16);      // sc1
    GbdRecord r;
    r.stopped = ok ? (int)v[0] : 1; r.step_size = __uint_as_float(v[1]); r.bc2_sqrt = __uint_as_float(v[2]);
    return r;
}

// L2IN (round 6): the D role of the SAME launch goes on to the next hidden activation (what k_l2 computed a launch later) and needs the next
// encoder activation this role produces: the tile leaves as 16-byte write-through stores, the wave drains them, and the workgroup counts
// itself in W.sync[1] -- a monotonic count of B-role arrivals since the train began (nB per optimizer step; k_prep / k_set_state set it).
template <int KW, bool X, bool FUSED = false, bool L2IN = false>              // W2 rows per wave: H2 = 8 KW; X: more than 64 input features ('6d': 72) -- lanes i4 < 2 of a row take a second float4
__device__ __forceinline__ void bwd2_role(const Dims& D, const Ws& W, int epoch, int blk, float* sh, unsigned long long bd_entry) {
    constexpr int NS = KW / 4;                 // k-steps of a wave
    constexpr bool V4 = KW % 16 == 0;          // A operand as 16-byte loads
    constexpr int NA = V4 ? NS / 4 : NS;       // A loads per tile and lane
    float* red = sh;                                   // [8][32][16] per-wave partial tiles of the pass
    float* encs = red + B2_WAVES * B2_RB * B2_CB;      // [K][IN] MLP input features
    float* gxs = encs + D.K * D.IN;                    // [K][16] g_x1 of this block's columns,
    float* xt = gxs + D.K * B2_CB;                     // [K][16] next encoder activation of this block's units
    float* w2s = xt + D.K * B2_CB;                     // [H2][16] the block's slab of W2 (16-byte aligned: every size above is a multiple of 4 floats)
    // (no early exit on `stopped` before the loads are issued: every store below is gated; an exit branch here would let hipcc
    //  sink the loads below it and serialise them)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, lj = lane & 15, kk = lane >> 4;
    // The two roles are one function to hipcc: its wait-count pass lets the OTHER role's requests count as possibly pending here and
    // put a vmcnt(0) in the middle of this role's own requests (before the first redefinition of a register the other role loads
    // into).  A wait it can see -- free at run time: nothing is outstanding when a role starts -- clears that state.
    __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0)
    const int c0 = blk * B2_CB, par = epoch & 1;
    const float* x1cur = par ? W.x1[1] : W.x1[0];     // (a runtime index into the shifted struct would go to scratch)
    float* x1next = par ? W.x1[0] : W.x1[1];
    const float* Pc = par ? W.P1 : W.P;               // this epoch's parameters; the updated rows go to the other buffer
    float* Pn = par ? W.P : W.P1;
    BD_T0
    // every load of the first pass is requested before the first wait
    {   // this wave's KW rows of the slab: lane l moves the 16 bytes W2[row 16 i + (l >> 2)][c0 + 4 (l & 3) ..] to slab row-major
        const float* src = Pc + D.oW2 + (size_t)(wv * KW + (lane >> 2)) * D.H + c0 + 4 * (lane & 3);
        float* dst = w2s + wv * KW * B2_CB;
#pragma unroll
        for (int i = 0; i < KW; i += 16)
            if (i + (lane >> 2) < KW)                  // (KW = 8, 12, 24: the last instruction is partial; inactive lanes write nothing)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)i * D.H),
                                                 (__attribute__((address_space(3))) void*)(dst + i * B2_CB), 16, 0, 0);
    }
    const float* gbase = W.g_h2 + wv * KW + (V4 ? 4 * kk : kk);        // + row * H2 (+ 16 g | 4 s)
    const __amdgpu_buffer_rsrc_t rGH = buf_rsrc(W.g_h2, D.KP * D.H2 * 4);
    auto load_a = [&](int row, float* dst) {                           // one tile's A operands of this lane: pose row `row` (clamped)
        if constexpr (FUSED) {                                         // written by the gradient role of THIS launch: sc1 loads
            const int off = (min(row, D.K - 1) * D.H2 + wv * KW + (V4 ? 4 * kk : kk)) * 4;
#pragma unroll
            for (int q = 0; q < NA; ++q) {
                if constexpr (V4) {
                    const u32x4v v = __builtin_amdgcn_raw_buffer_load_b128(rGH, off + 64 * q, 0, 16);
                    dst[4 * q] = __uint_as_float(v[0]); dst[4 * q + 1] = __uint_as_float(v[1]); dst[4 * q + 2] = __uint_as_float(v[2]); dst[4 * q + 3] = __uint_as_float(v[3]);
                } else dst[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rGH, off + 16 * q, 0, 16));
            }
        } else {
            const float* g = gbase + (size_t)min(row, D.K - 1) * D.H2;
#pragma unroll
            for (int q = 0; q < NA; ++q) {
                if constexpr (V4) { const float4 v = *(const float4*)(g + 16 * q); dst[4 * q] = v.x; dst[4 * q + 1] = v.y; dst[4 * q + 2] = v.z; dst[4 * q + 3] = v.w; }
                else dst[q] = g[4 * q];
            }
        }
    };
    float a0[NS], a1[NS];
    const bool two0 = D.K > 16;                        // workgroup-uniform
    if constexpr (!FUSED) {
        load_a(lj, a0);
        if (two0) load_a(16 + lj, a1);
    }
    // encoder rows: thread = (half, row of the block, float4 of its IN inputs); both halves hold the row (same operands,
    // same Adam result) and share the pose rows of the next-activation loop
    const int half = tid >> 8, t2 = tid & 255, row = t2 >> 4, i4 = t2 & 15, hu = c0 + row;
    const bool ain = 4 * i4 < D.IN;
    const int ei = min(4 * i4, D.IN - 4);
    const size_t wi = (size_t)D.oW1 + (size_t)hu * D.IN + ei;
    float4 pw = *(const float4*)(Pc + wi), pm = *(const float4*)(W.AM + wi), pv = *(const float4*)(W.AV + wi);
    const bool ain2 = X && 64 + 4 * i4 < D.IN;         // the features past 64
    const int ei2 = min(64 + 4 * i4, D.IN - 4);
    const size_t wi2 = (size_t)D.oW1 + (size_t)hu * D.IN + ei2;
    float4 pw2, pm2, pv2;
    if constexpr (X) { pw2 = *(const float4*)(Pc + wi2); pm2 = *(const float4*)(W.AM + wi2); pv2 = *(const float4*)(W.AV + wi2); }
    float pb = Pc[D.ob1 + hu], mb = W.AM[D.ob1 + hu], vb = W.AV[D.ob1 + hu];
    float xv0 = x1cur[(size_t)min(tid >> 4, D.K - 1) * D.H + c0 + (tid & 15)];      // post-activation of this thread's (pose row, column) output of pass 0
    {   // the features [K][IN] by LDS-DMA, straight-line (at most eight 16-byte pieces per thread: K x IN <= 16384 values; K = 20 needs one) and AFTER the other requests:
        // behind stage_issue's loop hipcc puts a full vmcnt(0) -- in front of everything requested after it
        const int n = D.K * D.IN / 4;
        constexpr int NPC = 8;                         // 16-byte pieces per thread: K * IN / 4 <= 512 NPC (round 4: 5 / 6 pieces, K <= 160; a piece past the end costs one branch)
#pragma unroll
        for (int c = 0; c < NPC; ++c) {
            const int i = c * B2_THREADS + tid;
            if (i < n)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const float4*)W.enc + i),
                                                 (__attribute__((address_space(3))) void*)((float4*)encs + c * B2_THREADS + (tid & ~63)), 16, 0, 0);
        }
    }
    __builtin_amdgcn_sched_barrier(0);                 // the loads above stay above
    // the state is requested LAST: hipcc makes `stopped` wave-uniform (v_readfirstlane) the moment it can, i.e. it waits for this load
    // right where it is issued -- at the top of the role that put a whole memory round trip in front of every other request
    GbdRecord S;
    if constexpr (FUSED) {
        const bool ok = gbd_wait(W, D.K);              // everything above is in flight or has landed; now the gradients exist
        load_a(lj, a0);
        if (two0) load_a(16 + lj, a1);
        __builtin_amdgcn_sched_barrier(0);
        S = gbd_record(W, ok);
    } else {
        const TrainState St = W.state[(epoch + 1) & 1];
        S.stopped = St.stopped; S.step_size = St.step_size; S.bc2_sqrt = St.bc2_sqrt;
    }
    const bool live = !S.stopped;
    __builtin_amdgcn_sched_barrier(0);
    BD_T                                               // B1: everything requested
    stage_wait();                                      // this wave's slab rows (and the features, and its A operands) have landed
    float wb[NS];                                      // B operands: this wave's W2 rows, column c0 + lj
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int o = V4 ? 16 * (s >> 2) + 4 * kk + (s & 3) : 4 * s + kk;
        wb[s] = w2s[(wv * KW + o) * B2_CB + lj];
    }
    for (int r0 = 0; r0 < D.K; r0 += B2_RB) {          // K <= 32: one pass
        const int nr = min(B2_RB, D.K - r0);
        const bool two = nr > 16;                      // workgroup-uniform
        float xv = xv0;
        if (r0) {
            load_a(r0 + lj, a0);
            if (two) load_a(r0 + 16 + lj, a1);
            xv = x1cur[(size_t)min(r0 + (tid >> 4), D.K - 1) * D.H + c0 + (tid & 15)];
        }
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        if (two) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[s], wb[s], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], wb[s], acc1, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int s = 0; s < NS; ++s) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[s], wb[s], acc0, 0, 0, 0);
        }
        if (r0) __syncthreads();                       // the previous pass has read red
        {   // D[4 kk + v][lj] of tile t -> red[wave][16 t + 4 kk + v][lj]
            float* o = red + (wv * B2_RB + 4 * kk) * B2_CB + lj;
#pragma unroll
            for (int v = 0; v < 4; ++v) { o[v * B2_CB] = acc0[v]; if (two) o[(16 + v) * B2_CB] = acc1[v]; }
        }
        stage_wait();                                  // (the features' LDS-DMA; every other load has been consumed)
        __syncthreads();
        if (r0 == 0) { BD_T }                          // B2: operands landed, MFMAs, partial tiles in LDS
        if (!live) return;                             // a stopped train: nothing below may be stored (workgroup-uniform)
        if (tid < nr * B2_CB) {                        // tid = (pose row of the pass) * 16 + column
            float sum = red[tid];
#pragma unroll
            for (int w2 = 1; w2 < B2_WAVES; ++w2) sum += red[w2 * B2_RB * B2_CB + tid];
            gxs[r0 * B2_CB + tid] = sum * act_grad(xv, D.slope);
        }
    }
    __syncthreads();
    BD_T                                               // B3: cross-wave reduction, activation gradient, g_x1 into LDS (+ further passes at K > 32)
    // ---- encoder rows: dW1 = g_x1^T enc, Adam, next x1
    float4 ag = make_float4(0.f, 0.f, 0.f, 0.f), ag2 = make_float4(0.f, 0.f, 0.f, 0.f);
    float gsum = 0.f;
    for (int r = 0; r < D.K; ++r) {
        const float g = gxs[r * B2_CB + row];
        const float4 e = *(const float4*)(encs + r * D.IN + ei);
        ag.x = fmaf(g, e.x, ag.x); ag.y = fmaf(g, e.y, ag.y); ag.z = fmaf(g, e.z, ag.z); ag.w = fmaf(g, e.w, ag.w);
        if constexpr (X) {
            const float4 e2 = *(const float4*)(encs + r * D.IN + ei2);
            ag2.x = fmaf(g, e2.x, ag2.x); ag2.y = fmaf(g, e2.y, ag2.y); ag2.z = fmaf(g, e2.z, ag2.z); ag2.w = fmaf(g, e2.w, ag2.w);
        }
        gsum += g;
    }
    float4 nw;
    nw.x = adam_value(pw.x, pm.x, pv.x, ag.x, S.step_size, S.bc2_sqrt);
    nw.y = adam_value(pw.y, pm.y, pv.y, ag.y, S.step_size, S.bc2_sqrt);
    nw.z = adam_value(pw.z, pm.z, pv.z, ag.z, S.step_size, S.bc2_sqrt);
    nw.w = adam_value(pw.w, pm.w, pv.w, ag.w, S.step_size, S.bc2_sqrt);
    float4 nw2 = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (X) {
        nw2.x = adam_value(pw2.x, pm2.x, pv2.x, ag2.x, S.step_size, S.bc2_sqrt);
        nw2.y = adam_value(pw2.y, pm2.y, pv2.y, ag2.y, S.step_size, S.bc2_sqrt);
        nw2.z = adam_value(pw2.z, pm2.z, pv2.z, ag2.z, S.step_size, S.bc2_sqrt);
        nw2.w = adam_value(pw2.w, pm2.w, pv2.w, ag2.w, S.step_size, S.bc2_sqrt);
    }
    pb = adam_value(pb, mb, vb, gsum, S.step_size, S.bc2_sqrt);        // every lane of the row (same operands, same result)
    if (live && half == 0) {
        if (ain) { st4_wt(Pn, wi, nw); st4_wt(W.AM, wi, pm); st4_wt(W.AV, wi, pv); }
        if constexpr (X) if (ain2) { st4_wt(Pn, wi2, nw2); st4_wt(W.AM, wi2, pm2); st4_wt(W.AV, wi2, pv2); }
        if (i4 == 0) { Pn[D.ob1 + hu] = pb; W.AM[D.ob1 + hu] = mb; W.AV[D.ob1 + hu] = vb; }
    }
    BD_T                                               // B4: dW1 over the pose rows, Adam, the encoder rows' stores issued
    for (int r = half; r < D.K; r += 2) {
        const float4 e = *(const float4*)(encs + r * D.IN + ei);
        float v = ain ? fmaf(nw.w, e.w, fmaf(nw.z, e.z, fmaf(nw.y, e.y, nw.x * e.x))) : 0.f;
        if constexpr (X) {
            const float4 e2 = *(const float4*)(encs + r * D.IN + ei2);
            if (ain2) v = fmaf(nw2.w, e2.w, fmaf(nw2.z, e2.z, fmaf(nw2.y, e2.y, fmaf(nw2.x, e2.x, v))));
        }
        v = row_sum16(v) + pb;
        if (i4 == 0) xt[r * B2_CB + row] = act_f(v, D.slope);
    }
    __syncthreads();
    if constexpr (L2IN) {
        if (live) {                             // 64-byte row segments as four 16-byte write-through stores: the D role reads them in this launch
            for (int t = tid; t < D.K * (B2_CB / 4); t += B2_THREADS) {
                const int r = t >> 2, c4 = 4 * (t & 3);
                st4_wt(x1next, r * D.H + c0 + c4, *(const float4*)(xt + r * B2_CB + c4));
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this thread's write-through stores have been acknowledged ...
            __syncthreads();                                       // ... every thread's
            if (tid == 0) __hip_atomic_fetch_add(W.sync + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else if (live)                            // the tile goes out as 64-byte row segments, 8 bytes per lane
        for (int t = tid; t < D.K * (B2_CB / 2); t += B2_THREADS) {
            const int r = t >> 3, c2 = 2 * (t & 7);
            *(nn_f2*)(x1next + (size_t)r * D.H + c0 + c2) = nn_f2{xt[r * B2_CB + c2], xt[r * B2_CB + c2 + 1]};
        }
    BD_TEND(0, epoch);                                 // B5: the next activation tile, its stores and the encoder rows' stores acknowledged
}


// ------------------------------------------------------------------------------------------ dW + Adam of the hidden / output rows
// Round 4: the weight gradient of a block of rows, dW[u][i] = sum_r g[r][u] act[r][i], is a [inputs x K] . [K x units] GEMM and runs
// on the matrix cores (v_mfma_f32_16x16x4_f32: exact float32, a k-ordered fmaf chain over the pose rows -- the same order as round 3's
// per-row VALU chain).  A workgroup owns 16 units (parameter rows), wave w their inputs [64 w, 64 w + 64) as four 16 x 16 tiles with
// M = inputs, N = units, contraction over the pose rows r: lane l holds D[i = 4 (l >> 4) + v][j = l & 15], gradients of unit
// u0 + (l & 15) -- the very bytes of the row it loads, updates (Adam, torch's arithmetic) and stores (write-through) as dwordx4 (how
// tiles map to inputs: in the body).  Operands straight from global memory: A = act[r = 4 s + (l >> 4)][..] (the K-row
// activation matrix, 40 KB per problem, L2-resident after the first touch), B[k][j] = g[r][u0 + (l & 15)] (rows K .. KP-1 are zero:
// the contraction is padded to a multiple of 20 rows).  Round 3 staged the whole activation matrix in LDS per workgroup (40 KB LDS-DMA, a wait
// for ALL loads, a barrier) and then ran 20 dependent steps of 2 ds_read_b128 + 8 FMA per wave; now there is no LDS, no barrier, and
// a wave's tiles are consumed in the order their loads were issued, so the stores of its first tiles overlap the loads of its last.
// Blocks of a problem: H2 / 16 hidden blocks, then one for the output rows of decoder_1 (width HA, 'q' only) and one for those of
// decoder_2 / the single decoder (width HB): the same code with fewer than 16 live units.
constexpr int DW_UNITS = 16;          // parameter rows per block (one MFMA N-tile)
constexpr int DW_TILES = 4;           // 16-input tiles per wave: 8 waves x 64 inputs = 512
constexpr int DW_KC = 5;              // k-steps (4 pose rows each) per chunk of operand loads: K = 20 is one chunk
__host__ __device__ inline int dw_blocks(const Dims& D) { return D.H2 / DW_UNITS + 2; }

template <bool FUSED = false, bool L2IN = false>
__device__ __forceinline__ void dw_role(const Dims& D, const Ws& W, int epoch, int blk, unsigned long long bd_entry, float* sh = nullptr) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, lj = lane & 15, q = lane >> 4;
    __builtin_amdgcn_s_waitcnt(0x0F70);             // vmcnt(0): see bwd2_role
    const int par = epoch & 1;
    const float* Pc = par ? W.P1 : W.P;             // this epoch's parameters; the updated rows go to the other buffer
    float* Pn = par ? W.P : W.P1;
    // (two selects of values, kept apart: hipcc folds a nested select over the four neighbouring members of the shifted
    //  struct into ONE load with a computed offset -- which puts the whole struct, 40 pointers, into scratch memory)
    const float* x1cur = par ? W.x1[1] : W.x1[0];
    const float* h2cur = par ? W.h2[1] : W.h2[0];
    asm volatile("" : "+s"(x1cur), "+s"(h2cur));
    // block -> (units, width, parameter offsets, activation matrix, gradient matrix); everything block-uniform
    const int nb2 = D.H2 / DW_UNITS;
    int u0, nu, n_in, oW, ob, astride, gstride;
    const float *amat, *gmat;
    if (blk < nb2) { u0 = blk * DW_UNITS; nu = DW_UNITS; n_in = D.H; oW = D.oW2 + u0 * D.H; ob = D.ob2 + u0; amat = x1cur; astride = D.H; gmat = W.g_h2 + u0; gstride = D.H2; }
    else if (blk == nb2) { u0 = 0; nu = D.OA; n_in = D.HA; oW = D.oW3A; ob = D.ob3A; amat = h2cur; astride = D.H2; gmat = W.g_out; gstride = 16; }
    else { u0 = 0; nu = D.OB; n_in = D.HB; oW = D.oW3B; ob = D.ob3B; amat = h2cur + D.HA; astride = D.H2; gmat = W.g_out + 4; gstride = 16; }
    if (nu == 0) return;                            // no decoder_1 ('dq'): block-uniform
    if (64 * wv >= n_in) {                          // a wave past the row's width (wave-uniform)
        if constexpr (FUSED) (void)gbd_wait(W, D.K);        // (it still owes the workgroup's barrier inside the wait)
        return;
    }
    BD_T0
    const int unit = min(lj, nu - 1);               // lanes past the live units mirror the last one and store nothing
    const bool ulive = lj < nu;
    // Every load is a raw buffer load (base in SGPRs, one byte offset per lane and matrix, pose rows as scalar offsets): with 64-bit
    // global addresses the loads of a lane needed so many address pairs that hipcc split them into two batches with a full wait in
    // between.  And every load but the gradients' is 16 bytes wide: a CU's address path takes ~16 cycles per wave-instruction
    // whatever its width (first build: 20 dword loads of the activations per lane -- the workgroup spent 2.2 us ISSUING its loads).
    // Hence the tiles are INTERLEAVED: a lane's A operands of one k-step are ONE float4 of the activation row, component t going to
    // tile t.  Which float4: the accumulator register v of tile t in lane (q, j) is M index 4 q + v of that tile, and it has to be
    // the gradient of input i0 + 16 v + 4 q + t of unit j -- then float4 v = (acc[0][v] .. acc[3][v]) of the lane sits at inputs
    // i0 + 16 v + 4 q .. + 3, and the four lanes q of a unit cover 64 CONTIGUOUS bytes per load / store instruction.  (A lane owning 64
    // consecutive bytes instead -- float4 v at i0 + 16 q + 4 v -- makes every instruction touch 64 different 64-byte segments, 16 bytes
    // of each: the write-through stores then took 7 us instead of 3.6.)  So M index a = 4 q + v loads the float4 at i0 + 16 (a & 3) +
    // 4 (a >> 2): the 16 lanes of a k index read the wave's 256 bytes of the row in a permuted order.
    const int i0 = 64 * wv;                         // this wave's first input
    const int po = (oW + unit * n_in + i0 + 4 * q) * 4;              // byte offset of this lane's float4 0 in the parameter arrays; float4 v is 64 v bytes on
    const __amdgpu_buffer_rsrc_t rP = buf_rsrc(Pc, D.NPAR * 4), rM = buf_rsrc(W.AM, D.NPAR * 4), rV = buf_rsrc(W.AV, D.NPAR * 4);
    // Request order = order of use, because requests return in order: (1) the MFMA operands of the first 20 pose rows, (2) the
    // state, (3) the biases, (4) the parameter / moment float4s v = 0 .. 3.  The MFMAs then run as soon as the few operand bytes
    // are in, and float4 v is updated and STORED while the later ones are still arriving -- the workgroup's write-through stores
    // (48 KB) overlap its loads (first build: parameters first, operands last, every store behind the last load: 2.0 us of loads,
    // then 4.7 us of Adam + stores).
    f32x4 acc[DW_TILES];
#pragma unroll
    for (int t = 0; t < DW_TILES; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float gb = 0.f;                                 // this lane's share of the bias gradient: sum over its pose rows r = 4 s + q
    // the contraction runs over KP rows: rows K .. KP-1 of g are zero (k_prep), those of the activations finite (zero) -- no select
    // after a load (it made hipcc wait for every gradient load on the spot) and no clamp
    const __amdgpu_buffer_rsrc_t rA = buf_rsrc(amat, (D.KP * astride - (int)(amat - (blk < nb2 ? x1cur : h2cur))) * 4);
    const __amdgpu_buffer_rsrc_t rG = buf_rsrc(gmat, (D.KP * gstride - (blk < nb2 ? u0 : (blk == nb2 ? 0 : 4))) * 4);
    const int aoff = (q * astride + i0 + 16 * (lj & 3) + 4 * (lj >> 2)) * 4, goff = (q * gstride + unit) * 4;
    float bop[DW_KC];
    float4 aop[DW_KC];
    auto load_b = [&](int s0) {                     // the gradients (FUSED: written by the gradient role of THIS launch -- sc1 loads)
#pragma unroll
        for (int s = 0; s < DW_KC; ++s)
            bop[s] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rG, goff, 4 * (s0 + s) * gstride * 4, FUSED ? 16 : 0));
    };
    auto load_a = [&](int s0) {                     // the activations
#pragma unroll
        for (int s = 0; s < DW_KC; ++s) aop[s] = ld_buf4(rA, aoff, 4 * (s0 + s) * astride * 4);   // (inputs past the row's width read the next row or 0: their gradients are not used)
    };
    auto load_ops = [&](int s0) {
#pragma unroll
        for (int s = 0; s < DW_KC; ++s) {
            const int r4 = 4 * (s0 + s);             // scalar: rows r4 + q
            bop[s] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rG, goff, r4 * gstride * 4, FUSED ? 16 : 0));
            aop[s] = ld_buf4(rA, aoff, r4 * astride * 4);
        }
    };
    auto mfma_ops = [&]() {
#pragma unroll
        for (int s = 0; s < DW_KC; ++s) {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(aop[s].x, bop[s], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(aop[s].y, bop[s], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(aop[s].z, bop[s], acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(aop[s].w, bop[s], acc[3], 0, 0, 0);
        }
#pragma unroll
        for (int s = 0; s < DW_KC; ++s) gb += bop[s];
    };
    GbdRecord S;
    if constexpr (FUSED) load_a(0); else load_ops(0);
    __builtin_amdgcn_sched_barrier(0);              // (the scheduler otherwise interleaves requests with the MFMAs and their waits)
    if constexpr (!FUSED) {
        const TrainState St = W.state[(epoch + 1) & 1];
        S.stopped = St.stopped; S.step_size = St.step_size; S.bc2_sqrt = St.bc2_sqrt; S.step = St.step;
    }
    float pb = Pc[ob + unit], mb = W.AM[ob + unit], vb = W.AV[ob + unit];
    __builtin_amdgcn_sched_barrier(0);
    float4 pw[DW_TILES], pm[DW_TILES], pv[DW_TILES];
    bool vl[DW_TILES];
#pragma unroll
    for (int v = 0; v < DW_TILES; ++v) {
        vl[v] = ulive && i0 + 16 * v + 4 * q < n_in;                 // (HA = 32 at hidden 64: half a wave's inputs exist)
        pw[v] = ld_buf4(rP, po + 64 * v, 0); pm[v] = ld_buf4(rM, po + 64 * v, 0); pv[v] = ld_buf4(rV, po + 64 * v, 0);      // (past the arrays: 0)
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (FUSED) {
        const bool ok = gbd_wait(W, D.K);           // the parameter rows and the activations are here; now the gradients exist
        load_b(0);
        __builtin_amdgcn_sched_barrier(0);
        S = gbd_record(W, ok);
        __builtin_amdgcn_sched_barrier(0);
    }
    BD_T                                            // D1: everything requested
    mfma_ops();
    for (int s0 = DW_KC; 4 * s0 < D.KP; s0 += DW_KC) {               // K > 20: further chunks of 20 pose rows (a round trip each)
        load_ops(s0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_ops();
    }
    BD_T                                            // D2: operands landed, MFMAs
    if (S.stopped) return;                          // a stopped train: nothing below may be stored (block-uniform)
#pragma unroll
    for (int v = 0; v < DW_TILES; ++v) {
        float4 nw;
        nw.x = adam_value(pw[v].x, pm[v].x, pv[v].x, acc[0][v], S.step_size, S.bc2_sqrt);
        nw.y = adam_value(pw[v].y, pm[v].y, pv[v].y, acc[1][v], S.step_size, S.bc2_sqrt);
        nw.z = adam_value(pw[v].z, pm[v].z, pv[v].z, acc[2][v], S.step_size, S.bc2_sqrt);
        nw.w = adam_value(pw[v].w, pm[v].w, pv[v].w, acc[3][v], S.step_size, S.bc2_sqrt);
        if (vl[v]) { const int e = po / 4 + 16 * v; st4_wt(Pn, e, nw); st4_wt(W.AM, e, pm[v]); st4_wt(W.AV, e, pv[v]); }
        if constexpr (L2IN) pw[v] = nw;             // the updated rows stay in registers: the B operands of the next hidden activation below
    }
    if (wv == 0) {                                  // the units' biases: gradient = sum over the pose rows of g[r][u], the four lane groups q in order
        float sum = gb;
        sum += __shfl_xor(gb, 16, 64);              // (q, q ^ 1)
        sum += __shfl_xor(sum, 32, 64);             // ((0,1), (2,3))
        pb = adam_value(pb, mb, vb, sum, S.step_size, S.bc2_sqrt);
        if (q == 0 && ulive) { Pn[ob + unit] = pb; W.AM[ob + unit] = mb; W.AV[ob + unit] = vb; }
    }
    if constexpr (L2IN) {
        // ---- the NEXT hidden activation of this block's 16 units, h2' = act(x1' . W2'^T + b2'), inside the backward launch (round 6): what k_l2
        // computed a launch later -- from the updated rows this wave still holds (the very B operands k_l2 loaded back: lane (q, unit) holds
        // inputs i0 + 16 v + 4 q + t as component t of float4 v, k_l2's k order) and the next encoder activation x1' of the B role of THIS launch
        // (write-through stores there, sc1 loads here, one counter in between).  Same tiles, same k order, same cross-wave sum in wave order:
        // bit for bit k_l2's result.  Only the hidden blocks take part (the output rows' blocks have left), all eight waves of them (H = 512).
        if (blk < nb2) {                            // block-uniform
            float* red = sh;                        // [8 waves][32 rows][16 units] partial tiles, then the 16 new biases
            float* sbias = red + (BD_THREADS / 64) * L2_RB * 16;
            if (wv == 0 && q == 0) sbias[lj] = pb;
            const int nB = D.H / B2_CB;
            if (threadIdx.x == 0) {                 // the B role's workgroups of this problem have counted themselves in: nB per optimizer step
                int it = 0;
                const int want = nB * S.step;
                while (__hip_atomic_load(W.sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want && ++it < (1 << 20)) __builtin_amdgcn_s_sleep(2);
                if (it >= (1 << 20)) __hip_atomic_store(W.sync + 3, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (cannot happen while workgroups are dispatched in index order: see gbd_wait)
            }
            __syncthreads();
            const float* x1n = par ? W.x1[0] : W.x1[1];
            float* h2n = par ? W.h2[0] : W.h2[1];
            const __amdgpu_buffer_rsrc_t rX = buf_rsrc(x1n, D.KP * D.H * 4);
            const int tid = threadIdx.x;
            for (int r0 = 0; r0 < D.K; r0 += L2_RB) {
                const int nr = min(L2_RB, D.K - r0);
                const bool two = nr > 16;           // block-uniform
                u32x4v xa[DW_TILES], xb[DW_TILES];
                const int o0 = (min(r0 + lj, D.K - 1) * D.H + i0 + 4 * q) * 4, o1 = (min(r0 + 16 + lj, D.K - 1) * D.H + i0 + 4 * q) * 4;
#pragma unroll
                for (int v = 0; v < DW_TILES; ++v) {
                    xa[v] = __builtin_amdgcn_raw_buffer_load_b128(rX, o0 + 64 * v, 0, 16);      // sc1: written through by the B role of this launch
                    if (two) xb[v] = __builtin_amdgcn_raw_buffer_load_b128(rX, o1 + 64 * v, 0, 16);
                }
                f32x4 c0v = {0.f, 0.f, 0.f, 0.f}, c1v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int v = 0; v < DW_TILES; ++v) {
                    const float bq[4] = {pw[v].x, pw[v].y, pw[v].z, pw[v].w};
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        c0v = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(xa[v][t]), bq[t], c0v, 0, 0, 0);
                        if (two) c1v = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(xb[v][t]), bq[t], c1v, 0, 0, 0);
                    }
                }
                if (r0) __syncthreads();            // the previous pass has read red
                {
                    float* o = red + (wv * L2_RB + 4 * q) * 16 + lj;
#pragma unroll
                    for (int v = 0; v < 4; ++v) { o[v * 16] = c0v[v]; if (two) o[(16 + v) * 16] = c1v[v]; }
                }
                __syncthreads();
                if (tid < nr * 16) {
                    float sum = red[tid];
#pragma unroll
                    for (int w2 = 1; w2 < BD_THREADS / 64; ++w2) sum += red[w2 * L2_RB * 16 + tid];
                    h2n[(size_t)(r0 + (tid >> 4)) * D.H2 + u0 + (tid & 15)] = act_f(sum + sbias[tid & 15], D.slope);
                }
            }
        }
    }
    BD_TEND(1, epoch);                              // D3: Adam, the write-through stores acknowledged
}

// ------------------------------------------------------------------------------------------ the backward launch: k_bd
// The backward to the encoder (B, H / 16 blocks per problem) and the weight gradients + Adam of the hidden / output rows (D,
// H2 / 8 + 1 blocks per problem) both need only what k_gradc leaves (g_out, g_h2, the advanced state) and this epoch's
// parameters, which nobody overwrites (the updates go to the other parameter buffer): two INDEPENDENT roles, so they are one
// launch of 512-thread workgroups told apart by their block index -- nothing is handed over inside it.  Round 3 first ran them
// as k_bwd2 (11.7 us) -> k_dw (14.9 us, which also computed the next hidden activation from the rows in its registers); side by
// side the launch takes what the longer role takes, and the next hidden activation, the one thing that needs BOTH results (the
// next encoder activation from B, the updated hidden rows from D), is k_l2 again, a launch boundary later.
// grid.x = (H / 16 + H2 / 8 + 1) * problems: the B blocks of ALL problems first (the longer chain of dependent phases).
template <int NC, int KW, bool X = false, bool L2IN = false>
__global__ __launch_bounds__(BD_THREADS, 4) void k_bd(Dims D, Ws W0, int epoch, size_t bstride, int nz) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef CREG_BD_STAMPS
    const unsigned long long bd_entry = wall_clock64();
#else
    const unsigned long long bd_entry = 0;
#endif
#ifndef CREG_BD_NO_TOUCH
    // Every kernel argument the roles read, fetched in the ENTRY block with one wait: hipcc otherwise loads arguments where a basic
    // block first needs them -- three dependent scalar round trips to the argument buffer (cold for every node of a replayed graph)
    // before the first request of a role went out.
    asm volatile("" :: "s"(W0.P), "s"(W0.P1), "s"(W0.AM), "s"(W0.AV), "s"(W0.enc), "s"(W0.x1[0]), "s"(W0.x1[1]), "s"(W0.h2[0]), "s"(W0.h2[1]),
                 "s"(W0.g_out), "s"(W0.g_h2), "s"(W0.state), "s"(D.K), "s"(D.KP), "s"(D.IN), "s"(D.H), "s"(D.H2), "s"(D.HA), "s"(D.HB), "s"(D.OA),
                 "s"(D.OB), "s"(D.oW1), "s"(D.ob1), "s"(D.oW2), "s"(D.ob2), "s"(D.oW3A), "s"(D.ob3A), "s"(D.oW3B), "s"(D.ob3B), "s"(D.NPAR),
                 "s"(D.slope), "s"(epoch), "s"(bstride), "s"(nz));
#endif
    const int nB = D.H / B2_CB, nD = dw_blocks(D);
    int i = blockIdx.x;
    const bool roleB = i < nB * nz;
    if (!roleB) i -= nB * nz;
    const int n = roleB ? nB : nD, z = i / n, blk = i - z * n;
    const Ws W = ws_shift(W0, (size_t)z * bstride);
#ifdef CREG_BD_ONLY                                   // measurement build: one role alone (1: backward to the encoder, 2: dW + Adam)
    if ((CREG_BD_ONLY == 1) != roleB) return;
#endif
    if (roleB) bwd2_role<KW, X, false, L2IN>(D, W, epoch, blk, (float*)smem, bd_entry);
    else dw_role<false, L2IN>(D, W, epoch, blk, bd_entry, (float*)smem);
}

// ------------------------------------------------------------------------------------------ the fused backward launch: k_gbd (round 5)
// k_gradc and k_bd as ONE launch: grid.x = (K + H / 16 + H2 / 16 + 2) * problems, the gradient role's blocks first.  What the
// boundary between the two launches serialised is not the arithmetic but the INTAKE: a B block pulls 110 KB and a D block 96 KB
// through a CU that takes ~11 bytes per cycle, and all but the few KB of gradients of it -- parameter rows, Adam moments, W2 slab,
// activations -- is known before k_gradc has started.  Here the consumers request all of that at once, then wait for the K gradient
// blocks (gbd_wait), then fetch the gradients.  One hand-off (gradient role -> {B, D} side by side), write-through stores on one side,
// sc1 loads on the other, no fence; bit-identical to the two launches (same arithmetic in the same order).
template <int NC, int KW, bool X = false>
__global__ __launch_bounds__(BD_THREADS, 4) void k_gbd(Dims D, Ws W0, int epoch, int nbx, int nby, size_t bstride, int nz) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    asm volatile("" :: "s"(W0.P), "s"(W0.P1), "s"(W0.AM), "s"(W0.AV), "s"(W0.enc), "s"(W0.x1[0]), "s"(W0.x1[1]), "s"(W0.h2[0]), "s"(W0.h2[1]),
                 "s"(W0.g_out), "s"(W0.g_h2), "s"(W0.state), "s"(W0.sync), "s"(D.K), "s"(D.KP), "s"(D.IN), "s"(D.H), "s"(D.H2), "s"(D.HA), "s"(D.HB), "s"(D.OA),
                 "s"(D.OB), "s"(D.oW1), "s"(D.ob1), "s"(D.oW2), "s"(D.ob2), "s"(D.oW3A), "s"(D.ob3A), "s"(D.oW3B), "s"(D.ob3B), "s"(D.NPAR),
                 "s"(D.slope), "s"(epoch), "s"(bstride), "s"(nz));
    asm volatile("" :: "s"(W0.head_save), "s"(W0.m2), "s"(W0.gm2), "s"(W0.pts4), "s"(W0.pred4), "s"(W0.sgn_x), "s"(W0.cnt4), "s"(W0.lossp_x), "s"(W0.lossp_y),
                 "s"(W0.bc1), "s"(W0.bc2s), "s"(W0.best_m), "s"(W0.best_pred), "s"(W0.loss_hist), "s"(W0.lr_hist), "s"(W0.result), "s"(W0.off), "s"(W0.hyper),
                 "s"(D.rot), "s"(D.NP), "s"(D.NT), "s"(D.epochs), "s"(nbx), "s"(nby));
    const int nG = D.K, nB = D.H / B2_CB, nD = dw_blocks(D);
    int i = blockIdx.x;
    if (i < nG * nz) {
        if (threadIdx.x >= 256) return;             // the gradient role is a 256-thread role
        const int z = i / nG;
        const Ws W = ws_shift(W0, (size_t)z * bstride);
        gradc_role<true>(D, W, epoch, nbx, nby, i - z * nG);
        return;
    }
    i -= nG * nz;
    const bool roleB = i < nB * nz;
    if (!roleB) i -= nB * nz;
    const int n = roleB ? nB : nD, z = i / n, blk = i - z * n;
    const Ws W = ws_shift(W0, (size_t)z * bstride);
    if (roleB) bwd2_role<KW, X, true>(D, W, epoch, blk, (float*)smem, 0ull);
    else dw_role<true>(D, W, epoch, blk, 0ull);
}

// after the last epoch: the parameters of an odd number of optimizer steps sit in the second buffer; the copy-out reads the first
// The parameters change buffer with every optimizer step: after a run that took an odd number of them they sit in P1.  `sidx`: the
// state the run ended in (epochs enqueued & 1), `step0`: the step count it started from (0 unless resumed from a caller's state).
__global__ __launch_bounds__(256) void k_params_home(Dims D, Ws W0, size_t bstride, int sidx, int step0) {
    const Ws W = ws_shift(W0, blockIdx.z * bstride);
    if (((W.state[sidx].step - step0) & 1) == 0) return;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < D.NPAR; i += gridDim.x * 256) W.P[i] = W.P1[i];
}

// creg_train_plan_resume: the caller's optimizer / control state over what k_prep staged (both copies of the double-buffered struct),
// and the state a run ended in as twelve doubles (the layout include/creg.h documents).
struct ResumeState { double lr, sched_best; int step, epochs_run, sched_bad, count, best_epoch, stopped; float min_loss; };
__global__ void k_set_state(Dims D, Ws W, ResumeState rs) {
    if (threadIdx.x || blockIdx.x) return;
    TrainState s = W.state[0];
    s.lr = rs.lr; s.sched_best = rs.sched_best; s.sched_bad = rs.sched_bad; s.count = rs.count; s.stopped = rs.stopped; s.step = rs.step;
    s.min_loss = rs.min_loss; s.epochs_run = rs.epochs_run; s.best_epoch = rs.best_epoch;
    s.next_bc1 = W.bc1[min(rs.step + 1, D.epochs)]; s.next_bc2s = W.bc2s[min(rs.step + 1, D.epochs)];       // (k_prep, the launch before, filled the tables)
    W.state[0] = s; W.state[1] = s;
    W.sync[1] = (D.H / B2_CB) * rs.step;          // (k_bd with the next hidden activation inside: B-role arrivals so far = nB per optimizer step)
    W.result[0] = s.min_loss; W.result[1] = (float)s.epochs_run; W.result[2] = (float)s.lr; W.result[3] = (float)s.best_epoch;
}
__global__ void k_get_state(Ws W, int sidx, double* out) {
    if (threadIdx.x || blockIdx.x) return;
    const TrainState s = W.state[sidx];
    out[0] = s.step; out[1] = s.epochs_run; out[2] = s.lr; out[3] = s.sched_best; out[4] = s.sched_bad; out[5] = s.count;
    out[6] = (double)s.min_loss; out[7] = s.best_epoch; out[8] = s.stopped; out[9] = (double)s.last_loss; out[10] = 0.0; out[11] = 0.0;
}

// ------------------------------------------------------------------------------------------ host side
struct Plan {
    creg_train_shape shape;
    Dims D;
    Ws W;
    char* base;
    size_t bytes;
    hipGraphExec_t gexec;     // two epochs (parity 0 then 1)
    bool graph_ready;
    int graph_epochs;
    int B;                    // problems the workspace holds
    int nz;                   // problems per launch right now (grid.z): B for run, 1 for probe / profile
    size_t bstride;           // bytes between consecutive problems' workspaces
    int smem_bd;
    bool fused;               // gradient reduction + backward as ONE launch (k_gbd)
    bool l2in;                // the next hidden activation inside the backward launch (k_bd<.., true>): no k_l2 launch per epoch
    int branches;             // parallel chains in the captured graph (groups of problems)
    // chain-stream mode (creg_train_shape.graph_branches < 0): every chain is its OWN linear graph on its OWN stream, forked from /
    // joined to the caller's stream once per train by events -- the chains' hardware queues are then the streams', not what the
    // runtime picks for the branches of one graph at every launch
    bool chain_streams;
    hipStream_t cst[8];
    hipGraphExec_t cexec[8];
    hipEvent_t cfork, cjoin[8];
    hipStream_t cal_stream;   // the caller's stream the chain streams were picked against (pick_chain_streams)
    bool cal_done;
    int probe_us;             // pick_chain_streams: the slowest ACCEPTED joint spin of the chain streams with the caller's (150 us kernels: < 250 = concurrent); -1: a chain had to
                              //   take a stream that shares a hardware queue; 0: not probed (one chain, or the probe is switched off)
    bool target_blocks_valid; // ys4 / ybox hold the k-d leaves of every problem's last run_batch frame (probe / profile overwrite them)
};

static bool make_dims(const creg_train_shape* s, Dims* D) {
    if (!s || s->rot < 0 || s->rot > 3 || s->k < 1 || s->k > 256 || (s->hidden != 64 && s->hidden != 128 && s->hidden != 256 && s->hidden != 512) || s->epochs < 1 || s->n_pred < 1 || s->n_tgt < 1 || s->n_pred >= (1ll << 31) ||
        s->n_tgt >= (1ll << 31))
        return false;
    memset(D, 0, sizeof(*D));
    D->rot = s->rot; D->K = s->k; D->KP = (s->k + 19) / 20 * 20; D->H = s->hidden; D->NP = (int)s->n_pred; D->NT = (int)s->n_tgt; D->epochs = s->epochs;
    // model_utils.py: QRegMLP :103-168 (7 inputs x 8 sin / cos features), DQRegMLP :65-101 (ReLU, one decoder), RRegMLP :170-214, RegMLP :216-281
    if (s->rot == ROT_Q) { D->IN = 56; D->HA = D->H / 2; D->HB = D->H; D->OA = 3; D->OB = 4; D->slope = 0.01f; }
    else if (s->rot == ROT_DQ) { D->IN = 64; D->HA = 0; D->HB = D->H; D->OA = 0; D->OB = 8; D->slope = 0.f; }
    else if (s->rot == ROT_6D) { D->IN = 72; D->HA = D->H / 2; D->HB = D->H; D->OA = 3; D->OB = 6; D->slope = 0.01f; }
    else { D->IN = 48; D->HA = D->H / 2; D->HB = D->H; D->OA = 3; D->OB = 3; D->slope = 0.01f; }
    if (D->K * D->IN / 4 > 8 * 512) return false;      // k_bd stages the [K][IN] features with at most eight 16-byte pieces per thread ('6d', 72 features: K <= 227)
    D->H2 = D->HA + D->HB;
    if (sizeof(float) * b2_smem_floats(D->K, D->IN, D->H2) > 160 * 1024) return false;      // ... in one CU's LDS beside its slab of W2 ('6d' at hidden 512: K <= 193)
    int o = 0;
    D->oW1 = o; o += D->H * D->IN; D->ob1 = o; o += D->H;
    D->oW2 = o; o += D->H2 * D->H; D->ob2 = o; o += D->H2;
    D->oW3A = o; o += D->OA * D->HA; D->ob3A = o; o += D->OA;
    D->oW3B = o; o += D->OB * D->HB; D->ob3B = o; o += D->OB;
    D->NPAR = o;
    if (D->H2 % 32 || D->H % B2_CB || D->H2 > 256 * GC_QMAX) return false;
    const NnGrid g = nn_grid(D->NP, D->NT, true, true, 4);
    D->nbx = g.blocksA; D->nby = g.blocksB;
    // block-pruned search: 4 queries per wave, a lane holds nbt boxes of the (static) target frame / nbp of the predicted cloud
    // (clusters padded to whole blocks: up to 8 x 64 of them; round 2 stopped at 128 and sent K > 64 at N = 4096 to the
    // exhaustive kernel).  Blocks of 64 points (one per lane and visit) up to 4096 targets, of 256 points (four per lane and
    // visit) beyond: measured at the franka shape (N = 16384, K = 40), 64-point blocks with 4 + 5 boxes per lane take 67.6 us
    // per NN launch against 46 for 64 + 104 blocks of 256 -- evaluating nine boxes per query and lane costs more than the
    // finer blocks save.  The target frame is cut into k-d leaves in LDS, 16384 points per workgroup (k_sort_y; up to four
    // chunks), a cluster by one workgroup (k_sort_p): n_tgt <= 65536, n_pred < 65535 (16-bit indices in the sort keys) and
    // clusters of at most 16384 points are this form's limits; outside them the plan runs the exhaustive kernel per direction.
    // Round 5: nn_search 0 takes sixteen queries per wave (nn_l1_rows) wherever BOTH clouds fit the 64-point block layouts -- up to 256
    // target blocks (four boxes per lane: n_tgt <= 16384, one chunk of k-d leaves) and PS_MAXB blocks of the padded predicted cloud;
    // nn_search 2 = the four-queries-per-wave search of rounds 2-4 (256-point blocks above 4096 targets), nn_search 1 = exhaustive.
    const bool rows_fit = D->NT <= 16384 && D->NP < 65535 && (D->NP + 63) / 64 + D->K <= PS_MAXB && g.qw == 4;
    // CREG_NN_ROWS: 0 = never, 1 = wherever it fits, unset = where it measured faster (profiles/r05_nn_rows_ab.log)
    const char* rows_env = getenv("CREG_NN_ROWS");
    const bool rows_pays = D->NT > 64 * 64;
    D->rows = s->nn_search == 0 && rows_fit && (rows_env ? rows_env[0] == '1' : rows_pays);
    D->ppl = (D->rows || D->NT <= 64 * 64) ? 1 : 4;
    const int BS = 64 * D->ppl;
    D->nyb = (s->nn_search != 1 && D->NT <= 4 * YS_CHUNK && g.qw == 4) ? (D->NT + BS - 1) / BS : 0;      // (chunks of 16384: k_sort_y)
    D->npb = (D->nyb && D->NP < 65535 && (D->NP + BS - 1) / BS + D->K <= PS_MAXB) ? (D->NP + BS - 1) / BS + D->K : 0;
    const int nt = (D->nyb + 63) / 64, np = (D->npb + 63) / 64;
    D->nbt = nt <= 1 ? 1 : (nt <= 2 ? 2 : 4);
    D->nbp = np <= 2 ? 2 : (np <= 3 ? 3 : (np <= 4 ? 4 : (np <= 5 ? 5 : (np <= 6 ? 6 : 8))));
    if (D->rows) { D->nbx = 4 * D->npb; D->nby = 4 * D->nyb; }          // loss partials per 16-slot group of the query clouds
    return true;
}

static size_t carve(const Dims& D, char* base, Ws* W) {
    size_t o = 0;
    auto take = [&](size_t bytes) -> char* { char* r = base ? base + o : nullptr; o = align_up(o + bytes, 256); return r; };
    const size_t f = sizeof(float);
    Ws w;
    w.P = (float*)take(f * D.NPAR); w.P1 = (float*)take(f * D.NPAR); w.AM = (float*)take(f * D.NPAR); w.AV = (float*)take(f * D.NPAR);
    w.pose_in = (float*)take(f * POSE_IN * D.K); w.enc = (float*)take(f * D.K * D.IN);
    w.x1[0] = (float*)take(f * D.KP * D.H); w.x1[1] = (float*)take(f * D.KP * D.H);
    w.h2[0] = (float*)take(f * D.KP * D.H2); w.h2[1] = (float*)take(f * D.KP * D.H2); w.head_save = (float*)take(f * 16 * D.K);
    w.m2 = (float*)take(f * 16 * D.K); w.gm2 = (float*)take(f * 16 * D.K);
    w.pts4 = (float4*)take(sizeof(float4) * D.NP); w.y4 = (float4*)take(sizeof(float4) * D.NT);
    w.pred4 = (float4*)take(sizeof(float4) * D.NP);
    const size_t bs = 64 * (size_t)D.ppl;
    w.ys4 = (float4*)take(sizeof(float4) * bs * (D.nyb ? D.nyb : 1)); w.ybox = (float*)take(f * 6 * 64 * D.nbt);
    w.psl4 = (float4*)take(sizeof(float4) * bs * (D.npb ? D.npb : 1)); w.ps4 = (float4*)take(sizeof(float4) * bs * (D.npb ? D.npb : 1));
    w.pbox = (float*)take(f * 6 * 64 * D.nbp); w.sb = (int*)take(sizeof(int) * (D.K + 1));
    w.sgn_x = (int*)take(sizeof(int) * D.NP);
    w.cnt4 = (int4*)take(sizeof(int4) * D.NP);
    w.lossp_x = (float*)take(f * D.nbx); w.lossp_y = (float*)take(f * D.nby);
    w.g_out = (float*)take(f * 16 * D.KP); w.g_h2 = (float*)take(f * D.KP * D.H2);
    w.state = (TrainState*)take(sizeof(TrainState) * 2);
    w.bc1 = (double*)take(sizeof(double) * (D.epochs + 1)); w.bc2s = (float*)take(f * (D.epochs + 1));
    w.best_m = (float*)take(f * 16 * D.K); w.best_pred = (float*)take(f * 3 * D.NP);
    w.loss_hist = (float*)take(f * D.epochs); w.lr_hist = (float*)take(f * D.epochs); w.result = (float*)take(f * 4);
    w.off = (int*)take(sizeof(int) * (D.K + 1)); w.hyper = (Hyper*)take(sizeof(Hyper));
    w.sync = (int*)take(sizeof(int) * 64);
    w.ysum = (unsigned long long*)take(sizeof(unsigned long long) * 4);
    if (W) *W = w;
    return o;
}

template <typename F>
static void by_nc(int H, F f) {            // H in {64, 128, 256, 512}
    switch (H / 64) {
        case 1: f(std::integral_constant<int, 1>{}); break;
        case 2: f(std::integral_constant<int, 2>{}); break;
        case 4: f(std::integral_constant<int, 4>{}); break;
        default: f(std::integral_constant<int, 8>{}); break;
    }
}
static void launch_l2(Plan* P, int par, hipStream_t s) {
    const Dims& D = P->D; const Ws& W = P->W;
    by_nc(D.H, [&](auto nc) {
        hipLaunchKernelGGL((k_l2<decltype(nc)::value>), dim3(D.H2 / 16, 1, P->nz), dim3(L2_THREADS), 0, s, D, W, par, P->bstride); });
}
static void launch_head(Plan* P, int par, hipStream_t s) {
    const Dims& D = P->D; const Ws& W = P->W;
    by_nc(D.H, [&](auto nc) {
        if (D.OA + D.OB > 8) hipLaunchKernelGGL((k_head<decltype(nc)::value, true>), dim3(D.K, 1, P->nz), dim3(512), 0, s, D, W, par, P->bstride);
        else hipLaunchKernelGGL((k_head<decltype(nc)::value>), dim3(D.K, 1, P->nz), dim3(512), 0, s, D, W, par, P->bstride);
    });
}
// the k_bd instance of a shape: NC = H / 64; the B role's 8 waves own KW = H2 / 8 rows of W2 each ('q': H2 = 96 NC, 'dq': 64 NC)
template <typename F>
static void by_bd(const Dims& D, F f) {
    by_nc(D.H, [&](auto nc) {
        constexpr int NC = decltype(nc)::value;
        if (D.IN > 64) f(k_bd<NC, 12 * NC, true>);             // '6d': 72 input features
        else if (D.HA) f(k_bd<NC, 12 * NC>);                   // two decoders ('q', 'rpy'): H2 = H / 2 + H
        else f(k_bd<NC, 8 * NC>);
    });
}
static void launch_bd(Plan* P, int epoch, hipStream_t s) {
    const Dims& D = P->D; const Ws& W = P->W;
    const int per = D.H / B2_CB + dw_blocks(D);
    if (P->l2in) {                                   // (hidden 512, 'q' / 'rpy' / 'dq' widths: see creg_train_plan_create)
        if (D.HA) hipLaunchKernelGGL((k_bd<8, 96, false, true>), dim3(per * P->nz), dim3(BD_THREADS), P->smem_bd, s, D, W, epoch, P->bstride, P->nz);
        else hipLaunchKernelGGL((k_bd<8, 64, false, true>), dim3(per * P->nz), dim3(BD_THREADS), P->smem_bd, s, D, W, epoch, P->bstride, P->nz);
        return;
    }
    by_bd(D, [&](auto kern) { hipLaunchKernelGGL(kern, dim3(per * P->nz), dim3(BD_THREADS), P->smem_bd, s, D, W, epoch, P->bstride, P->nz); });
}
template <typename F>
static void by_gbd(const Dims& D, F f) {
    by_nc(D.H, [&](auto nc) {
        constexpr int NC = decltype(nc)::value;
        if (D.IN > 64) f(k_gbd<NC, 12 * NC, true>);
        else if (D.HA) f(k_gbd<NC, 12 * NC>);
        else f(k_gbd<NC, 8 * NC>);
    });
}
static void launch_gbd(Plan* P, int epoch, hipStream_t s) {
    const Dims& D = P->D; const Ws& W = P->W;
    const int per = D.K + D.H / B2_CB + dw_blocks(D);
    by_gbd(D, [&](auto kern) { hipLaunchKernelGGL(kern, dim3(per * P->nz), dim3(BD_THREADS), P->smem_bd, s, D, W, epoch, D.nbx, D.nby, P->bstride, P->nz); });
}
static void launch_nn(const Dims& D, const Ws& W, size_t bstride, int nz, hipStream_t s, int par = 0) {
    const EngineEpi epi{W.sgn_x, W.cnt4, W.lossp_x, W.lossp_y, bstride, &W.state[par].stopped};
    if (D.rows) {
        const NnBlocks yb{W.ys4, W.ybox, D.nyb, nullptr}, pb{W.ps4, W.pbox, D.npb, W.sb + D.K};      // (pb.nblk: the host's bound, what ps4 / lossp_x are carved for)
        const int blocksA = cdiv(64 * D.npb, NN_ROW_SLOTS), blocksB = cdiv(64 * D.nyb, NN_ROW_SLOTS);
        const dim3 grid((blocksA + blocksB) * nz);
        auto go = [&](auto kern) {
            hipLaunchKernelGGL(kern, grid, dim3(NN_ROWS_BLOCK), 0, s, (const float*)W.pred4, (const float*)W.y4, blocksA, blocksB, epi, yb, pb, bstride);
        };
        auto pick_p = [&](auto nbt) {
            constexpr int NBT = decltype(nbt)::value;
            switch (D.nbp) {
                case 2: go(k_nn_rows<NBT, 2>); break;
                case 3: go(k_nn_rows<NBT, 3>); break;
                case 4: go(k_nn_rows<NBT, 4>); break;
                case 5: go(k_nn_rows<NBT, 5>); break;
                case 6: go(k_nn_rows<NBT, 6>); break;
                default: go(k_nn_rows<NBT, 8>); break;
            }
        };
        switch (D.nbt) {
            case 1: pick_p(std::integral_constant<int, 1>{}); break;
            case 2: pick_p(std::integral_constant<int, 2>{}); break;
            default: pick_p(std::integral_constant<int, 4>{}); break;
        }
    } else if (D.nyb) {
        const NnGrid g = nn_grid(D.NP, D.NT, true, true, 4);
        const NnBlocks yb{W.ys4, W.ybox, D.nyb, nullptr}, pb{W.ps4, W.pbox, 0, W.sb + D.K};
        const dim3 grid((g.blocksA + g.blocksB) * nz);
        auto go = [&](auto kern, int smem) {
            hipLaunchKernelGGL(kern, grid, dim3(NN_BLOCK), smem, s, (const float*)W.pred4, D.NP, (const float*)W.y4, D.NT,
                               g.blocksA, g.blocksB, epi, yb, pb, bstride);
        };
        // the instance of this shape: boxes per lane are template arguments (register arrays)
        auto pick_p = [&](auto ppl, auto nbt) {
            constexpr int PPL = decltype(ppl)::value, NBT = decltype(nbt)::value;
            if (!D.npb) { go(k_nn_plan<false, PPL, NBT, 2>, g.smem); return; }
            switch (D.nbp) {
                case 2: go(k_nn_plan<true, PPL, NBT, 2>, 0); break;
                case 3: go(k_nn_plan<true, PPL, NBT, 3>, 0); break;
                case 4: go(k_nn_plan<true, PPL, NBT, 4>, 0); break;
                case 5: go(k_nn_plan<true, PPL, NBT, 5>, 0); break;
                case 6: go(k_nn_plan<true, PPL, NBT, 6>, 0); break;
                default: go(k_nn_plan<true, PPL, NBT, 8>, 0); break;
            }
        };
        auto pick_t = [&](auto ppl) {
            switch (D.nbt) {
                case 1: pick_p(ppl, std::integral_constant<int, 1>{}); break;
                case 2: pick_p(ppl, std::integral_constant<int, 2>{}); break;
                default: pick_p(ppl, std::integral_constant<int, 4>{}); break;
            }
        };
        if (D.ppl == 1) pick_t(std::integral_constant<int, 1>{}); else pick_t(std::integral_constant<int, 4>{});
    } else {
        launch_nn_l1<int>((const float*)W.pred4, D.NP, 4, (const float*)W.y4, D.NT, 4, nullptr, nullptr, nullptr, nullptr,
                          true, true, epi, s, nz, bstride, 4);
    }
}
constexpr int NKERN = 5;
// `ev` (optional): NKERN + 1 events recorded before kernel 0 and after each kernel.
// Entering epoch e, x1[e & 1] and h2[e & 1] hold the activations of the current parameters (k_l1 / k_l2 for epoch 0,
// k_bwd2 / k_dw of the previous epoch afterwards).
static void enqueue_epoch(Plan* P, int epoch, hipStream_t s, hipEvent_t* ev = nullptr) {
    const Dims& D = P->D; const Ws& W = P->W;
    const int par = epoch & 1;
    auto mark = [&](int i) { if (ev) (void)hipEventRecord(ev[i], s); };
    mark(0);
    launch_head(P, par, s); mark(1);
    launch_nn(D, W, P->bstride, P->nz, s, par); mark(2);
    if (P->fused) { launch_gbd(P, epoch, s); mark(3); mark(4); }
    else {
        hipLaunchKernelGGL(k_gradc, dim3(D.K, 1, P->nz), dim3(256), 0, s, D, W, epoch, D.nbx, D.nby, P->bstride); mark(3);
        launch_bd(P, epoch, s); mark(4);
    }
    if (!P->l2in) launch_l2(P, par ^ 1, s);   // the next epoch's hidden activation: next encoder activation (B) x updated hidden rows (D)
    mark(5);                                  //   (l2in: computed at the end of k_bd's D role)
}

// One launch copies up to 16 (source, destination, dword count) ranges: a problem's 10 parameter tensors + offsets in,
// parameters + results out -- instead of one hipMemcpyAsync (a ~2.5 us blit kernel each) per tensor, 270 per frame round.
constexpr int COPY_MAX = 150;      // ranges per launch (3.1 KB of kernel arguments)
struct CopyTable { const unsigned* src[COPY_MAX]; unsigned* dst[COPY_MAX]; int n[COPY_MAX]; int count; };
__global__ __launch_bounds__(256) void k_copy_table(CopyTable T) {
    const unsigned* __restrict__ s = T.src[blockIdx.y]; unsigned* __restrict__ d = T.dst[blockIdx.y];
    const int n = T.n[blockIdx.y];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) d[i] = s[i];
}
static void copy_table_launch(CopyTable& T, hipStream_t s) {
    if (T.count) hipLaunchKernelGGL(k_copy_table, dim3(32, T.count), dim3(256), 0, s, T);
    T.count = 0;
}
static void copy_table_add(CopyTable& T, const void* src, void* dst, size_t bytes, hipStream_t s) {   // launches when full
    if (T.count == COPY_MAX) copy_table_launch(T, s);
    T.src[T.count] = (const unsigned*)src; T.dst[T.count] = (unsigned*)dst; T.n[T.count] = (int)(bytes / 4); ++T.count;
}

struct ParamMap { int off, count; };
static int param_map(const Dims& D, ParamMap* pm) {
    if (D.rot != ROT_DQ) {                 // QRegMLP / RRegMLP / RegMLP: encoder, decoder_1 (two layers), decoder_2 (two layers)
        const ParamMap m[10] = {{D.oW1, D.H * D.IN}, {D.ob1, D.H}, {D.oW2, D.HA * D.H}, {D.ob2, D.HA},
                                {D.oW3A, D.OA * D.HA}, {D.ob3A, D.OA}, {D.oW2 + D.HA * D.H, D.HB * D.H},
                                {D.ob2 + D.HA, D.HB}, {D.oW3B, D.OB * D.HB}, {D.ob3B, D.OB}};
        memcpy(pm, m, sizeof(m));
        return 10;
    }
    const ParamMap m[6] = {{D.oW1, D.H * D.IN}, {D.ob1, D.H}, {D.oW2, D.HB * D.H}, {D.ob2, D.HB}, {D.oW3B, D.OB * D.HB}, {D.ob3B, D.OB}};
    memcpy(pm, m, sizeof(m));
    return 6;
}

// After the problems' inputs are staged: the once-per-train block layouts of the two clouds, all problems in one launch each.
static int ys_sort_npow(const Dims& D) { return pow2_at_least(D.NT < YS_CHUNK ? D.NT : YS_CHUNK); }
static int ys_sort_smem(const Dims& D) {       // keys + [group][6] box scratch of ONE chunk's blocks
    const int per_chunk = YS_CHUNK / (64 * D.ppl);
    return ys_sort_npow(D) * 8 + 6 * 4 * (D.nyb < per_chunk ? D.nyb : per_chunk) * D.ppl;
}
static int ps_sort_smem(const Dims& D) {       // a cluster can hold every point
    const int BS = 64 * D.ppl;
    const int np = pow2_at_least((D.NP + BS - 1) / BS * BS);
    return (np < YS_CHUNK ? np : YS_CHUNK) * 8;
}
static void launch_sorts(Plan* P, hipStream_t s, int nz, bool keep_target_blocks = false) {
    const Dims& D = P->D;
    // (keep_target_blocks: the launch only verifies the caller's claim -- see k_sort_y)
    if (D.nyb) hipLaunchKernelGGL(k_sort_y, dim3(cdiv(D.NT, YS_CHUNK), 1, nz), dim3(1024), ys_sort_smem(D), s, D, P->W, P->bstride, ys_sort_npow(D), keep_target_blocks ? 1 : 0);
    if (D.npb) hipLaunchKernelGGL(k_sort_p, dim3(D.K, 1, nz), dim3(512), ps_sort_smem(D), s, D, P->W, P->bstride);
}

// Inputs of `n` problems into their workspaces: the parameter / offset copies of all of them in one launch (per
// COPY_MAX ranges), then one k_prep and one k_l1 launch with the problems in grid.z.
static int stage_inputs(Plan* P, const creg_train_args* args, int n, hipStream_t s) {
    const Dims& D = P->D;
    ParamMap pm[10];
    const int np = param_map(D, pm);
    CopyTable T; T.count = 0;
    PrepBatch pb;
    for (int b = 0; b < n; ++b) {
        const creg_train_args* a = args + b;
        const Ws W = ws_shift(P->W, (size_t)b * P->bstride);
        for (int i = 0; i < np; ++i) {
            CREG_REQUIRE(a->params[i], "creg_train: params[%d] of problem %d is null", i, b);
            copy_table_add(T, a->params[i], W.P + pm[i].off, sizeof(float) * pm[i].count, s);
        }
        copy_table_add(T, a->seg_offsets, W.off, sizeof(int) * (D.K + 1), s);
        pb.m[b] = a->m; pb.y[b] = a->y; pb.pts[b] = a->local_pts;
        pb.hy[b] = Hyper{a->lr, a->sched_factor, a->sched_patience, a->stop};
    }
    copy_table_launch(T, s);
    const int mx = D.NP > D.NT ? D.NP : D.NT;
    int blocks = cdiv(mx > D.NPAR ? mx : D.NPAR, 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_prep, dim3(blocks, 1, n), dim3(256), 0, s, D, P->W, P->bstride, pb);
    hipLaunchKernelGGL(k_l1, dim3(cdiv(D.H, 4), 1, n), dim3(256), 0, s, D, P->W, 0, P->bstride);
    {   // the first epoch's hidden activation (later ones come out of k_dw)
        const int nz_keep = P->nz;
        P->nz = n;
        launch_l2(P, 0, s);
        P->nz = nz_keep;
    }
    CREG_LAUNCH_CHECK();
    return CREG_OK;
}

}  // namespace creg
using namespace creg;

// EPG consecutive epochs captured as one graph (G independent branches over contiguous groups of problems).  On any
// failure the capture is ended, and every stream / event / graph created here is destroyed before returning.
static int capture_epochs(Plan* P, int epg) {
    // captured on a private stream (torch's current stream is usually the null stream, which cannot be captured); the
    // instantiated graph is then launched on the caller's stream.
    const int G = P->branches;
    hipStream_t cs = nullptr, cs2[16] = {nullptr};
    hipEvent_t ef = nullptr, ej[16] = {nullptr};
    hipGraph_t g = nullptr;
    bool capturing = false;
    const Ws Wall = P->W;
    const int nz_all = P->nz;
    hipError_t err = hipSuccess;
    const char* what = "";
#define CAP_TRY(call) do { if (err == hipSuccess) { err = (call); if (err != hipSuccess) what = #call; } } while (0)
    CAP_TRY(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    CAP_TRY(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
    capturing = err == hipSuccess;
    if (G > 1) {
        // fork / join through events, so the instantiated graph has parallel chains: the runtime feeds them to
        // different hardware queues and the latency-bound kernels of one group overlap the long launches of the
        // other.  Problems never interact, so results do not depend on G (stress-tested: tests/test_gpu_parity.py).
        CAP_TRY(hipEventCreateWithFlags(&ef, hipEventDisableTiming));
        CAP_TRY(hipEventRecord(ef, cs));
        int first = 0;
        for (int gi = 0; gi < G && err == hipSuccess; ++gi) {
            const int cnt = P->B / G + (gi < P->B % G ? 1 : 0);
            hipStream_t st = cs;
            if (gi > 0) {
                CAP_TRY(hipStreamCreateWithFlags(&cs2[gi], hipStreamNonBlocking));
                CAP_TRY(hipStreamWaitEvent(cs2[gi], ef, 0));
                st = cs2[gi];
            }
            if (err != hipSuccess) break;
            P->W = ws_shift(Wall, (size_t)first * P->bstride); P->nz = cnt;
            for (int i = 0; i < epg; ++i) enqueue_epoch(P, i, st);
            if (gi > 0) {
                CAP_TRY(hipEventCreateWithFlags(&ej[gi], hipEventDisableTiming));
                CAP_TRY(hipEventRecord(ej[gi], st));
            }
            first += cnt;
        }
        P->W = Wall; P->nz = nz_all;
        for (int gi = 1; gi < G; ++gi) if (ej[gi]) CAP_TRY(hipStreamWaitEvent(cs, ej[gi], 0));
    } else {
        if (err == hipSuccess) for (int i = 0; i < epg; ++i) enqueue_epoch(P, i, cs);
    }
    if (capturing) {
        const hipError_t e2 = hipStreamEndCapture(cs, &g);          // always end the capture, also after a failure
        if (err == hipSuccess && e2 != hipSuccess) { err = e2; what = "hipStreamEndCapture"; }
    }
    CAP_TRY(hipGraphInstantiate(&P->gexec, g, nullptr, nullptr, 0));
#undef CAP_TRY
    if (g) (void)hipGraphDestroy(g);
    for (int gi = 1; gi < 16; ++gi) { if (cs2[gi]) (void)hipStreamDestroy(cs2[gi]); if (ej[gi]) (void)hipEventDestroy(ej[gi]); }
    if (ef) (void)hipEventDestroy(ef);
    if (cs) (void)hipStreamDestroy(cs);
    if (err != hipSuccess) {
        (void)hipGetLastError();
        if (P->gexec) { (void)hipGraphExecDestroy(P->gexec); P->gexec = nullptr; }
        creg::set_error("creg_train_plan_run_batch: graph capture failed: %s: %s", what, hipGetErrorString(err));
        return CREG_EHIP;
    }
    P->graph_ready = true;
    P->graph_epochs = epg;
    return CREG_OK;
}

// ---- chain streams must sit in DIFFERENT hardware queues (round 5) -------------------------------------------------------------
// The runtime multiplexes streams onto GPU_MAX_HW_QUEUES (4) hardware queues: the first four streams of a process get a queue each,
// every later one SHARES the queue with the fewest users -- possibly the caller's.  Two chains in one queue run one after the other
// (75 us per epoch instead of 45.8: the default bench line fell 182 -> 112 frames/s the moment a world-1 RCCL group, which owns
// streams of its own, was created before the plan; profiles/r05_rccl_vs_chains.log).  HIP does not say which queue a stream got, so
// it is measured: a stream is accepted as a chain's when a 150 us spin kernel on it runs CONCURRENTLY with one on the caller's stream
// and on the chains picked before it; rejected candidates stay alive until the pick is over (so that the next one lands elsewhere).
__global__ void k_spin(unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}

static double spin_together_us(hipStream_t* st, int n, unsigned long long ticks) {
    for (int i = 0; i < n; ++i) if (hipStreamSynchronize(st[i]) != hipSuccess) return -1.0;
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st[i], ticks);
    if (hipGetLastError() != hipSuccess) return -1.0;
    for (int i = 0; i < n; ++i) if (hipStreamSynchronize(st[i]) != hipSuccess) return -1.0;
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
}

// (Re)creates P->cst[1 .. branches-1] so that they overlap with `s` and with each other; keeps what it has when no better stream exists.
static int pick_chain_streams(Plan* P, hipStream_t s) {
    static const bool probe = !(getenv("CREG_NO_QUEUE_PROBE") && getenv("CREG_NO_QUEUE_PROBE")[0] == '1');
    const unsigned long long ticks = 15000;                 // wall_clock64 runs at 100 MHz: 150 us
    P->probe_us = 0;
    for (int gi = 1; gi < P->branches; ++gi) {
        hipStream_t set[9], rejected[8];
        int ns = 0, nr = 0;
        set[ns++] = s;
        for (int j = 1; j < gi; ++j) set[ns++] = P->cst[j];
        hipStream_t chosen = nullptr;
        if (P->cst[gi]) {                                    // the stream of an earlier pick (another caller stream): keep it if it still fits
            set[ns] = P->cst[gi];
            double t = spin_together_us(set, ns + 1, ticks);
            const double t2 = t < 0 || t >= 250.0 ? spin_together_us(set, ns + 1, ticks) : t;
            if (t2 >= 0 && (t < 0 || t2 < t)) t = t2;
            if (!probe || (t >= 0 && t < 250.0)) { chosen = P->cst[gi]; if (probe && P->probe_us >= 0 && (int)t > P->probe_us) P->probe_us = (int)t; }
            else { (void)hipStreamSynchronize(P->cst[gi]); rejected[nr++] = P->cst[gi]; P->cst[gi] = nullptr; }
        }
        for (int attempt = 0; !chosen && attempt < 7; ++attempt) {
            hipStream_t cs = nullptr;
            if (hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) != hipSuccess) break;
            if (!probe) { chosen = cs; break; }
            set[ns] = cs;
            double t = spin_together_us(set, ns + 1, ticks);
            if (t < 0 || t >= 250.0) {                       // a late host thread looks like a shared queue: ask twice
                const double t2 = spin_together_us(set, ns + 1, ticks);
                if (t2 >= 0 && (t < 0 || t2 < t)) t = t2;
            }
            if (t >= 0 && t < 250.0) { chosen = cs; if (P->probe_us >= 0 && (int)t > P->probe_us) P->probe_us = (int)t; }
            else if (nr < 8) rejected[nr++] = cs;
            else (void)hipStreamDestroy(cs);
        }
        if (!chosen && nr) { chosen = rejected[--nr]; P->probe_us = -1; }      // every queue is shared with somebody: any stream is as good as another
        for (int i = 0; i < nr; ++i) (void)hipStreamDestroy(rejected[i]);
        if (!chosen) {
            (void)hipGetLastError();
            creg::set_error("creg_train_plan_run_batch: cannot create a stream for chain %d", gi);
            return CREG_EHIP;
        }
        P->cst[gi] = chosen;
    }
    P->cal_stream = s; P->cal_done = true;
    return CREG_OK;
}

// chain-stream mode: chain gi's EPG epochs as a linear graph (captured on a temporary stream: an instantiated graph runs on any)
static int chain_first(const Plan* P, int gi) { int f = 0; for (int i = 0; i < gi; ++i) f += P->B / P->branches + (i < P->B % P->branches ? 1 : 0); return f; }
static int chain_count(const Plan* P, int gi) { return P->B / P->branches + (gi < P->B % P->branches ? 1 : 0); }
static int capture_chains(Plan* P, int epg) {
    const Ws Wall = P->W;
    const int nz_all = P->nz;
    hipError_t err = hipSuccess;
    const char* what = "";
#define CAP_TRY(call) do { if (err == hipSuccess) { err = (call); if (err != hipSuccess) what = #call; } } while (0)
    if (!P->cfork) CAP_TRY(hipEventCreateWithFlags(&P->cfork, hipEventDisableTiming));
    hipStream_t cs = nullptr;
    CAP_TRY(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    for (int gi = 0; gi < P->branches && err == hipSuccess; ++gi) {
        hipGraph_t g = nullptr;
        if (gi > 0 && !P->cjoin[gi]) CAP_TRY(hipEventCreateWithFlags(&P->cjoin[gi], hipEventDisableTiming));
        if (P->cexec[gi]) { (void)hipGraphExecDestroy(P->cexec[gi]); P->cexec[gi] = nullptr; }      // (a retry after a failed capture)
        CAP_TRY(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
        if (err == hipSuccess) {
            P->W = ws_shift(Wall, (size_t)chain_first(P, gi) * P->bstride); P->nz = chain_count(P, gi);
            for (int i = 0; i < epg; ++i) enqueue_epoch(P, i, cs);
            P->W = Wall; P->nz = nz_all;
            const hipError_t e2 = hipStreamEndCapture(cs, &g);
            if (e2 != hipSuccess) { err = e2; what = "hipStreamEndCapture"; }
        }
        CAP_TRY(hipGraphInstantiate(&P->cexec[gi], g, nullptr, nullptr, 0));
        if (g) (void)hipGraphDestroy(g);
    }
    if (cs) (void)hipStreamDestroy(cs);
#undef CAP_TRY
    if (err != hipSuccess) {
        (void)hipGetLastError();
        for (int gi = 0; gi < 8; ++gi) if (P->cexec[gi]) { (void)hipGraphExecDestroy(P->cexec[gi]); P->cexec[gi] = nullptr; }
        creg::set_error("creg_train_plan_run_batch: chain graph capture failed: %s: %s", what, hipGetErrorString(err));
        return CREG_EHIP;
    }
    P->graph_ready = true;
    P->graph_epochs = epg;
    return CREG_OK;
}

static int batch_of(const creg_train_shape* s) { return s->batch >= 1 ? s->batch : 1; }

extern "C" size_t creg_train_workspace_bytes(const creg_train_shape* shape) {
    Dims D;
    if (!make_dims(shape, &D) || shape->batch < 0 || shape->batch > 64) return 0;
    return align_up(carve(D, nullptr, nullptr), 256) * batch_of(shape);
}

extern "C" int creg_train_plan_create(const creg_train_shape* shape, void* workspace, size_t workspace_bytes,
                                      creg_train_plan** plan) {
    Dims D;
    CREG_REQUIRE(plan && workspace, "creg_train_plan_create: null pointer");
    CREG_REQUIRE(make_dims(shape, &D), "creg_train_plan_create: unsupported shape (rot in 0..3, hidden in {64, 128, 256, 512}, 1 <= k <= 256 with the K x IN features + the B role's tiles inside 160 KB of LDS, sizes >= 1)");
    CREG_REQUIRE(((uintptr_t)workspace & 255) == 0, "creg_train_plan_create: workspace must be 256-byte aligned");
    CREG_REQUIRE(shape->batch >= 0 && shape->batch <= 64, "creg_train_plan_create: batch must be in [0, 64]");
    const size_t one = align_up(carve(D, nullptr, nullptr), 256), need = one * batch_of(shape);
    CREG_REQUIRE(workspace_bytes >= need, "creg_train_plan_create: workspace too small (%zu < %zu)", workspace_bytes, need);
    Plan* P = new Plan();
    P->shape = *shape; P->D = D; P->base = (char*)workspace; P->bytes = workspace_bytes;
    carve(D, P->base, &P->W);
    P->B = batch_of(shape); P->nz = P->B; P->bstride = one;
    // How the problems of a batch share the GPU (round 4, profiles/r04_chains_by_batch.log; bench.py --graph-branches):
    //   graph_branches < 0: -n CHAIN STREAMS -- the batch is cut into n contiguous groups, every group's epochs are a LINEAR graph
    //     (replayed from pre-built packets: 0.6-1.2 ms of host time per 300-epoch train) on its own stream; chain 0 uses the
    //     caller's stream, the others fork from / join it once per train.
    //   graph_branches > 0: n parallel branches inside ONE graph (rounds 2-3).  The runtime enqueues every node of a multi-branch
    //     graph from the host at each replay -- 9 ms per train with two branches, 14 of the train's 15 ms with three, which is why
    //     three branches ran "bimodal" (host-bound: any hiccup of the enqueuing thread stalls a queue).
    //   0 = auto: chain streams; one chain up to 4 problems (47.9 / 92.9 / 127 / 153 frames/s for 1-4 sequences against
    //     - / 84 / 119 / 152 with two), two for 5-7 (182 / 201 / 219 against 175 / 188 / 189 with one and 172 / 198 / 209 with
    //     three), three from 8 (230 against 226); frames above 4096 points (several points per lane in the NN kernels, whose tails
    //     leave CUs idle) take three from 3 problems on (franka shape 79.4 against 74.8 / 74.0 with two / one).  Four chains need a
    //     fifth hardware queue and collapse (112 at 6 sequences).  Results never depend on any of this.
    const int auto_chains = (D.NT > 64 * 64 && P->B >= 3) ? 3 : (P->B >= 8 ? 3 : (P->B >= 5 ? 2 : 1));
    P->chain_streams = shape->graph_branches <= 0;
    P->branches = shape->graph_branches > 0 ? shape->graph_branches : (shape->graph_branches < 0 ? -shape->graph_branches : auto_chains);
    if (P->branches > P->B) P->branches = P->B;
    if (P->branches > 8) P->branches = 8;
    P->gexec = nullptr; P->graph_ready = false; P->target_blocks_valid = false;
    P->cfork = nullptr; P->cal_stream = nullptr; P->cal_done = false; P->probe_us = 0;
    for (int i = 0; i < 8; ++i) { P->cst[i] = nullptr; P->cexec[i] = nullptr; P->cjoin[i] = nullptr; }
    {   // the dynamic LDS of k_bd is the B role's (71 KB at K = 20, hidden 512: its 48 KB slab of W2 + 23 KB; the D role uses none)
        P->smem_bd = (int)(sizeof(float) * b2_smem_floats(D.K, D.IN, D.H2));
    }
    CREG_REQUIRE(P->smem_bd <= 160 * 1024, "creg_train_plan_create: k_bd needs %d B of LDS (K too large)", P->smem_bd);
    {   // the dynamic-LDS limit is per kernel AND per device: raise it at every plan creation (a process may drive several
        // GPUs; a cached "already set" flag would leave the second device at 64 KB)
        hipError_t e2 = hipSuccess;
        by_bd(D, [&](auto kern) { e2 = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
        CREG_REQUIRE(e2 == hipSuccess, "creg_train_plan_create: cannot raise the dynamic LDS limit of k_bd");
        // the fused launch carries the gradient role's static LDS (4 KB) beside the B role's dynamic block
        const char* fe = getenv("CREG_FUSED_GBD");
        // Measured (profiles/r05_fused_gbd_ab.log): bit-identical, and SLOWER -- 174 against 181.5 frames/s at five sequences, 48.5
        // against 48.8 at one, franka 77.2 against 80.9: what the consumers do AFTER the gradients exist (B: a 61 KB read of g_h2, the
        // MFMAs, the cross-wave sum, dW1, Adam, the next activation: ~7 us of dependent phases) is the critical path, not the intake
        // the prefetch hides, and the hand-off costs what the boundary did.  Off unless CREG_FUSED_GBD=1 (kept as a tested experiment).
        P->fused = (fe ? fe[0] == '1' : false) && P->smem_bd + 4608 <= 160 * 1024;
        // Round 6: the next hidden activation inside the backward launch (no k_l2 launch per epoch).  Hidden 512 without the '6d' input width;
        // CREG_L2_IN_BD=0 / 1 overrides (A/B).  Not together with the fused gradient launch.
        const char* le = getenv("CREG_L2_IN_BD");
        P->l2in = !P->fused && D.H == 512 && D.IN <= 64 && (le ? le[0] == '1' : CREG_L2_IN_BD_DEFAULT);
        if (P->l2in) {
            if (D.HA) e2 = hipFuncSetAttribute((const void*)k_bd<8, 96, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            else e2 = hipFuncSetAttribute((const void*)k_bd<8, 64, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e2 != hipSuccess) { (void)hipGetLastError(); P->l2in = false; }
        }
        if (P->fused) {
            by_gbd(D, [&](auto kern) { e2 = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4608); });
            if (e2 != hipSuccess) { (void)hipGetLastError(); P->fused = false; }
        }
    }
    // the limit is per kernel, not per plan: always raise it to the largest any plan can ask for (16384 keys + boxes),
    // so that a small plan created later does not lower it under a large one
    if (D.npb) CREG_HIP(hipFuncSetAttribute((const void*)k_sort_p, hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 8));
    if (D.nyb) CREG_HIP(hipFuncSetAttribute((const void*)k_sort_y, hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 8 + 6 * 4 * 256));
    *plan = (creg_train_plan*)P;
    return CREG_OK;
}

extern "C" int creg_train_plan_run_batch(creg_train_plan* plan, const creg_train_args* args, int32_t n,
                                         creg_stream_t stream) {
    Plan* P = (Plan*)plan;
    CREG_REQUIRE(P && args, "creg_train_plan_run_batch: null pointer");
    CREG_REQUIRE(n == P->B, "creg_train_plan_run_batch: the plan was created for a batch of %d problems, got %d", P->B, n);
    hipStream_t s = (hipStream_t)stream;
    const Dims& D = P->D;
    for (int b = 0; b < n; ++b) {
        const creg_train_args* a = args + b;
        CREG_REQUIRE(a->m && a->y && a->local_pts && a->seg_offsets && a->params && a->best_m && a->best_pred && a->result,
                     "creg_train_plan_run_batch: null pointer in problem %d", b);
    }
    {
        int rc = stage_inputs(P, args, n, s);
        if (rc) return rc;
    }
    {   // the target frame's k-d leaves survive from the previous run when the caller vouches for every problem's frame
        bool keep = P->target_blocks_valid;
        for (int b = 0; b < n; ++b) keep = keep && args[b].y_unchanged != 0;
        launch_sorts(P, s, P->B, keep);
        P->target_blocks_valid = true;
    }
    P->nz = P->B;
    int e = 0;
    if (P->shape.use_graph && D.epochs >= 2) {
        // One graph holds EPG consecutive epochs (even count: the epoch number enters the kernels only
        // through its parity, the history index comes from the device-side counter state.epochs_run),
        // so a 300-epoch train is 6 graph launches instead of 1800 kernel launches.
        int epg = P->shape.use_graph > 1 ? P->shape.use_graph : 50;
        if (epg > D.epochs) epg = D.epochs;
        epg &= ~1;
        if (!P->graph_ready) {
            const int rc = P->chain_streams ? capture_chains(P, epg) : capture_epochs(P, epg);
            if (rc) return rc;
        }
        if (P->chain_streams) {
            // fork once, every chain runs the whole train on its own stream (graph replays, then the epochs that do not fill a
            // graph), join once.  Chain 0 runs on the CALLER's stream: a stream that only waits for the others would hold a barrier
            // packet in its hardware queue for the whole train, and a parked barrier slows the chains in the other queues by ~10 %
            // (measured, tests/measure/frame_phases_by_chains.py).
            if (P->branches > 1 && (!P->cal_done || P->cal_stream != s)) {
                const int rc = pick_chain_streams(P, s);           // streams in hardware queues of their own (measured, see there)
                if (rc) { P->target_blocks_valid = false; return rc; }
            }
            const Ws Wall = P->W;
            auto st_of = [&](int gi) { return gi == 0 ? s : P->cst[gi]; };
            hipError_t cerr = hipSuccess;
            const char* cwhat = "";
#define CH_TRY(call) do { if (cerr == hipSuccess) { cerr = (call); if (cerr != hipSuccess) cwhat = #call; } } while (0)
            CH_TRY(hipEventRecord(P->cfork, s));
            for (int gi = 1; gi < P->branches; ++gi) CH_TRY(hipStreamWaitEvent(P->cst[gi], P->cfork, 0));
            for (; cerr == hipSuccess && e + P->graph_epochs <= D.epochs; e += P->graph_epochs)
                for (int gi = 0; gi < P->branches; ++gi) CH_TRY(hipGraphLaunch(P->cexec[gi], st_of(gi)));
            for (int gi = 0; cerr == hipSuccess && gi < P->branches; ++gi) {
                P->W = ws_shift(Wall, (size_t)chain_first(P, gi) * P->bstride); P->nz = chain_count(P, gi);
                for (int e2 = e; e2 < D.epochs; ++e2) enqueue_epoch(P, e2, st_of(gi));
                P->W = Wall; P->nz = P->B;
            }
            // the join is enqueued ALSO after a failure: chains that did start must not be left running on the workspace, un-joined,
            // beside whatever the caller does next (a retry would stage inputs under them)
            for (int gi = 1; gi < P->branches; ++gi) {
                if (hipEventRecord(P->cjoin[gi], P->cst[gi]) != hipSuccess || hipStreamWaitEvent(s, P->cjoin[gi], 0) != hipSuccess) {
                    (void)hipStreamSynchronize(P->cst[gi]);
                    if (cerr == hipSuccess) { cerr = hipErrorUnknown; cwhat = "joining a chain stream"; }
                }
            }
#undef CH_TRY
            if (cerr != hipSuccess) {
                (void)hipGetLastError();
                P->target_blocks_valid = false;                    // nothing of this run may be vouched for by the next one
                creg::set_error("creg_train_plan_run_batch: %s: %s", cwhat, hipGetErrorString(cerr));
                return CREG_EHIP;
            }
            e = D.epochs;
        } else {
            for (; e + P->graph_epochs <= D.epochs; e += P->graph_epochs) CREG_HIP(hipGraphLaunch(P->gexec, s));
        }
    }
    for (; e < D.epochs; ++e) enqueue_epoch(P, e, s);
    hipLaunchKernelGGL(k_params_home, dim3(64, 1, P->B), dim3(256), 0, s, D, P->W, P->bstride, D.epochs & 1, 0);
    CREG_LAUNCH_CHECK();
    // results out, parameters back into the callers' tensors
    ParamMap pm[10];
    const int np = param_map(D, pm);
    CopyTable T; T.count = 0;
    for (int b = 0; b < n; ++b) {
        const creg_train_args* a = args + b;
        const Ws W = ws_shift(P->W, (size_t)b * P->bstride);
        for (int i = 0; i < np; ++i) copy_table_add(T, W.P + pm[i].off, a->params[i], sizeof(float) * pm[i].count, s);
        copy_table_add(T, W.best_m, a->best_m, sizeof(float) * 16 * D.K, s);
        copy_table_add(T, W.best_pred, a->best_pred, sizeof(float) * 3 * D.NP, s);
        copy_table_add(T, W.result, a->result, sizeof(float) * 4, s);
        if (a->loss_hist) copy_table_add(T, W.loss_hist, a->loss_hist, sizeof(float) * D.epochs, s);
        if (a->lr_hist) copy_table_add(T, W.lr_hist, a->lr_hist, sizeof(float) * D.epochs, s);
    }
    copy_table_launch(T, s);
    return CREG_OK;
}

extern "C" int creg_train_plan_run(creg_train_plan* plan, const creg_train_args* a, creg_stream_t stream) {
    return creg_train_plan_run_batch(plan, a, 1, stream);
}

// Round 6: a train continued from a caller-supplied optimizer / control state (torch.optim.Adam's exp_avg / exp_avg_sq / step,
// ReduceLROnPlateau's best / num_bad_epochs / lr, train()'s min_loss / count: mlp_reg.py:41-50,96-119) -- checkpoint / resume of a
// train, and what the teacher-forced late-epoch parity tests are built on (the oracle's state entering epoch e, ONE epoch of the plan).
// Problem slot 0, eager launches (the kernels are the graph's; graph == eager is tested elsewhere).
extern "C" int creg_train_plan_resume(creg_train_plan* plan, const creg_train_args* a, const creg_train_state* from, int32_t n_epochs,
                                      float* const* exp_avg_out, float* const* exp_avg_sq_out, double* state_out, creg_stream_t stream) {
    Plan* P = (Plan*)plan;
    CREG_REQUIRE(P && a && from, "creg_train_plan_resume: null pointer");
    CREG_REQUIRE(a->m && a->y && a->local_pts && a->seg_offsets && a->params && a->best_m && a->best_pred && a->result, "creg_train_plan_resume: null pointer in the problem");
    CREG_REQUIRE(from->exp_avg && from->exp_avg_sq, "creg_train_plan_resume: the state needs both Adam moment arrays");
    const Dims& D = P->D;
    CREG_REQUIRE(n_epochs >= 1 && from->step >= 0 && from->epochs_run >= 0 && from->step + n_epochs <= D.epochs && from->epochs_run + n_epochs <= D.epochs,
                 "creg_train_plan_resume: step %d / epochs_run %d + %d epochs exceed the plan's %d", from->step, from->epochs_run, n_epochs, D.epochs);
    CREG_REQUIRE(from->lr >= 0.0 && from->sched_bad >= 0 && from->count >= 0, "creg_train_plan_resume: negative lr or counter");
    hipStream_t s = (hipStream_t)stream;
    int rc = stage_inputs(P, a, 1, s);          // parameters into the parity-0 buffer, the activations of epoch "0" from them, state and moments zeroed
    if (rc) return rc;
    P->target_blocks_valid = false;
    launch_sorts(P, s, 1);
    ParamMap pm[10];
    const int np = param_map(D, pm);
    {
        CopyTable T; T.count = 0;
        for (int i = 0; i < np; ++i) {
            CREG_REQUIRE(from->exp_avg[i] && from->exp_avg_sq[i], "creg_train_plan_resume: moment tensor %d is null", i);
            copy_table_add(T, from->exp_avg[i], P->W.AM + pm[i].off, sizeof(float) * pm[i].count, s);
            copy_table_add(T, from->exp_avg_sq[i], P->W.AV + pm[i].off, sizeof(float) * pm[i].count, s);
        }
        if (from->best_epoch >= 0) {            // the best so far is the caller's (kept when none of the resumed epochs improves on min_loss)
            copy_table_add(T, a->best_m, P->W.best_m, sizeof(float) * 16 * D.K, s);
            copy_table_add(T, a->best_pred, P->W.best_pred, sizeof(float) * 3 * D.NP, s);
        }
        copy_table_launch(T, s);
    }
    const ResumeState rs{from->lr, from->sched_best, from->step, from->epochs_run, from->sched_bad, from->count, from->best_epoch, from->stopped ? 1 : 0, from->min_loss};
    hipLaunchKernelGGL(k_set_state, dim3(1), dim3(64), 0, s, D, P->W, rs);
    const int nz_keep = P->nz;
    P->nz = 1;
    for (int e = 0; e < n_epochs; ++e) enqueue_epoch(P, e, s);
    P->nz = nz_keep;
    hipLaunchKernelGGL(k_params_home, dim3(64, 1, 1), dim3(256), 0, s, D, P->W, P->bstride, n_epochs & 1, from->step);
    if (state_out) hipLaunchKernelGGL(k_get_state, dim3(1), dim3(64), 0, s, P->W, n_epochs & 1, state_out);
    CREG_LAUNCH_CHECK();
    CopyTable T; T.count = 0;
    for (int i = 0; i < np; ++i) {
        copy_table_add(T, P->W.P + pm[i].off, a->params[i], sizeof(float) * pm[i].count, s);
        if (exp_avg_out && exp_avg_out[i]) copy_table_add(T, P->W.AM + pm[i].off, exp_avg_out[i], sizeof(float) * pm[i].count, s);
        if (exp_avg_sq_out && exp_avg_sq_out[i]) copy_table_add(T, P->W.AV + pm[i].off, exp_avg_sq_out[i], sizeof(float) * pm[i].count, s);
    }
    copy_table_add(T, P->W.best_m, a->best_m, sizeof(float) * 16 * D.K, s);
    copy_table_add(T, P->W.best_pred, a->best_pred, sizeof(float) * 3 * D.NP, s);
    copy_table_add(T, P->W.result, a->result, sizeof(float) * 4, s);
    if (a->loss_hist) copy_table_add(T, P->W.loss_hist, a->loss_hist, sizeof(float) * D.epochs, s);
    if (a->lr_hist) copy_table_add(T, P->W.lr_hist, a->lr_hist, sizeof(float) * D.epochs, s);
    copy_table_launch(T, s);
    CREG_LAUNCH_CHECK();
    return CREG_OK;
}

extern "C" int creg_train_plan_probe(creg_train_plan* plan, const creg_train_args* a, float* m2, float* pred,
                                     float* loss, float* grad_m2, creg_stream_t stream) {
    Plan* P = (Plan*)plan;
    CREG_REQUIRE(P && a && a->m && a->y && a->local_pts && a->seg_offsets && a->params, "creg_train_plan_probe: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const Dims& D = P->D; const Ws& W = P->W;
    int rc = stage_inputs(P, a, 1, s);
    if (rc) return rc;
    P->target_blocks_valid = false;
    launch_sorts(P, s, 1);
    P->nz = 1;
    launch_head(P, 0, s);
    launch_nn(D, W, 0, 1, s);
    hipLaunchKernelGGL(k_gradc, dim3(D.K, 1, 1), dim3(256), 0, s, D, W, 0, D.nbx, D.nby, (size_t)0);
    CREG_LAUNCH_CHECK();
    if (m2) CREG_HIP(hipMemcpyAsync(m2, W.m2, sizeof(float) * 16 * D.K, hipMemcpyDeviceToDevice, s));
    if (pred) CREG_HIP(hipMemcpyAsync(pred, W.best_pred, sizeof(float) * 3 * D.NP, hipMemcpyDeviceToDevice, s));
    if (loss) CREG_HIP(hipMemcpyAsync(loss, W.loss_hist, sizeof(float), hipMemcpyDeviceToDevice, s));
    if (grad_m2) CREG_HIP(hipMemcpyAsync(grad_m2, W.gm2, sizeof(float) * 16 * D.K, hipMemcpyDeviceToDevice, s));
    return CREG_OK;
}

extern "C" int creg_train_plan_profile(creg_train_plan* plan, const creg_train_args* a, int32_t n_epochs,
                                       float* us_out, creg_stream_t stream) {
    Plan* P = (Plan*)plan;
    CREG_REQUIRE(P && a && a->m && a->y && a->local_pts && a->seg_offsets && a->params && us_out && n_epochs >= 1,
                 "creg_train_plan_profile: bad argument");
    hipStream_t s = (hipStream_t)stream;
    // the launches as the timed region issues them: the problems of the first (largest) graph branch in grid.z, EVERY one of them
    // staged from `a` and advanced through the bracketed epochs.  (Until the end of round 3 only problem 0 was staged: the others
    // kept the state of the caller's last trains, and a train that had stopped early leaves `stopped` set -- its workgroups in the
    // back-to-back launches below requested their loads and returned, so those launches carried less work than their label said.)
    const int br = P->shape.use_graph ? P->branches : 1;
    const int nzb = P->B / br + (P->B % br ? 1 : 0);
    std::vector<creg_train_args> all((size_t)nzb, *a);
    int rc = stage_inputs(P, all.data(), nzb, s);
    if (rc) return rc;
    P->target_blocks_valid = false;
    launch_sorts(P, s, nzb);
    P->nz = nzb;
    std::vector<hipEvent_t> ev((size_t)(NKERN + 1) * n_epochs);
    for (auto& e : ev) CREG_HIP(hipEventCreate(&e));
    for (int e = 0; e < n_epochs; ++e) enqueue_epoch(P, e, s, ev.data() + (NKERN + 1) * e);
    CREG_LAUNCH_CHECK();
    CREG_HIP(hipStreamSynchronize(s));
    double acc[NKERN] = {0};
    for (int e = 0; e < n_epochs; ++e)
        for (int k = 0; k < NKERN; ++k) {
            float ms = 0.f;
            CREG_HIP(hipEventElapsedTime(&ms, ev[(NKERN + 1) * e + k], ev[(NKERN + 1) * e + k + 1]));
            acc[k] += ms;
        }
    us_out[0] = 0.f;                    // (round 2's k_l2 slot: the hidden forward has no launch of its own any more)
    for (int k = 0; k < NKERN; ++k) us_out[1 + k] = (float)(acc[k] * 1000.0 / n_epochs);
    // us_out[6]: the nearest-neighbour kernel alone, REP back-to-back launches between two events
    // (per-kernel event brackets carry ~7 us of event overhead; this one carries only the
    // launch-to-launch gap, so it upper-bounds the rocprofv3 kernel duration by ~1 us).
    const int REP = 200;
    const Dims& D = P->D; const Ws& W = P->W;
    CREG_HIP(hipEventRecord(ev[0], s));
    for (int i = 0; i < REP; ++i)
        launch_nn(D, W, P->bstride, P->nz, s);
    CREG_HIP(hipEventRecord(ev[1], s));
    CREG_HIP(hipStreamSynchronize(s));
    float ms = 0.f;
    CREG_HIP(hipEventElapsedTime(&ms, ev[0], ev[1]));
    us_out[6] = ms * 1000.f / REP;
    us_out[7] = (float)P->nz;       // problems carried by that launch
    // us_out[8..11]: k_bd, k_l2, k_head, k_gradc the same way (measurement hooks: every k_bd launch is one more Adam step on
    // the plan's copies of the parameters, alternating between the two buffers; the caller's tensors are not written back)
    auto b2b = [&](auto launch, float* out) -> int {
        CREG_HIP(hipEventRecord(ev[0], s));
        for (int i = 0; i < REP; ++i) launch(i);
        CREG_HIP(hipEventRecord(ev[1], s));
        CREG_HIP(hipStreamSynchronize(s));
        CREG_HIP(hipEventElapsedTime(&ms, ev[0], ev[1]));
        *out = ms * 1000.f / REP;
        return CREG_OK;
    };
#ifdef CREG_BD_STAMPS
    {   BdStamps* z = new BdStamps; memset(z, 0, sizeof(*z)); memset(z->first, 0xff, sizeof(z->first)); memset(z->startB, 0xff, sizeof(z->startB));
        memset(z->startD, 0xff, sizeof(z->startD)); z->armed = 1;
        CREG_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_bd_st), z, sizeof(*z))); delete z; }
#endif
    if (int rc2 = b2b([&](int i) { launch_bd(P, i, s); }, us_out + 8)) return rc2;
#ifdef CREG_BD_STAMPS
    { const int zero = 0; CREG_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_bd_st), &zero, sizeof(int), offsetof(BdStamps, armed))); }
#endif
    if (int rc2 = b2b([&](int i) { launch_l2(P, i & 1, s); }, us_out + 9)) return rc2;
    if (int rc2 = b2b([&](int i) { launch_head(P, i & 1, s); }, us_out + 10)) return rc2;
    if (int rc2 = b2b([&](int i) { hipLaunchKernelGGL(k_gradc, dim3(D.K, 1, P->nz), dim3(256), 0, s, D, W, i, D.nbx, D.nby, P->bstride); }, us_out + 11)) return rc2;
    CREG_LAUNCH_CHECK();
    P->nz = P->B;
    for (auto& e : ev) (void)hipEventDestroy(e);
    return CREG_OK;
}

#ifdef CREG_BD_STAMPS
// out[0..4]: mean launch span, B start, B end, D start, D end relative to the launch's first workgroup (us); out[5], out[6]: B / D
// workgroups per launch; out[8..15] / out[16..23]: mean phase durations of a B / D workgroup (us)
extern "C" int creg_debug_bd_stamps(double* out) {
    BdStamps* z = new BdStamps;
    if (hipMemcpyFromSymbol(z, HIP_SYMBOL(g_bd_st), sizeof(*z)) != hipSuccess) { delete z; return CREG_EHIP; }
    for (int i = 0; i < 24; ++i) out[i] = 0.0;
    int n = 0;
    for (int sl = 0; sl < 256; ++sl) {
        if (z->first[sl] == ~0ull || !z->endB[sl] || !z->endD[sl]) continue;
        ++n;
        const double f = (double)z->first[sl];
        out[0] += ((double)(z->endB[sl] > z->endD[sl] ? z->endB[sl] : z->endD[sl]) - f) / 100.0;
        out[1] += ((double)z->startB[sl] - f) / 100.0; out[2] += ((double)z->endB[sl] - f) / 100.0;
        out[3] += ((double)z->startD[sl] - f) / 100.0; out[4] += ((double)z->endD[sl] - f) / 100.0;
    }
    for (int i = 0; i < 5; ++i) out[i] /= n ? n : 1;
    out[5] = n ? (double)z->blocks[0] / n : 0; out[6] = n ? (double)z->blocks[1] / n : 0; out[7] = n;
    for (int r = 0; r < 2; ++r)
        for (int k = 0; k < 8; ++k) out[8 + 8 * r + k] = z->blocks[r] ? (double)z->phase[r][k] / (double)z->blocks[r] / 100.0 : 0.0;
    delete z;
    return CREG_OK;
}
#endif

#ifdef CREG_NN_STATS
extern "C" int creg_debug_nn_stats(unsigned long long* out8, int reset) {
    CREG_HIP(hipMemcpyFromSymbol(out8, HIP_SYMBOL(creg::g_nn_stats), sizeof(unsigned long long) * 8));
    if (reset) { unsigned long long z[8] = {0}; CREG_HIP(hipMemcpyToSymbol(HIP_SYMBOL(creg::g_nn_stats), z, sizeof(z))); }
    return CREG_OK;
}
#endif

#ifdef CREG_NN_WAVE_STAMPS
// out: up to `cap` records of 8 x u64 {t0, t1, t2, t3, visits, candidates, direction, block}; returns the number recorded; reset != 0 clears
extern "C" long long creg_debug_nn_waves(unsigned long long* out, long long cap, int reset) {
    static unsigned n[creg::NN_WAVE_SHARDS * 32];
    if (hipMemcpyFromSymbol(n, HIP_SYMBOL(creg::g_nn_wave_n), sizeof(n)) != hipSuccess) return -1;
    long long total = 0;
    if (out && cap > 0) {
        std::vector<creg::NnWaveRec> h((size_t)creg::NN_WAVE_CAP);
        if (hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(creg::g_nn_wave), sizeof(creg::NnWaveRec) * (size_t)creg::NN_WAVE_CAP) != hipSuccess) return -1;
        for (unsigned sh = 0; sh < creg::NN_WAVE_SHARDS; ++sh) {
            const unsigned m = n[32 * sh] < creg::NN_WAVE_PER ? n[32 * sh] : creg::NN_WAVE_PER;
            for (unsigned i = 0; i < m && total < cap; ++i, ++total) {
                const creg::NnWaveRec& r = h[(size_t)sh * creg::NN_WAVE_PER + i];
                unsigned long long* o = out + 8 * total;
                o[0] = r.t0; o[1] = r.t1; o[2] = r.t2; o[3] = r.t3; o[4] = (unsigned long long)r.visits; o[5] = (unsigned long long)r.cands;
                o[6] = (unsigned long long)r.dir; o[7] = (unsigned long long)r.blk;
            }
        }
    } else for (unsigned sh = 0; sh < creg::NN_WAVE_SHARDS; ++sh) total += n[32 * sh];
    if (reset) { static const unsigned z[creg::NN_WAVE_SHARDS * 32] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(creg::g_nn_wave_n), z, sizeof(z)) != hipSuccess) return -1; }
    return total;
}
#endif

#ifdef CREG_XCD_PROBE
// ---- Stage 1 of the XCD-resident train: a MEASUREMENT build (python -m autourdf_amd.build --variant xcd with
// CREG_EXTRA_FLAGS=-DCREG_XCD_PROBE; tests/measure/xcd_stage1.py).  Two questions, two kernels:
//   (a) what does a barrier among the workgroups of ONE XCD cost, a 4 KB hand-off included (every member writes its share with plain
//       stores, everybody reads all of it)?  Members find each other at run time: a workgroup reads HW_REG_XCC_ID and takes a ticket
//       from its XCD's counter -- nothing assumes blockIdx % 8.
//   (b) what does the plan's nearest-neighbour launch cost when ONE problem's blocks are confined to ONE XCD's 32 CUs (problem =
//       XCC_ID, the blocks handed out by a per-XCD queue)?
namespace creg {
__device__ __forceinline__ int xcc_id() { return (int)(__builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) & 7u); }      // HW_REG_XCC_ID[3:0]
__device__ __forceinline__ int ld_sc1(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// ctl: 128-byte lines of ints.  line x (0..7): XCD x's arrival counter; line 8 + x: its member count; line 16: all members counted.
// MODE 0 "XCD-local": plain payload stores, vmcnt(0), an atomic that STAYS in the XCD's L2 (workgroup scope: no sc1), sc1 polls and
//   sc1 payload loads (L2-served: the vector L1 of the reading CU is bypassed) -- sound only because every member IS on this XCD.
// MODE 1 the placement-independent recipe: plain stores, agent release, agent counter; relaxed poll, agent acquire, plain loads.
// MODE 2 write-through: sc1 payload stores, vmcnt(0), agent counter; sc1 polls and loads.
template <int MODE>
__global__ __launch_bounds__(256) void k_xcd_barrier(int* ctl, float* payload, int rounds, int skew, unsigned long long* ticks, int* errs) {
    __shared__ int s_t, s_n;
    const int tid = threadIdx.x;
    const int x = xcc_id();
    if (tid == 0) {
        s_t = __hip_atomic_fetch_add(ctl + 32 * (8 + x), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(ctl + 32 * 16, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int it = 0;
        while (ld_sc1(ctl + 32 * 16) < (int)gridDim.x && ++it < (1 << 22)) __builtin_amdgcn_s_sleep(8);
        s_n = it < (1 << 22) ? ld_sc1(ctl + 32 * (8 + x)) : 0;
    }
    __syncthreads();
    const int t = s_t, n = s_n;
    if (n == 0) { if (tid == 0) { ticks[blockIdx.x] = ~0ull; atomicAdd(errs, 1 << 20); } return; }      // not co-resident: no barrier possible
    const int per = 1024 / n, lo = t * per, hi = t == n - 1 ? 1024 : lo + per;      // this member's floats of the 4 KB
    int bad = 0;
    const unsigned long long t0 = wall_clock64();
    for (int r = 0; r < rounds; ++r) {
        float* buf = payload + (size_t)(2 * x + (r & 1)) * 1024;
        if (skew && (t & 3) == 1) for (int i = 0; i < skew; ++i) __builtin_amdgcn_s_sleep(32);      // uneven arrival
        for (int i = lo + tid; i < hi; i += 256) {
            const float v = (float)(r * 4096 + i + 1);
            if (MODE == 2) __hip_atomic_store(buf + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else buf[i] = v;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            if (MODE == 1) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            if (MODE == 0) __hip_atomic_fetch_add(ctl + 32 * x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else __hip_atomic_fetch_add(ctl + 32 * x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int it = 0;
            while (ld_sc1(ctl + 32 * x) < n * (r + 1) && ++it < (1 << 22)) __builtin_amdgcn_s_sleep(1);
            if (it >= (1 << 22)) atomicAdd(errs, 1 << 20);
            if (MODE == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        float4 v;
        if (MODE == 1) v = ((const float4*)buf)[tid];
        else { const u32x4v w = __builtin_amdgcn_raw_buffer_load_b128(buf_rsrc(buf, 4096), 16 * tid, 0, 16); v = make_float4(__uint_as_float(w[0]), __uint_as_float(w[1]), __uint_as_float(w[2]), __uint_as_float(w[3])); }
        const float e = (float)(r * 4096 + 4 * tid + 1);
        bad += (v.x != e) + (v.y != e + 1.f) + (v.z != e + 2.f) + (v.w != e + 3.f);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = wall_clock64();
    if (tid == 0) ticks[blockIdx.x] = ((t1 - t0) << 8) | (unsigned)(x << 5) | (unsigned)min(n - 1, 31);
    if (bad) atomicAdd(errs, bad);
}

// (b) ctl: line x of half `par`: XCD x's block queue head; line 8 + x: members seen (statistics).  The other half is zeroed for the
// next launch.  WSCOPE: the queue atomics stay in the XCD's L2.
template <bool ROWS, int NBT, int NBP, bool WSCOPE>
__global__ __launch_bounds__(NN_BLOCK) void k_nn_xcd(const float* A, int na, const float* B, int nb, int blocksA, int blocksB, EngineEpi epi, NnBlocks yb,
                                                     NnBlocks pb, size_t zstride, int nz, int* ctl, int par) {
    __shared__ int s_bx;
    const int x = xcc_id();
    int* mine = ctl + 32 * (16 * par + x);
    if (threadIdx.x == 0) {
        ctl[32 * (16 * (par ^ 1) + x)] = 0;
        ctl[32 * (16 * (par ^ 1) + 8 + x)] = 0;
        __hip_atomic_fetch_add(mine + 32 * 8, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (x >= nz) return;
    const size_t zb = x * zstride;
    A = (const float*)((const char*)A + zb);
    B = (const float*)((const char*)B + zb);
    yb.ts4 = (const float4*)((const char*)yb.ts4 + zb);
    yb.tbox = (const float*)((const char*)yb.tbox + zb);
    pb.ts4 = (const float4*)((const char*)pb.ts4 + zb);
    pb.tbox = (const float*)((const char*)pb.tbox + zb);
    pb.nblk_dev = (const int*)((const char*)pb.nblk_dev + zb);
    epi.shift(x);
    for (;;) {
        __syncthreads();                                   // the previous block's LDS partials have been read
        if (threadIdx.x == 0)
            s_bx = WSCOPE ? __hip_atomic_fetch_add(mine, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
                          : __hip_atomic_fetch_add(mine, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const int bx = s_bx;
        if (bx >= blocksA + blocksB) return;
        if constexpr (ROWS) {
            if (bx < blocksA) nn_l1_rows<NBT, EngineEpi, false, true>(pb.ts4, 0, pb.nblk_dev, pb.nblk, yb, 0, epi, bx, epi.stopped, B, epi.lossp_x);
            else nn_l1_rows<NBP, EngineEpi, true, false>(yb.ts4, yb.nblk, nullptr, yb.nblk, pb, 1, epi, bx - blocksA, epi.stopped, A, epi.lossp_y);
        } else {
            if (bx < blocksA) nn_l1_block_pruned<NBT, 1, EngineEpi, true, false>(A, na, 4, yb, 0, epi, bx, epi.stopped, B, 4);
            else nn_l1_block_pruned<NBP, 1, EngineEpi, true, true>(B, nb, 4, pb, 1, epi, bx - blocksA, epi.stopped, A, 4);
        }
    }
}
// Stage 2 pre-check: the hidden-layer launch (k_l2) with one problem's H2 / 16 workgroups confined to ONE XCD (problem = XCC_ID, blocks from the
// per-XCD queue) -- what the W2 slabs (1.5 MB per problem, written through by k_bd a launch earlier) cost through one XCD's fabric port.
template <int NC>
__global__ __launch_bounds__(L2_THREADS) void k_l2_xcd(Dims D, Ws W0, int par, size_t bstride, int nz, int* ctl, int cpar) {
    __shared__ int s_bx;
    const int x = xcc_id();
    int* mine = ctl + 32 * (16 * cpar + x);
    if (threadIdx.x == 0) ctl[32 * (16 * (cpar ^ 1) + x)] = 0;
    if (x >= nz) return;
    const Ws W = ws_shift(W0, x * bstride);
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) s_bx = __hip_atomic_fetch_add(mine, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const int bx = s_bx;
        if (bx >= D.H2 / 16) return;
        l2_body<NC>(D, W, par, bx);
    }
}
}  // namespace creg

// out (HOST, 64 doubles):
//  [0..2]  barrier + 4 KB hand-off per round (us, slowest member), 256 workgroups on an idle chip: MODE 0 / 1 / 2;   [3..5] the same with uneven arrival
//  [6..8]  payload words that arrived wrong, per MODE (both runs);   [9] members of the smallest XCD group, [10] of the largest
//  [16] the plan's NN launch, one problem on the whole chip (us, 200 back to back)   [17] the same with `nz` problems in grid.z
//  [18] XCD-confined, nz problems, agent-scope queue   [19] workgroup-scope queue   [20] 1 problem confined   [21] outputs identical to the plan's launch (1 / 0)
//  [22] workgroups per XCD used   [24..31] workgroups that landed on XCD 0..7 in the last confined launch
//  [32] k_l2 chip-wide, nz problems (us, 200 back to back)   [33] k_l2 confined, problem = XCC_ID   [34] h2 identical (1 / 0)   [35] k_l2 chip-wide, 1 problem   [36] confined, 1 problem
extern "C" int creg_debug_xcd_stage1(creg_train_plan* plan, const creg_train_args* a, int32_t nz, int32_t wg_per_cu, double* out, creg_stream_t stream) {
    Plan* P = (Plan*)plan;
    CREG_REQUIRE(P && a && out && nz >= 1 && nz <= 8 && nz <= P->B && wg_per_cu >= 1 && wg_per_cu <= 8, "creg_debug_xcd_stage1: bad argument");
    const Dims& D = P->D; const Ws& W = P->W;
    CREG_REQUIRE(D.nyb && D.npb && D.ppl == 1 && D.nbt == 1 && D.nbp == 2, "creg_debug_xcd_stage1: built for the configs[1] instance (64-point blocks, 1 + 2 boxes per lane)");
    hipStream_t s = (hipStream_t)stream;
    for (int i = 0; i < 64; ++i) out[i] = 0.0;
    int* ctl = nullptr; float* payload = nullptr; unsigned long long* ticks = nullptr; int* errs = nullptr;
    CREG_HIP(hipMalloc(&ctl, 4 * 32 * 32)); CREG_HIP(hipMalloc(&payload, 16 * 4096)); CREG_HIP(hipMalloc(&ticks, 8 * 1024)); CREG_HIP(hipMalloc(&errs, 4));
    hipEvent_t e0, e1;
    CREG_HIP(hipEventCreate(&e0)); CREG_HIP(hipEventCreate(&e1));
    // ---- (a)
    const int rounds = 200;
    for (int skew = 0; skew < 2; ++skew)
        for (int mode = 0; mode < 3; ++mode) {
            CREG_HIP(hipMemsetAsync(ctl, 0, 4 * 32 * 32, s)); CREG_HIP(hipMemsetAsync(errs, 0, 4, s)); CREG_HIP(hipMemsetAsync(payload, 0, 16 * 4096, s));
            if (mode == 0) hipLaunchKernelGGL(k_xcd_barrier<0>, dim3(256), dim3(256), 0, s, ctl, payload, rounds, skew ? 8 : 0, ticks, errs);
            else if (mode == 1) hipLaunchKernelGGL(k_xcd_barrier<1>, dim3(256), dim3(256), 0, s, ctl, payload, rounds, skew ? 8 : 0, ticks, errs);
            else hipLaunchKernelGGL(k_xcd_barrier<2>, dim3(256), dim3(256), 0, s, ctl, payload, rounds, skew ? 8 : 0, ticks, errs);
            CREG_LAUNCH_CHECK();
            CREG_HIP(hipStreamSynchronize(s));
            unsigned long long th[256]; int eh = 0;
            CREG_HIP(hipMemcpy(th, ticks, sizeof(th), hipMemcpyDeviceToHost)); CREG_HIP(hipMemcpy(&eh, errs, 4, hipMemcpyDeviceToHost));
            unsigned long long mx = 0; int nmin = 99, nmax = 0;
            for (int i = 0; i < 256; ++i) { if (th[i] == ~0ull) continue; if ((th[i] >> 8) > mx) mx = th[i] >> 8; const int n = (int)(th[i] & 31) + 1; if (n < nmin) nmin = n; if (n > nmax) nmax = n; }
            out[3 * skew + mode] = (double)mx / 100.0 / rounds;
            out[6 + mode] += eh;
            out[9] = nmin; out[10] = nmax;
        }
    // ---- (b)
    std::vector<creg_train_args> all((size_t)nz, *a);
    int rc = stage_inputs(P, all.data(), nz, s);
    if (rc) return rc;
    P->target_blocks_valid = false;
    launch_sorts(P, s, nz);
    const int nz_keep = P->nz;
    P->nz = nz;
    for (int e = 0; e < 20; ++e) enqueue_epoch(P, e, s);          // a few epochs in: the clouds as a train sees them
    launch_head(P, 0, s);
    CREG_LAUNCH_CHECK();
    const int REP = 200;
    float ms = 0.f;
    auto timed = [&](auto launch, double* o) -> int {
        launch(0);
        CREG_HIP(hipEventRecord(e0, s));
        for (int i = 0; i < REP; ++i) launch(i + 1);
        CREG_HIP(hipEventRecord(e1, s));
        CREG_HIP(hipStreamSynchronize(s));
        CREG_HIP(hipEventElapsedTime(&ms, e0, e1));
        *o = ms * 1000.0 / REP;
        return CREG_OK;
    };
    if (int r2 = timed([&](int) { launch_nn(D, W, P->bstride, 1, s); }, out + 16)) return r2;
    if (int r2 = timed([&](int) { launch_nn(D, W, P->bstride, nz, s); }, out + 17)) return r2;
    const EngineEpi epi{W.sgn_x, W.cnt4, W.lossp_x, W.lossp_y, P->bstride, &W.state[0].stopped};
    const NnBlocks yb{W.ys4, W.ybox, D.nyb, nullptr}, pb{W.ps4, W.pbox, D.rows ? D.npb : 0, W.sb + D.K};
    int blocksA, blocksB;
    if (D.rows) { blocksA = cdiv(64 * D.npb, NN_ROW_SLOTS); blocksB = cdiv(64 * D.nyb, NN_ROW_SLOTS); }
    else { const NnGrid g = nn_grid(D.NP, D.NT, true, true, 4); blocksA = g.blocksA; blocksB = g.blocksB; }
    const int grid = 8 * 32 * wg_per_cu;
    auto xl = [&](int i, int nzz, bool wscope) {
        const int par = i & 1;
#define CREG_XCD_GO(R, WS) hipLaunchKernelGGL((k_nn_xcd<R, 1, 2, WS>), dim3(grid), dim3(R ? NN_ROWS_BLOCK : NN_BLOCK), 0, s, (const float*)W.pred4, D.NP, (const float*)W.y4, D.NT, \
                                              blocksA, blocksB, epi, yb, pb, P->bstride, nzz, ctl, par)
        if (D.rows) { if (wscope) CREG_XCD_GO(true, true); else CREG_XCD_GO(true, false); }
        else { if (wscope) CREG_XCD_GO(false, true); else CREG_XCD_GO(false, false); }
#undef CREG_XCD_GO
    };
    CREG_HIP(hipMemsetAsync(ctl, 0, 4 * 32 * 32, s));
    if (int r2 = timed([&](int i) { xl(i, nz, false); }, out + 18)) return r2;
    CREG_HIP(hipMemsetAsync(ctl, 0, 4 * 32 * 32, s));
    if (int r2 = timed([&](int i) { xl(i, nz, true); }, out + 19)) return r2;
    CREG_HIP(hipMemsetAsync(ctl, 0, 4 * 32 * 32, s));
    if (int r2 = timed([&](int i) { xl(i, 1, true); }, out + 20)) return r2;
    {   // identical outputs: loss partials, sign bits and scatter counters of every problem after ONE launch of either kind
        const size_t nl = (size_t)D.nbx + D.nby;
        std::vector<float> l0(nl * nz), l1(nl * nz); std::vector<int> s0((size_t)D.NP * nz), s1((size_t)D.NP * nz), c0((size_t)4 * D.NP * nz), c1((size_t)4 * D.NP * nz);
        auto grab = [&](std::vector<float>& l, std::vector<int>& sg, std::vector<int>& c) -> int {
            CREG_HIP(hipStreamSynchronize(s));
            for (int z = 0; z < nz; ++z) {
                const Ws Wz = ws_shift(W, (size_t)z * P->bstride);
                CREG_HIP(hipMemcpy(l.data() + nl * z, Wz.lossp_x, 4 * (size_t)D.nbx, hipMemcpyDeviceToHost));
                CREG_HIP(hipMemcpy(l.data() + nl * z + D.nbx, Wz.lossp_y, 4 * (size_t)D.nby, hipMemcpyDeviceToHost));
                CREG_HIP(hipMemcpy(sg.data() + (size_t)D.NP * z, Wz.sgn_x, 4 * (size_t)D.NP, hipMemcpyDeviceToHost));
                CREG_HIP(hipMemcpy(c.data() + (size_t)4 * D.NP * z, Wz.cnt4, 16 * (size_t)D.NP, hipMemcpyDeviceToHost));
            }
            return CREG_OK;
        };
        launch_head(P, 0, s); launch_nn(D, W, P->bstride, nz, s);
        if (int r2 = grab(l0, s0, c0)) return r2;
        launch_head(P, 0, s);
        CREG_HIP(hipMemsetAsync(ctl, 0, 4 * 32 * 32, s));
        xl(0, nz, true);
        if (int r2 = grab(l1, s1, c1)) return r2;
        out[21] = (memcmp(l0.data(), l1.data(), 4 * l0.size()) == 0 && s0 == s1 && c0 == c1) ? 1.0 : 0.0;
        int ch[32 * 32];
        CREG_HIP(hipMemcpy(ch, ctl, sizeof(ch), hipMemcpyDeviceToHost));
        for (int x = 0; x < 8; ++x) out[24 + x] = ch[32 * (8 + x)];
    }
    out[22] = 32 * wg_per_cu;
    CREG_LAUNCH_CHECK();
    if (D.H == 512) {   // ---- Stage 2 pre-check: k_l2
        const int l2grid = 8 * (D.H2 / 16);
        auto l2x = [&](int i, int nzz) { hipLaunchKernelGGL((k_l2_xcd<8>), dim3(l2grid), dim3(L2_THREADS), 0, s, D, W, 0, P->bstride, nzz, ctl, i & 1); };
        P->nz = nz;
        if (int r2 = timed([&](int) { launch_l2(P, 0, s); }, out + 32)) return r2;
        CREG_HIP(hipMemsetAsync(ctl, 0, 4 * 32 * 32, s));
        if (int r2 = timed([&](int i) { l2x(i, nz); }, out + 33)) return r2;
        P->nz = 1;
        if (int r2 = timed([&](int) { launch_l2(P, 0, s); }, out + 35)) return r2;
        CREG_HIP(hipMemsetAsync(ctl, 0, 4 * 32 * 32, s));
        if (int r2 = timed([&](int i) { l2x(i, 1); }, out + 36)) return r2;
        P->nz = nz;
        const size_t nh = (size_t)D.KP * D.H2;
        std::vector<float> h0(nh * nz), h1(nh * nz);
        launch_l2(P, 0, s);
        CREG_HIP(hipStreamSynchronize(s));
        for (int z = 0; z < nz; ++z) CREG_HIP(hipMemcpy(h0.data() + nh * z, ws_shift(W, (size_t)z * P->bstride).h2[0], 4 * nh, hipMemcpyDeviceToHost));
        for (int z = 0; z < nz; ++z) CREG_HIP(hipMemsetAsync(ws_shift(W, (size_t)z * P->bstride).h2[0], 0xff, 4 * (size_t)D.K * D.H2, s));
        CREG_HIP(hipMemsetAsync(ctl, 0, 4 * 32 * 32, s));
        l2x(0, nz);
        CREG_HIP(hipStreamSynchronize(s));
        for (int z = 0; z < nz; ++z) CREG_HIP(hipMemcpy(h1.data() + nh * z, ws_shift(W, (size_t)z * P->bstride).h2[0], 4 * nh, hipMemcpyDeviceToHost));
        bool same = true;
        for (int z = 0; z < nz && same; ++z) same = memcmp(h0.data() + nh * z, h1.data() + nh * z, 4 * (size_t)D.K * D.H2) == 0;
        out[34] = same ? 1.0 : 0.0;
        CREG_LAUNCH_CHECK();
    }
    P->nz = nz_keep;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(ctl); (void)hipFree(payload); (void)hipFree(ticks); (void)hipFree(errs);
    return CREG_OK;
}
#endif

extern "C" int creg_train_plan_info(const creg_train_plan* plan, creg_train_plan_info_t* info) {
    CREG_REQUIRE(plan && info, "creg_train_plan_info: null pointer");
    const Plan* P = (const Plan*)plan;
    info->pruned_target_search = P->D.rows ? 2 : (P->D.nyb > 0);
    info->pruned_predicted_search = P->D.npb > 0;
    info->graph_branches = P->branches;
    info->batch = P->B;
    info->epochs_per_graph = P->shape.use_graph ? P->graph_epochs : 0;
    info->nn_points_per_lane = P->D.ppl;
    info->nn_boxes_target = P->D.nbt;
    info->nn_boxes_predicted = P->D.nbp;
    info->chain_probe_us = P->probe_us;
    return CREG_OK;
}

extern "C" int creg_train_plan_destroy(creg_train_plan* plan) {
    Plan* P = (Plan*)plan;
    if (!P) return CREG_OK;
    if (P->gexec) (void)hipGraphExecDestroy(P->gexec);
    for (int i = 0; i < 8; ++i) {
        if (P->cst[i]) (void)hipStreamSynchronize(P->cst[i]);
        if (P->cexec[i]) (void)hipGraphExecDestroy(P->cexec[i]);
        if (P->cjoin[i]) (void)hipEventDestroy(P->cjoin[i]);
        if (P->cst[i]) (void)hipStreamDestroy(P->cst[i]);
    }
    if (P->cfork) (void)hipEventDestroy(P->cfork);
    delete P;
    return CREG_OK;
}
