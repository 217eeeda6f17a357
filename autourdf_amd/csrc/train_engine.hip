// train_engine.hip -- A1: the whole `train` loop of the reference (PointCloud/mlp_reg.py:17-152)
// as a device-resident plan.  One epoch = eight small launches, no host round trip:
//
//   k_l2      hidden layer(s) of the pose MLP            (model_utils.py:152-159 / :94-99)
//   k_head    output layer(s) + residual + pose assembly + calculate_pc (mlp_reg.py:62-94,155-170)
//   k_nn      L1 nearest neighbour both ways             (chamfer_distance, mlp_reg.py:96)
//   k_post    loss partial sums + sign scatter of the y->x term (integer atomics: exact)
//   k_ctrl    loss, best tracking (mlp_reg.py:102-111), ReduceLROnPlateau, Adam scalars, early stop
//   k_gradc   per-cluster reduction to dL/dR, dL/dt, backward through the pose head
//   k_bwd2    backward through the output / hidden layers to the encoder activation
//   k_dw      weight gradients fused with the Adam update (no gradient buffer), and -- for the
//             encoder rows -- the NEXT epoch's encoder activation from the just-updated weights
//
// Everything is fp32 like the reference; every reduction has a fixed order (bit-reproducible
// run to run).  State that the reference keeps in Python (min_loss, count, scheduler, lr) lives in
// a double-buffered device struct indexed by epoch parity.
#include <cfloat>
#include <cmath>
#include <cstring>
#include <type_traits>
#include <vector>
#include "creg_common.h"
#include "nn_l1.h"

namespace creg {

struct Dims {
    int rot, K, IN, H, HA, HB, H2, OA, OB, NP, NT, epochs;
    float slope;
    // flat parameter offsets
    int oW1, ob1, oW2, ob2, oW3A, ob3A, oW3B, ob3B, NPAR;
    int OC;        // o-chunks of k_bwd2
    int nblk_post; // blocks of k_post (= loss partial count)
};

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
};

struct Ws {         // device pointers into the caller's workspace
    float *P, *AM, *AV;
    float *pose_in, *enc, *x1[2], *h2, *head_save, *m_in, *m2, *gm2;
    float4 *pts4, *y4, *pred4;
    float *dist_x, *dist_y;
    int *idx_x, *idx_y;
    int4* cnt4;
    float *lossp_x, *lossp_y;
    float *g_out, *g_h2, *gx1_part;
    TrainState* state;
    float *best_m, *best_pred, *loss_hist, *lr_hist, *result;
    int* off;
    Hyper* hyper;
};

__device__ __forceinline__ float act_f(float v, float slope) { return v > 0.f ? v : v * slope; }
__device__ __forceinline__ float act_grad(float post, float slope) { return post > 0.f ? 1.f : slope; }

__device__ __forceinline__ int seg_of(const int* __restrict__ off, int k, int n) {
    int lo = 0, hi = k;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (off[mid] <= n) lo = mid; else hi = mid; }
    return lo;
}

// sum_i w[i] * a[i], i < n, lanes strided, fixed order (per-lane ascending chunks, then butterfly)
__device__ __forceinline__ float wave_dot(const float* __restrict__ w, const float* __restrict__ a, int n, int lane) {
    float s = 0.f;
    for (int i = lane; i < n; i += 64) s = fmaf(w[i], a[i], s);
    return wave_sum(s);
}

// ------------------------------------------------------------------------------------------ prep
__global__ __launch_bounds__(256) void k_prep(Dims D, Ws W, Hyper hy, const float* __restrict__ m,
                                              const float* __restrict__ y, const float* __restrict__ pts) {
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
    if (blockIdx.x == 0) {
        for (int r = threadIdx.x; r < D.K; r += 256) {
            const float* M = m + 16 * r;
            for (int i = 0; i < 16; ++i) W.m_in[16 * r + i] = M[i];
            const float R[9] = {M[0], M[1], M[2], M[4], M[5], M[6], M[8], M[9], M[10]};
            const float tr[3] = {M[3], M[7], M[11]};
            float in[8];
            int nin;
            if (D.rot == 0) {                     // cat([t, matrix_to_quaternion(R)])  mlp_reg.py:65-66
                float q[4];
                matrix_to_quat(R, q);
                in[0] = tr[0]; in[1] = tr[1]; in[2] = tr[2]; in[3] = q[0]; in[4] = q[1]; in[5] = q[2]; in[6] = q[3]; in[7] = 0.f;
                nin = 7;
            } else {                              // transform_to_dualquat  mlp_reg.py:80
                se3_to_dq(R, tr, in, FLT_EPSILON);
                nin = 8;
            }
            for (int i = 0; i < 8; ++i) W.pose_in[8 * r + i] = in[i];
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
            W.state[0] = s; W.state[1] = s;
        }
    }
}

// ------------------------------------------------------------------------------------------ layer 1 (epoch 0 only)
__global__ __launch_bounds__(256) void k_l1(Dims D, Ws W, int par) {
    const int lane = threadIdx.x & 63;
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= D.H) return;
    const float* w = W.P + D.oW1 + (size_t)o * D.IN;
    const float b = W.P[D.ob1 + o];
    for (int r = 0; r < D.K; ++r) {
        const float v = wave_dot(w, W.enc + (size_t)r * D.IN, D.IN, lane) + b;
        if (lane == 0) W.x1[par][(size_t)r * D.H + o] = act_f(v, D.slope);
    }
}

// ------------------------------------------------------------------------------------------ layer 2
__global__ __launch_bounds__(256) void k_l2(Dims D, Ws W, int par) {
    const int lane = threadIdx.x & 63;
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= D.H2) return;
    const float* w = W.P + D.oW2 + (size_t)o * D.H;
    const float b = W.P[D.ob2 + o];
    // weights of this row stay in registers (H/64 per lane), activations stream from L2
    float wr[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) wr[c] = (c * 64 + lane < D.H) ? w[c * 64 + lane] : 0.f;
    const int nc = D.H / 64;
    for (int r = 0; r < D.K; ++r) {
        const float* a = W.x1[par] + (size_t)r * D.H;
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < 16; ++c) if (c < nc) s = fmaf(wr[c], a[c * 64 + lane], s);
        s = wave_sum(s) + b;
        if (lane == 0) W.h2[(size_t)r * D.H2 + o] = act_f(s, D.slope);
    }
}

// ------------------------------------------------------------------------------------------ head + transform
// Every block recomputes the K-row output layer (tiny) so the transformed cloud can follow in the
// same launch; block 0 publishes m2 / head_save.
__global__ __launch_bounds__(1024) void k_head(Dims D, Ws W) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* outs = (float*)smem;              // [K][8]  raw output-layer values
    float* m2s = outs + 8 * D.K;             // [K][12] rows of [R|t]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int NO = D.OA + D.OB;
    for (int id = wave; id < D.K * NO; id += 16) {
        const int r = id / NO, o = id % NO;
        float v;
        if (o < D.OA)
            v = wave_dot(W.P + D.oW3A + (size_t)o * D.HA, W.h2 + (size_t)r * D.H2, D.HA, lane) + W.P[D.ob3A + o];
        else
            v = wave_dot(W.P + D.oW3B + (size_t)(o - D.OA) * D.HB, W.h2 + (size_t)r * D.H2 + D.HA, D.HB, lane) +
                W.P[D.ob3B + (o - D.OA)];
        if (lane == 0) outs[8 * r + o] = v;
    }
    __syncthreads();
    for (int r = threadIdx.x; r < D.K; r += 1024) {
        const float* in = W.pose_in + 8 * r;
        float R[9], t[3], save[16];
        for (int i = 0; i < 16; ++i) save[i] = 0.f;
        if (D.rot == 0) {
            // xyz + orig[:, :3] ; normalize(q + orig[:, 3:])   model_utils.py:159
            for (int i = 0; i < 3; ++i) t[i] = outs[8 * r + i] + in[i];
            float v[4], n2 = 0.f;
            for (int i = 0; i < 4; ++i) { v[i] = outs[8 * r + 3 + i] + in[3 + i]; n2 = fmaf(v[i], v[i], n2); }
            const float nrm = sqrtf(n2), den = fmaxf(nrm, 1e-12f);
            float u[4];
            for (int i = 0; i < 4; ++i) u[i] = v[i] / den;
            quat_to_matrix(u, R);
            for (int i = 0; i < 4; ++i) save[i] = u[i];
            save[4] = nrm;
        } else {
            float dq[8];
            for (int i = 0; i < 8; ++i) dq[i] = outs[8 * r + i] + in[i];     // x + orig  model_utils.py:99
            dq_to_se3(dq, R, t);
            for (int i = 0; i < 8; ++i) save[i] = dq[i];
        }
        for (int a = 0; a < 3; ++a) { m2s[12 * r + 4 * a] = R[3 * a]; m2s[12 * r + 4 * a + 1] = R[3 * a + 1];
                                      m2s[12 * r + 4 * a + 2] = R[3 * a + 2]; m2s[12 * r + 4 * a + 3] = t[a]; }
        if (blockIdx.x == 0) {
            for (int i = 0; i < 12; ++i) W.m2[16 * r + i] = m2s[12 * r + i];
            W.m2[16 * r + 12] = 0.f; W.m2[16 * r + 13] = 0.f; W.m2[16 * r + 14] = 0.f; W.m2[16 * r + 15] = 1.f;
            for (int i = 0; i < 16; ++i) W.head_save[16 * r + i] = save[i];
        }
    }
    __syncthreads();
    const int n = blockIdx.x * 1024 + threadIdx.x;
    if (n < D.NP) {
        const float4 p = W.pts4[n];
        const float* T = m2s + 12 * __float_as_int(p.w);
        float o[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) o[a] = fmaf(p.z, T[4 * a + 2], fmaf(p.y, T[4 * a + 1], p.x * T[4 * a])) + T[4 * a + 3];
        W.pred4[n] = make_float4(o[0], o[1], o[2], 0.f);
        W.cnt4[n] = make_int4(0, 0, 0, 0);
    }
}

// ------------------------------------------------------------------------------------------ post-NN
__global__ __launch_bounds__(256) void k_post(Dims D, Ws W) {
    __shared__ float sc[4];
    const int t = blockIdx.x * 256 + threadIdx.x;
    const float a = t < D.NP ? W.dist_x[t] : 0.f;
    const float b = t < D.NT ? W.dist_y[t] : 0.f;
    if (t < D.NT) {
        const int i = W.idx_y[t];
        const float4 yv = W.y4[t], xv = W.pred4[i];
        // knn(p1=y, p2=x) backward: grad_p2 -= g * sign, sign = (p1 > p2 ? 1 : -1)
        atomicAdd(&W.cnt4[i].x, (yv.x > xv.x) ? -1 : 1);
        atomicAdd(&W.cnt4[i].y, (yv.y > xv.y) ? -1 : 1);
        atomicAdd(&W.cnt4[i].z, (yv.z > xv.z) ? -1 : 1);
    }
    const float sa = block_sum<float, 256>(a, sc);
    const float sb = block_sum<float, 256>(b, sc);
    if (threadIdx.x == 0) { W.lossp_x[blockIdx.x] = sa; W.lossp_y[blockIdx.x] = sb; }
}

// ------------------------------------------------------------------------------------------ control
// Every block derives the same loss / decision; block 0 advances the state, all blocks copy their
// slice of the prediction when the loss improved.
__global__ __launch_bounds__(256) void k_ctrl(Dims D, Ws W, int epoch) {
    __shared__ float sc[4];
    __shared__ float s_loss;
    const TrainState S = W.state[epoch & 1];
    if (S.stopped) {
        if (blockIdx.x == 0 && threadIdx.x == 0) W.state[(epoch + 1) & 1] = S;
        return;
    }
    float a = 0.f, b = 0.f;
    for (int i = threadIdx.x; i < D.nblk_post; i += 256) { a += W.lossp_x[i]; b += W.lossp_y[i]; }
    a = block_sum<float, 256>(a, sc);
    b = block_sum<float, 256>(b, sc);
    if (threadIdx.x == 0) s_loss = a / (float)D.NP + b / (float)D.NT;
    __syncthreads();
    const float loss = s_loss;
    const bool improved = loss < S.min_loss;
    const int e = S.epochs_run;          // device-side epoch counter (the launch argument only carries parity)
    if (improved) {
        for (int n = blockIdx.x * 256 + threadIdx.x; n < D.NP; n += gridDim.x * 256) {
            const float4 p = W.pred4[n];
            W.best_pred[3 * (size_t)n] = p.x; W.best_pred[3 * (size_t)n + 1] = p.y; W.best_pred[3 * (size_t)n + 2] = p.z;
        }
        if (blockIdx.x == 0)
            for (int i = threadIdx.x; i < 16 * D.K; i += 256) W.best_m[i] = W.m2[i];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        TrainState N = S;
        N.last_loss = loss;
        N.epochs_run = e + 1;
        W.loss_hist[e] = loss;
        W.lr_hist[e] = (float)S.lr;
        if (improved) { N.min_loss = loss; N.count = 0; N.best_epoch = e; }
        else { N.count = S.count + 1; if (N.count > W.hyper->stop) N.stopped = 1; }   // mlp_reg.py:107-111
        if (!N.stopped) {
            // optimizer.step() of this epoch uses S.lr (torch.optim.Adam, betas (0.9, 0.999), eps 1e-8)
            N.step = S.step + 1;
            const double bc1 = 1.0 - pow(0.9, (double)N.step), bc2 = 1.0 - pow(0.999, (double)N.step);
            N.step_size = (float)(S.lr / bc1);
            N.bc2_sqrt = (float)sqrt(bc2);
            // scheduler.step(loss): ReduceLROnPlateau(mode='min', threshold 1e-4 rel, cooldown 0, min_lr 0, eps 1e-8)
            const double cur = (double)loss;
            if (cur < S.sched_best * (1.0 - 1e-4)) { N.sched_best = cur; N.sched_bad = 0; }
            else N.sched_bad = S.sched_bad + 1;
            if (N.sched_bad > W.hyper->patience) {
                const double nl = fmax(S.lr * (double)W.hyper->factor, 0.0);
                if (S.lr - nl > 1e-8) N.lr = nl;
                N.sched_bad = 0;
            }
        }
        W.state[(epoch + 1) & 1] = N;
        W.result[0] = N.min_loss; W.result[1] = (float)N.epochs_run; W.result[2] = (float)N.lr;
        W.result[3] = (float)N.best_epoch;
    }
}

// ------------------------------------------------------------------------------------------ cluster grads + head backward
__global__ __launch_bounds__(256) void k_gradc(Dims D, Ws W, int epoch) {
    __shared__ float sc[4];
    __shared__ float red[12];
    if (W.state[(epoch + 1) & 1].stopped) return;
    const int k = blockIdx.x, b = W.off[k], e = W.off[k + 1];
    const float gx = 1.0f / (float)D.NP, gy = 1.0f / (float)D.NT;
    float acc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = 0.f;
    for (int n = b + threadIdx.x; n < e; n += 256) {
        const float4 xv = W.pred4[n], yv = W.y4[W.idx_x[n]], p = W.pts4[n];
        const int4 c = W.cnt4[n];
        const float g[3] = {((xv.x > yv.x) ? gx : -gx) + gy * (float)c.x,
                            ((xv.y > yv.y) ? gx : -gx) + gy * (float)c.y,
                            ((xv.z > yv.z) ? gx : -gx) + gy * (float)c.z};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            acc[4 * a] = fmaf(g[a], p.x, acc[4 * a]); acc[4 * a + 1] = fmaf(g[a], p.y, acc[4 * a + 1]);
            acc[4 * a + 2] = fmaf(g[a], p.z, acc[4 * a + 2]); acc[4 * a + 3] += g[a];
        }
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        const float r = block_sum<float, 256>(acc[i], sc);
        if (threadIdx.x == 0) red[i] = r;
    }
    if (threadIdx.x != 0) return;
    for (int i = 0; i < 16; ++i) W.gm2[16 * k + i] = i < 12 ? red[i] : 0.f;
    const float G[9] = {red[0], red[1], red[2], red[4], red[5], red[6], red[8], red[9], red[10]};
    const float gt[3] = {red[3], red[7], red[11]};
    float* go = W.g_out + 16 * k;          // [0..2] branch A, [4..11] branch B
    const float* sv = W.head_save + 16 * k;
    if (D.rot == 0) {
        go[0] = gt[0]; go[1] = gt[1]; go[2] = gt[2];
        float gu[4];
        quat_to_matrix_vjp(sv, G, gu);
        const float nrm = sv[4];
        if (nrm > 1e-12f) {
            const float dot = sv[0] * gu[0] + sv[1] * gu[1] + sv[2] * gu[2] + sv[3] * gu[3];
            for (int i = 0; i < 4; ++i) go[4 + i] = (gu[i] - sv[i] * dot) / nrm;
        } else {
            for (int i = 0; i < 4; ++i) go[4 + i] = gu[i] / 1e-12f;
        }
    } else {
        float gdq[8];
        dq_to_se3_vjp(sv, G, gt, gdq);
        for (int i = 0; i < 8; ++i) go[4 + i] = gdq[i];
    }
}

// ------------------------------------------------------------------------------------------ backward to x1
// grid (H/256, OC), 4 waves: wave = 64 columns of x1, block = one chunk of hidden rows.
constexpr int RT = 32;      // rows of K handled per register tile
__global__ __launch_bounds__(256) void k_bwd2(Dims D, Ws W, int epoch) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* gs = (float*)smem;                 // [rows][K]
    if (W.state[(epoch + 1) & 1].stopped) return;
    const int rows = D.H2 / D.OC, o0 = blockIdx.y * rows;
    for (int id = threadIdx.x; id < rows * D.K; id += 256) {
        const int ol = id / D.K, r = id % D.K, o = o0 + ol;
        const float* go = W.g_out + 16 * r;
        float s = 0.f;
        if (o < D.HA) { for (int j = 0; j < D.OA; ++j) s = fmaf(go[j], W.P[D.oW3A + (size_t)j * D.HA + o], s); }
        else { for (int j = 0; j < D.OB; ++j) s = fmaf(go[4 + j], W.P[D.oW3B + (size_t)j * D.HB + (o - D.HA)], s); }
        s *= act_grad(W.h2[(size_t)r * D.H2 + o], D.slope);
        gs[ol * D.K + r] = s;
        if (blockIdx.x == 0) W.g_h2[(size_t)r * D.H2 + o] = s;
    }
    __syncthreads();
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= D.H) return;
    for (int r0 = 0; r0 < D.K; r0 += RT) {
        float acc[RT];
#pragma unroll
        for (int r = 0; r < RT; ++r) acc[r] = 0.f;
        for (int ol = 0; ol < rows; ++ol) {
            const float w = W.P[D.oW2 + (size_t)(o0 + ol) * D.H + col];
            const float* g = gs + ol * D.K + r0;
#pragma unroll
            for (int r = 0; r < RT; ++r) if (r0 + r < D.K) acc[r] = fmaf(g[r], w, acc[r]);
        }
#pragma unroll
        for (int r = 0; r < RT; ++r)
            if (r0 + r < D.K) W.gx1_part[((size_t)blockIdx.y * D.K + r0 + r) * D.H + col] = acc[r];
    }
}

// ------------------------------------------------------------------------------------------ dW + Adam (+ next x1)
__device__ __forceinline__ void adam_update(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                            float g, float step_size, float bc2_sqrt) {
    // torch single-tensor Adam: exp_avg.lerp_(g, 1-b1); exp_avg_sq.mul_(b2).addcmul_(g, g, 1-b2);
    // denom = sqrt(v)/sqrt(bc2) + eps; p.addcdiv_(exp_avg, denom, value=-step_size)
    const float b1w = (float)(1.0 - 0.9), b2 = 0.999f, b2w = (float)(1.0 - 0.999);
    float mm = *m, vv = *v;
    mm = mm + b1w * (g - mm);
    vv = vv * b2 + (b2w * g) * g;
    const float denom = sqrtf(vv) / bc2_sqrt + 1e-8f;
    *p = *p + (-step_size * mm) / denom;
    *m = mm; *v = vv;
}

__global__ __launch_bounds__(256) void k_dw(Dims D, Ws W, int epoch) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const TrainState S = W.state[(epoch + 1) & 1];
    if (S.stopped) return;                          // block-uniform
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    float* g = (float*)smem + wib * D.K;            // per-wave gradient column g[r]
    const int row = blockIdx.x * 4 + wib;
    const int nrows = D.H2 + D.OA + D.OB + D.H;
    const bool active = row < nrows;
    const int par = epoch & 1;
    // decode the row: 0: W2, 1: W3A, 2: W3B, 3: W1
    int oW = 0, ob = 0, o = 0, n_in = 0, astride = 0, kind = 0; const float* act = nullptr;
    if (row < D.H2) { kind = 0; o = row; oW = D.oW2 + o * D.H; ob = D.ob2 + o; n_in = D.H; act = W.x1[par]; astride = D.H; }
    else if (row < D.H2 + D.OA) { kind = 1; o = row - D.H2; oW = D.oW3A + o * D.HA; ob = D.ob3A + o; n_in = D.HA; act = W.h2; astride = D.H2; }
    else if (row < D.H2 + D.OA + D.OB) { kind = 2; o = row - D.H2 - D.OA; oW = D.oW3B + o * D.HB; ob = D.ob3B + o; n_in = D.HB; act = W.h2 + D.HA; astride = D.H2; }
    else if (active) { kind = 3; o = row - D.H2 - D.OA - D.OB; oW = D.oW1 + o * D.IN; ob = D.ob1 + o; n_in = D.IN; act = W.enc; astride = D.IN; }
    // gradient of this output unit for every pose row
    if (active) {
        for (int r = lane; r < D.K; r += 64) {
            float v;
            if (kind == 0) v = W.g_h2[(size_t)r * D.H2 + o];
            else if (kind == 1) v = W.g_out[16 * r + o];
            else if (kind == 2) v = W.g_out[16 * r + 4 + o];
            else {
                float s = 0.f;
                for (int c = 0; c < D.OC; ++c) s += W.gx1_part[((size_t)c * D.K + r) * D.H + o];
                v = s * act_grad(W.x1[par][(size_t)r * D.H + o], D.slope);
            }
            g[r] = v;
        }
    }
    __syncthreads();
    if (!active) return;
    // weights: lanes over the input index; the (single) chunk of an encoder row stays in a register
    float wnew = 0.f;
    for (int i = lane; i < n_in; i += 64) {
        float s = 0.f;
        for (int r = 0; r < D.K; ++r) s = fmaf(g[r], act[(size_t)r * astride + i], s);
        adam_update(W.P + oW + i, W.AM + oW + i, W.AV + oW + i, s, S.step_size, S.bc2_sqrt);
        wnew = W.P[oW + i];
    }
    float bnew = 0.f;
    if (lane == 0) {
        float s = 0.f;
        for (int r = 0; r < D.K; ++r) s += g[r];
        adam_update(W.P + ob, W.AM + ob, W.AV + ob, s, S.step_size, S.bc2_sqrt);
        bnew = W.P[ob];
    }
    if (kind == 3) {
        // next epoch's encoder activation from the updated row held in registers (IN <= 64: one
        // weight per lane).  The MLP input is the same every epoch: m.clone() of the same m
        // (mlp_reg.py:62), so only the weights moved.
        bnew = __shfl(bnew, 0, 64);
        for (int r = 0; r < D.K; ++r) {
            float v = (lane < D.IN) ? wnew * W.enc[(size_t)r * D.IN + lane] : 0.f;
            v = wave_sum(v) + bnew;
            if (lane == 0) W.x1[par ^ 1][(size_t)r * D.H + o] = act_f(v, D.slope);
        }
    }
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
    int smem_head, smem_bwd2, smem_dw;
};

static bool make_dims(const creg_train_shape* s, Dims* D) {
    if (!s || (s->rot != 0 && s->rot != 1) || s->k < 1 || s->k > 4096 || s->hidden < 64 || s->hidden > 1024 ||
        s->hidden % 64 || s->epochs < 1 || s->n_pred < 1 || s->n_tgt < 1 || s->n_pred >= (1ll << 31) ||
        s->n_tgt >= (1ll << 31))
        return false;
    memset(D, 0, sizeof(*D));
    D->rot = s->rot; D->K = s->k; D->H = s->hidden; D->NP = (int)s->n_pred; D->NT = (int)s->n_tgt; D->epochs = s->epochs;
    if (s->rot == 0) { D->IN = 56; D->HA = D->H / 2; D->HB = D->H; D->OA = 3; D->OB = 4; D->slope = 0.01f; }
    else { D->IN = 64; D->HA = 0; D->HB = D->H; D->OA = 0; D->OB = 8; D->slope = 0.f; }
    D->H2 = D->HA + D->HB;
    int o = 0;
    D->oW1 = o; o += D->H * D->IN; D->ob1 = o; o += D->H;
    D->oW2 = o; o += D->H2 * D->H; D->ob2 = o; o += D->H2;
    D->oW3A = o; o += D->OA * D->HA; D->ob3A = o; o += D->OA;
    D->oW3B = o; o += D->OB * D->HB; D->ob3B = o; o += D->OB;
    D->NPAR = o;
    D->OC = 16;
    while (D->H2 % D->OC) D->OC /= 2;
    const int mx = D->NP > D->NT ? D->NP : D->NT;
    D->nblk_post = (mx + 255) / 256;
    return true;
}

static size_t carve(const Dims& D, char* base, Ws* W) {
    size_t o = 0;
    auto take = [&](size_t bytes) -> char* { char* r = base ? base + o : nullptr; o = align_up(o + bytes, 256); return r; };
    const size_t f = sizeof(float);
    Ws w;
    w.P = (float*)take(f * D.NPAR); w.AM = (float*)take(f * D.NPAR); w.AV = (float*)take(f * D.NPAR);
    w.pose_in = (float*)take(f * 8 * D.K); w.enc = (float*)take(f * D.K * D.IN);
    w.x1[0] = (float*)take(f * D.K * D.H); w.x1[1] = (float*)take(f * D.K * D.H);
    w.h2 = (float*)take(f * D.K * D.H2); w.head_save = (float*)take(f * 16 * D.K);
    w.m_in = (float*)take(f * 16 * D.K); w.m2 = (float*)take(f * 16 * D.K); w.gm2 = (float*)take(f * 16 * D.K);
    w.pts4 = (float4*)take(sizeof(float4) * D.NP); w.y4 = (float4*)take(sizeof(float4) * D.NT);
    w.pred4 = (float4*)take(sizeof(float4) * D.NP);
    w.dist_x = (float*)take(f * D.NP); w.dist_y = (float*)take(f * D.NT);
    w.idx_x = (int*)take(sizeof(int) * D.NP); w.idx_y = (int*)take(sizeof(int) * D.NT);
    w.cnt4 = (int4*)take(sizeof(int4) * D.NP);
    w.lossp_x = (float*)take(f * D.nblk_post); w.lossp_y = (float*)take(f * D.nblk_post);
    w.g_out = (float*)take(f * 16 * D.K); w.g_h2 = (float*)take(f * D.K * D.H2);
    w.gx1_part = (float*)take(f * (size_t)D.OC * D.K * D.H);
    w.state = (TrainState*)take(sizeof(TrainState) * 2);
    w.best_m = (float*)take(f * 16 * D.K); w.best_pred = (float*)take(f * 3 * D.NP);
    w.loss_hist = (float*)take(f * D.epochs); w.lr_hist = (float*)take(f * D.epochs); w.result = (float*)take(f * 4);
    w.off = (int*)take(sizeof(int) * (D.K + 1)); w.hyper = (Hyper*)take(sizeof(Hyper));
    if (W) *W = w;
    return o;
}

// `ev` (optional): 9 events recorded before kernel 0 and after each of the 8 kernels.
static void enqueue_epoch(Plan* P, int epoch, hipStream_t s, hipEvent_t* ev = nullptr) {
    const Dims& D = P->D; const Ws& W = P->W;
    const int par = epoch & 1;
    auto mark = [&](int i) { if (ev) (void)hipEventRecord(ev[i], s); };
    mark(0);
    hipLaunchKernelGGL(k_l2, dim3(cdiv(D.H2, 4)), dim3(256), 0, s, D, W, par); mark(1);
    hipLaunchKernelGGL(k_head, dim3(cdiv(D.NP, 1024)), dim3(1024), P->smem_head, s, D, W); mark(2);
    launch_nn_l1_bidir<int>((const float*)W.pred4, D.NP, 4, (const float*)W.y4, D.NT, 4, W.dist_x, W.idx_x, W.dist_y,
                            W.idx_y, s); mark(3);
    hipLaunchKernelGGL(k_post, dim3(D.nblk_post), dim3(256), 0, s, D, W); mark(4);
    hipLaunchKernelGGL(k_ctrl, dim3(cdiv(D.NP, 1024)), dim3(256), 0, s, D, W, epoch); mark(5);
    hipLaunchKernelGGL(k_gradc, dim3(D.K), dim3(256), 0, s, D, W, epoch); mark(6);
    hipLaunchKernelGGL(k_bwd2, dim3(cdiv(D.H, 256), D.OC), dim3(256), P->smem_bwd2, s, D, W, epoch); mark(7);
    hipLaunchKernelGGL(k_dw, dim3(cdiv(D.H2 + D.OA + D.OB + D.H, 4)), dim3(256), P->smem_dw, s, D, W, epoch); mark(8);
}

struct ParamMap { int off, count; };
static int param_map(const Dims& D, ParamMap* pm) {
    if (D.rot == 0) {
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

static int stage_inputs(Plan* P, const creg_train_args* a, hipStream_t s) {
    const Dims& D = P->D; const Ws& W = P->W;
    ParamMap pm[10];
    const int np = param_map(D, pm);
    for (int i = 0; i < np; ++i) {
        CREG_REQUIRE(a->params[i], "creg_train: params[%d] is null", i);
        CREG_HIP(hipMemcpyAsync(W.P + pm[i].off, a->params[i], sizeof(float) * pm[i].count, hipMemcpyDeviceToDevice, s));
    }
    CREG_HIP(hipMemcpyAsync(W.off, a->seg_offsets, sizeof(int) * (D.K + 1), hipMemcpyDeviceToDevice, s));
    Hyper h = {a->lr, a->sched_factor, a->sched_patience, a->stop};
    const int mx = D.NP > D.NT ? D.NP : D.NT;
    int blocks = cdiv(mx > D.NPAR ? mx : D.NPAR, 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_prep, dim3(blocks), dim3(256), 0, s, D, W, h, a->m, a->y, a->local_pts);
    hipLaunchKernelGGL(k_l1, dim3(cdiv(D.H, 4)), dim3(256), 0, s, D, W, 0);
    CREG_LAUNCH_CHECK();
    return CREG_OK;
}

}  // namespace creg
using namespace creg;

extern "C" size_t creg_train_workspace_bytes(const creg_train_shape* shape) {
    Dims D;
    if (!make_dims(shape, &D)) return 0;
    return carve(D, nullptr, nullptr);
}

extern "C" int creg_train_plan_create(const creg_train_shape* shape, void* workspace, size_t workspace_bytes,
                                      creg_train_plan** plan) {
    Dims D;
    CREG_REQUIRE(plan && workspace, "creg_train_plan_create: null pointer");
    CREG_REQUIRE(make_dims(shape, &D), "creg_train_plan_create: unsupported shape (rot in {0,1}, 64 <= hidden <= 1024 multiple of 64, sizes >= 1)");
    CREG_REQUIRE(((uintptr_t)workspace & 255) == 0, "creg_train_plan_create: workspace must be 256-byte aligned");
    const size_t need = carve(D, nullptr, nullptr);
    CREG_REQUIRE(workspace_bytes >= need, "creg_train_plan_create: workspace too small (%zu < %zu)", workspace_bytes, need);
    Plan* P = new Plan();
    P->shape = *shape; P->D = D; P->base = (char*)workspace; P->bytes = workspace_bytes;
    carve(D, P->base, &P->W);
    P->gexec = nullptr; P->graph_ready = false;
    P->smem_head = (int)(sizeof(float) * (8 + 12) * D.K);
    P->smem_bwd2 = (int)(sizeof(float) * (D.H2 / D.OC) * D.K);
    P->smem_dw = (int)(sizeof(float) * 4 * D.K);
    if (P->smem_head > 65536)
        CREG_HIP(hipFuncSetAttribute((const void*)k_head, hipFuncAttributeMaxDynamicSharedMemorySize, P->smem_head));
    if (P->smem_bwd2 > 65536)
        CREG_HIP(hipFuncSetAttribute((const void*)k_bwd2, hipFuncAttributeMaxDynamicSharedMemorySize, P->smem_bwd2));
    *plan = (creg_train_plan*)P;
    return CREG_OK;
}

extern "C" int creg_train_plan_run(creg_train_plan* plan, const creg_train_args* a, creg_stream_t stream) {
    Plan* P = (Plan*)plan;
    CREG_REQUIRE(P && a && a->m && a->y && a->local_pts && a->seg_offsets && a->params && a->best_m && a->best_pred &&
                     a->result, "creg_train_plan_run: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const Dims& D = P->D; const Ws& W = P->W;
    int rc = stage_inputs(P, a, s);
    if (rc) return rc;
    int e = 0;
    if (P->shape.use_graph && D.epochs >= 2) {
        if (!P->graph_ready) {
            // capture one even + one odd epoch: the epoch number enters the kernels only through its
            // parity (the history index comes from the device-side counter state.epochs_run).
            // captured on a private stream (torch's current stream is usually the null stream, which
            // cannot be captured); the instantiated graph is then launched on the caller's stream.
            hipGraph_t g;
            hipStream_t cs;
            CREG_HIP(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
            CREG_HIP(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
            enqueue_epoch(P, 0, cs);
            enqueue_epoch(P, 1, cs);
            CREG_HIP(hipStreamEndCapture(cs, &g));
            CREG_HIP(hipGraphInstantiate(&P->gexec, g, nullptr, nullptr, 0));
            CREG_HIP(hipGraphDestroy(g));
            CREG_HIP(hipStreamDestroy(cs));
            P->graph_ready = true;
        }
        for (; e + 2 <= D.epochs; e += 2) CREG_HIP(hipGraphLaunch(P->gexec, s));
    }
    for (; e < D.epochs; ++e) enqueue_epoch(P, e, s);
    CREG_LAUNCH_CHECK();
    // results out, parameters back into the caller's tensors
    ParamMap pm[10];
    const int np = param_map(D, pm);
    for (int i = 0; i < np; ++i)
        CREG_HIP(hipMemcpyAsync(a->params[i], W.P + pm[i].off, sizeof(float) * pm[i].count, hipMemcpyDeviceToDevice, s));
    CREG_HIP(hipMemcpyAsync(a->best_m, W.best_m, sizeof(float) * 16 * D.K, hipMemcpyDeviceToDevice, s));
    CREG_HIP(hipMemcpyAsync(a->best_pred, W.best_pred, sizeof(float) * 3 * D.NP, hipMemcpyDeviceToDevice, s));
    CREG_HIP(hipMemcpyAsync(a->result, W.result, sizeof(float) * 4, hipMemcpyDeviceToDevice, s));
    if (a->loss_hist) CREG_HIP(hipMemcpyAsync(a->loss_hist, W.loss_hist, sizeof(float) * D.epochs, hipMemcpyDeviceToDevice, s));
    if (a->lr_hist) CREG_HIP(hipMemcpyAsync(a->lr_hist, W.lr_hist, sizeof(float) * D.epochs, hipMemcpyDeviceToDevice, s));
    return CREG_OK;
}

extern "C" int creg_train_plan_probe(creg_train_plan* plan, const creg_train_args* a, float* m2, float* pred,
                                     float* loss, float* grad_m2, creg_stream_t stream) {
    Plan* P = (Plan*)plan;
    CREG_REQUIRE(P && a && a->m && a->y && a->local_pts && a->seg_offsets && a->params, "creg_train_plan_probe: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const Dims& D = P->D; const Ws& W = P->W;
    int rc = stage_inputs(P, a, s);
    if (rc) return rc;
    hipLaunchKernelGGL(k_l2, dim3(cdiv(D.H2, 4)), dim3(256), 0, s, D, W, 0);
    hipLaunchKernelGGL(k_head, dim3(cdiv(D.NP, 1024)), dim3(1024), P->smem_head, s, D, W);
    launch_nn_l1_bidir<int>((const float*)W.pred4, D.NP, 4, (const float*)W.y4, D.NT, 4, W.dist_x, W.idx_x, W.dist_y, W.idx_y, s);
    hipLaunchKernelGGL(k_post, dim3(D.nblk_post), dim3(256), 0, s, D, W);
    hipLaunchKernelGGL(k_ctrl, dim3(cdiv(D.NP, 1024)), dim3(256), 0, s, D, W, 0);
    hipLaunchKernelGGL(k_gradc, dim3(D.K), dim3(256), 0, s, D, W, 0);
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
    int rc = stage_inputs(P, a, s);
    if (rc) return rc;
    std::vector<hipEvent_t> ev((size_t)9 * n_epochs);
    for (auto& e : ev) CREG_HIP(hipEventCreate(&e));
    for (int e = 0; e < n_epochs; ++e) enqueue_epoch(P, e, s, ev.data() + 9 * e);
    CREG_LAUNCH_CHECK();
    CREG_HIP(hipStreamSynchronize(s));
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int e = 0; e < n_epochs; ++e)
        for (int k = 0; k < 8; ++k) {
            float ms = 0.f;
            CREG_HIP(hipEventElapsedTime(&ms, ev[9 * e + k], ev[9 * e + k + 1]));
            acc[k] += ms;
        }
    for (int k = 0; k < 8; ++k) us_out[k] = (float)(acc[k] * 1000.0 / n_epochs);
    for (auto& e : ev) (void)hipEventDestroy(e);
    return CREG_OK;
}

extern "C" int creg_train_plan_destroy(creg_train_plan* plan) {
    Plan* P = (Plan*)plan;
    if (!P) return CREG_OK;
    if (P->gexec) (void)hipGraphExecDestroy(P->gexec);
    delete P;
    return CREG_OK;
}
