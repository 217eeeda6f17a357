// nn_l1.h -- the K=1 L1 nearest-neighbour kernel (K1), shared by the standalone C-ABI entry points
// (nn_l1.hip) and the fused train plan (train_engine.hip).
//
// Mapping (N=4096: 256 workgroups = one per CU, 8 waves each):
//   * a workgroup stages the target cloud into LDS once (float4 per point, 64 KB per 4096 points,
//     chunked for larger clouds) -- without that every wave streamed the whole cloud from L2
//     (128 MB per launch at N=4096, the v1 kernel was L2-bandwidth bound);
//   * a wave owns QW consecutive queries (wave-uniform -> SGPRs); its lanes stride over the staged
//     targets (conflict-free ds_read_b128), each lane keeping the FIRST minimum over its own
//     ascending targets; a (distance, index) lexicographic butterfly yields the global first min;
//   * no atomics, no cross-workgroup reduction: results are final inside the launch, so the
//     epilogue can consume them (the train plan's loss partials and sign scatter).
// Pair cost: 3 sub + 2 add(|.|) + cmp + 2 cndmask = 8 VALU lane-ops; d = (|dx|+|dy|)+|dz| in fp32,
// the pytorch3d knn_cpu accumulation order (bit-exact against the oracle).
#pragma once
#include <type_traits>
#include "creg_dev.h"

namespace creg {

constexpr int NN_TCH = 4096;          // targets staged per LDS chunk (64 KB)
constexpr int NN_BLOCK = 512;         // 8 waves

struct NnEpilogueNone {
    __device__ __forceinline__ void operator()(int, int, int, float, int, float&) const {}
    __device__ __forceinline__ void finish(int, int, float, float*) const {}
};

// A/B: point arrays with sa/sb floats per point (3 = packed xyz, 4 = float4 padded).
// Blocks [0, blocksA) search B for A's queries (direction 0), the rest search A for B's (direction 1).
// Epi::operator()(dir, q, lane, dist, idx, acc) runs on lane u for the wave's u-th query; Epi::finish(dir, blk, sum, scratch)
// once per block with the block's fixed-order sum of `acc`.
template <int QW, typename IdxT, typename Epi>
__global__ __launch_bounds__(NN_BLOCK) void k_nn_l1(
    const float* __restrict__ A, int na, int sa, const float* __restrict__ B, int nb, int sb,
    float* __restrict__ dA, IdxT* __restrict__ iA, float* __restrict__ dB, IdxT* __restrict__ iB,
    int blocksA, Epi epi) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float4* sT = (float4*)smem_raw;
    __shared__ float s_part[NN_BLOCK / 64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int blk = blockIdx.x, dir = 0;
    const float* Q = A; const float* T = B;
    int nq = na, nt = nb, sq = sa, st = sb;
    float* dO = dA; IdxT* iO = iA;
    if (blk >= blocksA) { blk -= blocksA; dir = 1; Q = B; T = A; nq = nb; nt = na; sq = sb; st = sa; dO = dB; iO = iB; }
    const int q0 = (blk * (NN_BLOCK / 64) + wave) * QW;

    float qx[QW], qy[QW], qz[QW], best[QW];
    int bidx[QW];
#pragma unroll
    for (int u = 0; u < QW; ++u) {
        const int qi = min(q0 + u, nq - 1);                  // wave-uniform: scalar loads
        qx[u] = Q[(size_t)qi * sq]; qy[u] = Q[(size_t)qi * sq + 1]; qz[u] = Q[(size_t)qi * sq + 2];
        best[u] = INFINITY; bidx[u] = 0x7fffffff;
    }
    for (int t0 = 0; t0 < nt; t0 += NN_TCH) {
        const int cnt = min(NN_TCH, nt - t0);
        const int padded = (cnt + 255) & ~255;               // pad with +inf points: never selected
        if (t0) __syncthreads();
        // all of a thread's (<= 8) target loads are issued before the first LDS write
        {
            float4 v[NN_TCH / NN_BLOCK];
#pragma unroll
            for (int q = 0; q < NN_TCH / NN_BLOCK; ++q) {
                const int j = min(q * NN_BLOCK + tid, cnt - 1);
                if (st == 4) v[q] = *(const float4*)(T + (size_t)(t0 + j) * 4);
                else { const float* p = T + (size_t)(t0 + j) * st; v[q] = make_float4(p[0], p[1], p[2], 0.f); }
            }
#pragma unroll
            for (int q = 0; q < NN_TCH / NN_BLOCK; ++q) {
                const int j = q * NN_BLOCK + tid;
                if (j < padded) sT[j] = (j < cnt) ? v[q] : make_float4(INFINITY, INFINITY, INFINITY, 0.f);
            }
        }
        __syncthreads();
        for (int jj = lane; jj < padded; jj += 256) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float4 t = sT[jj + 64 * e];
                const int j = t0 + jj + 64 * e;
#pragma unroll
                for (int u = 0; u < QW; ++u) {
                    const float d = l1_dist(qx[u], qy[u], qz[u], t.x, t.y, t.z);
                    const bool lt = d < best[u];
                    best[u] = lt ? d : best[u];
                    bidx[u] = lt ? j : bidx[u];
                }
            }
        }
    }
    // results: lane u keeps query u's (distance, index) so the QW stores / epilogues run side by side
    float acc = 0.f, mv = 0.f;
    int mi = 0;
#pragma unroll
    for (int u = 0; u < QW; ++u) {
        float v = best[u]; int i = bidx[u];
        wave_argmin(v, i);
        if (lane == u) { mv = v; mi = i; }
    }
    if (lane < QW && q0 + lane < nq) {
        if (dO) { dO[q0 + lane] = mv; iO[q0 + lane] = (IdxT)mi; }
        epi(dir, q0 + lane, lane, mv, mi, acc);
    }
    if constexpr (!std::is_same<Epi, NnEpilogueNone>::value) {
        acc = wave_sum_fast(acc);
        if (lane == 0) s_part[wave] = acc;
        __syncthreads();
        if (tid == 0) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < NN_BLOCK / 64; ++w) s += s_part[w];
            epi.finish(dir, blk, s, nullptr);
        }
    }
}

struct NnGrid { int qw, blocksA, blocksB, smem; };
inline NnGrid nn_grid(int na, int nb, bool doA, bool doB) {
    // queries per wave: 4 fills all 256 CUs at N=4096; larger clouds amortise LDS reads with 8
    const long total = (long)(doA ? na : 0) + (long)(doB ? nb : 0);
    NnGrid g;
    g.qw = (total / (8 * 8) >= 2048) ? 8 : 4;
    const int per = (NN_BLOCK / 64) * g.qw;
    g.blocksA = doA ? (na + per - 1) / per : 0;
    g.blocksB = doB ? (nb + per - 1) / per : 0;
    const int mx = na > nb ? na : nb;
    const int cnt = mx < NN_TCH ? mx : NN_TCH;
    g.smem = ((cnt + 255) & ~255) * (int)sizeof(float4);
    return g;
}

template <typename IdxT, typename Epi>
inline void launch_nn_l1(const float* A, int na, int sa, const float* B, int nb, int sb, float* dA, IdxT* iA,
                         float* dB, IdxT* iB, bool doA, bool doB, Epi epi, hipStream_t s) {
    const NnGrid g = nn_grid(na, nb, doA, doB);
    if (g.blocksA + g.blocksB == 0) return;
    if (g.qw == 8)
        hipLaunchKernelGGL((k_nn_l1<8, IdxT, Epi>), dim3(g.blocksA + g.blocksB), dim3(NN_BLOCK), g.smem, s, A, na, sa, B,
                           nb, sb, dA, iA, dB, iB, g.blocksA, epi);
    else
        hipLaunchKernelGGL((k_nn_l1<4, IdxT, Epi>), dim3(g.blocksA + g.blocksB), dim3(NN_BLOCK), g.smem, s, A, na, sa, B,
                           nb, sb, dA, iA, dB, iB, g.blocksA, epi);
}

}  // namespace creg
