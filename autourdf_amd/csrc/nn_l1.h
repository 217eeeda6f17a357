// nn_l1.h -- device routine + launcher for the K=1 L1 nearest-neighbour kernel (shared by the
// standalone C-ABI entry points and the fused train plan).
#pragma once
#include "creg_dev.h"

namespace creg {

// One wave owns QW consecutive queries (wave-uniform, held in SGPRs); its 64 lanes stride over the
// targets, each lane keeping the first minimum over its own ascending targets; a lexicographic
// (distance, index) butterfly then yields the global first minimum.  No LDS, no atomics,
// deterministic.  Pair cost: 3 sub + 2 add(|.|) + cmp + 2 cndmask = 8 VALU lane-ops.
//
// A/B are point arrays with `sa`/`sb` floats per point (3 = packed xyz as the reference stores
// clouds, 4 = float4-padded engine layout).  Two directions are fused into one launch:
// waves [0, wavesA) search B for the queries of A; waves [wavesA, ...) search A for B's.
template <int QW, typename IdxT>
__global__ __launch_bounds__(256) void k_nn_l1_bidir(
    const float* __restrict__ A, int na, int sa, const float* __restrict__ B, int nb, int sb,
    float* __restrict__ dA, IdxT* __restrict__ iA, float* __restrict__ dB, IdxT* __restrict__ iB,
    int wavesA) {
    const int lane = threadIdx.x & 63;
    int wave = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float* Q = A; const float* T = B;
    int nq = na, nt = nb, sq = sa, st = sb;
    float* dO = dA; IdxT* iO = iA;
    if (wave >= wavesA) {                       // second direction
        wave -= wavesA;
        Q = B; T = A; nq = nb; nt = na; sq = sb; st = sa; dO = dB; iO = iB;
    }
    const int q0 = wave * QW;
    if (q0 >= nq || dO == nullptr) return;

    float qx[QW], qy[QW], qz[QW], best[QW];
    int bidx[QW];
#pragma unroll
    for (int u = 0; u < QW; ++u) {
        const int qi = min(q0 + u, nq - 1);     // wave-uniform -> scalar loads
        qx[u] = Q[(size_t)qi * sq]; qy[u] = Q[(size_t)qi * sq + 1]; qz[u] = Q[(size_t)qi * sq + 2];
        best[u] = INFINITY; bidx[u] = 0x7fffffff;
    }
    // Targets are consumed in chunks of TC per lane, double-buffered in registers so the next
    // chunk's loads are in flight while the current one is evaluated (at N=4096 only ~2 waves sit
    // on a SIMD, so latency must be hidden inside the wave).  Out-of-range slots re-read target nt-1.
    constexpr int TC = 8;
    float cx[TC], cy[TC], cz[TC], nx_[TC], ny_[TC], nz_[TC];
    auto load_chunk = [&](int base, float (&ox)[TC], float (&oy)[TC], float (&oz)[TC]) {
#pragma unroll
        for (int e = 0; e < TC; ++e) {
            // clamp instead of predicate: a duplicate of target nt-1 under an index > nt-1 can
            // never beat the real one in the (distance, index) order, and keeps the load free of
            // a dependent select so it really stays in flight during the compute below.
            const int j = min(base + e * 64 + lane, nt - 1);
            const float* t = T + (size_t)j * st;
            ox[e] = t[0]; oy[e] = t[1]; oz[e] = t[2];
        }
    };
    load_chunk(0, cx, cy, cz);
    for (int base = 0; base < nt; base += TC * 64) {
        const bool more = base + TC * 64 < nt;          // wave-uniform
        if (more) load_chunk(base + TC * 64, nx_, ny_, nz_);
#pragma unroll
        for (int e = 0; e < TC; ++e) {
            const int j = base + e * 64 + lane;
#pragma unroll
            for (int u = 0; u < QW; ++u) {
                const float d = l1_dist(qx[u], qy[u], qz[u], cx[e], cy[e], cz[e]);
                const bool lt = d < best[u];
                best[u] = lt ? d : best[u];
                bidx[u] = lt ? j : bidx[u];
            }
        }
        if (more) {
#pragma unroll
            for (int e = 0; e < TC; ++e) { cx[e] = nx_[e]; cy[e] = ny_[e]; cz[e] = nz_[e]; }
        }
    }
#pragma unroll
    for (int u = 0; u < QW; ++u) {
        float v = best[u]; int i = bidx[u];
        wave_argmin(v, i);
        if (lane == 0 && q0 + u < nq) { dO[q0 + u] = v; iO[q0 + u] = (IdxT)i; }
    }
}

template <typename IdxT>
inline void launch_nn_l1_bidir(const float* A, int na, int sa, const float* B, int nb, int sb,
                               float* dA, IdxT* iA, float* dB, IdxT* iB, hipStream_t s) {
    // pick queries-per-wave so the launch has >= ~2048 waves (256 CUs x 4 SIMDs x 2)
    const long total = (long)(dA ? na : 0) + (long)(dB ? nb : 0);
    int qw = 1;
    while (qw < 16 && total / (qw * 2) >= 2048) qw *= 2;
    auto go = [&](auto QWc) {
        constexpr int QW = decltype(QWc)::value;
        const int wavesA = dA ? (na + QW - 1) / QW : 0;
        const int wavesB = dB ? (nb + QW - 1) / QW : 0;
        const int blocks = (wavesA + wavesB + 3) / 4;
        if (blocks > 0)
            hipLaunchKernelGGL((k_nn_l1_bidir<QW, IdxT>), dim3(blocks), dim3(256), 0, s, A, na, sa, B,
                               nb, sb, dA, iA, dB, iB, wavesA);
    };
    switch (qw) {
        case 1: go(std::integral_constant<int, 1>{}); break;
        case 2: go(std::integral_constant<int, 2>{}); break;
        case 4: go(std::integral_constant<int, 4>{}); break;
        case 8: go(std::integral_constant<int, 8>{}); break;
        default: go(std::integral_constant<int, 16>{}); break;
    }
}

}  // namespace creg
