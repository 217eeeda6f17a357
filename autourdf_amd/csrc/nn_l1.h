// nn_l1.h -- the K=1 L1 nearest-neighbour kernels (K1), shared by the standalone C-ABI entry points
// (nn_l1.hip) and the fused train plan (train_engine.hip): the exhaustive search (nn_l1_block / k_nn_l1) and
// the exact block-pruned search over k-d leaf blocks with boxes (nn_l1_block_pruned, the plan's default; the
// plan builds the block layouts).  Both return the same (distance, first index) bit for bit.
//
// Exhaustive mapping (N=4096: 256 workgroups = one per CU, 8 waves each):
//   * a workgroup stages the target cloud into LDS once as three coordinate planes (48 KB per 4096
//     points, chunked for larger clouds) -- without that every wave streamed the whole cloud from
//     L2 (128 MB per launch at N=4096: the v1 kernel was L2-bandwidth bound);
//   * a wave owns QW consecutive queries (wave-uniform -> SGPRs); its lanes stride over the staged
//     targets in groups of 4 (ds_read2st64_b32 returns two targets' coordinate in one register
//     pair, which is what v_pk_add_f32 wants);
//   * per (group, query): 6 packed subs + 8 adds with |.| modifiers give the four distances
//     d = (|dx|+|dy|)+|dz| (pytorch3d knn_cpu order, bit-exact); v_min3 + v_min reduce them and only
//     the GROUP of the running first minimum is tracked (cmp + cndmask + min): 4.75 VALU ops per
//     pair instead of 8 for per-pair index tracking (VALU issue is the bound of this kernel);
//   * after the sweep each lane re-evaluates the 4 targets of its best group to recover the exact
//     first index, then a DPP (distance, index) lexicographic reduction yields the global first min;
//   * no atomics, no cross-workgroup reduction: results are final inside the launch, so the
//     epilogue can consume them (the train plan's loss partials and sign scatter).
#pragma once
#include <type_traits>
#include "creg_dev.h"

namespace creg {

constexpr int NN_TCH = 4096;          // targets staged per LDS chunk (3 planes x 16 KB)
constexpr int NN_BLOCK = 512;         // 8 waves
typedef float nn_f2 __attribute__((ext_vector_type(2)));

struct NnEpilogueNone {
    __device__ __forceinline__ void finish(int, int, float, float*) const {}
    __device__ __forceinline__ void shift(int) {}
};

// A/B: point arrays with sa/sb floats per point (3 = packed xyz, 4 = float4 padded).
// Blocks [0, blocksA) search B for A's queries (direction 0), the rest search A for B's (direction 1).
// Epi::operator()(dir, q, idx, dist, query xyz, matched-target xyz, acc) runs on lane u for the wave's u-th query;
// Epi::finish(dir, blk, sum, scratch) once per block with the block's fixed-order sum of `acc`.
template <int QW, typename IdxT, typename Epi>
__device__ __forceinline__ void nn_l1_block(
    const float* A, int na, int sa, const float* B, int nb, int sb,
    float* __restrict__ dA, IdxT* __restrict__ iA, float* __restrict__ dB, IdxT* __restrict__ iB,
    int blocksA, Epi& epi, int bx) {      // bx: block index in [0, blocksA + blocksB)
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* sX = (float*)smem_raw;                 // planes of `padded` floats each
    __shared__ float s_part[NN_BLOCK / 64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int blk = bx, dir = 0;
    const float* Q = A; const float* T = B;
    int nq = na, nt = nb, sq = sa, st = sb;
    float* dO = dA; IdxT* iO = iA;
    if (blk >= blocksA) { blk -= blocksA; dir = 1; Q = B; T = A; nq = nb; nt = na; sq = sb; st = sa; dO = dB; iO = iB; }
    const int q0 = (blk * (NN_BLOCK / 64) + wave) * QW;

    float qx[QW], qy[QW], qz[QW], best[QW];
    int bgrp[QW];                                  // first target index of the group holding the lane's first min
#pragma unroll
    for (int u = 0; u < QW; ++u) {
        const int qi = min(q0 + u, nq - 1);                  // wave-uniform: scalar loads
        qx[u] = Q[(size_t)qi * sq]; qy[u] = Q[(size_t)qi * sq + 1]; qz[u] = Q[(size_t)qi * sq + 2];
        best[u] = INFINITY; bgrp[u] = -1;
    }
    for (int t0 = 0; t0 < nt; t0 += NN_TCH) {
        const int cnt = min(NN_TCH, nt - t0);
        const int padded = (cnt + 255) & ~255;               // pad with +inf points: never selected
        float* sY = sX + padded; float* sZ = sY + padded;
        if (t0) __syncthreads();
        {   // all of a thread's (<= 8) target loads are issued before the first LDS write
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
                const bool real = j < cnt;
                if (j < padded) { sX[j] = real ? v[q].x : INFINITY; sY[j] = real ? v[q].y : INFINITY; sZ[j] = real ? v[q].z : INFINITY; }
            }
        }
        __syncthreads();
        for (int jj = lane; jj < padded; jj += 256) {
            const nn_f2 x01 = {sX[jj], sX[jj + 64]}, x23 = {sX[jj + 128], sX[jj + 192]};
            const nn_f2 y01 = {sY[jj], sY[jj + 64]}, y23 = {sY[jj + 128], sY[jj + 192]};
            const nn_f2 z01 = {sZ[jj], sZ[jj + 64]}, z23 = {sZ[jj + 128], sZ[jj + 192]};
            const int grp = t0 + jj;
#pragma unroll
            for (int u = 0; u < QW; ++u) {
                const nn_f2 qxx = {qx[u], qx[u]}, qyy = {qy[u], qy[u]}, qzz = {qz[u], qz[u]};
                const nn_f2 ax = qxx - x01, bx = qxx - x23, ay = qyy - y01, by = qyy - y23, az = qzz - z01, bz = qzz - z23;
                const float d0 = (fabsf(ax.x) + fabsf(ay.x)) + fabsf(az.x);
                const float d1 = (fabsf(ax.y) + fabsf(ay.y)) + fabsf(az.y);
                const float d2 = (fabsf(bx.x) + fabsf(by.x)) + fabsf(bz.x);
                const float d3 = (fabsf(bx.y) + fabsf(by.y)) + fabsf(bz.y);
                const float g = fminf(__builtin_fminf(__builtin_fminf(d0, d1), d2), d3);
                const bool lt = g < best[u];
                bgrp[u] = lt ? grp : bgrp[u];
                best[u] = lt ? g : best[u];
            }
        }
    }
    // exact first index inside the best group: re-evaluate its 4 targets.  With a single staged chunk
    // (nt <= NN_TCH, the N=4096 case) they are still in LDS; otherwise one batch of global loads.
    const bool in_lds = nt <= NN_TCH;
    const int padded0 = (min(NN_TCH, nt) + 255) & ~255;
    float acc = 0.f, mv = 0.f;
    int mi = 0;
    float rx[QW][4], ry[QW][4], rz[QW][4];
#pragma unroll
    for (int u = 0; u < QW; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = min(max(bgrp[u], 0) + 64 * e, nt - 1);
            if (in_lds) { rx[u][e] = sX[j]; ry[u][e] = sX[padded0 + j]; rz[u][e] = sX[2 * padded0 + j]; }
            else { const float* p = T + (size_t)j * st; rx[u][e] = p[0]; ry[u][e] = p[1]; rz[u][e] = p[2]; }
        }
    float mqx = 0.f, mqy = 0.f, mqz = 0.f;
#pragma unroll
    for (int u = 0; u < QW; ++u) {
        int bi = 0x7fffffff;
#pragma unroll
        for (int e = 3; e >= 0; --e) {
            const int j = bgrp[u] + 64 * e;
            const float d = l1_dist(qx[u], qy[u], qz[u], rx[u][e], ry[u][e], rz[u][e]);
            if (bgrp[u] >= 0 && j < nt && d == best[u]) bi = j;      // descending e: the smallest j survives
        }
        float v = best[u]; int i = bi;
        wave_argmin_fast(v, i);
        if (lane == u) { mv = v; mi = i; mqx = qx[u]; mqy = qy[u]; mqz = qz[u]; }
    }
    if (lane < QW && q0 + lane < nq) {
        if (dO) { dO[q0 + lane] = mv; iO[q0 + lane] = (IdxT)mi; }
        if constexpr (!std::is_same<Epi, NnEpilogueNone>::value) {
            // coordinates of the matched target: LDS when staged, else global
            float tx, ty, tz;
            if (in_lds) { tx = sX[mi]; ty = sX[padded0 + mi]; tz = sX[2 * padded0 + mi]; }
            else { const float* p = T + (size_t)mi * st; tx = p[0]; ty = p[1]; tz = p[2]; }
            epi(dir, q0 + lane, mi, mv, mqx, mqy, mqz, tx, ty, tz, acc);
        }
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

// ---- exact search over a block-sorted target cloud --------------------------------------------------
// The targets are laid out in blocks of 64 * PPL consecutive slots that are spatially compact (leaves of a k-d tree)
// with an axis-aligned box each: `ts4[slot] = (x, y, z, bits(original index))`, +inf padding slots carry index
// INT_MAX, boxes as six planes `tbox[c * 64 * NB + b]`, c = lo x,y,z, hi x,y,z (coalesced: a lane reads its box with
// six loads that each cover consecutive words), at most 64 * NB blocks so a wave holds NB boxes per lane.
// A wave owns 4 queries as in the exhaustive kernel.  Per query: every lane evaluates the L1 distance from the
// query to its box(es) -- a lower bound of the distance to every target inside, and in float arithmetic too:
// each |q - t| is a monotone function of t on either side of the box and rounding is monotone; the block with
// the smallest bound is visited first (one target per lane, one coalesced 1 KB read), the wave minimum becomes
// the pruning radius, and only blocks whose bound is <= the radius (ties included) are visited after it.
// The result is the exhaustive kernel's: smallest distance, then smallest ORIGINAL index.  On the registration
// clouds 2-3 of the 64-80 blocks are visited.
struct NnBlocks { const float4* ts4; const float* tbox; int nblk; const int* nblk_dev; };

// stop_flag (optional, wave-uniform address): non-zero = the caller's train has stopped early -- the block returns before
// its search (the flag is requested with the first loads and tested after the box bounds, so a live search pays nothing).
// T / st (PPL > 1 only): the targets in their ORIGINAL order (the same values the blocks hold).  With several points per lane and
// visit the launch is VALU-issue bound, so the running best is one 64-bit key (distance bits : original index -- distances are
// non-negative, their bit patterns order like the values) compared once per point, and the winner's coordinates are not carried
// along but read from T by its index at the end.
// STOP / NBDEV: the flag / the device-side block count exist (compile time: as run-time null tests each became a branch with its own
// scalar round trip in front of the first box load -- two dependent memory round trips per launch before the search had requested anything).
template <int NB, int PPL, typename Epi, bool STOP = false, bool NBDEV = false>
__device__ __forceinline__ void nn_l1_block_pruned(const float* Q, int nq, int sq, NnBlocks tb, int dir, Epi& epi, int blk,
                                                   const int* stop_flag = nullptr, const float* T = nullptr, int st = 0) {
    constexpr int QW = 4;
    int stop = 0;
    if constexpr (STOP) stop = *stop_flag;
    __shared__ float s_partp[NN_BLOCK / 64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q0 = (blk * (NN_BLOCK / 64) + wave) * QW;
    int nblk = tb.nblk;
    if constexpr (NBDEV) nblk = *tb.nblk_dev;
    float lox[NB], loy[NB], loz[NB], hix[NB], hiy[NB], hiz[NB];
#pragma unroll
    for (int g = 0; g < NB; ++g) {
        // (the box table is allocated for 64 * NB entries: no dependence of these loads on nblk_dev's round trip;
        //  entries past nblk get an infinite bound below)
        const float* bx = tb.tbox + (NBDEV ? 64 * g + lane : min(64 * g + lane, nblk - 1));
        constexpr int BP = 64 * NB;                             // plane stride
        lox[g] = bx[0]; loy[g] = bx[BP]; loz[g] = bx[2 * BP]; hix[g] = bx[3 * BP]; hiy[g] = bx[4 * BP]; hiz[g] = bx[5 * BP];
    }
    float qx[QW], qy[QW], qz[QW], lb[QW][NB], bd[QW], tx[QW], ty[QW], tz[QW], wb[QW];
    int bi[QW];
    unsigned long long pend[QW][NB];          // (wave-uniform masks: SGPR pairs.  A visited block's bound is set to +inf
                                              //  instead of keeping a second mask array: QW x NB x 2 masks ran out of SGPRs at NB > 2)
#pragma unroll
    for (int u = 0; u < QW; ++u) {
        const int qi = min(q0 + u, nq - 1);
        qx[u] = Q[(size_t)qi * sq]; qy[u] = Q[(size_t)qi * sq + 1]; qz[u] = Q[(size_t)qi * sq + 2];
    }
    if constexpr (STOP || NBDEV) {
        // wave-uniform values that arrive through vector loads (requested together with the boxes) go back to SGPRs: as VGPRs the four
        // queries' coordinates cost the launch a wave per SIMD (82 registers instead of 71 at one point per lane)
        stop = __builtin_amdgcn_readfirstlane(stop); nblk = __builtin_amdgcn_readfirstlane(nblk);
#pragma unroll
        for (int u = 0; u < QW; ++u) {
            qx[u] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(qx[u])));
            qy[u] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(qy[u])));
            qz[u] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(qz[u])));
        }
    }
#pragma unroll
    for (int u = 0; u < QW; ++u) {
        float lm = INFINITY;
#pragma unroll
        for (int g = 0; g < NB; ++g) {
            const float ex = fmaxf(fmaxf(lox[g] - qx[u], qx[u] - hix[g]), 0.f);
            const float ey = fmaxf(fmaxf(loy[g] - qy[u], qy[u] - hiy[g]), 0.f);
            const float ez = fmaxf(fmaxf(loz[g] - qz[u], qz[u] - hiz[g]), 0.f);
            const float bound = (ex + ey) + ez;                 // same association as l1_dist
            lb[u][g] = 64 * g + lane < nblk ? bound : INFINITY; // (slots past the last block: never visited)
            lm = __builtin_fminf(lm, lb[u][g]);
        }
        const float m = wave_min_fast(lm);                      // start from the block with the smallest bound
        bool found = false;
#pragma unroll
        for (int g = 0; g < NB; ++g) {
            const unsigned long long f = __ballot(lb[u][g] == m && 64 * g + lane < nblk);
            pend[u][g] = (!found && f) ? (f & (~f + 1ull)) : 0ull;       // lowest set bit
            found = found || f != 0ull;
        }
        if (!found) pend[u][0] = 1ull;                          // NaN bound: any block, nothing will be taken
        bd[u] = INFINITY; bi[u] = 0x7fffffff; tx[u] = ty[u] = tz[u] = 0.f; wb[u] = INFINITY;
    }
    if (stop) return;                                           // workgroup-uniform
    bool more = true;
    while (more) {
        int b[QW];
        bool act[QW];
#pragma unroll
        for (int u = 0; u < QW; ++u) {
            b[u] = 0; act[u] = false;
#pragma unroll
            for (int g = NB - 1; g >= 0; --g)
                if (pend[u][g]) { b[u] = 64 * g + __builtin_ctzll(pend[u][g]); act[u] = true; }
        }
        float4 v1[QW];
        if constexpr (PPL == 1) {                             // the loads of all four queries are in flight together
#pragma unroll
            for (int u = 0; u < QW; ++u) v1[u] = tb.ts4[(size_t)b[u] * 64 + lane];
        }
        more = false;
#pragma unroll
        for (int u = 0; u < QW; ++u) {
            if (act[u]) {                                    // wave-uniform
                float4 v[PPL];
                if constexpr (PPL == 1) v[0] = v1[u];
                else {                                       // big clouds: PPL points per lane, enough waves to hide the reads
#pragma unroll
                    for (int k = 0; k < PPL; ++k) v[k] = tb.ts4[((size_t)b[u] * PPL + k) * 64 + lane];
                }
#pragma unroll
                for (int g = 0; g < NB; ++g)
                    if ((b[u] >> 6) == g && lane == (b[u] & 63)) lb[u][g] = INFINITY;     // visited: never again
                if constexpr (PPL > 1) {
                    unsigned long long key = ((unsigned long long)__float_as_uint(bd[u]) << 32) | (unsigned)bi[u];
#pragma unroll
                    for (int k = 0; k < PPL; ++k) {
                        const float d = l1_dist(qx[u], qy[u], qz[u], v[k].x, v[k].y, v[k].z);
                        // (a NaN distance has the bits of a huge key and is never taken, like the comparisons of the other form;
                        //  -0.0 cannot occur: a sum of absolute values)
                        const unsigned long long kk = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)__float_as_int(v[k].w);
                        key = kk < key ? kk : key;
                    }
                    bd[u] = __uint_as_float((unsigned)(key >> 32)); bi[u] = (int)(unsigned)key;
                } else {
#pragma unroll
                    for (int k = 0; k < PPL; ++k) {
                        const float d = l1_dist(qx[u], qy[u], qz[u], v[k].x, v[k].y, v[k].z);
                        const int oi = __float_as_int(v[k].w);
                        const unsigned long long key = ((unsigned long long)__float_as_uint(bd[u]) << 32) | (unsigned)bi[u];
                        const unsigned long long kk = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)oi;
                        const bool take = kk < key;
                        bd[u] = take ? d : bd[u]; bi[u] = take ? oi : bi[u];
                        tx[u] = take ? v[k].x : tx[u]; ty[u] = take ? v[k].y : ty[u]; tz[u] = take ? v[k].z : tz[u];
                    }
                }
                wb[u] = wave_min_fast(bd[u]);
#pragma unroll
                for (int g = 0; g < NB; ++g) {
                    pend[u][g] = __ballot(lb[u][g] <= wb[u] && lb[u][g] < INFINITY);
                    more |= pend[u][g] != 0ull;
                }
            }
        }
    }
    float acc = 0.f, mv = 0.f, mqx = 0.f, mqy = 0.f, mqz = 0.f, mtx = 0.f, mty = 0.f, mtz = 0.f;
    int mi = 0;
#pragma unroll
    for (int u = 0; u < QW; ++u) {
        const int i = wave_min_fast(bd[u] == wb[u] ? bi[u] : 0x7fffffff);
        const unsigned long long win = __ballot(bd[u] == wb[u] && bi[u] == i);
        const int src = win ? __builtin_ctzll(win) : 0;
        const float fx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tx[u]), src));
        const float fy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ty[u]), src));
        const float fz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tz[u]), src));
        if (lane == u) { mv = wb[u]; mi = i; mqx = qx[u]; mqy = qy[u]; mqz = qz[u]; mtx = fx; mty = fy; mtz = fz; }
    }
    if constexpr (PPL > 1) {
        if (lane < QW && q0 + lane < nq && mi != 0x7fffffff) { const float* p = T + (size_t)mi * st; mtx = p[0]; mty = p[1]; mtz = p[2]; }
    }
    if (lane < QW && q0 + lane < nq) epi(dir, q0 + lane, mi, mv, mqx, mqy, mqz, mtx, mty, mtz, acc);
    acc = wave_sum_fast(acc);
    if (lane == 0) s_partp[wave] = acc;
    __syncthreads();
    if (tid == 0) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < NN_BLOCK / 64; ++w) s += s_partp[w];
        epi.finish(dir, blk, s, nullptr);
    }
}

// ---- exact search, sixteen queries per wave (round 5) -------------------------------------------------------------------------
// nn_l1_block_pruned gives a wave FOUR queries and evaluates one target per lane: every per-query step -- the box bounds, the wave
// minima, the ballots, the bookkeeping of visited blocks -- is executed by all 64 lanes for one query at a time, and the pair
// evaluations are a sixth of its ~800 VALU instructions per wave (200 per query; the launch is VALU-issue bound: 6 waves per SIMD at
// three problems).  Here a wave takes SIXTEEN queries that are neighbours in space -- 16 consecutive slots of a k-d leaf of the
// QUERY cloud (both clouds are block-sorted: the target frame once per frame, the predicted cloud by k_head every epoch) -- as
// lane = 16 g + q: query q, target quarter g.  Per wave, once: the box of its queries (row reductions), its L1 box-to-box bound to
// every target block (lane b <-> block b: a lower bound for every (query, target) pair, in float arithmetic too -- subtraction and
// addition are monotone), the block with the smallest bound is visited first.  A visit: lane (g, q) evaluates targets 16 g .. 16 g + 15
// of the block for its query (16-byte loads, one address per row), running minimum as ONE 64-bit key (distance bits : original index),
// then the four quarters of a query are combined by two row swaps.  After the first visit the remaining blocks are taken in order of
// their box-to-box bound while that bound is <= the largest running distance among the wave's queries; a block is visited only if
// some query's own point-to-box bound is <= its running distance (ties included: a smaller original index may win).  Exact: the
// result is the exhaustive kernel's (smallest distance, then smallest ORIGINAL index).  Per-wave instruction count ~900 for 16
// queries (~55 per query).  The epilogue runs per lane (row 0), the wave's loss partial is the row sum of its 16 distances in slot
// order: lossp[16-slot group] -- another summation order than the four-queries-per-wave kernels' (per 32 original indices), i.e. the
// loss can differ from theirs in the last bit; indices, distances, signs and counters are identical.
__device__ __forceinline__ float row_min16(float v) {
    asm("s_nop 4\n\t"
        "v_min_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_min_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_min_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_min_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1"
        : "+v"(v));
    return v;
}
__device__ __forceinline__ float row_max16(float v) {
    asm("s_nop 4\n\t"
        "v_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1"
        : "+v"(v));
    return v;
}
__device__ __forceinline__ float row_sum16_f(float v) {
    CREG_DPP_STEP(v, 0xB1, 0xF);    // quad_perm [1,0,3,2]
    CREG_DPP_STEP(v, 0x4E, 0xF);    // quad_perm [2,3,0,1]
    CREG_DPP_STEP(v, 0x141, 0xF);   // row_half_mirror
    CREG_DPP_STEP(v, 0x140, 0xF);   // row_mirror
    return v;
}
// min over the lanes l, l ^ 16, l ^ 32, l ^ 48 of a 64-bit key (gfx950 row / half swaps: one VALU instruction per word and step)
__device__ __forceinline__ unsigned long long key_min_rows(unsigned long long k) {
    {   const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)(k >> 32), (unsigned)(k >> 32), false, false);
        const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)k, (unsigned)k, false, false);
        const unsigned long long a = ((unsigned long long)hi[0] << 32) | lo[0], b = ((unsigned long long)hi[1] << 32) | lo[1];
        k = a < b ? a : b; }
    {   const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)(k >> 32), (unsigned)(k >> 32), false, false);
        const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)k, (unsigned)k, false, false);
        const unsigned long long a = ((unsigned long long)hi[0] << 32) | lo[0], b = ((unsigned long long)hi[1] << 32) | lo[1];
        k = a < b ? a : b; }
    return k;
}
#ifdef CREG_NN_STATS
__device__ unsigned long long g_nn_stats[8];       // waves, candidates tested, visits, ... (measurement build: tests/measure/nn_rows_stats.py)
#define NN_STAT(i, v) do { if (lane == 0) atomicAdd(&g_nn_stats[i], (unsigned long long)(v)); } while (0)
#else
#define NN_STAT(i, v) do { } while (0)
#endif
#ifdef CREG_NN_WAVE_STAMPS
// Measurement build (tests/measure/nn_rows_waves.py): every wave of nn_l1_rows records when it started, when its box
// bounds were done, when its visits were done and when it left (100 MHz wall clock), with its visit / candidate counts.
struct NnWaveRec { unsigned long long t0, t1, t2, t3; int visits, cands, dir, blk; };
constexpr unsigned NN_WAVE_CAP = 1u << 19, NN_WAVE_SHARDS = 256, NN_WAVE_PER = NN_WAVE_CAP / NN_WAVE_SHARDS;
__device__ NnWaveRec g_nn_wave[NN_WAVE_CAP];
__device__ unsigned g_nn_wave_n[NN_WAVE_SHARDS * 32];      // one slot counter per shard (blockIdx & 255), a 128-byte line each: thousands of waves taking
                                                           //   their record slot from ONE word serialise in the L2 (47 us per launch: the first version measured itself)
#define NN_WSTAMP(...) __VA_ARGS__
#else
#define NN_WSTAMP(...)
#endif
constexpr int NN_ROWQ = 16;                        // queries per wave
#ifndef NN_ROW_FIRST
#define NN_ROW_FIRST 2                             // blocks loaded in the first trip: the nearest by bound + the next ones, speculatively
#endif
#ifndef NN_ROW_BATCH
#define NN_ROW_BATCH 3                             // candidate blocks whose loads are in flight together after the first visit
#endif
// Workgroups of FOUR waves (round 6).  nn_l1_rows has no workgroup-level state (no LDS, no barrier), so the workgroup only sets the
// granule in which waves are admitted to a CU: an 8-wave workgroup takes two wave slots per SIMD, and at the franka shape's instance (94
// VGPRs: five waves per SIMD) a CU held two of them -- four of the five slots -- so the 552 workgroups of a two-problem launch needed a
// second round on 512 places: the last ~40 started 11 us into the launch (measured per wave, profiles/r06_nn_rows_waves_franka.log).
// Four-wave workgroups fill all five slots and the whole launch is resident at once.
#ifndef NN_ROWS_BLOCK
#define NN_ROWS_BLOCK 256
#endif
constexpr int NN_ROW_SLOTS = (NN_ROWS_BLOCK / 64) * NN_ROWQ;      // query slots per workgroup

// qs4: the QUERY cloud in slot order (xyz, bits(original index); padding = index INT_MAX), 64 * nqblk slots in use (nqblk_dev: the
// device-side count when the host only knows an upper bound); tb: the target blocks of 64 slots with their boxes (NB per lane);
// T: the targets in ORIGINAL order, 4 floats per point (the winner's coordinates for the epilogue); lossp[group]: the wave's partial.
// nqblk_host: the HOST's upper bound of the query cloud's blocks -- what the slot array and lossp are allocated for (4 groups per block).
// Waves past that bound leave before any load or store (with 128-slot workgroups and an odd bound the last workgroup's upper four
// waves used to read qs4 and write lossp[4 nqblk_host ..] into the carve padding; workgroups are one block of 64 slots since round 6).
template <int NB, typename Epi, bool NBDEV, bool NQDEV>
__device__ __forceinline__ void nn_l1_rows(const float4* __restrict__ qs4, int nqblk, const int* nqblk_dev, int nqblk_host, NnBlocks tb, int dir, Epi& epi, int blk,
                                           const int* stop_flag, const float* __restrict__ T, float* __restrict__ lossp) {
    const int tid = threadIdx.x, lane = tid & 63, q = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = blk * (NN_ROWS_BLOCK / 64) + wave;        // this wave's 16-slot group of the query cloud
    if (grp >= 4 * nqblk_host) return;                        // wave-uniform; nothing below synchronises the workgroup
    NN_WSTAMP(const unsigned long long ws_t0 = (unsigned long long)wall_clock64(); unsigned long long ws_t1 = 0, ws_t2 = 0; int ws_vis = 0, ws_cand = 0;)
    int stop = *stop_flag;
    int nblk = tb.nblk;
    if constexpr (NBDEV) nblk = *tb.nblk_dev;
    if constexpr (NQDEV) nqblk = *nqblk_dev;
    float lox[NB], loy[NB], loz[NB], hix[NB], hiy[NB], hiz[NB];
#pragma unroll
    for (int gi = 0; gi < NB; ++gi) {          // (the box table holds 64 * NB entries: no dependence on the block counts' round trip)
        const float* bx = tb.tbox + 64 * gi + lane;
        constexpr int BP = 64 * NB;
        lox[gi] = bx[0]; loy[gi] = bx[BP]; loz[gi] = bx[2 * BP]; hix[gi] = bx[3 * BP]; hiy[gi] = bx[4 * BP]; hiz[gi] = bx[5 * BP];
    }
    const int slot = NN_ROWQ * grp + q;
    const float4 qv = qs4[slot];               // (inside the slot array: grp < 4 nqblk_host, the bound it is allocated for)
    stop = __builtin_amdgcn_readfirstlane(stop); nblk = __builtin_amdgcn_readfirstlane(nblk); nqblk = __builtin_amdgcn_readfirstlane(nqblk);
    const int qi = __float_as_int(qv.w);
    const bool valid = slot < 64 * nqblk && qi != 0x7fffffff;
    const float qx = qv.x, qy = qv.y, qz = qv.z;
    if (stop) return;                                            // workgroup-uniform
    if (!__ballot(valid)) { if (lane == 0) lossp[grp] = 0.f; return; }      // a group of padding / past the cloud: an empty partial
    // the box of the wave's queries (every row holds the same sixteen)
    const float qlx = row_min16(valid ? qx : INFINITY), qly = row_min16(valid ? qy : INFINITY), qlz = row_min16(valid ? qz : INFINITY);
    const float qhx = row_max16(valid ? qx : -INFINITY), qhy = row_max16(valid ? qy : -INFINITY), qhz = row_max16(valid ? qz : -INFINITY);
    float bnd[NB];                                               // box-to-box bound of block 64 gi + lane; +inf: visited, rejected or none
    float lm = INFINITY;
#pragma unroll
    for (int gi = 0; gi < NB; ++gi) {
        const float ex = fmaxf(fmaxf(lox[gi] - qhx, qlx - hix[gi]), 0.f);
        const float ey = fmaxf(fmaxf(loy[gi] - qhy, qly - hiy[gi]), 0.f);
        const float ez = fmaxf(fmaxf(loz[gi] - qhz, qlz - hiz[gi]), 0.f);
        const float b = (ex + ey) + ez;                          // same association as l1_dist
        bnd[gi] = (64 * gi + lane < nblk && b == b) ? b : INFINITY;     // (a NaN box: never a candidate; nothing could be taken from it)
        lm = __builtin_fminf(lm, bnd[gi]);
    }
    unsigned long long key = (0x7f800000ull << 32) | 0x7fffffffu;      // (+inf : INT_MAX): nothing found yet
    NN_STAT(0 + 4 * dir, 1);
    NN_WSTAMP(asm volatile("" :: "v"(bnd[0]), "v"(lm)); ws_t1 = wall_clock64();)
    // One block's 64 targets against the wave's queries: lane l holds target l (ONE coalesced 16-byte load per lane and visit); lane
    // (g, q) meets the sixteen targets of ITS row by rotating the row (DPP row_ror: the rotated operand feeds the subtraction
    // directly) -- every lane sees them in another order, which the key (distance : original index) does not care about.
    auto visit = [&](const float4 v) {
#define CREG_NN_PAIR(tx_, ty_, tz_, ti_) {                                                                                   \
            const float d = l1_dist(qx, qy, qz, tx_, ty_, tz_);                                                               \
            /* (a NaN distance has the bits of a huge key and is never taken; -0.0 cannot occur: a sum of absolute values) */ \
            const unsigned long long kk = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)(ti_);                   \
            key = kk < key ? kk : key; }
#define CREG_NN_ROT(R) CREG_NN_PAIR(__int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v.x), 0x120 + R, 0xF, 0xF, true)),   \
                                    __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v.y), 0x120 + R, 0xF, 0xF, true)),   \
                                    __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v.z), 0x120 + R, 0xF, 0xF, true)),   \
                                    __builtin_amdgcn_update_dpp(0, __float_as_int(v.w), 0x120 + R, 0xF, 0xF, true))
        CREG_NN_PAIR(v.x, v.y, v.z, __float_as_int(v.w))
        CREG_NN_ROT(1) CREG_NN_ROT(2) CREG_NN_ROT(3) CREG_NN_ROT(4) CREG_NN_ROT(5) CREG_NN_ROT(6) CREG_NN_ROT(7) CREG_NN_ROT(8)
        CREG_NN_ROT(9) CREG_NN_ROT(10) CREG_NN_ROT(11) CREG_NN_ROT(12) CREG_NN_ROT(13) CREG_NN_ROT(14) CREG_NN_ROT(15)
#undef CREG_NN_ROT
#undef CREG_NN_PAIR
        key = key_min_rows(key);                                 // the four rows of every query
    };
    // the remaining block with the smallest box-to-box bound <= r: its index (or -1), its box into (bl, bh), its bound retired
    auto take_next = [&](float r, float (&bl)[3], float (&bh)[3]) -> int {
        const float m = wave_min_fast(lm);
        if (!(m <= r)) return -1;                                // (also when nothing is left: m = +inf, r < +inf -- or both +inf: below)
        if (!(m < INFINITY)) return -1;
        int b = 0;
        bool found = false;
#pragma unroll
        for (int gi = 0; gi < NB; ++gi) {
            const unsigned long long f = __ballot(bnd[gi] == m);
            if (!found && f) { b = 64 * gi + __builtin_ctzll(f); found = true; }
        }
        lm = INFINITY;
#pragma unroll
        for (int gi = 0; gi < NB; ++gi) {
            if ((b >> 6) == gi) {                                // wave-uniform
                const int src = b & 63;
                bl[0] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lox[gi]), src)); bl[1] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(loy[gi]), src));
                bl[2] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(loz[gi]), src)); bh[0] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(hix[gi]), src));
                bh[1] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(hiy[gi]), src)); bh[2] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(hiz[gi]), src));
                if (lane == src) bnd[gi] = INFINITY;             // taken: never again
            }
            lm = __builtin_fminf(lm, bnd[gi]);
        }
        NN_STAT(1 + 4 * dir, 1);
        NN_WSTAMP(++ws_cand;)
        return b;
    };
    auto needed = [&](const float (&bl)[3], const float (&bh)[3]) -> bool {      // does any query's own point-to-box bound reach its running distance?
        const float ex = fmaxf(fmaxf(bl[0] - qx, qx - bh[0]), 0.f), ey = fmaxf(fmaxf(bl[1] - qy, qy - bh[1]), 0.f), ez = fmaxf(fmaxf(bl[2] - qz, qz - bh[2]), 0.f);
        const float pb = (ex + ey) + ez;
        return __ballot(valid && pb <= __uint_as_float((unsigned)(key >> 32))) != 0ull;
    };
    {   // the first trip: the block with the smallest bound (always evaluated) and, their loads in flight with its own, the next
        // NN_ROW_FIRST - 1 by bound (evaluated if some query still needs them afterwards) -- one dependent round trip instead of two
        float bl[NN_ROW_FIRST][3], bh[NN_ROW_FIRST][3];
        int bb[NN_ROW_FIRST];
        bool open = true;
#pragma unroll
        for (int j = 0; j < NN_ROW_FIRST; ++j) {
            bb[j] = open ? take_next(INFINITY, bl[j], bh[j]) : -1;
            open = bb[j] >= 0;
        }
        float4 v[NN_ROW_FIRST];
#pragma unroll
        for (int j = 0; j < NN_ROW_FIRST; ++j) v[j] = tb.ts4[(size_t)max(bb[j], 0) * 64 + lane];
        if (bb[0] >= 0) { NN_STAT(2 + 4 * dir, 1); NN_WSTAMP(++ws_vis;) visit(v[0]); }
#pragma unroll
        for (int j = 1; j < NN_ROW_FIRST; ++j)
            if (bb[j] >= 0 && needed(bl[j], bh[j])) { NN_STAT(2 + 4 * dir, 1); NN_WSTAMP(++ws_vis;) visit(v[j]); }
    }
    // then batches of up to NN_ROW_BATCH: the next blocks by bound while the bound is <= the largest running distance of the wave's
    // queries (taken from the distances BEFORE the batch: a superset, r only shrinks), their loads in flight together; a block is
    // evaluated only if some query still needs it when its turn comes
    for (;;) {
        const float r = -wave_min_fast(valid ? -__uint_as_float((unsigned)(key >> 32)) : INFINITY);
        float bl[NN_ROW_BATCH][3], bh[NN_ROW_BATCH][3];
        int bb[NN_ROW_BATCH];
        bool open = true;
#pragma unroll
        for (int j = 0; j < NN_ROW_BATCH; ++j) {
            bb[j] = open ? take_next(r, bl[j], bh[j]) : -1;
            open = bb[j] >= 0;
        }
        if (bb[0] < 0) break;
        float4 v[NN_ROW_BATCH];
#pragma unroll
        for (int j = 0; j < NN_ROW_BATCH; ++j) v[j] = tb.ts4[(size_t)max(bb[j], 0) * 64 + lane];
#pragma unroll
        for (int j = 0; j < NN_ROW_BATCH; ++j)
            if (bb[j] >= 0 && needed(bl[j], bh[j])) { NN_STAT(2 + 4 * dir, 1); NN_WSTAMP(++ws_vis;) visit(v[j]); }
        if (!open) break;                                        // the batch was not full: nothing is left within r
    }
    NN_WSTAMP(asm volatile("" :: "v"(key)); ws_t2 = wall_clock64();)
    // ---- per query (row 0): outputs, epilogue, the wave's partial
    const float bd = __uint_as_float((unsigned)(key >> 32));
    const int bi = (int)(unsigned)key;
    float acc = 0.f;
    if (g == 0 && valid) {
        float tx = 0.f, ty = 0.f, tz = 0.f;
        if (bi != 0x7fffffff) { const float* p = T + (size_t)bi * 4; tx = p[0]; ty = p[1]; tz = p[2]; }
        epi(dir, qi, bi, bd, qx, qy, qz, tx, ty, tz, acc);
    }
    acc = row_sum16_f(acc);
    if (lane == 0) lossp[grp] = acc;
    NN_WSTAMP(asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
              const unsigned long long ws_t3 = (unsigned long long)wall_clock64();
              if (lane == 0) { const unsigned sh_ = blockIdx.x & (NN_WAVE_SHARDS - 1);
                               const unsigned i_ = atomicAdd(&g_nn_wave_n[32 * sh_], 1u);
                               if (i_ < NN_WAVE_PER) g_nn_wave[sh_ * NN_WAVE_PER + i_] = NnWaveRec{ws_t0, ws_t1, ws_t2, ws_t3, ws_vis, ws_cand, dir, blk}; })
}

template <int QW, typename IdxT, typename Epi>
__global__ __launch_bounds__(NN_BLOCK) void k_nn_l1(
    const float* A, int na, int sa, const float* B, int nb, int sb,
    float* __restrict__ dA, IdxT* __restrict__ iA, float* __restrict__ dB, IdxT* __restrict__ iB,
    int blocksA, Epi epi, size_t zstride) {
    // grid.z = independent problems of a batch: point arrays and epilogue outputs of problem z live
    // zstride bytes further on (0 for the standalone entry points)
    A = (const float*)((const char*)A + blockIdx.z * zstride);
    B = (const float*)((const char*)B + blockIdx.z * zstride);
    epi.shift(blockIdx.z);
    nn_l1_block<QW, IdxT, Epi>(A, na, sa, B, nb, sb, dA, iA, dB, iB, blocksA, epi, (int)blockIdx.x);
}

struct NnGrid { int qw, blocksA, blocksB, smem; };
inline NnGrid nn_grid(int na, int nb, bool doA, bool doB, int force_qw = 0) {
    // queries per wave: 4 fills all 256 CUs at N=4096; larger clouds amortise LDS reads with 8 (force_qw: the train plan
    // always cuts its queries in fours, so that its loss partials do not depend on the search it runs)
    const long total = (long)(doA ? na : 0) + (long)(doB ? nb : 0);
    NnGrid g;
    g.qw = force_qw ? force_qw : ((total / (8 * 8) >= 2048) ? 8 : 4);
    const int per = (NN_BLOCK / 64) * g.qw;
    g.blocksA = doA ? (na + per - 1) / per : 0;
    g.blocksB = doB ? (nb + per - 1) / per : 0;
    const int mx = na > nb ? na : nb;
    const int cnt = mx < NN_TCH ? mx : NN_TCH;
    g.smem = ((cnt + 255) & ~255) * 3 * (int)sizeof(float);
    return g;
}

template <typename IdxT, typename Epi>
inline void launch_nn_l1(const float* A, int na, int sa, const float* B, int nb, int sb, float* dA, IdxT* iA,
                         float* dB, IdxT* iB, bool doA, bool doB, Epi epi, hipStream_t s, int nz = 1,
                         size_t zstride = 0, int force_qw = 0) {
    const NnGrid g = nn_grid(na, nb, doA, doB, force_qw);
    if (g.blocksA + g.blocksB == 0) return;
    if (g.qw == 8)
        hipLaunchKernelGGL((k_nn_l1<8, IdxT, Epi>), dim3(g.blocksA + g.blocksB, 1, nz), dim3(NN_BLOCK), g.smem, s, A, na, sa,
                           B, nb, sb, dA, iA, dB, iB, g.blocksA, epi, zstride);
    else
        hipLaunchKernelGGL((k_nn_l1<4, IdxT, Epi>), dim3(g.blocksA + g.blocksB, 1, nz), dim3(NN_BLOCK), g.smem, s, A, na, sa,
                           B, nb, sb, dA, iA, dB, iB, g.blocksA, epi, zstride);
}

}  // namespace creg
