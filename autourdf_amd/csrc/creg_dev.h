// creg_dev.h -- device-side math shared by the kernels (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>

namespace creg {

constexpr int WAVE = 64;

// ---- wave / block reductions with a fixed combination order (deterministic) ------------------
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, WAVE);
    return v;
}

// DPP row reductions (gfx9 family): 4 intra-row steps + row_bcast15/31; total lands in lane 63.
// ~6 VALU ops instead of 6 LDS-crossbar bpermutes; the combination order is fixed.
#define CREG_DPP_STEP(v, ctrl, mask) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, mask, 0xF, false))
// sum over the wave, result broadcast to every lane (wave-uniform)
__device__ __forceinline__ float wave_sum_fast(float v) {
    CREG_DPP_STEP(v, 0xB1, 0xF);    // quad_perm [1,0,3,2]
    CREG_DPP_STEP(v, 0x4E, 0xF);    // quad_perm [2,3,0,1]
    CREG_DPP_STEP(v, 0x141, 0xF);   // row_half_mirror
    CREG_DPP_STEP(v, 0x140, 0xF);   // row_mirror      -> every lane holds its row's sum
    CREG_DPP_STEP(v, 0x142, 0xA);   // row_bcast15 into rows 1,3
    CREG_DPP_STEP(v, 0x143, 0xC);   // row_bcast31 into rows 2,3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// fp64 flavour: the two halves travel through the same DPP moves (old = 0 in both halves is +0.0)
#define CREG_DPP_STEP_F64(v, ctrl, mask)                                                                   \
    {                                                                                                      \
        const int lo_ = __builtin_amdgcn_update_dpp(0, __double2loint(v), ctrl, mask, 0xF, false);         \
        const int hi_ = __builtin_amdgcn_update_dpp(0, __double2hiint(v), ctrl, mask, 0xF, false);         \
        v += __hiloint2double(hi_, lo_);                                                                   \
    }
__device__ __forceinline__ double wave_sum_fast(double v) {
    CREG_DPP_STEP_F64(v, 0xB1, 0xF)
    CREG_DPP_STEP_F64(v, 0x4E, 0xF)
    CREG_DPP_STEP_F64(v, 0x141, 0xF)
    CREG_DPP_STEP_F64(v, 0x140, 0xF)
    CREG_DPP_STEP_F64(v, 0x142, 0xA)
    CREG_DPP_STEP_F64(v, 0x143, 0xC)
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

// Cooperative global -> LDS copy of n float4 by NT threads through the LDS-DMA path
// (global_load_lds_dwordx4: 1 KiB per wave-instruction, no VGPR round trip).  Every request is in
// flight before the single wait, so the copy costs ONE memory round trip.  The register-staged
// alternative (`v[q] = src[..]; ...; dst[..] = v[q]`) does not survive hipcc: it sinks each load to
// its LDS write and serialises the pairs (one ~1-2 us cold round trip each), and fences either get
// bypassed (__restrict__) or push the array to scratch.
// dst must be 16-byte aligned and have room for n float4 (lanes past n are masked off).
// B is kept as a template argument for call-site documentation only.
template <int NT>
__device__ __forceinline__ void stage_issue(float4* dst, const float4* src, int n) {
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
    for (int c = wave; c * 64 < n; c += NT / 64) {
        const int i = c * 64 + lane;
        // lanes past the end stay inactive: an LDS-DMA instruction writes lane l's 16 bytes at base + 16 l for the
        // ACTIVE lanes only, so nothing lands beyond dst + n.  (v1 clamped the source index instead and let the last
        // instruction write a full 1 KiB: in k_bwd2 that tail spilled over the NEXT LDS array, and whether the DMA
        // or the other waves' ds_writes to that array landed last was a race -- lost only under memory contention.)
        if (i < n)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i),
                                             (__attribute__((address_space(3))) void*)(dst + c * 64), 16, 0, 0);
    }
}
// wait for this wave's outstanding DMA (and loads); follow with __syncthreads() before other waves read
__device__ __forceinline__ void stage_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <int NT, int B>
__device__ __forceinline__ void stage_f4(float4* dst, const float4* src, int n) {
    stage_issue<NT>(dst, src, n);
    stage_wait();
}

// (value, index) lexicographic minimum over the wave with DPP moves instead of LDS-crossbar
// bpermutes (same tree as wave_sum_fast); the result is returned wave-uniform.
#define CREG_DPP_ARGMIN_STEP(ctrl, mask)                                                              \
    {                                                                                                 \
        const float ov = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), ctrl, mask, 0xF, false)); \
        const int oi = __builtin_amdgcn_update_dpp(i, i, ctrl, mask, 0xF, false);                      \
        const bool take = (ov < v) || (ov == v && oi < i);                                            \
        v = take ? ov : v;                                                                            \
        i = take ? oi : i;                                                                            \
    }
__device__ __forceinline__ void wave_argmin_fast(float& v, int& i) {
    CREG_DPP_ARGMIN_STEP(0xB1, 0xF)
    CREG_DPP_ARGMIN_STEP(0x4E, 0xF)
    CREG_DPP_ARGMIN_STEP(0x141, 0xF)
    CREG_DPP_ARGMIN_STEP(0x140, 0xF)
    CREG_DPP_ARGMIN_STEP(0x142, 0xA)
    CREG_DPP_ARGMIN_STEP(0x143, 0xC)
    v = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
    i = __builtin_amdgcn_readlane(i, 63);
}

// Wave-uniform minimum of a float / int over the 64 lanes (same DPP tree as wave_argmin_fast), as v_min_*_dpp instructions: through
// __builtin_amdgcn_update_dpp + fminf hipcc emits FIVE instructions a step (v_mov_b32_dpp, a canonicalising v_max, v_min, a copy, s_nop),
// and the nearest-neighbour search runs ~17 of these reductions per wave -- a third of its instructions.  Hazards are ours inside the
// asm: a DPP operand written by the previous VALU instruction needs 2 wait states (s_nop 1 between the steps), a DPP instruction
// after an EXEC write 5 (the leading s_nop 4 covers both for whatever precedes).  row_mask limits the last two steps to the rows
// that receive a broadcast, the others keep their value -- what update_dpp(old = v) did.
__device__ __forceinline__ float wave_min_fast(float v) {
    asm("s_nop 4\n\t"
        "v_min_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_min_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_min_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_min_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_min_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
        "v_min_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"
        : "+v"(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ int wave_min_fast(int v) {
    asm("s_nop 4\n\t"
        "v_min_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_min_i32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_min_i32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_min_i32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_min_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
        "v_min_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"
        : "+v"(v));
    return __builtin_amdgcn_readlane(v, 63);
}

// (value, index) lexicographic minimum across the wave: smallest value, then smallest index.
__device__ __forceinline__ void wave_argmin(float& v, int& i) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        float ov = __shfl_xor(v, off, WAVE);
        int oi = __shfl_xor(i, off, WAVE);
        bool take = (ov < v) || (ov == v && oi < i);
        v = take ? ov : v;
        i = take ? oi : i;
    }
}

// Sum `v` over a block of NT threads; result valid in thread 0. `scratch` >= NT/64 entries.
template <typename T, int NT>
__device__ __forceinline__ T block_sum(T v, T* scratch) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    T r = T(0);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NT / 64; ++i) r += scratch[i];
    }
    return r;
}

// ---- L1 distance in the canonical order (pytorch3d knn: dist += |diff| for d = 0,1,2) ---------
__device__ __forceinline__ float l1_dist(float ax, float ay, float az, float bx, float by, float bz) {
    float d = fabsf(ax - bx);
    d = d + fabsf(ay - by);
    d = d + fabsf(az - bz);
    return d;
}

// ---- quaternion helpers, pytorch3d.transforms conventions (real first) -----------------------
template <typename T>
__device__ __forceinline__ void quat_to_matrix(const T q[4], T R[9]) {
    const T w = q[0], x = q[1], y = q[2], z = q[3];
    const T s = T(2) / (w * w + x * x + y * y + z * z);
    R[0] = T(1) - s * (y * y + z * z); R[1] = s * (x * y - z * w); R[2] = s * (x * z + y * w);
    R[3] = s * (x * y + z * w); R[4] = T(1) - s * (x * x + z * z); R[5] = s * (y * z - x * w);
    R[6] = s * (x * z - y * w); R[7] = s * (y * z + x * w); R[8] = T(1) - s * (x * x + y * y);
}

// vector-Jacobian product of quat_to_matrix: G = dL/dR (row-major 3x3) -> gq = dL/dq.
template <typename T>
__device__ __forceinline__ void quat_to_matrix_vjp(const T q[4], const T G[9], T gq[4]) {
    const T w = q[0], x = q[1], y = q[2], z = q[3];
    const T n = w * w + x * x + y * y + z * z;
    const T s = T(2) / n;
    // R = diag-part + s*N(q); dL/ds = sum G_ij N_ij
    const T N00 = -(y * y + z * z), N01 = x * y - z * w, N02 = x * z + y * w;
    const T N10 = x * y + z * w, N11 = -(x * x + z * z), N12 = y * z - x * w;
    const T N20 = x * z - y * w, N21 = y * z + x * w, N22 = -(x * x + y * y);
    const T gs = G[0] * N00 + G[1] * N01 + G[2] * N02 + G[3] * N10 + G[4] * N11 + G[5] * N12 +
                 G[6] * N20 + G[7] * N21 + G[8] * N22;
    const T dw = -z * G[1] + y * G[2] + z * G[3] - x * G[5] - y * G[6] + x * G[7];
    const T dx = y * (G[1] + G[3]) + z * (G[2] + G[6]) - T(2) * x * (G[4] + G[8]) + w * (G[7] - G[5]);
    const T dy = -T(2) * y * (G[0] + G[8]) + x * (G[1] + G[3]) + w * (G[2] - G[6]) + z * (G[5] + G[7]);
    const T dz = -T(2) * z * (G[0] + G[4]) + w * (G[3] - G[1]) + x * (G[2] + G[6]) + y * (G[5] + G[7]);
    const T k = -gs * s * s;     // ds/dq_c = -s^2 q_c
    gq[0] = s * dw + k * w; gq[1] = s * dx + k * x; gq[2] = s * dy + k * y; gq[3] = s * dz + k * z;
}

template <typename T>
__device__ __forceinline__ T sqrt_pos(T v) { return v > T(0) ? sqrt(v) : T(0); }

// R row-major 3x3 -> q (w >= 0), best-conditioned of the four candidates (first max on ties).
template <typename T>
__device__ __forceinline__ void matrix_to_quat(const T R[9], T q[4]) {
    const T m00 = R[0], m01 = R[1], m02 = R[2], m10 = R[3], m11 = R[4], m12 = R[5], m20 = R[6],
            m21 = R[7], m22 = R[8];
    T qa[4] = {sqrt_pos(T(1) + m00 + m11 + m22), sqrt_pos(T(1) + m00 - m11 - m22),
               sqrt_pos(T(1) - m00 + m11 - m22), sqrt_pos(T(1) - m00 - m11 + m22)};
    int c = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i) if (qa[i] > qa[c]) c = i;
    T cand[4];
    if (c == 0) { cand[0] = qa[0] * qa[0]; cand[1] = m21 - m12; cand[2] = m02 - m20; cand[3] = m10 - m01; }
    else if (c == 1) { cand[0] = m21 - m12; cand[1] = qa[1] * qa[1]; cand[2] = m10 + m01; cand[3] = m02 + m20; }
    else if (c == 2) { cand[0] = m02 - m20; cand[1] = m10 + m01; cand[2] = qa[2] * qa[2]; cand[3] = m12 + m21; }
    else { cand[0] = m10 - m01; cand[1] = m20 + m02; cand[2] = m21 + m12; cand[3] = qa[3] * qa[3]; }
    const T den = T(2) * (qa[c] > T(0.1) ? qa[c] : T(0.1));
    const bool neg = (cand[0] / den) < T(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) { T v = cand[i] / den; q[i] = neg ? -v : v; }
}

// Hamilton product a (x) b, real first.
template <typename T>
__device__ __forceinline__ void quat_mul(const T a[4], const T b[4], T o[4]) {
    o[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    o[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    o[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
    o[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
}

// (R,t) -> dual quaternion, dq_func.py:72-98: q = m2q(R) / max(|q|, eps); dual = 0.5 (0,t) (x) q.
template <typename T>
__device__ __forceinline__ void se3_to_dq(const T R[9], const T t[3], T dq[8], T eps) {
    T q[4];
    matrix_to_quat(R, q);
    T n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    n = n > eps ? n : eps;
#pragma unroll
    for (int i = 0; i < 4; ++i) dq[i] = q[i] / n;
    const T p[4] = {T(0), t[0], t[1], t[2]};
    T d[4];
    quat_mul(p, dq, d);
#pragma unroll
    for (int i = 0; i < 4; ++i) dq[4 + i] = T(0.5) * d[i];
}

// dual quaternion -> (R,t), dq_func.py:148-168: R = q2m(real); t = (2 dual (x) conj(real))[1:].
template <typename T>
__device__ __forceinline__ void dq_to_se3(const T dq[8], T R[9], T t[3]) {
    quat_to_matrix(dq, R);
    const T c[4] = {dq[0], -dq[1], -dq[2], -dq[3]};
    T o[4];
    quat_mul(dq + 4, c, o);
    t[0] = T(2) * o[1]; t[1] = T(2) * o[2]; t[2] = T(2) * o[3];
}

// VJP of dq_to_se3: (G = dL/dR, gt = dL/dt) -> gdq (8).
template <typename T>
__device__ __forceinline__ void dq_to_se3_vjp(const T dq[8], const T G[9], const T gt[3], T gdq[8]) {
    quat_to_matrix_vjp(dq, G, gdq);
    const T rw = dq[0], rx = dq[1], ry = dq[2], rz = dq[3];
    const T dw = dq[4], dx = dq[5], dy = dq[6], dz = dq[7];
    const T a = gt[0], b = gt[1], c = gt[2];
    gdq[4] = T(2) * (-rx * a - ry * b - rz * c);
    gdq[5] = T(2) * (rw * a + rz * b - ry * c);
    gdq[6] = T(2) * (-rz * a + rw * b + rx * c);
    gdq[7] = T(2) * (ry * a - rx * b + rw * c);
    gdq[0] += T(2) * (dx * a + dy * b + dz * c);
    gdq[1] += T(2) * (-dw * a - dz * b + dy * c);
    gdq[2] += T(2) * (dz * a - dw * b - dx * c);
    gdq[3] += T(2) * (-dy * a + dx * b - dw * c);
}

// ---- the two optional pose representations of mlp_reg.py:72-76 / 86-90 (--r 6d, --r rpy), pytorch3d.transforms conventions ----
// rotation_6d_to_matrix: rows b1 = normalize(a1), b2 = normalize(a2 - (b1 . a2) b1), b3 = b1 x b2  (F.normalize: v / max(|v|, 1e-12))
template <typename T>
__device__ __forceinline__ void rot6d_to_matrix(const T d6[6], T R[9]) {
    const T n1 = sqrt(d6[0] * d6[0] + d6[1] * d6[1] + d6[2] * d6[2]), i1 = T(1) / (n1 > T(1e-12) ? n1 : T(1e-12));
    const T b1[3] = {d6[0] * i1, d6[1] * i1, d6[2] * i1};
    const T dot = b1[0] * d6[3] + b1[1] * d6[4] + b1[2] * d6[5];
    const T u[3] = {d6[3] - dot * b1[0], d6[4] - dot * b1[1], d6[5] - dot * b1[2]};
    const T n2 = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]), i2 = T(1) / (n2 > T(1e-12) ? n2 : T(1e-12));
    const T b2[3] = {u[0] * i2, u[1] * i2, u[2] * i2};
    R[0] = b1[0]; R[1] = b1[1]; R[2] = b1[2]; R[3] = b2[0]; R[4] = b2[1]; R[5] = b2[2];
    R[6] = b1[1] * b2[2] - b1[2] * b2[1]; R[7] = b1[2] * b2[0] - b1[0] * b2[2]; R[8] = b1[0] * b2[1] - b1[1] * b2[0];
}

// vector-Jacobian product of rot6d_to_matrix: G = dL/dR (row-major) -> g6 = dL/d(d6)
template <typename T>
__device__ __forceinline__ void rot6d_to_matrix_vjp(const T d6[6], const T G[9], T g6[6]) {
    const T n1 = sqrt(d6[0] * d6[0] + d6[1] * d6[1] + d6[2] * d6[2]), c1 = n1 > T(1e-12) ? n1 : T(1e-12);
    const T b1[3] = {d6[0] / c1, d6[1] / c1, d6[2] / c1};
    const T dot = b1[0] * d6[3] + b1[1] * d6[4] + b1[2] * d6[5];
    const T u[3] = {d6[3] - dot * b1[0], d6[4] - dot * b1[1], d6[5] - dot * b1[2]};
    const T n2 = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]), c2 = n2 > T(1e-12) ? n2 : T(1e-12);
    const T b2[3] = {u[0] / c2, u[1] / c2, u[2] / c2};
    const T* g3 = G + 6;
    // b3 = b1 x b2:  dL/db1 += b2 x g3,  dL/db2 += g3 x b1
    T gb1[3] = {G[0] + (b2[1] * g3[2] - b2[2] * g3[1]), G[1] + (b2[2] * g3[0] - b2[0] * g3[2]), G[2] + (b2[0] * g3[1] - b2[1] * g3[0])};
    const T gb2[3] = {G[3] + (g3[1] * b1[2] - g3[2] * b1[1]), G[4] + (g3[2] * b1[0] - g3[0] * b1[2]), G[5] + (g3[0] * b1[1] - g3[1] * b1[0])};
    // b2 = u / max(|u|, eps)
    T gu[3];
    if (n2 > T(1e-12)) {
        const T d = b2[0] * gb2[0] + b2[1] * gb2[1] + b2[2] * gb2[2];
        for (int i = 0; i < 3; ++i) gu[i] = (gb2[i] - b2[i] * d) / n2;
    } else {
        for (int i = 0; i < 3; ++i) gu[i] = gb2[i] / T(1e-12);
    }
    // u = a2 - (b1 . a2) b1
    const T gdot = -(gu[0] * b1[0] + gu[1] * b1[1] + gu[2] * b1[2]);
    for (int i = 0; i < 3; ++i) { gb1[i] += gdot * d6[3 + i] - dot * gu[i]; g6[3 + i] = gu[i] + gdot * b1[i]; }
    // b1 = a1 / max(|a1|, eps)
    if (n1 > T(1e-12)) {
        const T d = b1[0] * gb1[0] + b1[1] * gb1[1] + b1[2] * gb1[2];
        for (int i = 0; i < 3; ++i) g6[i] = (gb1[i] - b1[i] * d) / n1;
    } else {
        for (int i = 0; i < 3; ++i) g6[i] = gb1[i] / T(1e-12);
    }
}

// euler_angles_to_matrix(e, "XYZ") = Rx(a) Ry(b) Rz(c)
template <typename T>
__device__ __forceinline__ void euler_xyz_to_matrix(const T e[3], T R[9]) {
    const T sa = sin(e[0]), ca = cos(e[0]), sb = sin(e[1]), cb = cos(e[1]), sc = sin(e[2]), cc = cos(e[2]);
    R[0] = cb * cc; R[1] = -cb * sc; R[2] = sb;
    R[3] = ca * sc + sa * sb * cc; R[4] = ca * cc - sa * sb * sc; R[5] = -sa * cb;
    R[6] = sa * sc - ca * sb * cc; R[7] = sa * cc + ca * sb * sc; R[8] = ca * cb;
}

template <typename T>
__device__ __forceinline__ void euler_xyz_to_matrix_vjp(const T e[3], const T G[9], T ge[3]) {
    const T sa = sin(e[0]), ca = cos(e[0]), sb = sin(e[1]), cb = cos(e[1]), sc = sin(e[2]), cc = cos(e[2]);
    T R[9];
    euler_xyz_to_matrix(e, R);
    // d/da: row 1 -> -row 2, row 2 -> row 1
    ge[0] = -(G[3] * R[6] + G[4] * R[7] + G[5] * R[8]) + (G[6] * R[3] + G[7] * R[4] + G[8] * R[5]);
    ge[1] = G[0] * (-sb * cc) + G[1] * (sb * sc) + G[2] * cb + G[3] * (sa * cb * cc) + G[4] * (-sa * cb * sc) + G[5] * (sa * sb) +
            G[6] * (-ca * cb * cc) + G[7] * (ca * cb * sc) + G[8] * (-ca * sb);
    ge[2] = G[0] * (-cb * sc) + G[1] * (-cb * cc) + G[3] * (ca * cc - sa * sb * sc) + G[4] * (-ca * sc - sa * sb * cc) +
            G[6] * (sa * cc + ca * sb * sc) + G[7] * (-sa * sc + ca * sb * cc);
}

// matrix_to_euler_angles(R, "XYZ"): R02 = sin b, R12 = -sin a cos b, R22 = cos a cos b, R01 = -cos b sin c, R00 = cos b cos c
template <typename T>
__device__ __forceinline__ void matrix_to_euler_xyz(const T R[9], T e[3]) {
    e[0] = atan2(-R[5], R[8]); e[1] = asin(R[2]); e[2] = atan2(-R[1], R[0]);
}

// 16 bytes written through (sc1): the line goes to the memory side while the kernel is still running instead of waiting,
// dirty, for the write-back the end of the kernel performs -- a kernel that rewrites tens of MB (the Adam state) otherwise
// pays that drain at its boundary (MI355X_MICROARCH.md, "boundary": + B / 6 TB/s behind B dirty bytes).
// `base` must be wave-uniform (it becomes the buffer descriptor), `elem` is the lane's float index from it.  A buffer store
// with sc1 (aux bit 4) is the 16-byte write-through store the compiler counts like any other (the agent-scope atomic store is
// 8 bytes at most; an inline-asm global_store ... sc1 would be invisible to the s_waitcnt insertion).  Measured on the train
// plan's k_bd (15 MB of parameters and Adam moments per launch): headline 145.7 -> 150.1 frames/s; a non-temporal store: no gain.
__device__ __forceinline__ void st4_wt(float* base, int elem, float4 v) {
#ifdef CREG_ST4_PLAIN                                            // A/B measurement build only
    *(float4*)(base + elem) = v; return;
#endif
    typedef int int4v __attribute__((ext_vector_type(4)));
    const int4v x = {__float_as_int(v.x), __float_as_int(v.y), __float_as_int(v.z), __float_as_int(v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(x, __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000), elem * 4, 0, 16);
}


// Raw buffer loads: base in four SGPRs, ONE 32-bit byte offset per lane, a scalar offset and an immediate -- a dozen loads of a
// strided pattern share one address VGPR (global_load needs a 64-bit address pair per distinct row), and bytes past `bytes` read as 0.
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_rsrc(const void* base, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float ld_buf(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ float4 ld_buf4(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    const u32x4v v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
}

}  // namespace creg
