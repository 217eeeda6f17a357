// se3.hip -- K5: SE(3) <-> dual quaternion rows and the pytorch3d.transforms rows on the path.
// One thread per row; these are K-row (20..128) conversions, latency not bandwidth matters.
// Replaces dq_func.py:29-257 and pytorch3d matrix_to_quaternion / quaternion_to_matrix.
#include <cfloat>
#include "creg_common.h"
#include "creg_dev.h"

namespace creg {

enum Op { SE3_TO_DQ, DQ_TO_SE3, DQ_TO_SE3_BWD, DQ_MUL, DQ_INV, DQ_TO_QT, QT_TO_DQ, M_TO_Q, Q_TO_M };

template <int OP>
__global__ __launch_bounds__(64) void k_rows(const float* __restrict__ a, const float* __restrict__ b,
                                             int k, float* __restrict__ o, float* __restrict__ o2) {
    const int r = blockIdx.x * 64 + threadIdx.x;
    if (r >= k) return;
    if constexpr (OP == SE3_TO_DQ) {
        const float* M = a + 16 * r;
        const float R[9] = {M[0], M[1], M[2], M[4], M[5], M[6], M[8], M[9], M[10]};
        const float t[3] = {M[3], M[7], M[11]};
        float dq[8];
        se3_to_dq(R, t, dq, FLT_EPSILON);
        for (int i = 0; i < 8; ++i) o[8 * r + i] = dq[i];
    } else if constexpr (OP == DQ_TO_SE3) {
        float dq[8], R[9], t[3];
        for (int i = 0; i < 8; ++i) dq[i] = a[8 * r + i];
        dq_to_se3(dq, R, t);
        float* M = o + 16 * r;
        for (int i = 0; i < 3; ++i) { M[4 * i] = R[3 * i]; M[4 * i + 1] = R[3 * i + 1]; M[4 * i + 2] = R[3 * i + 2]; M[4 * i + 3] = t[i]; }
        M[12] = 0.f; M[13] = 0.f; M[14] = 0.f; M[15] = 1.f;
    } else if constexpr (OP == DQ_TO_SE3_BWD) {
        float dq[8], G[9], gt[3], gdq[8];
        for (int i = 0; i < 8; ++i) dq[i] = a[8 * r + i];
        const float* gM = b + 16 * r;
        for (int i = 0; i < 3; ++i) { G[3 * i] = gM[4 * i]; G[3 * i + 1] = gM[4 * i + 1]; G[3 * i + 2] = gM[4 * i + 2]; gt[i] = gM[4 * i + 3]; }
        dq_to_se3_vjp(dq, G, gt, gdq);
        for (int i = 0; i < 8; ++i) o[8 * r + i] = gdq[i];
    } else if constexpr (OP == DQ_MUL) {
        const float* x = a + 8 * r; const float* y = b + 8 * r;
        float re[4], d1[4], d2[4];
        quat_mul(x, y, re); quat_mul(x, y + 4, d1); quat_mul(x + 4, y, d2);
        for (int i = 0; i < 4; ++i) { o[8 * r + i] = re[i]; o[8 * r + 4 + i] = d1[i] + d2[i]; }
    } else if constexpr (OP == DQ_INV) {
        // dq_func.py:213-236: conj(real)/|real|^2 ; conj(dual)/|real|^2 - 2 conj(real) <real,dual>/|real|^4
        const float* x = a + 8 * r;
        const float nrm = sqrtf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3]);
        float n2 = nrm * nrm;
        n2 = n2 > FLT_EPSILON ? n2 : FLT_EPSILON;
        const float dot = (x[0] * x[4] + x[1] * x[5] + x[2] * x[6] + x[3] * x[7]) / (n2 * n2);
        for (int i = 0; i < 4; ++i) {
            const float sg = i == 0 ? 1.f : -1.f;
            o[8 * r + i] = sg * x[i] / n2;
            o[8 * r + 4 + i] = sg * x[4 + i] / n2 - 2.f * (sg * x[i]) * dot;
        }
    } else if constexpr (OP == DQ_TO_QT) {
        // dq_func.py:126-146 (q is real (x) dual, as the reference computes it)
        const float* x = a + 8 * r;
        float q[4], tt[4];
        quat_mul(x, x + 4, q);
        const float c[4] = {x[0], -x[1], -x[2], -x[3]};
        quat_mul(x + 4, c, tt);
        for (int i = 0; i < 4; ++i) o[4 * r + i] = q[i];
        for (int i = 0; i < 3; ++i) o2[3 * r + i] = 2.f * tt[1 + i];
    } else if constexpr (OP == QT_TO_DQ) {
        const float* q = a + 4 * r;
        const float p[4] = {0.f, b[3 * r], b[3 * r + 1], b[3 * r + 2]};
        float d[4];
        quat_mul(p, q, d);
        for (int i = 0; i < 4; ++i) { o[8 * r + i] = q[i]; o[8 * r + 4 + i] = 0.5f * d[i]; }
    } else if constexpr (OP == M_TO_Q) {
        float R[9], q[4];
        for (int i = 0; i < 9; ++i) R[i] = a[9 * r + i];
        matrix_to_quat(R, q);
        for (int i = 0; i < 4; ++i) o[4 * r + i] = q[i];
    } else if constexpr (OP == Q_TO_M) {
        float q[4], R[9];
        for (int i = 0; i < 4; ++i) q[i] = a[4 * r + i];
        quat_to_matrix(q, R);
        for (int i = 0; i < 9; ++i) o[9 * r + i] = R[i];
    }
}

template <int OP>
static int run(const float* a, const float* b, int k, float* o, float* o2, creg_stream_t s, const char* name) {
    if (!a || !o || k < 0) { set_error("%s: bad argument", name); return CREG_EINVAL; }
    if (k == 0) return CREG_OK;
    hipLaunchKernelGGL((k_rows<OP>), dim3(cdiv(k, 64)), dim3(64), 0, (hipStream_t)s, a, b, k, o, o2);
    CREG_LAUNCH_CHECK();
    return CREG_OK;
}

}  // namespace creg
using namespace creg;

extern "C" {
int creg_se3_to_dq_f32(const float* M, int32_t k, float* dq, creg_stream_t s) { return run<SE3_TO_DQ>(M, nullptr, k, dq, nullptr, s, __func__); }
int creg_dq_to_se3_f32(const float* dq, int32_t k, float* M, creg_stream_t s) { return run<DQ_TO_SE3>(dq, nullptr, k, M, nullptr, s, __func__); }
int creg_dq_to_se3_bwd_f32(const float* dq, const float* gM, int32_t k, float* gdq, creg_stream_t s) {
    CREG_REQUIRE(gM, "creg_dq_to_se3_bwd_f32: null grad_M");
    return run<DQ_TO_SE3_BWD>(dq, gM, k, gdq, nullptr, s, __func__);
}
int creg_dq_multiply_f32(const float* a, const float* b, int32_t k, float* o, creg_stream_t s) {
    CREG_REQUIRE(b, "creg_dq_multiply_f32: null operand");
    return run<DQ_MUL>(a, b, k, o, nullptr, s, __func__);
}
int creg_dq_invert_f32(const float* dq, int32_t k, float* o, creg_stream_t s) { return run<DQ_INV>(dq, nullptr, k, o, nullptr, s, __func__); }
int creg_dq_to_quat_trans_f32(const float* dq, int32_t k, float* q, float* t, creg_stream_t s) {
    CREG_REQUIRE(t, "creg_dq_to_quat_trans_f32: null output");
    return run<DQ_TO_QT>(dq, nullptr, k, q, t, s, __func__);
}
int creg_quat_trans_to_dq_f32(const float* q, const float* t, int32_t k, float* dq, creg_stream_t s) {
    CREG_REQUIRE(t, "creg_quat_trans_to_dq_f32: null operand");
    return run<QT_TO_DQ>(q, t, k, dq, nullptr, s, __func__);
}
int creg_matrix_to_quat_f32(const float* R, int32_t k, float* q, creg_stream_t s) { return run<M_TO_Q>(R, nullptr, k, q, nullptr, s, __func__); }
int creg_quat_to_matrix_f32(const float* q, int32_t k, float* R, creg_stream_t s) { return run<Q_TO_M>(q, nullptr, k, R, nullptr, s, __func__); }
}
