// sample.hip -- N4: area-weighted surface sampling of an articulated triangle-mesh robot, fp64.
// The data-generation side of the path (reference Sim/sim_data.py:246-370 renders depth images of the
// PyBullet model from 20 cameras and fuses them; this build samples the same URDF + mesh surfaces
// directly): one thread per output point picks a triangle by inverting the cumulative-area table,
// a uniform barycentric point on it, and moves it by its link's pose of the current joint state.
//   f   = first triangle with cum_area[f] > u0 * cum_area[F-1]
//   s   = sqrt(u1);  p = (1 - s) A + (s (1 - u2)) B + (s u2) C         (uniform on the triangle)
//   out = R_link p + t_link
// Every operation is written out in a fixed order without contraction, so the host restatement
// (oracle/sim_data.py, numpy) reproduces the points bit for bit.
#include "creg_common.h"
#include "creg_dev.h"

namespace creg {

__global__ __launch_bounds__(256) void k_sample_mesh(const double* __restrict__ tri, const double* __restrict__ cum_area,
                                                     const int* __restrict__ tri_link, const double* __restrict__ link_T,
                                                     const double* __restrict__ u, int64_t n, int F, int n_links,
                                                     double* __restrict__ out, int* __restrict__ link_out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double u0 = u[3 * i], u1 = u[3 * i + 1], u2 = u[3 * i + 2];
    const double target = u0 * cum_area[F - 1];
    int lo = 0, hi = F - 1;                              // upper bound: first f with cum_area[f] > target
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cum_area[mid] > target) hi = mid; else lo = mid + 1;
    }
    const double* t = tri + 9 * (size_t)lo;
    const double s = sqrt(u1);
    const double b0 = 1.0 - s, b1 = s * (1.0 - u2), b2 = s * u2;
    double p[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) p[d] = (b0 * t[d] + b1 * t[3 + d]) + b2 * t[6 + d];
    const int l = min(max(tri_link[lo], 0), n_links - 1);   // a link index outside [0, n_links) would read past link_T: clamped (bad input stays bounded)
    const double* T = link_T + 16 * (size_t)l;
#pragma unroll
    for (int d = 0; d < 3; ++d) out[3 * i + d] = ((T[4 * d] * p[0] + T[4 * d + 1] * p[1]) + T[4 * d + 2] * p[2]) + T[4 * d + 3];
    if (link_out) link_out[i] = l;
}

// ---- camera-ring visibility (the occlusion the reference's rendered depth cameras impose, sim_data.py:88-116,246-306) ----
// The reference fuses depth images of `num_cameras` pinhole cameras on a sphere around the robot (fov 60 degrees, aspect 1,
// near 0.1, far 4, 800 x 800): a surface point is in its data only if some camera sees it.  Here the posed triangles
// are rasterised into one depth buffer per camera (linear depth along the view axis, perspective-correct at the pixel
// centres, fp64, minimum by integer atomics on the bit pattern of the positive doubles: order independent) and a sample
// survives if, for at least one camera, it projects inside the image between near and far and is not more than `eps`
// behind the buffer at its pixel.  cams (C,12) = eye | forward | right | up (unit vectors).
// All arithmetic is spelled out without contraction so the numpy restatement (oracle/sim_data.py) gives the same buffers.
struct CamParams { double tan_half, aspect, near_v, far_v; int W, H; };

__device__ __forceinline__ void cam_project(const double* cam, const double* p, double& xc, double& yc, double& d) {
    const double r0 = p[0] - cam[0], r1 = p[1] - cam[1], r2 = p[2] - cam[2];
    d = (r0 * cam[3] + r1 * cam[4]) + r2 * cam[5];
    xc = (r0 * cam[6] + r1 * cam[7]) + r2 * cam[8];
    yc = (r0 * cam[9] + r1 * cam[10]) + r2 * cam[11];
}
// continuous pixel coordinates (pixel centres at +0.5)
__device__ __forceinline__ void cam_pixel(const CamParams& c, double xc, double yc, double d, double& px, double& py) {
    const double nx = xc / ((d * c.tan_half) * c.aspect), ny = yc / (d * c.tan_half);
    px = (nx * 0.5 + 0.5) * (double)c.W;
    py = (1.0 - (ny * 0.5 + 0.5)) * (double)c.H;
}

__global__ __launch_bounds__(256) void k_depth_clear(unsigned long long* __restrict__ z, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) z[i] = 0x7FF0000000000000ull;            // +inf
}

// grid (ceil(F / 256), C): thread = (triangle, camera)
__global__ __launch_bounds__(256) void k_raster_depth(const double* __restrict__ tri, const int* __restrict__ tri_link, int F, int n_links,
                                                      const double* __restrict__ link_T, const double* __restrict__ cams,
                                                      CamParams c, unsigned long long* __restrict__ zbuf) {
    const int f = blockIdx.x * 256 + threadIdx.x, cam_id = blockIdx.y;
    if (f >= F) return;
    const double* cam = cams + 12 * cam_id;
    const double* T = link_T + 16 * (size_t)min(max(tri_link[f], 0), n_links - 1);
    double X[3], Y[3], D[3];
#pragma unroll
    for (int v = 0; v < 3; ++v) {
        const double* q = tri + 9 * (size_t)f + 3 * v;
        double w[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) w[a] = ((T[4 * a] * q[0] + T[4 * a + 1] * q[1]) + T[4 * a + 2] * q[2]) + T[4 * a + 3];
        double xc, yc;
        cam_project(cam, w, xc, yc, D[v]);
        if (!(D[v] >= c.near_v)) return;                // a vertex at or behind the near plane: the facet is not drawn (NO clipping
                                                        // against the near plane: a large facet that reaches behind it stops occluding;
                                                        // the camera ring stands 10+ near distances off the robot, so none does)
        cam_pixel(c, xc, yc, D[v], X[v], Y[v]);
    }
    const double area = (X[1] - X[0]) * (Y[2] - Y[0]) - (X[2] - X[0]) * (Y[1] - Y[0]);
    if (area == 0.0) return;
    const double xmin = fmin(X[0], fmin(X[1], X[2])), xmax = fmax(X[0], fmax(X[1], X[2]));
    const double ymin = fmin(Y[0], fmin(Y[1], Y[2])), ymax = fmax(Y[0], fmax(Y[1], Y[2]));
    // pixel bounds clamped BEFORE the integer conversion (a huge projected coordinate would overflow the cast)
    const int x0 = (int)fmax(0.0, floor(xmin - 0.5)), x1 = (int)fmin((double)(c.W - 1), ceil(xmax - 0.5));
    const int y0 = (int)fmax(0.0, floor(ymin - 0.5)), y1 = (int)fmin((double)(c.H - 1), ceil(ymax - 0.5));
    if (!(xmin - 0.5 <= (double)c.W) || !(ymin - 0.5 <= (double)c.H) || !(xmax >= -1.0) || !(ymax >= -1.0)) return;   // off screen / NaN
    const double ia = 1.0 / area, i0 = 1.0 / D[0], i1 = 1.0 / D[1], i2 = 1.0 / D[2];
    unsigned long long* z = zbuf + (size_t)cam_id * c.W * c.H;
    for (int y = y0; y <= y1; ++y)
        for (int x = x0; x <= x1; ++x) {
            const double cx = (double)x + 0.5, cy = (double)y + 0.5;
            const double b0 = ((X[1] - cx) * (Y[2] - cy) - (X[2] - cx) * (Y[1] - cy)) * ia;
            const double b1 = ((X[2] - cx) * (Y[0] - cy) - (X[0] - cx) * (Y[2] - cy)) * ia;
            const double b2 = ((X[0] - cx) * (Y[1] - cy) - (X[1] - cx) * (Y[0] - cy)) * ia;
            if (b0 < 0.0 || b1 < 0.0 || b2 < 0.0) continue;
            const double d = 1.0 / ((b0 * i0 + b1 * i1) + b2 * i2);      // 1/depth is linear in screen space
            if (d >= c.near_v && d <= c.far_v) atomicMin(&z[(size_t)y * c.W + x], (unsigned long long)__double_as_longlong(d));
        }
}

__global__ __launch_bounds__(256) void k_visible(const double* __restrict__ pts, int64_t n, const double* __restrict__ cams, int C,
                                                 CamParams c, const unsigned long long* __restrict__ zbuf, double eps,
                                                 unsigned char* __restrict__ vis) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double p[3] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
    unsigned char seen = 0;
    for (int k = 0; k < C && !seen; ++k) {
        double xc, yc, d, px, py;
        cam_project(cams + 12 * k, p, xc, yc, d);
        if (!(d >= c.near_v && d <= c.far_v)) continue;
        cam_pixel(c, xc, yc, d, px, py);
        const int x = (int)floor(px), y = (int)floor(py);
        if (x < 0 || x >= c.W || y < 0 || y >= c.H) continue;
        const double zb = __longlong_as_double((long long)zbuf[((size_t)k * c.H + y) * c.W + x]);
        if (d <= zb + eps) seen = 1;
    }
    vis[i] = seen;
}

}  // namespace creg
using namespace creg;

extern "C" size_t creg_visibility_workspace_bytes(int32_t n_cams, int32_t width, int32_t height) {
    if (n_cams < 1 || width < 1 || height < 1) return 0;
    return sizeof(unsigned long long) * (size_t)n_cams * width * height;
}

extern "C" int creg_visibility_f64(const double* tri, const int32_t* tri_link, int32_t n_tri, const double* link_T, int32_t n_links,
                                   const double* cams, int32_t n_cams, double fov_deg, double aspect, double near_val, double far_val,
                                   int32_t width, int32_t height, const double* pts, int64_t n, double eps, uint8_t* visible,
                                   void* workspace, size_t workspace_bytes, creg_stream_t stream) {
    CREG_REQUIRE(tri && tri_link && link_T && cams && pts && visible && workspace, "creg_visibility_f64: null pointer");
    CREG_REQUIRE(n_tri >= 1 && n_links >= 1 && n_cams >= 1 && width >= 1 && height >= 1 && n >= 1 && fov_deg > 0 && fov_deg < 180 &&
                 near_val > 0 && far_val > near_val, "creg_visibility_f64: bad argument");
    const size_t need = creg_visibility_workspace_bytes(n_cams, width, height);
    CREG_REQUIRE(workspace_bytes >= need, "creg_visibility_f64: workspace too small (%zu < %zu)", workspace_bytes, need);
    hipStream_t s = (hipStream_t)stream;
    const size_t npx = (size_t)n_cams * width * height;
    CamParams c{tan(fov_deg * 3.14159265358979323846 / 360.0), aspect, near_val, far_val, width, height};
    hipLaunchKernelGGL(k_depth_clear, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, s, (unsigned long long*)workspace, npx);
    hipLaunchKernelGGL(k_raster_depth, dim3(cdiv(n_tri, 256), n_cams), dim3(256), 0, s, tri, tri_link, (int)n_tri, (int)n_links, link_T, cams, c,
                       (unsigned long long*)workspace);
    hipLaunchKernelGGL(k_visible, dim3((unsigned)cdiv(n, (int64_t)256)), dim3(256), 0, s, pts, n, cams, (int)n_cams, c,
                       (const unsigned long long*)workspace, eps, visible);
    CREG_LAUNCH_CHECK();
    return CREG_OK;
}

extern "C" int creg_sample_mesh_f64(const double* tri, const double* cum_area, const int32_t* tri_link, int32_t n_tri,
                                    const double* link_T, int32_t n_links, const double* u, int64_t n, double* out,
                                    int32_t* link_out, creg_stream_t stream) {
    CREG_REQUIRE(tri && cum_area && tri_link && link_T && u && out, "creg_sample_mesh_f64: null pointer");
    CREG_REQUIRE(n_tri >= 1 && n_links >= 1 && n >= 1, "creg_sample_mesh_f64: bad size");
    hipLaunchKernelGGL(k_sample_mesh, dim3((unsigned)cdiv(n, (int64_t)256)), dim3(256), 0, (hipStream_t)stream, tri, cum_area,
                       tri_link, link_T, u, n, (int)n_tri, (int)n_links, out, link_out);
    CREG_LAUNCH_CHECK();
    return CREG_OK;
}
