// What a launch that ONLY moves k_bd's bytes costs on this box: the practical floor next to which DESIGN.md section 5 reads the
// roofline fraction of the train plan's dominant kernel (k_bd: parameters + both Adam moments read, all three written; the
// headline shape moves 36.3 MB per launch of 3 problems).  No arithmetic beyond one fma per element, no LDS, no barrier.
//   hipcc -O3 --offload-arch=gfx950 tests/measure/stream_floor.hip -o /tmp/stream_floor && /tmp/stream_floor
// Prints microseconds per launch (HIP events around 200 back-to-back launches) for the grid k_bd uses and for finer / coarser ones,
// with plain and with write-through (sc1) 16-byte stores, then the same bytes as a read-only and a write-only launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef int int4v __attribute__((ext_vector_type(4)));
template <bool WT>
__device__ __forceinline__ void st4(float* base, int elem, float4 v) {
    if constexpr (WT) {
        const int4v x = {__float_as_int(v.x), __float_as_int(v.y), __float_as_int(v.z), __float_as_int(v.w)};
        __builtin_amdgcn_raw_buffer_store_b128(x, __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000), elem * 4, 0, 16);
    } else *(float4*)(base + elem) = v;
}

// MODE 0: read w, m, v, write w', m, v (the Adam traffic); 1: read only; 2: write only.  A wave owns `PER` slabs of 256 floats.
template <bool WT, int PER, int MODE>
__global__ __launch_bounds__(512) void k_stream(const float* __restrict__ w, float* __restrict__ wn, float* __restrict__ m, float* __restrict__ v, int n, float* sink) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * 512 + threadIdx.x) >> 6;
    float4 a[PER], b[PER], c[PER];
#pragma unroll
    for (int p = 0; p < PER; ++p) {
        const int i = (wave * PER + p) * 256 + lane * 4;
        if (MODE != 2 && i < n) { a[p] = *(const float4*)(w + i); b[p] = *(const float4*)(m + i); c[p] = *(const float4*)(v + i); }
        else a[p] = b[p] = c[p] = make_float4(1.f, 2.f, 3.f, 4.f);
    }
    float s = 0.f;
#pragma unroll
    for (int p = 0; p < PER; ++p) {
        const int i = (wave * PER + p) * 256 + lane * 4;
        a[p].x += fmaf(b[p].x, 0.999f, c[p].x); a[p].y += fmaf(b[p].y, 0.999f, c[p].y); a[p].z += fmaf(b[p].z, 0.999f, c[p].z); a[p].w += fmaf(b[p].w, 0.999f, c[p].w);
        if (MODE == 1) s += a[p].x + a[p].y + a[p].z + a[p].w;
        else if (i < n) { st4<WT>(wn, i, a[p]); st4<WT>(m, i, b[p]); st4<WT>(v, i, c[p]); }
    }
    if (MODE == 1 && s == 12345.678f) *sink = s;
}

template <bool WT, int PER, int MODE>
static int run(const char* what, int n, float* w, float* wn, float* m, float* v, float* sink) {
    const int waves = (n / 256 + PER - 1) / PER, blocks = (waves + 7) / 8;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k_stream<WT, PER, MODE>), dim3(blocks), dim3(512), 0, 0, (i & 1) ? wn : w, (i & 1) ? w : wn, m, v, n, sink);
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL((k_stream<WT, PER, MODE>), dim3(blocks), dim3(512), 0, 0, (i & 1) ? wn : w, (i & 1) ? w : wn, m, v, n, sink);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms = 0.f; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1000.0 / 200, bytes = (MODE == 0 ? 24.0 : 12.0) * n;
    printf("%-34s %5d workgroups x 512, %d slab(s) per wave: %6.2f us per launch, %5.2f TB/s of %.1f MB\n", what, blocks, PER, us, bytes / us * 1e-6, bytes * 1e-6);
    return 0;
}

int main() {
    // headline: 3 problems x 504 567 parameters in the rows k_bd owns (36.3 MB / 24 B); franka shape: 2 problems x K = 40
    for (int n : {3 * 504576, 2 * 504576, 1 * 504576, 8 * 504576}) {
        float *w, *wn, *m, *v, *sink;
        CK(hipMalloc(&w, 4ull * n)); CK(hipMalloc(&wn, 4ull * n)); CK(hipMalloc(&m, 4ull * n)); CK(hipMalloc(&v, 4ull * n)); CK(hipMalloc(&sink, 4));
        CK(hipMemset(w, 0, 4ull * n)); CK(hipMemset(wn, 0, 4ull * n)); CK(hipMemset(m, 0, 4ull * n)); CK(hipMemset(v, 0, 4ull * n));
        printf("-- %d floats per array\n", n);
        if (run<false, 2, 0>("adam traffic, plain stores", n, w, wn, m, v, sink)) return 1;      // k_bd's shape: a wave owns one row of 512
        if (run<true, 2, 0>("adam traffic, write-through", n, w, wn, m, v, sink)) return 1;
        if (run<true, 1, 0>("adam traffic, write-through", n, w, wn, m, v, sink)) return 1;
        if (run<true, 4, 0>("adam traffic, write-through", n, w, wn, m, v, sink)) return 1;
        if (run<true, 8, 0>("adam traffic, write-through", n, w, wn, m, v, sink)) return 1;
        if (run<false, 2, 1>("read only (w, m, v)", n, w, wn, m, v, sink)) return 1;
        if (run<true, 2, 2>("write only, write-through", n, w, wn, m, v, sink)) return 1;
        if (run<false, 2, 2>("write only, plain", n, w, wn, m, v, sink)) return 1;
        CK(hipFree(w)); CK(hipFree(wn)); CK(hipFree(m)); CK(hipFree(v)); CK(hipFree(sink));
    }
    return 0;
}
