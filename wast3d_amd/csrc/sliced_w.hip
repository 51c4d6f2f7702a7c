// C1  sliced-Wasserstein loss head (NOT in the reference: parity unpinned; known answer = 1-D optimal transport).
// Inputs are already projected onto L unit directions (a plain library GEMM on the host side): px [L,N], and the style
// projections sorted once (they are constant): py_sorted [L,N].  One segmented radix sort (rocPRIM) orders every projection of the
// content cloud; the head then is elementwise: loss += w * mean((sort(px) - py_sorted)^2), g_px[l, perm] = 2 w diff / (L N).
#include "common.h"
#include <rocprim/device/device_segmented_radix_sort.hpp>

namespace {
__global__ void sw_iota_offsets_kernel(int32_t *vals, int32_t *offsets, int L, int N) {
    const int64_t n = (int64_t)L * N;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) vals[i] = (int32_t)(i % N);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= L; i += (int64_t)gridDim.x * blockDim.x) offsets[i] = (int32_t)(i * N);
}
__global__ __launch_bounds__(256) void sw_head_kernel(const float *__restrict__ sx, const int32_t *__restrict__ perm, const float *__restrict__ sy, int L, int N,
                                                      float w, float *__restrict__ loss, float *__restrict__ g_px) {
    __shared__ float red[16];
    const int64_t n = (int64_t)L * N;
    const float inv = 1.0f / (float)n;
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float d = sx[i] - sy[i];
        acc += d * d;
        if (g_px) g_px[(i / N) * N + perm[i]] = 2.0f * w * inv * d;
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) atomicAdd(loss, w * inv * acc);
}
}  // namespace

extern "C" size_t w3d_sw_workspace_bytes(int L, int N) {
    const size_t n = (size_t)L * N;
    size_t tmp = 0;
    rocprim::segmented_radix_sort_pairs(nullptr, tmp, (const float *)nullptr, (float *)nullptr, (const int32_t *)nullptr, (int32_t *)nullptr, (unsigned)n, (unsigned)L,
                                        (const int32_t *)nullptr, (const int32_t *)nullptr, 0u, 32u, (hipStream_t)0);
    return n * 4 * 3 + ((size_t)L + 1) * 4 + tmp + 1024;
}

extern "C" int w3d_sw_loss_fwd_bwd(const float *px, const float *py_sorted, int L, int N, float weight, float *loss_accum, float *g_px, void *workspace,
                                   size_t workspace_bytes, w3d_stream_t stream) {
    if (!px || !py_sorted || !loss_accum || !workspace || L <= 0 || N <= 0) return W3D_ERR_ARG;
    if (workspace_bytes < w3d_sw_workspace_bytes(L, N)) return W3D_ERR_WORKSPACE;
    const size_t n = (size_t)L * N;
    hipStream_t st = w3d_s(stream);
    char *p = reinterpret_cast<char *>(workspace);
    float *sx = reinterpret_cast<float *>(p); p += n * 4;
    int32_t *iota = reinterpret_cast<int32_t *>(p); p += n * 4;
    int32_t *perm = reinterpret_cast<int32_t *>(p); p += n * 4;
    int32_t *offs = reinterpret_cast<int32_t *>(p); p += ((size_t)L + 1) * 4;
    p = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(p) + 255) & ~(uintptr_t)255);
    size_t tmp = 0;
    rocprim::segmented_radix_sort_pairs(nullptr, tmp, (const float *)nullptr, (float *)nullptr, (const int32_t *)nullptr, (int32_t *)nullptr, (unsigned)n, (unsigned)L,
                                        (const int32_t *)nullptr, (const int32_t *)nullptr, 0u, 32u, (hipStream_t)0);
    hipLaunchKernelGGL(sw_iota_offsets_kernel, dim3(w3d_grid((int64_t)n, 256)), dim3(256), 0, st, iota, offs, L, N);
    if (rocprim::segmented_radix_sort_pairs(p, tmp, px, sx, (const int32_t *)iota, perm, (unsigned)n, (unsigned)L, (const int32_t *)offs, (const int32_t *)(offs + 1), 0u, 32u,
                                            st) != hipSuccess)
        return W3D_ERR_LAUNCH;
    hipLaunchKernelGGL(sw_head_kernel, dim3(w3d_grid((int64_t)n, 256, 1024)), dim3(256), 0, st, sx, perm, py_sorted, L, N, weight, loss_accum, g_px);
    W3D_CHECK_LAUNCH();
    return W3D_OK;
}
