// C1  sliced-Wasserstein loss head (NOT in the reference: parity unpinned; known answer = 1-D optimal transport of two equally sized clouds:
// W2^2 = mean((sort(x) - sort(y))^2), which is ot.emd2 with uniform weights on a line — SURVEY 8c).
//
// Round 6: the head is hand-written end to end for clouds that fit LDS (N <= 16 384 points: BASELINE C1 is 2 000 x 64-d x 64 projections):
//   sw_project_sort_kernel   one workgroup per projection l: p_i = <F_i, theta_l> (16 lanes per feature row, coalesced 64-B segments, DPP row sum),
//                            (orderable key, index) pairs as one u64 each in LDS, bitonic sort in place — ties ordered by index, i.e. numpy's stable
//                            argsort — then EITHER the sorted projections are written out (the constant style side: once per style set) OR the head runs
//                            on the spot: diff_r = sx_r - sy_r, the projection's share of the loss, dL/dp scattered back through the permutation.
//   sw_feature_grad_kernel   dL/dF [N,d] = dL/dp^T [N,L] x theta [L,d]  (theta through LDS) and the loss total (the L partial sums added in a fixed order: the
//                            result is bit-reproducible, unlike one float atomic per workgroup).
// Nothing but these two launches per iteration; no library GEMM, no library sort.  Larger clouds keep the round-1 path below (caller-side projections + rocPRIM's
// segmented radix sort + sw_head_kernel).
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

// float bits -> u32 whose unsigned order is the float order (-0 < +0; NaNs beyond +inf: they sort last like numpy's)
__device__ __forceinline__ uint32_t sw_orderable(float f) {
    const uint32_t u = __float_as_uint(f);
    return u ^ ((uint32_t)((int32_t)u >> 31) | 0x80000000u);
}
__device__ __forceinline__ float sw_from_orderable(uint32_t k) {
    const uint32_t u = (k & 0x80000000u) ? (k ^ 0x80000000u) : ~k;
    return __uint_as_float(u);
}

constexpr int SW_BLOCK = 1024;
constexpr int SW_MAX_N = 16384;       // 8 B per padded element: 128 KB of the CU's 160 KB

// HEAD = false: out_sorted [L,N] (+ out_perm [L,N], may be NULL).   HEAD = true: sy [L,N] sorted style projections in; g_pt [N,L] (dL/dp, transposed: a row per point) and
// loss_part [L] out.
template <bool HEAD>
__global__ __launch_bounds__(SW_BLOCK) void sw_project_sort_kernel(const float *__restrict__ F, const float *__restrict__ proj, int N, int d, int L, int Npad,
                                                                   float *__restrict__ out_sorted, int32_t *__restrict__ out_perm, const float *__restrict__ sy, float scale,
                                                                   float *__restrict__ g_pt, float *__restrict__ loss_part) {
    extern __shared__ unsigned long long sw_keys[];          // [Npad]
    __shared__ float red[16];
    const int l = blockIdx.x, tid = threadIdx.x;
    const float *th = proj + (size_t)l * d;
    // ---- projections: 16 lanes per feature row (64 rows per pass of the block), lane j takes features j, j + 16, ... — every load instruction reads 64-B segments
    const int j = tid & 15, grp = tid >> 4;
    for (int i0 = 0; i0 < Npad; i0 += SW_BLOCK / 16) {
        const int i = i0 + grp;
        float acc = 0.f;
        if (i < N) {
            const float *row = F + (size_t)i * d;
            for (int c = j; c < d; c += 16) acc = fmaf(row[c], th[c], acc);
        }
        acc = dpp_add_step<0x111, 0xf>(acc);      // row_shr 1, 2, 4, 8: lane 15 of the 16-lane row holds the row's sum
        acc = dpp_add_step<0x112, 0xf>(acc);
        acc = dpp_add_step<0x114, 0xf>(acc);
        acc = dpp_add_step<0x118, 0xf>(acc);
        if (j == 15 && i < Npad) sw_keys[i] = i < N ? (((unsigned long long)sw_orderable(acc) << 32) | (uint32_t)i) : ~0ull;
    }
    __syncthreads();
    // ---- bitonic sort of the u64 keys, ascending (padding = all ones sorts last)
    for (int k = 2; k <= Npad; k <<= 1) {
        for (int s = k >> 1; s > 0; s >>= 1) {
            for (int t = tid; t < (Npad >> 1); t += SW_BLOCK) {
                const int a = ((t & ~(s - 1)) << 1) | (t & (s - 1));      // lower index of the pair
                const int b = a | s;
                const bool up = (a & k) == 0;
                const unsigned long long x = sw_keys[a], y = sw_keys[b];
                if ((x > y) == up) { sw_keys[a] = y; sw_keys[b] = x; }
            }
            __syncthreads();
        }
    }
    if (!HEAD) {
        for (int r = tid; r < N; r += SW_BLOCK) {
            const unsigned long long key = sw_keys[r];
            out_sorted[(size_t)l * N + r] = sw_from_orderable((uint32_t)(key >> 32));
            if (out_perm) out_perm[(size_t)l * N + r] = (int32_t)(uint32_t)key;
        }
        return;
    }
    float acc = 0.f;
    for (int r = tid; r < N; r += SW_BLOCK) {
        const unsigned long long key = sw_keys[r];
        const float diff = sw_from_orderable((uint32_t)(key >> 32)) - sy[(size_t)l * N + r];
        acc = fmaf(diff, diff, acc);
        if (g_pt) g_pt[(size_t)(uint32_t)key * L + l] = scale * diff;
    }
    acc = block_sum(acc, red);
    if (tid == 0) loss_part[l] = acc;
}

// g_F [N,d] = g_pt [N,L] x proj [L,d]; block 0 also folds the L loss parts (fixed order) into loss_accum
__global__ __launch_bounds__(256) void sw_feature_grad_kernel(const float *__restrict__ g_pt, const float *__restrict__ proj, int N, int d, int L, float *__restrict__ g_F,
                                                              const float *__restrict__ loss_part, float loss_scale, float *__restrict__ loss_accum, int proj_in_lds) {
    extern __shared__ float sw_proj[];                       // [L * d] when it fits
    if (proj_in_lds) {
        for (int t = threadIdx.x; t < L * d; t += 256) sw_proj[t] = proj[t];
        __syncthreads();
    }
    const float *th = proj_in_lds ? sw_proj : proj;
    if (g_F) {
        const int64_t n = (int64_t)N * d;
        for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < n; o += (int64_t)gridDim.x * 256) {
            const int i = (int)(o / d), c = (int)(o - (int64_t)i * d);
            const float *g = g_pt + (size_t)i * L;
            float acc = 0.f;
            for (int l = 0; l < L; ++l) acc = fmaf(g[l], th[l * d + c], acc);
            g_F[o] = acc;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        float s = 0.f;
        for (int l = 0; l < L; ++l) s += loss_part[l];
        loss_accum[0] += loss_scale * s;
    }
}

inline int sw_npad(int N) {
    int p = 2;
    while (p < N) p <<= 1;
    return p;
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

extern "C" int w3d_sw_max_points(void) { return SW_MAX_N; }

extern "C" int w3d_sw_project_sort(const float *F, const float *proj, int N, int d, int L, float *out_sorted, int32_t *out_perm, w3d_stream_t stream) {
    if (!F || !proj || !out_sorted || N <= 0 || d <= 0 || L <= 0) return W3D_ERR_ARG;
    if (N > SW_MAX_N) return W3D_ERR_UNSUPPORTED;
    const int Npad = sw_npad(N);
    const size_t lds = (size_t)Npad * 8;
    static bool attr_done = false;
    if (!attr_done) {
        W3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(sw_project_sort_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, SW_MAX_N * 8));
        W3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(sw_project_sort_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, SW_MAX_N * 8));
        attr_done = true;
    }
    hipLaunchKernelGGL(sw_project_sort_kernel<false>, dim3(L), dim3(SW_BLOCK), lds, w3d_s(stream), F, proj, N, d, L, Npad, out_sorted, out_perm, (const float *)nullptr, 0.f,
                       (float *)nullptr, (float *)nullptr);
    W3D_CHECK_LAUNCH();
    return W3D_OK;
}

extern "C" size_t w3d_sw_features_workspace_bytes(int L, int N) { return ((size_t)L * N + (size_t)L) * 4 + 256; }

extern "C" int w3d_sw_features_fwd_bwd(const float *FX, const float *proj, const float *py_sorted, int N, int d, int L, float weight, float *loss_accum, float *g_FX,
                                       void *workspace, size_t workspace_bytes, w3d_stream_t stream) {
    if (!FX || !proj || !py_sorted || !loss_accum || !workspace || N <= 0 || d <= 0 || L <= 0) return W3D_ERR_ARG;
    if (N > SW_MAX_N) return W3D_ERR_UNSUPPORTED;
    if (workspace_bytes < w3d_sw_features_workspace_bytes(L, N)) return W3D_ERR_WORKSPACE;
    const int Npad = sw_npad(N);
    const size_t lds = (size_t)Npad * 8;
    static bool attr_done = false;
    if (!attr_done) {
        W3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(sw_project_sort_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, SW_MAX_N * 8));
        W3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(sw_feature_grad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
        attr_done = true;
    }
    float *g_pt = reinterpret_cast<float *>(workspace);
    float *loss_part = g_pt + (size_t)L * N;
    const float inv = 1.0f / ((float)L * (float)N);
    hipStream_t st = w3d_s(stream);
    hipLaunchKernelGGL(sw_project_sort_kernel<true>, dim3(L), dim3(SW_BLOCK), lds, st, FX, proj, N, d, L, Npad, (float *)nullptr, (int32_t *)nullptr, py_sorted,
                       2.0f * weight * inv, g_FX ? g_pt : (float *)nullptr, loss_part);
    const size_t pbytes = (size_t)L * d * 4;
    const int in_lds = pbytes <= 64 * 1024;
    hipLaunchKernelGGL(sw_feature_grad_kernel, dim3(g_FX ? w3d_grid((int64_t)N * d, 256, 2048) : 1), dim3(256), in_lds ? pbytes : 0, st, g_pt, proj, N, d, L, g_FX, loss_part,
                       weight * inv, loss_accum, in_lds);
    W3D_CHECK_LAUNCH();
    return W3D_OK;
}
