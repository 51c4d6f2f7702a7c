// Shared helpers for the gfx950 kernels of libwast3d_hip.so.  wave = 64 lanes everywhere.
#pragma once
#include <cstring>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/wast3d_hip.h"

#define W3D_WAVE 64

#define W3D_CHECK_LAUNCH()                                        \
    do {                                                          \
        hipError_t e__ = hipGetLastError();                       \
        if (e__ != hipSuccess) return W3D_ERR_LAUNCH;             \
    } while (0)

#define W3D_HIP(expr)                                             \
    do {                                                          \
        if ((expr) != hipSuccess) return W3D_ERR_LAUNCH;          \
    } while (0)

static inline hipStream_t w3d_s(w3d_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// memory-bound grid: cap at 256 CUs x 8 blocks and grid-stride the rest
static inline int w3d_grid(int64_t work_items, int block, int max_blocks = 2048) {
    int64_t g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > max_blocks) g = max_blocks;
    return (int)g;
}

// Wave-wide sum on the DPP path (no LDS crossbar): 6 v_add_f32 with row_shr:1,2,4,8 / row_bcast:15 / row_bcast:31 modifiers leave the total in
// lane 63, one v_readlane makes it wave-uniform (all 64 lanes must be active).  A __shfl_xor butterfly costs 6 ds_bpermute_b32 + 6 adds
// per value: in the blend backward (10 sums per Gaussian and wave) that was the kernel — 2.8 -> 1.85 ms.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add_step(float x) {
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, ROW_MASK, 0xf, false);   // lanes without a source read 0
    return x + __int_as_float(moved);
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v = dpp_add_step<0x111, 0xf>(v);   // row_shr:1
    v = dpp_add_step<0x112, 0xf>(v);   // row_shr:2
    v = dpp_add_step<0x114, 0xf>(v);   // row_shr:4
    v = dpp_add_step<0x118, 0xf>(v);   // row_shr:8   -> lane 15 of every row holds the row total
    v = dpp_add_step<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
    v = dpp_add_step<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave total
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// N independent wave sums with the DPP steps interleaved (each step depends on the previous one of the same value: interleaving fills the
// DPP wait states instead of s_nop)
template <int N>
__device__ __forceinline__ void wave_sum_dpp_n(float (&v)[N]) {
#pragma unroll
    for (int q = 0; q < N; ++q) v[q] = dpp_add_step<0x111, 0xf>(v[q]);
#pragma unroll
    for (int q = 0; q < N; ++q) v[q] = dpp_add_step<0x112, 0xf>(v[q]);
#pragma unroll
    for (int q = 0; q < N; ++q) v[q] = dpp_add_step<0x114, 0xf>(v[q]);
#pragma unroll
    for (int q = 0; q < N; ++q) v[q] = dpp_add_step<0x118, 0xf>(v[q]);
#pragma unroll
    for (int q = 0; q < N; ++q) v[q] = dpp_add_step<0x142, 0xa>(v[q]);
#pragma unroll
    for (int q = 0; q < N; ++q) v[q] = dpp_add_step<0x143, 0xc>(v[q]);
#pragma unroll
    for (int q = 0; q < N; ++q) v[q] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v[q]), 63));
}
__device__ __forceinline__ float wave_sum(float v) { return wave_sum_dpp(v); }
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// block-wide sum for blockDim.x a multiple of 64, <= 1024; result valid in thread 0
__device__ __forceinline__ float block_sum(float v, float *lds /* >= 16 floats */) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) lds[wid] = v;
    __syncthreads();
    float r = 0.f;
    if (wid == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        r = lane < nw ? lds[lane] : 0.f;
        r = wave_sum(r);
    }
    __syncthreads();
    return r;
}

__device__ __forceinline__ float bf16_to_f32(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }
// round-to-nearest-even; NaN stays NaN (plain cast lowers to v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
    __bf16 b = (__bf16)f;
    return *reinterpret_cast<uint16_t *>(&b);
}

// torch.cdist's matmul form on short vectors (ATen _euclidean_dist): sqrt(clamp_min(|x|^2 + |y|^2 - 2 x.y, 0)).
// ONE definition shared by every kernel that must agree bit-for-bit on a distance (k-th thresholds vs masks).
template <int D>
__device__ __forceinline__ float w3d_dist_mm(const float *x, float x2, const float *y) {
    float dot = 0.f, y2 = 0.f;
#pragma unroll
    for (int c = 0; c < D; ++c) { dot = fmaf(x[c], y[c], dot); y2 = fmaf(y[c], y[c], y2); }
    return sqrtf(fmaxf(x2 + y2 - 2.0f * dot, 0.f));
}
template <int D>
__device__ __forceinline__ float w3d_sqnorm(const float *x) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < D; ++c) s = fmaf(x[c], x[c], s);
    return s;
}
