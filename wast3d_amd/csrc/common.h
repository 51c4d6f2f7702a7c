// Shared helpers for the gfx950 kernels of libwast3d_hip.so.  wave = 64 lanes everywhere.
#pragma once
#include <cstring>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/wast3d_hip.h"

#define W3D_WAVE 64

// The HIP error behind the last W3D_ERR_LAUNCH of this thread (w3d_last_hip_error_string, api.hip): a bare return code says nothing about WHICH
// runtime call failed or why.
extern thread_local int w3d_last_hip_error_code;
static inline int w3d_launch_status() {
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return W3D_OK;
    w3d_last_hip_error_code = (int)e;
    return W3D_ERR_LAUNCH;
}

#define W3D_CHECK_LAUNCH()                                        \
    do {                                                          \
        if (w3d_launch_status() != W3D_OK) return W3D_ERR_LAUNCH; \
    } while (0)

#define W3D_HIP(expr)                                             \
    do {                                                          \
        const hipError_t e__ = (expr);                            \
        if (e__ != hipSuccess) { w3d_last_hip_error_code = (int)e__; return W3D_ERR_LAUNCH; } \
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
// Ten wave-wide sums at once ("reduce-scatter" over the lanes): lane l (l & 15 < 10) returns the total of v[l & 15] over all 64 lanes.
// Inside each 16-lane row four DPP butterfly stages (row_mirror, row_half_mirror, quad_perm [3,2,1,0], quad_perm [1,0,3,2]) halve the number
// of values a lane carries (10 -> 8 -> 4 -> 2 -> 1) while doubling the lanes summed; masked v_add_f32_dpp writes keep "my half" without any
// select in the first two stages.  27 VALU + 2 cross-row exchanges instead of 60 DPP adds + 10 v_readlane + 10 selects for ten separate
// wave_sum_dpp.  All 64 lanes must be active; v is destroyed.  DPP reads need 2 wait states after the VALU write of their source: the
// instruction order below keeps every such pair >= 3 instructions apart (one s_nop where it does not).
// ROW0_ONLY: the totals are valid in lanes 0-9 only (all the blend backward's 10-lane atomic needs).
template <bool ROW0_ONLY = false>
__device__ __forceinline__ float wave_sum10_scatter(float (&v)[10]) {
    float z;
    const unsigned long long m_bit1 = 0xccccccccccccccccull, m_bit0 = 0xaaaaaaaaaaaaaaaaull;
    asm volatile(
        "s_nop 1\n\t"                                        // the inputs may have been written by the VALU instructions just before this block
        // stage A: pairs (i, i+8) over row_mirror; lanes 8-15 keep index i+8 (only 8, 9 exist), lanes 0-7 index i
        "v_add_f32_dpp %8, %8, %8 row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %9, %9, %9 row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %2, %2, %2 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %3, %3 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %4, %4, %4 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %5, %5, %5 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %6, %6, %6 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %7, %7, %7 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %8, %0, %0 row_mirror row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %9, %1, %1 row_mirror row_mask:0xf bank_mask:0x3\n\t"
        // stage B: pairs (i, i+4) over row_half_mirror; lanes with bit 2 set (banks 1, 3) keep index i+4.  w0 = %8, w1 = %9, w2..w7 = %2..%7
        "v_add_f32_dpp %4, %4, %4 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %5, %5, %5 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %6, %6, %6 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %7, %7, %7 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %4, %8, %8 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %5, %9, %9 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %6, %2, %2 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %7, %3, %3 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        // stage C: pairs (i, i+2) inside quads; lanes with bit 1 set keep index i+2.  x0..x3 = %4..%7; temporaries %0..%3
        "v_add_f32_dpp %0, %4, %4 quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %6, %6 quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %5, %5 quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %7, %7 quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32 %0, %0, %1, %11\n\t"
        "v_cndmask_b32 %2, %2, %3, %11\n\t"
        "s_nop 1\n\t"
        // stage D: neighbours; lanes with bit 0 set keep index i+1
        "v_add_f32_dpp %1, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32 %10, %1, %3, %12"
        : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "=&v"(z)
        : "s"(m_bit1), "s"(m_bit0));
    if (ROW0_ONLY) {
        // gfx950's lane-swap instructions instead of two trips through the LDS crossbar (ds_bpermute): v_permlane32_swap exchanges lanes 32-63 of its first operand with
        // lanes 0-31 of its second, v_permlane16_swap the odd 16-lane rows of the first with the even rows of the second.  With both operands = z the second result holds,
        // in lanes 0-31 (rows 0, 2), what sat 32 (16) lanes higher: lanes 0-15 end with the sum over the four rows — (row0 + row2) + (row1 + row3).
        const int zi = __float_as_int(z);
        const auto r = __builtin_amdgcn_permlane32_swap(zi, zi, false, false);
        z += __int_as_float((int)r[1]);
        const int yi = __float_as_int(z);
        const auto q = __builtin_amdgcn_permlane16_swap(yi, yi, false, false);
        return z + __int_as_float((int)q[1]);
    }
    z += __shfl_xor(z, 16, 64);                 // the four rows
    z += __shfl_xor(z, 32, 64);
    return z;
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
