// Row-wise fp8 quantisation for the distance-GEMM (BASELINE config C5: "fp16 feats + fp8 dist-GEMM, e4m3, per-row scale").
// One power-of-two scale per row in E8M0 (the MX block-scale byte: value 2^(e-127)), chosen as the smallest power of two that maps the
// row's largest magnitude into e4m3's range (|q| <= 448); elements are converted round-to-nearest-even by v_cvt_pk_fp8_f32 (OCP e4m3fn
// on gfx950).  The scaled MFMA applies the scales in hardware, so quantised rows feed w3d_knn_fp8 directly.
#include "common.h"

namespace {

template <int DT>   // 0 f32, 1 bf16, 2 f16
__device__ __forceinline__ float ld(const void *p, int64_t i) {
    if (DT == 0) return reinterpret_cast<const float *>(p)[i];
    const uint16_t v = reinterpret_cast<const uint16_t *>(p)[i];
    if (DT == 1) return bf16_to_f32(v);
    _Float16 h; __builtin_memcpy(&h, &v, 2); return (float)h;
}

// one wave per row; lane handles elements 4*lane .. 4*lane+3 of every 256-element span (d % 4 == 0)
template <int DT>
__global__ __launch_bounds__(256) void quantize_rows_fp8_kernel(const void *__restrict__ F, int64_t R, int d, uint8_t *__restrict__ q, int32_t *__restrict__ scale,
                                                                float *__restrict__ motion2) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    for (int64_t r = wave0; r < R; r += nwaves) {
        float amax = 0.f;
        for (int c = lane * 4; c < d; c += 256)
#pragma unroll
            for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fabsf(ld<DT>(F, r * d + c + e)));
        amax = wave_max(amax);
        int ex = 0;
        const float m = frexpf(amax, &ex);                               // amax = m 2^ex, m in [0.5, 1)
        int e = (m <= 0.875f ? ex - 9 : ex - 8) + 127;                   // smallest e with amax 2^-(e-127) <= 448 = 0.875 2^9
        if (!(amax > 0.f)) e = 127;
        e = min(max(e, 1), 254);
        const float inv = __uint_as_float((uint32_t)(254 - e) << 23);     // 2^-(e-127)
        // motion2[r] = |x^_new - x^_old|^2 of the DEQUANTISED row (q and scale must hold the previous iteration's values): the kNN runs on
        // x^, so this — not the motion of the 16-bit features — is what the triangle-inequality warm start needs
        const float old_f = motion2 ? __uint_as_float((uint32_t)(scale[r] & 0xff) << 23) : 0.f;
        const float new_f = __uint_as_float((uint32_t)e << 23);
        float m2 = 0.f;
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) scale[r] = e;
        for (int c = lane * 4; c < d; c += 256) {
            float v[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) v[t] = fminf(fmaxf(ld<DT>(F, r * d + c + t) * inv, -448.f), 448.f);
            int packed = 0;
            packed = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], packed, false);
            packed = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], packed, true);
            if (motion2) {
                const int old = *reinterpret_cast<const int *>(q + r * d + c);
                const float o[4] = {__builtin_amdgcn_cvt_f32_fp8(old, 0), __builtin_amdgcn_cvt_f32_fp8(old, 1), __builtin_amdgcn_cvt_f32_fp8(old, 2), __builtin_amdgcn_cvt_f32_fp8(old, 3)};
                const float nw[4] = {__builtin_amdgcn_cvt_f32_fp8(packed, 0), __builtin_amdgcn_cvt_f32_fp8(packed, 1), __builtin_amdgcn_cvt_f32_fp8(packed, 2), __builtin_amdgcn_cvt_f32_fp8(packed, 3)};
#pragma unroll
                for (int t = 0; t < 4; ++t) { const float dd = nw[t] * new_f - o[t] * old_f; m2 = fmaf(dd, dd, m2); }
            }
            *reinterpret_cast<int *>(q + r * d + c) = packed;
        }
        if (motion2) { m2 = wave_sum(m2); if (lane == 0) motion2[r] = m2; }
    }
}

}  // namespace

extern "C" int w3d_quantize_rows_fp8(const void *F, int dtype, int64_t R, int d, void *q, int32_t *scale_e8m0, float *motion2, w3d_stream_t stream) {
    if (!F || !q || !scale_e8m0 || R < 0 || d < 1) return W3D_ERR_ARG;
    if (d % 4 || (((uintptr_t)q) & 3)) return W3D_ERR_UNSUPPORTED;
    if (R == 0) return W3D_OK;
    const dim3 grid(w3d_grid(R, 4, 16384)), block(256);
    uint8_t *qq = reinterpret_cast<uint8_t *>(q);
    if (dtype == W3D_DTYPE_F32) hipLaunchKernelGGL(quantize_rows_fp8_kernel<0>, grid, block, 0, w3d_s(stream), F, R, d, qq, scale_e8m0, motion2);
    else if (dtype == W3D_DTYPE_BF16) hipLaunchKernelGGL(quantize_rows_fp8_kernel<1>, grid, block, 0, w3d_s(stream), F, R, d, qq, scale_e8m0, motion2);
    else if (dtype == W3D_DTYPE_F16) hipLaunchKernelGGL(quantize_rows_fp8_kernel<2>, grid, block, 0, w3d_s(stream), F, R, d, qq, scale_e8m0, motion2);
    else return W3D_ERR_ARG;
    W3D_CHECK_LAUNCH();
    return W3D_OK;
}
