// a8  Gram-matrix style loss on feature maps (utils/loss_utils.py:178-205, calc_style_loss):
//     G = F F^T  (F = features viewed [C, H*W]),  A = S S^T,  loss_layer = mean((G - A)^2).
// Exact-f32 matrix cores: v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate — the same fma chain as a scalar loop), one
// 32x32 output tile per wave.  Shapes are the VGG taps at 112x112 input: C x HW = 64x12544 ... 512x49 (0.44 GFLOP in total),
// so the kernels are latency/L2-bound; the point of the split-K + fixed-order reduction is bit-reproducibility (no float atomics).
//   1. gram_partial : P[s] = F[:, ks] F[:, ks]^T for K-slice s                      (grid: tiles x slices)
//   2. gram_reduce  : G = sum_s P[s] (fixed order); optional loss: D = G - A, loss += w mean(D^2), Dw = (4 w / C^2) D
//   3. gram_grad    : gF += Dw F          (d/dF mean((FF^T - A)^2) = (dG + dG^T) F, dG = 2 (G - A) / C^2 symmetric)
#include "common.h"
#include <algorithm>

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int KSLICE_MIN = 64;     // smallest K range worth a wave

__device__ __forceinline__ int d_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// P[s, i, j] = sum_{k in slice s} F[i,k] F[j,k]
template <bool VEC>
__global__ __launch_bounds__(64) void gram_partial_kernel(const float *__restrict__ F, int C, int HW, int kslice, float *__restrict__ P) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    const int tj = blockIdx.x, ti = blockIdx.y, s = blockIdx.z;
    const int i = ti * 32 + r, j = tj * 32 + r;
    const int k_begin = s * kslice, k_end = min(HW, k_begin + kslice);
    const float *fi = F + (int64_t)min(i, C - 1) * HW, *fj = F + (int64_t)min(j, C - 1) * HW;
    const bool vi = i < C, vj = j < C;
    f32x16 acc = {0.f};
    for (int k0 = k_begin; k0 < k_end; k0 += 8) {
        const int kk = k0 + 4 * h;            // this lane's four k of the chunk (same assignment for both operands)
        float a[4], b[4];
        if (VEC && kk + 3 < k_end) {
            const float4 va = *reinterpret_cast<const float4 *>(fi + kk), vb = *reinterpret_cast<const float4 *>(fj + kk);
            a[0] = va.x; a[1] = va.y; a[2] = va.z; a[3] = va.w; b[0] = vb.x; b[1] = vb.y; b[2] = vb.z; b[3] = vb.w;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const bool in = kk + e < k_end; a[e] = in ? fi[kk + e] : 0.f; b[e] = in ? fj[kk + e] : 0.f; }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(vi ? a[e] : 0.f, vj ? b[e] : 0.f, acc, 0, 0, 0);
    }
    // D layout: col = lane & 31 (the B operand's row index j), row = d_row(reg) (the A operand's row index i)
    float *Ps = P + (int64_t)s * C * C;
    const int jc = tj * 32 + r;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int ir = ti * 32 + d_row(reg, lane);
        if (ir < C && jc < C) Ps[(int64_t)ir * C + jc] = acc[reg];
    }
}

// G = sum_s P[s]; with A: D = G - A, loss += weight * mean(D^2), Dw = (4 weight / C^2) * D  (written over G's slot in `out`)
__global__ __launch_bounds__(256) void gram_reduce_kernel(const float *__restrict__ P, int nslice, int64_t CC, const float *__restrict__ A, float weight,
                                                          float inv_cc, float *__restrict__ out, float *__restrict__ block_partial) {
    __shared__ float red[16];
    float local = 0.f;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < CC; e += (int64_t)gridDim.x * 256) {
        float g = 0.f;
        for (int s = 0; s < nslice; ++s) g += P[(int64_t)s * CC + e];
        if (A) {
            const float dlt = g - A[e];
            local = fmaf(dlt, dlt, local);
            out[e] = 4.0f * weight * inv_cc * dlt;
        } else {
            out[e] = g;
        }
    }
    if (A) {
        const float t = block_sum(local, red);
        if (threadIdx.x == 0) block_partial[blockIdx.x] = t;
    }
}

// fixed-order sum of the per-block partials -> loss_accum += weight * inv_cc * sum
__global__ __launch_bounds__(64) void gram_loss_final_kernel(const float *__restrict__ block_partial, int nb, float scale, float *__restrict__ loss_accum) {
    float s = 0.f;
    for (int i = threadIdx.x; i < nb; i += 64) s += block_partial[i];
    s = wave_sum(s);
    if (threadIdx.x == 0) loss_accum[0] += scale * s;
}

// gF[i, x] += sum_j Dw[i, j] F[j, x]
__global__ __launch_bounds__(64) void gram_grad_kernel(const float *__restrict__ Dw, const float *__restrict__ F, int C, int HW, float *__restrict__ gF) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    const int x = blockIdx.x * 32 + r, i = blockIdx.y * 32 + r;
    const bool vx = x < HW, vi = i < C;
    const float *drow = Dw + (int64_t)min(i, C - 1) * C;
    f32x16 acc = {0.f};
    for (int j0 = 0; j0 < C; j0 += 8) {
        const int jj = j0 + 4 * h;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool in = jj + e < C;
            const float a = (vi && in) ? drow[jj + e] : 0.f;
            const float b = (vx && in) ? F[(int64_t)(jj + e) * HW + x] : 0.f;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    }
    const int xc = blockIdx.x * 32 + r;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int ir = blockIdx.y * 32 + d_row(reg, lane);
        if (ir < C && xc < HW) gF[(int64_t)ir * HW + xc] += acc[reg];
    }
}

struct GramPlan { int tiles, nslice, kslice, nred; };

GramPlan plan(int C, int HW) {
    GramPlan p;
    p.tiles = (C + 31) / 32;
    const int want = std::max(1, 1024 / (p.tiles * p.tiles));                 // ~1024 waves in flight
    int ns = std::min(want, std::max(1, HW / KSLICE_MIN));
    int ks = ((HW + ns - 1) / ns + 7) & ~7;                                    // slices are multiples of the 8-wide k chunk
    p.kslice = ks;
    p.nslice = (HW + ks - 1) / ks;
    p.nred = (int)std::min<int64_t>(((int64_t)C * C + 255) / 256, 256);
    return p;
}

}  // namespace

extern "C" size_t w3d_gram_workspace_bytes(int C, int HW) {
    if (C < 1 || HW < 1) return 0;
    const GramPlan p = plan(C, HW);
    return ((size_t)p.nslice * C * C + (size_t)C * C + 256) * sizeof(float) + 256;
}

static int gram_run(const float *F, const float *A, int C, int HW, float weight, float *loss_accum, float *gF, float *G_out, void *ws, size_t ws_bytes,
                    hipStream_t st) {
    if (!F || C < 1 || HW < 1) return W3D_ERR_ARG;
    if (!ws || ws_bytes < w3d_gram_workspace_bytes(C, HW)) return W3D_ERR_WORKSPACE;
    const GramPlan p = plan(C, HW);
    float *P = reinterpret_cast<float *>(ws);
    float *Dw = P + (size_t)p.nslice * C * C;
    float *bp = Dw + (size_t)C * C;
    const dim3 grid(p.tiles, p.tiles, p.nslice);
    if ((HW & 3) == 0 && (((uintptr_t)F) & 15) == 0)
        hipLaunchKernelGGL(gram_partial_kernel<true>, grid, dim3(64), 0, st, F, C, HW, p.kslice, P);
    else
        hipLaunchKernelGGL(gram_partial_kernel<false>, grid, dim3(64), 0, st, F, C, HW, p.kslice, P);
    const int64_t CC = (int64_t)C * C;
    const float inv_cc = 1.0f / (float)CC;
    hipLaunchKernelGGL(gram_reduce_kernel, dim3(p.nred), dim3(256), 0, st, P, p.nslice, CC, A, weight, inv_cc, A ? Dw : G_out, bp);
    if (A) {
        if (loss_accum) hipLaunchKernelGGL(gram_loss_final_kernel, dim3(1), dim3(64), 0, st, bp, p.nred, weight * inv_cc, loss_accum);
        if (gF) hipLaunchKernelGGL(gram_grad_kernel, dim3((HW + 31) / 32, p.tiles), dim3(64), 0, st, Dw, F, C, HW, gF);
    }
    return w3d_launch_status();
}

extern "C" int w3d_gram_matrix(const float *F, int C, int HW, float *G, void *ws, size_t ws_bytes, w3d_stream_t stream) {
    if (!G) return W3D_ERR_ARG;
    return gram_run(F, nullptr, C, HW, 0.f, nullptr, nullptr, G, ws, ws_bytes, w3d_s(stream));
}

extern "C" int w3d_gram_style_loss_fwd_bwd(const float *F, const float *A, int C, int HW, float weight, float *loss_accum, float *g_F, void *ws,
                                           size_t ws_bytes, w3d_stream_t stream) {
    if (!A) return W3D_ERR_ARG;
    return gram_run(F, A, C, HW, weight, loss_accum, g_F, nullptr, ws, ws_bytes, w3d_s(stream));
}
