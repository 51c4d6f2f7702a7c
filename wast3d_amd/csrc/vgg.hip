// a8 (f4)  The VGG-19 feature stack of the image-style leg (utils/loss_utils.py:66-109: torchvision vgg19.features[:29] on a 112x112 nearest-resized render,
// taps at layers 0, 5, 10, 19, 28.  torchvision's ReLUs are inplace=True and the reference appends the conv OUTPUT TENSOR, which the next layer then rewrites in
// place: taps 0, 5, 10, 19 therefore hold relu(conv) — only tap 28, which no ReLU follows inside features[:29], is a pre-activation).  The reference runs it through cuDNN; here: f32 direct 3x3 convolution
// (pad 1, stride 1) with an LDS-staged input tile and weight slab, 2x2 max-pool, nearest resize, and their backward passes with respect to the IMAGE
// (the network is frozen: no weight gradients).  The whole stack is ~5 GFLOP forward + ~5 GFLOP backward per iteration — 1e-4 of the distance-GEMM's
// work — so the kernels are written for exact f32 parity with torch's CPU convolution, not for the matrix cores.
//   * activations are stored PRE-ReLU, the ReLU is applied when a convolution loads its input (and by w3d_relu for the four tapped maps the reference sees post-ReLU); max-pool commutes with ReLU, so pooled
//     layers store pool(z) and the next convolution applies the ReLU on load;
//   * backward-data of a convolution is the same kernel on flipped, transposed weights (prepared once by the host).
#include "common.h"

namespace {

constexpr int TP = 16;     // 16 x 16 output pixels per block
constexpr int CO_T = 16;   // output channels per block
constexpr int CI_T = 8;    // input channels per LDS stage

// out[co,y,x] = bias[co] + sum_{ci,ky,kx} act(in[ci, y+ky-1, x+kx-1]) * w[co,ci,ky,kx];  act = ReLU if RELU_IN.  bias may be NULL.  ACCUM: out +=
template <bool RELU_IN>
__global__ __launch_bounds__(TP * TP) void conv3x3_kernel(const float *__restrict__ in, const float *__restrict__ w, const float *__restrict__ bias, int Cin, int Cout, int H, int W,
                                                          float *__restrict__ out) {
    __shared__ float tile[CI_T][TP + 2][TP + 2];
    __shared__ float wt[CO_T][CI_T][9];
    const int tx = threadIdx.x % TP, ty = threadIdx.x / TP;
    const int x0 = blockIdx.x * TP, y0 = blockIdx.y * TP, co0 = blockIdx.z * CO_T;
    float acc[CO_T];
#pragma unroll
    for (int c = 0; c < CO_T; ++c) acc[c] = 0.f;
    for (int ci0 = 0; ci0 < Cin; ci0 += CI_T) {
        __syncthreads();
        for (int i = threadIdx.x; i < CI_T * (TP + 2) * (TP + 2); i += TP * TP) {
            const int ci = i / ((TP + 2) * (TP + 2)), r = i % ((TP + 2) * (TP + 2));
            const int yy = y0 + r / (TP + 2) - 1, xx = x0 + r % (TP + 2) - 1;
            float v = 0.f;
            if (ci0 + ci < Cin && yy >= 0 && yy < H && xx >= 0 && xx < W) { v = in[((size_t)(ci0 + ci) * H + yy) * W + xx]; if (RELU_IN) v = fmaxf(v, 0.f); }
            tile[ci][r / (TP + 2)][r % (TP + 2)] = v;
        }
        for (int i = threadIdx.x; i < CO_T * CI_T * 9; i += TP * TP) {
            const int co = i / (CI_T * 9), r = i % (CI_T * 9), ci = r / 9, kk = r % 9;
            wt[co][ci][kk] = (co0 + co < Cout && ci0 + ci < Cin) ? w[((size_t)(co0 + co) * Cin + ci0 + ci) * 9 + kk] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int ci = 0; ci < CI_T; ++ci) {
            float v[9];
#pragma unroll
            for (int kk = 0; kk < 9; ++kk) v[kk] = tile[ci][ty + kk / 3][tx + kk % 3];
#pragma unroll
            for (int co = 0; co < CO_T; ++co) {
#pragma unroll
                for (int kk = 0; kk < 9; ++kk) acc[co] = fmaf(v[kk], wt[co][ci][kk], acc[co]);
            }
        }
    }
    const int x = x0 + tx, y = y0 + ty;
    if (x < W && y < H) {
#pragma unroll
        for (int co = 0; co < CO_T; ++co)
            if (co0 + co < Cout) out[((size_t)(co0 + co) * H + y) * W + x] = acc[co] + (bias ? bias[co0 + co] : 0.f);
    }
}

__global__ __launch_bounds__(256) void maxpool2_kernel(const float *__restrict__ in, int C, int H, int W, float *__restrict__ out) {
    const int Ho = H / 2, Wo = W / 2;
    const int64_t n = (int64_t)C * Ho * Wo;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int x = (int)(i % Wo), y = (int)((i / Wo) % Ho), c = (int)(i / ((int64_t)Wo * Ho));
        const float *p = in + ((size_t)c * H + 2 * y) * W + 2 * x;
        out[i] = fmaxf(fmaxf(p[0], p[1]), fmaxf(p[W], p[W + 1]));
    }
}

// Gradient with respect to a layer's PRE-activation z from the gradient with respect to what the next convolution consumed (relu(z), or
// relu(pool(z)) at half resolution if POOL): gz (=|+=) g_act routed to the arg-max of each 2x2 cell (first maximum in row-major order, as
// torch's max_pool2d backward) times [z > 0].
template <bool POOL>
__global__ __launch_bounds__(256) void relu_pool_bwd_kernel(const float *__restrict__ z, const float *__restrict__ g_act, int C, int H, int W, int accumulate, float *__restrict__ gz) {
    if (!POOL) {
        const int64_t n = (int64_t)C * H * W;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
            const float g = z[i] > 0.f ? g_act[i] : 0.f;
            gz[i] = accumulate ? gz[i] + g : g;
        }
        return;
    }
    const int Ho = H / 2, Wo = W / 2;
    const int64_t n = (int64_t)C * Ho * Wo;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int x = (int)(i % Wo), y = (int)((i / Wo) % Ho), c = (int)(i / ((int64_t)Wo * Ho));
        const size_t b = ((size_t)c * H + 2 * y) * W + 2 * x;
        const float v[4] = {z[b], z[b + 1], z[b + W], z[b + W + 1]};
        int am = 0;
#pragma unroll
        for (int q = 1; q < 4; ++q) if (v[q] > v[am]) am = q;
        const float g = v[am] > 0.f ? g_act[i] : 0.f;
        const size_t off[4] = {b, b + 1, b + W, b + W + 1};
#pragma unroll
        for (int q = 0; q < 4; ++q) { const float gq = q == am ? g : 0.f; gz[off[q]] = accumulate ? gz[off[q]] + gq : gq; }
    }
}

// F.interpolate(mode='nearest') (utils/loss_utils.py:107): src = floor(dst * in / out), computed in float like ATen's nearest kernel
__device__ __forceinline__ int nearest_src(int dst, int in_size, int out_size) {
    const float scale = (float)in_size / (float)out_size;
    const int s = (int)floorf((float)dst * scale);
    return s < in_size - 1 ? s : in_size - 1;
}
__global__ __launch_bounds__(256) void nearest_resize_kernel(const float *__restrict__ in, int C, int H, int W, int Ho, int Wo, float *__restrict__ out) {
    const int64_t n = (int64_t)C * Ho * Wo;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int x = (int)(i % Wo), y = (int)((i / Wo) % Ho), c = (int)(i / ((int64_t)Wo * Ho));
        out[i] = in[((size_t)c * H + nearest_src(y, H, Ho)) * W + nearest_src(x, W, Wo)];
    }
}
__global__ __launch_bounds__(256) void nearest_resize_bwd_kernel(const float *__restrict__ g_out, int C, int H, int W, int Ho, int Wo, float *__restrict__ g_in) {
    const int64_t n = (int64_t)C * Ho * Wo;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int x = (int)(i % Wo), y = (int)((i / Wo) % Ho), c = (int)(i / ((int64_t)Wo * Ho));
        unsafeAtomicAdd(g_in + ((size_t)c * H + nearest_src(y, H, Ho)) * W + nearest_src(x, W, Wo), g_out[i]);      // up-sampling maps several outputs to one input
    }
}

}  // namespace

extern "C" int w3d_conv3x3(const float *in, const float *weight, const float *bias, int Cin, int Cout, int H, int W, int relu_input, float *out, w3d_stream_t stream) {
    if (!in || !weight || !out || Cin < 1 || Cout < 1 || H < 1 || W < 1) return W3D_ERR_ARG;
    const dim3 grid((W + TP - 1) / TP, (H + TP - 1) / TP, (Cout + CO_T - 1) / CO_T);
    if (relu_input) hipLaunchKernelGGL(conv3x3_kernel<true>, grid, dim3(TP * TP), 0, w3d_s(stream), in, weight, bias, Cin, Cout, H, W, out);
    else hipLaunchKernelGGL(conv3x3_kernel<false>, grid, dim3(TP * TP), 0, w3d_s(stream), in, weight, bias, Cin, Cout, H, W, out);
    W3D_CHECK_LAUNCH();
    return W3D_OK;
}

__global__ __launch_bounds__(256) void relu_kernel(const float *__restrict__ z, int64_t n, float *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = fmaxf(z[i], 0.f);
}

extern "C" int w3d_relu(const float *z, int64_t n, float *out, w3d_stream_t stream) {
    if (!z || !out || n < 1) return W3D_ERR_ARG;
    hipLaunchKernelGGL(relu_kernel, dim3(w3d_grid(n, 256, 4096)), dim3(256), 0, w3d_s(stream), z, n, out);
    W3D_CHECK_LAUNCH();
    return W3D_OK;
}

extern "C" int w3d_maxpool2(const float *in, int C, int H, int W, float *out, w3d_stream_t stream) {
    if (!in || !out || C < 1 || H < 2 || W < 2 || (H & 1) || (W & 1)) return W3D_ERR_ARG;
    hipLaunchKernelGGL(maxpool2_kernel, dim3(w3d_grid((int64_t)C * (H / 2) * (W / 2), 256, 4096)), dim3(256), 0, w3d_s(stream), in, C, H, W, out);
    W3D_CHECK_LAUNCH();
    return W3D_OK;
}

extern "C" int w3d_relu_pool_bwd(const float *z, const float *g_act, int C, int H, int W, int pooled, int accumulate, float *gz, w3d_stream_t stream) {
    if (!z || !g_act || !gz || C < 1 || H < 1 || W < 1 || (pooled && ((H & 1) || (W & 1)))) return W3D_ERR_ARG;
    if (pooled) hipLaunchKernelGGL(relu_pool_bwd_kernel<true>, dim3(w3d_grid((int64_t)C * (H / 2) * (W / 2), 256, 4096)), dim3(256), 0, w3d_s(stream), z, g_act, C, H, W, accumulate, gz);
    else hipLaunchKernelGGL(relu_pool_bwd_kernel<false>, dim3(w3d_grid((int64_t)C * H * W, 256, 4096)), dim3(256), 0, w3d_s(stream), z, g_act, C, H, W, accumulate, gz);
    W3D_CHECK_LAUNCH();
    return W3D_OK;
}

extern "C" int w3d_nearest_resize(const float *in, int C, int H, int W, int Ho, int Wo, float *out, w3d_stream_t stream) {
    if (!in || !out || C < 1 || H < 1 || W < 1 || Ho < 1 || Wo < 1) return W3D_ERR_ARG;
    hipLaunchKernelGGL(nearest_resize_kernel, dim3(w3d_grid((int64_t)C * Ho * Wo, 256, 4096)), dim3(256), 0, w3d_s(stream), in, C, H, W, Ho, Wo, out);
    W3D_CHECK_LAUNCH();
    return W3D_OK;
}

extern "C" int w3d_nearest_resize_bwd(const float *g_out, int C, int H, int W, int Ho, int Wo, float *g_in, w3d_stream_t stream) {
    if (!g_out || !g_in || C < 1 || H < 1 || W < 1 || Ho < 1 || Wo < 1) return W3D_ERR_ARG;
    hipLaunchKernelGGL(nearest_resize_bwd_kernel, dim3(w3d_grid((int64_t)C * Ho * Wo, 256, 4096)), dim3(256), 0, w3d_s(stream), g_out, C, H, W, Ho, Wo, g_in);
    W3D_CHECK_LAUNCH();
    return W3D_OK;
}
