// a8  image-space loss heads on the render, forward value + gradient in one streaming pass.
#include "common.h"

// tv_loss (utils/loss_utils.py:213-215): 0.5*(mean|d/dy| + mean|d/dx|) over [..., H, W]
__global__ __launch_bounds__(256) void tv_loss_kernel(const float *__restrict__ img, int C, int H, int W, float wy, float wx, int accumulate,
                                                      float *__restrict__ loss_accum, float *__restrict__ grad) {
    __shared__ float red[16];
    const int64_t n = (int64_t)C * H * W;
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int x = (int)(i % W);
        const int y = (int)((i / W) % H);
        const float v = img[i];
        float g = 0.f;
        // each pixel owns the differences to its right and lower neighbour; gradient gathers all four
        if (x + 1 < W) { const float d = img[i + 1] - v; s += wx * fabsf(d); g -= wx * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)); }
        if (y + 1 < H) { const float d = img[i + W] - v; s += wy * fabsf(d); g -= wy * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)); }
        if (x > 0) { const float d = v - img[i - 1]; g += wx * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)); }
        if (y > 0) { const float d = v - img[i - W]; g += wy * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)); }
        if (grad) { if (accumulate) grad[i] += g; else grad[i] = g; }
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0 && loss_accum) atomicAdd(loss_accum, s);
}

extern "C" int w3d_tv_loss_fwd_bwd(const float *img, int C, int H, int W, float weight, int accumulate, float *loss_accum, float *grad,
                                   w3d_stream_t stream) {
    if (!img || C <= 0 || H <= 1 || W <= 1) return W3D_ERR_ARG;
    const float wy = 0.5f * weight / ((float)C * (float)(H - 1) * (float)W);
    const float wx = 0.5f * weight / ((float)C * (float)H * (float)(W - 1));
    hipLaunchKernelGGL(tv_loss_kernel, dim3(w3d_grid((int64_t)C * H * W, 256, 1024)), dim3(256), 0, w3d_s(stream), img, C, H, W, wy, wx, accumulate,
                       loss_accum, grad);
    W3D_CHECK_LAUNCH();
    return W3D_OK;
}

// l2_loss (utils/loss_utils.py:23-24): mean((a-b)^2); grad wrt a
__global__ __launch_bounds__(256) void mse_loss_kernel(const float *__restrict__ a, const float *__restrict__ b, int64_t n, float w, int accumulate,
                                                       float *__restrict__ loss_accum, float *__restrict__ grad) {
    __shared__ float red[16];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float d = a[i] - b[i];
        s += d * d;
        if (grad) { const float g = 2.0f * w * d; if (accumulate) grad[i] += g; else grad[i] = g; }
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0 && loss_accum) atomicAdd(loss_accum, w * s);
}

extern "C" int w3d_mse_loss_fwd_bwd(const float *a, const float *b, int64_t n, float weight, int accumulate, float *loss_accum, float *grad_a,
                                    w3d_stream_t stream) {
    if (!a || !b || n <= 0) return W3D_ERR_ARG;
    hipLaunchKernelGGL(mse_loss_kernel, dim3(w3d_grid(n, 256, 1024)), dim3(256), 0, w3d_s(stream), a, b, n, weight / (float)n, accumulate, loss_accum,
                       grad_a);
    W3D_CHECK_LAUNCH();
    return W3D_OK;
}
