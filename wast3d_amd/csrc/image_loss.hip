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

// ssim (utils/loss_utils.py:34-64): 11x11 Gaussian window (sigma 1.5, zero padding, one window per plane: groups = channel), C1 = 0.01^2, C2 = 0.03^2.
//   mu1 = G*x, mu2 = G*y, e1 = G*x^2, e2 = G*y^2, e12 = G*xy;   s1 = e1 - mu1^2, s2 = e2 - mu2^2, s12 = e12 - mu1 mu2
//   S = (2 mu1 mu2 + C1)(2 s12 + C2) / ((mu1^2 + mu2^2 + C1)(s1 + s2 + C2)) = A1 A2 / (B1 B2)
// The window is separable (the reference builds it as the outer product of the normalised 1-D window): a 16x16 output tile stages its 26x26 inputs in LDS, runs the
// five row filters, then the five column filters — 2 x 11 taps per sum instead of 121.  Forward writes, next to the per-plane sums of S, the three maps the backward
// needs (dS/dmu1, dS/de1, dS/de12 with S expressed in mu1, e1, e12); backward filters those with the same window and folds them with x and y:
//   dL/dx = w_plane * ( G*a + 2 x (G*b) + y (G*c) ).
namespace {
constexpr int SSIM_R = 5, SSIM_T = 16, SSIM_IN = SSIM_T + 2 * SSIM_R;
struct SsimWindow { float g[2 * SSIM_R + 1]; };
SsimWindow ssim_window() {               // gaussian(11, 1.5) as the reference forms it: float32 values of the double exp, normalised in float32
    SsimWindow w; float s = 0.f;
    for (int i = 0; i <= 2 * SSIM_R; ++i) { w.g[i] = (float)exp(-(double)((i - SSIM_R) * (i - SSIM_R)) / (2.0 * 1.5 * 1.5)); s += w.g[i]; }
    for (int i = 0; i <= 2 * SSIM_R; ++i) w.g[i] /= s;
    return w;
}

__global__ __launch_bounds__(256) void ssim_fwd_kernel(const float *__restrict__ x, const float *__restrict__ y, int H, int W, SsimWindow win, float *__restrict__ plane_sums,
                                                       float *__restrict__ dmap, int64_t map_stride) {
    __shared__ float sx[SSIM_IN][SSIM_IN + 1], sy[SSIM_IN][SSIM_IN + 1];
    __shared__ float hz[5][SSIM_IN][SSIM_T + 1];
    __shared__ float red[16];
    const int tid = threadIdx.x, tx = tid % SSIM_T, ty = tid / SSIM_T;        // 1-D block of 256 (block_sum assumes it): thread (tx, ty) of the 16x16 tile
    const int x0 = blockIdx.x * SSIM_T - SSIM_R, y0 = blockIdx.y * SSIM_T - SSIM_R;
    const int64_t plane = (int64_t)blockIdx.z * H * W;
    for (int i = tid; i < SSIM_IN * SSIM_IN; i += 256) {
        const int r = i / SSIM_IN, c = i % SSIM_IN, gy = y0 + r, gx = x0 + c;
        const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
        sx[r][c] = in ? x[plane + (int64_t)gy * W + gx] : 0.f;
        sy[r][c] = in ? y[plane + (int64_t)gy * W + gx] : 0.f;
    }
    __syncthreads();
    for (int i = tid; i < SSIM_IN * SSIM_T; i += 256) {
        const int r = i / SSIM_T, c = i % SSIM_T;
        float a = 0.f, b = 0.f, aa = 0.f, bb = 0.f, ab = 0.f;
#pragma unroll
        for (int t = 0; t <= 2 * SSIM_R; ++t) {
            const float u = sx[r][c + t], v = sy[r][c + t], g = win.g[t];
            a = fmaf(g, u, a); b = fmaf(g, v, b); aa = fmaf(g, u * u, aa); bb = fmaf(g, v * v, bb); ab = fmaf(g, u * v, ab);
        }
        hz[0][r][c] = a; hz[1][r][c] = b; hz[2][r][c] = aa; hz[3][r][c] = bb; hz[4][r][c] = ab;
    }
    __syncthreads();
    const int px = blockIdx.x * SSIM_T + tx, py = blockIdx.y * SSIM_T + ty;
    float S = 0.f;
    if (px < W && py < H) {
        float m1 = 0.f, m2 = 0.f, e1 = 0.f, e2 = 0.f, e12 = 0.f;
#pragma unroll
        for (int t = 0; t <= 2 * SSIM_R; ++t) {
            const float g = win.g[t];
            m1 = fmaf(g, hz[0][ty + t][tx], m1); m2 = fmaf(g, hz[1][ty + t][tx], m2);
            e1 = fmaf(g, hz[2][ty + t][tx], e1); e2 = fmaf(g, hz[3][ty + t][tx], e2);
            e12 = fmaf(g, hz[4][ty + t][tx], e12);
        }
        const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
        const float A1 = 2.f * m1 * m2 + C1, A2 = 2.f * (e12 - m1 * m2) + C2, B1 = m1 * m1 + m2 * m2 + C1, B2 = (e1 - m1 * m1) + (e2 - m2 * m2) + C2;
        const float inv = 1.f / (B1 * B2);
        S = A1 * A2 * inv;
        if (dmap) {
            const int64_t o = plane + (int64_t)py * W + px;
            dmap[o] = 2.f * m2 * (A2 - A1) * inv - S * 2.f * m1 * (1.f / B1 - 1.f / B2);      // dS/dmu1 (mu1 also enters s1 and s12)
            dmap[map_stride + o] = -S / B2;                                                     // dS/de1
            dmap[2 * map_stride + o] = 2.f * A1 * inv;                                          // dS/de12
        }
    }
    S = block_sum(S, red);
    if (tid == 0) atomicAdd(&plane_sums[blockIdx.z], S);
}

__global__ __launch_bounds__(256) void ssim_bwd_kernel(const float *__restrict__ x, const float *__restrict__ y, const float *__restrict__ dmap, int64_t map_stride, int H, int W,
                                                       SsimWindow win, const float *__restrict__ plane_w, int accumulate, float *__restrict__ grad) {
    __shared__ float sm[3][SSIM_IN][SSIM_IN + 1];
    __shared__ float hz[3][SSIM_IN][SSIM_T + 1];
    const int tid = threadIdx.x, tx = tid % SSIM_T, ty = tid / SSIM_T;        // 1-D block of 256 (block_sum assumes it): thread (tx, ty) of the 16x16 tile
    const int x0 = blockIdx.x * SSIM_T - SSIM_R, y0 = blockIdx.y * SSIM_T - SSIM_R;
    const int64_t plane = (int64_t)blockIdx.z * H * W;
    for (int i = tid; i < SSIM_IN * SSIM_IN; i += 256) {
        const int r = i / SSIM_IN, c = i % SSIM_IN, gy = y0 + r, gx = x0 + c;
        const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;              // outside the image no S exists: its derivative maps are 0 there
        const int64_t o = plane + (int64_t)gy * W + gx;
#pragma unroll
        for (int m = 0; m < 3; ++m) sm[m][r][c] = in ? dmap[m * map_stride + o] : 0.f;
    }
    __syncthreads();
    for (int i = tid; i < SSIM_IN * SSIM_T; i += 256) {
        const int r = i / SSIM_T, c = i % SSIM_T;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int t = 0; t <= 2 * SSIM_R; ++t) { const float g = win.g[t]; s0 = fmaf(g, sm[0][r][c + t], s0); s1 = fmaf(g, sm[1][r][c + t], s1); s2 = fmaf(g, sm[2][r][c + t], s2); }
        hz[0][r][c] = s0; hz[1][r][c] = s1; hz[2][r][c] = s2;
    }
    __syncthreads();
    const int px = blockIdx.x * SSIM_T + tx, py = blockIdx.y * SSIM_T + ty;
    if (px < W && py < H) {
        float ga = 0.f, gb = 0.f, gc = 0.f;
#pragma unroll
        for (int t = 0; t <= 2 * SSIM_R; ++t) {
            const float g = win.g[t];
            ga = fmaf(g, hz[0][ty + t][tx], ga); gb = fmaf(g, hz[1][ty + t][tx], gb); gc = fmaf(g, hz[2][ty + t][tx], gc);
        }
        const int64_t o = plane + (int64_t)py * W + px;
        const float v = plane_w[blockIdx.z] * (ga + 2.f * x[o] * gb + y[o] * gc);
        if (accumulate) grad[o] += v; else grad[o] = v;
    }
}
}  // namespace

// planes = product of the leading dimensions (every [H,W] plane has its own window: groups = channel).  plane_sums [planes] += sum of the SSIM map of each plane (the
// caller zeroes it and divides by the element count: mean over everything, or per batch item for size_average = False); dmaps [3][planes*H*W] (may be NULL: value only).
extern "C" int w3d_ssim_fwd(const float *img1, const float *img2, int planes, int H, int W, float *plane_sums, float *dmaps, w3d_stream_t stream) {
    if (!img1 || !img2 || !plane_sums || planes <= 0 || planes > 65535 || H <= 0 || W <= 0) return W3D_ERR_ARG;
    hipLaunchKernelGGL(ssim_fwd_kernel, dim3((W + SSIM_T - 1) / SSIM_T, (H + SSIM_T - 1) / SSIM_T, planes), dim3(SSIM_T * SSIM_T), 0, w3d_s(stream), img1, img2, H, W, ssim_window(),
                       plane_sums, dmaps, (int64_t)planes * H * W);
    W3D_CHECK_LAUNCH();
    return W3D_OK;
}
// grad1 (+)= plane_w[plane] * d(sum of the plane's SSIM map)/d img1, from the maps w3d_ssim_fwd left in dmaps
extern "C" int w3d_ssim_bwd(const float *img1, const float *img2, const float *dmaps, int planes, int H, int W, const float *plane_w, int accumulate, float *grad1,
                            w3d_stream_t stream) {
    if (!img1 || !img2 || !dmaps || !plane_w || !grad1 || planes <= 0 || planes > 65535 || H <= 0 || W <= 0) return W3D_ERR_ARG;
    hipLaunchKernelGGL(ssim_bwd_kernel, dim3((W + SSIM_T - 1) / SSIM_T, (H + SSIM_T - 1) / SSIM_T, planes), dim3(SSIM_T * SSIM_T), 0, w3d_s(stream), img1, img2, dmaps,
                       (int64_t)planes * H * W, H, W, ssim_window(), plane_w, accumulate, grad1);
    W3D_CHECK_LAUNCH();
    return W3D_OK;
}
