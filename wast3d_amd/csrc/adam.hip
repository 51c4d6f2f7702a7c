// a9  fused multi-group Adam.  Pure streaming: 16 B read + 12 B write per parameter (28 B),
// +4 B when the gradient is zeroed in the same pass.  One launch covers all parameter groups: the group
// table travels as a kernel argument, each block walks a contiguous float4 span of one group.
#include "common.h"
#include <math.h>

struct AdamTable {
    W3dAdamGroup g[W3D_ADAM_MAX_GROUPS];
    int64_t vec_start[W3D_ADAM_MAX_GROUPS + 1];   // prefix sums of ceil(numel/4)
    float step_size[W3D_ADAM_MAX_GROUPS];         // lr / (1 - beta1^t), computed in double on the host like torch
    int n;
};

// torch/optim/adam.py::_single_tensor_adam order: lerp, mul+addcmul, sqrt/bc2sqrt + eps, addcdiv
__device__ __forceinline__ void adam1(float &p, float g, float &m, float &v, float lr_over_bc1, float one_m_b1, float b2, float one_m_b2,
                                      float inv_bc2_sqrt_is_div, float bc2_sqrt, float eps) {
    m = m + (g - m) * one_m_b1;
    v = v * b2 + one_m_b2 * g * g;
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    p = p - lr_over_bc1 * (m / denom);
    (void)inv_bc2_sqrt_is_div;
}

__global__ __launch_bounds__(256) void adam_kernel(AdamTable t, float bc2_sqrt, float one_m_b1, float b2, float one_m_b2, float eps, int zero_grad) {
    const int64_t total = t.vec_start[t.n];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        int gi = 0;
#pragma unroll
        for (int q = 1; q < W3D_ADAM_MAX_GROUPS; ++q) gi += (q < t.n && i >= t.vec_start[q]) ? 1 : 0;
        const W3dAdamGroup G = t.g[gi];
        const int64_t e = (i - t.vec_start[gi]) * 4;
        const float step = t.step_size[gi];
        if (e + 4 <= G.numel) {
            float4 p = *reinterpret_cast<float4 *>(G.param + e);
            float4 g = *reinterpret_cast<const float4 *>(G.grad + e);
            if (G.grad2) { const float4 g2 = *reinterpret_cast<const float4 *>(G.grad2 + e); g.x += g2.x; g.y += g2.y; g.z += g2.z; g.w += g2.w; }
            float4 m = *reinterpret_cast<float4 *>(G.exp_avg + e);
            float4 v = *reinterpret_cast<float4 *>(G.exp_avg_sq + e);
            adam1(p.x, g.x, m.x, v.x, step, one_m_b1, b2, one_m_b2, 0.f, bc2_sqrt, eps);
            adam1(p.y, g.y, m.y, v.y, step, one_m_b1, b2, one_m_b2, 0.f, bc2_sqrt, eps);
            adam1(p.z, g.z, m.z, v.z, step, one_m_b1, b2, one_m_b2, 0.f, bc2_sqrt, eps);
            adam1(p.w, g.w, m.w, v.w, step, one_m_b1, b2, one_m_b2, 0.f, bc2_sqrt, eps);
            *reinterpret_cast<float4 *>(G.param + e) = p;
            *reinterpret_cast<float4 *>(G.exp_avg + e) = m;
            *reinterpret_cast<float4 *>(G.exp_avg_sq + e) = v;
            if (zero_grad) {
                *reinterpret_cast<float4 *>(G.grad + e) = make_float4(0.f, 0.f, 0.f, 0.f);
                if (G.grad2) *reinterpret_cast<float4 *>(G.grad2 + e) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {
            for (int64_t j = e; j < G.numel; ++j) {
                float p = G.param[j], m = G.exp_avg[j], v = G.exp_avg_sq[j];
                adam1(p, G.grad[j] + (G.grad2 ? G.grad2[j] : 0.f), m, v, step, one_m_b1, b2, one_m_b2, 0.f, bc2_sqrt, eps);
                G.param[j] = p; G.exp_avg[j] = m; G.exp_avg_sq[j] = v;
                if (zero_grad) { G.grad[j] = 0.f; if (G.grad2) G.grad2[j] = 0.f; }
            }
        }
    }
}

extern "C" int w3d_adam_step(const W3dAdamGroup *groups_host, int n_groups, int step, double beta1, double beta2, double eps, int zero_grad,
                             w3d_stream_t stream) {
    if (!groups_host || n_groups <= 0 || n_groups > W3D_ADAM_MAX_GROUPS || step < 1) return W3D_ERR_ARG;
    AdamTable t;
    t.n = n_groups;
    t.vec_start[0] = 0;
    for (int i = 0; i < n_groups; ++i) {
        const W3dAdamGroup &g = groups_host[i];
        if (g.numel < 0 || (g.numel > 0 && (!g.param || !g.grad || !g.exp_avg || !g.exp_avg_sq))) return W3D_ERR_ARG;   // an empty group (numel 0, any pointers) is skipped
        // float4 path needs 16-B aligned bases (torch allocations are 256-B aligned)
        if (((uintptr_t)g.param | (uintptr_t)g.grad | (uintptr_t)g.exp_avg | (uintptr_t)g.exp_avg_sq | (uintptr_t)g.grad2) & 15) return W3D_ERR_ARG;
        t.g[i] = g;
        t.vec_start[i + 1] = t.vec_start[i] + (g.numel + 3) / 4;
    }
    for (int i = n_groups; i < W3D_ADAM_MAX_GROUPS; ++i) { t.g[i] = t.g[0]; t.vec_start[i + 1] = t.vec_start[n_groups]; }
    if (t.vec_start[n_groups] == 0) return W3D_OK;
    const double bc1 = 1.0 - pow(beta1, (double)step);
    for (int i = 0; i < W3D_ADAM_MAX_GROUPS; ++i) t.step_size[i] = (float)(t.g[i].lr / bc1);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    hipLaunchKernelGGL(adam_kernel, dim3(w3d_grid(t.vec_start[n_groups], 256, 4096)), dim3(256), 0, w3d_s(stream), t, (float)sqrt(bc2), (float)(1.0 - beta1),
                       (float)beta2, (float)(1.0 - beta2), (float)eps, zero_grad);
    W3D_CHECK_LAUNCH();
    return W3D_OK;
}
