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

// LEAN form (zero_grad bit 1): the same arithmetic, one launch per group with the group's pointers as kernel arguments (scalar bases + one 32-bit offset per lane) and a
// float2 per thread and pass: under 32 VGPRs and no LDS, so a wave of it fits on a SIMD NEXT TO the two 240-VGPR waves the persistent emit sweep keeps there — a pure HBM
// stream that runs UNDER the distance-GEMM (which leaves HBM idle) instead of next to the fused backward (scripts/under_sweep_probe.py).
#ifndef W3D_ADAM_NT
#define W3D_ADAM_NT 1
#endif
// (the lean kernel's accesses are non-temporal: 1.6 GB streamed once should not push the sweep's style tiles out of the last-level cache)
__device__ __forceinline__ float2 ntl(const float2 *p) {
#if W3D_ADAM_NT
    typedef float v2f __attribute__((ext_vector_type(2)));
    const v2f v = __builtin_nontemporal_load(reinterpret_cast<const v2f *>(p)); return make_float2(v.x, v.y);
#else
    return *p;
#endif
}
__device__ __forceinline__ void nts(float2 *p, float2 v) {
#if W3D_ADAM_NT
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f w; w.x = v.x; w.y = v.y; __builtin_nontemporal_store(w, reinterpret_cast<v2f *>(p));
#else
    *p = v;
#endif
}
__global__ __launch_bounds__(256) void adam_lean_kernel(float *__restrict__ param, float *__restrict__ grad, float *__restrict__ grad2, float *__restrict__ exp_avg,
                                                        float *__restrict__ exp_avg_sq, uint32_t numel, float step, float bc2_sqrt, float one_m_b1, float b2, float one_m_b2,
                                                        float eps, int zero_grad) {
    const uint32_t pairs = numel >> 1;
    // Round 6: TWO float2 per thread and pass, all their loads issued before the first use.  Under the sweep exactly one 32-VGPR wave fits on a SIMD (512 - 2 x 240), so the
    // stream's rate is bytes in flight per wave / latency: with one float2 per array in flight the 45-float group took the whole sweep and its tail ran next to the select kernel
    // (profiles/r06aq_two_stream_timeline.log).  The accesses are raw BUFFER loads / stores — a scalar resource per array and ONE 32-bit byte offset per element pair — because ten
    // 64-bit addresses alone would take 20 of the 32 VGPRs.  Same arithmetic per element.
    const uint32_t stride = gridDim.x * 256u;
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (numel <= 0x3fffffffu) {                                        // (byte offsets fit 32 bits)
        typedef int i32x2_t __attribute__((ext_vector_type(2)));
        const int nbytes = (int)(numel * 4u);
        const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(param, (short)0, nbytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(grad, (short)0, nbytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rg2 = __builtin_amdgcn_make_buffer_rsrc(grad2 ? grad2 : grad, (short)0, nbytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(exp_avg, (short)0, nbytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(exp_avg_sq, (short)0, nbytes, 0x00020000);
        constexpr int NT = W3D_ADAM_NT ? 2 : 0;                        // cache-policy bits of the buffer instructions: nt
        auto ld = [&](const __amdgpu_buffer_rsrc_t r, uint32_t off) -> float2 {
            const i32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, NT);
            return make_float2(__int_as_float(v[0]), __int_as_float(v[1]));
        };
        auto st = [&](const __amdgpu_buffer_rsrc_t r, uint32_t off, float2 v) {
            i32x2_t w; w[0] = __float_as_int(v.x); w[1] = __float_as_int(v.y);
            __builtin_amdgcn_raw_buffer_store_b64(w, r, (int)off, 0, NT);
        };
        for (; i + stride < pairs; i += 2u * stride) {
            const uint32_t o0 = i * 8u, o1 = (i + stride) * 8u;
            float2 p0 = ld(rp, o0), p1 = ld(rp, o1), g0 = ld(rg, o0), g1 = ld(rg, o1), m0 = ld(rm, o0), m1 = ld(rm, o1), v0 = ld(rv, o0), v1 = ld(rv, o1);
            if (grad2) { const float2 a = ld(rg2, o0), b = ld(rg2, o1); g0.x += a.x; g0.y += a.y; g1.x += b.x; g1.y += b.y; }
            adam1(p0.x, g0.x, m0.x, v0.x, step, one_m_b1, b2, one_m_b2, 0.f, bc2_sqrt, eps);
            adam1(p0.y, g0.y, m0.y, v0.y, step, one_m_b1, b2, one_m_b2, 0.f, bc2_sqrt, eps);
            st(rp, o0, p0); st(rm, o0, m0); st(rv, o0, v0);
            adam1(p1.x, g1.x, m1.x, v1.x, step, one_m_b1, b2, one_m_b2, 0.f, bc2_sqrt, eps);
            adam1(p1.y, g1.y, m1.y, v1.y, step, one_m_b1, b2, one_m_b2, 0.f, bc2_sqrt, eps);
            st(rp, o1, p1); st(rm, o1, m1); st(rv, o1, v1);
            if (zero_grad) {
                st(rg, o0, make_float2(0.f, 0.f)); st(rg, o1, make_float2(0.f, 0.f));
                if (grad2) { st(rg2, o0, make_float2(0.f, 0.f)); st(rg2, o1, make_float2(0.f, 0.f)); }
            }
        }
    }
    for (; i < pairs; i += stride) {
        float2 p = ntl(reinterpret_cast<float2 *>(param) + i);
        float2 g = ntl(reinterpret_cast<const float2 *>(grad) + i);
        if (grad2) { const float2 g2 = ntl(reinterpret_cast<const float2 *>(grad2) + i); g.x += g2.x; g.y += g2.y; }
        float2 m = ntl(reinterpret_cast<float2 *>(exp_avg) + i);
        float2 v = ntl(reinterpret_cast<float2 *>(exp_avg_sq) + i);
        adam1(p.x, g.x, m.x, v.x, step, one_m_b1, b2, one_m_b2, 0.f, bc2_sqrt, eps);
        adam1(p.y, g.y, m.y, v.y, step, one_m_b1, b2, one_m_b2, 0.f, bc2_sqrt, eps);
        nts(reinterpret_cast<float2 *>(param) + i, p);
        nts(reinterpret_cast<float2 *>(exp_avg) + i, m);
        nts(reinterpret_cast<float2 *>(exp_avg_sq) + i, v);
        if (zero_grad) {
            nts(reinterpret_cast<float2 *>(grad) + i, make_float2(0.f, 0.f));
            if (grad2) nts(reinterpret_cast<float2 *>(grad2) + i, make_float2(0.f, 0.f));
        }
    }
    if ((numel & 1u) && blockIdx.x == 0 && threadIdx.x == 0) {          // an odd group's last element
        const uint32_t j = numel - 1u;
        float p = param[j], m = exp_avg[j], v = exp_avg_sq[j];
        adam1(p, grad[j] + (grad2 ? grad2[j] : 0.f), m, v, step, one_m_b1, b2, one_m_b2, 0.f, bc2_sqrt, eps);
        param[j] = p; exp_avg[j] = m; exp_avg_sq[j] = v;
        if (zero_grad) { grad[j] = 0.f; if (grad2) grad2[j] = 0.f; }
    }
}

extern "C" int w3d_adam_step(const W3dAdamGroup *groups_host, int n_groups, int step, double beta1, double beta2, double eps, int zero_grad,
                             w3d_stream_t stream) {
    if (!groups_host || n_groups <= 0 || n_groups > W3D_ADAM_MAX_GROUPS || step < 1) return W3D_ERR_ARG;
    if (zero_grad < 0 || zero_grad > 3) return W3D_ERR_ARG;          // a bit field since ABI 5 (bit 0 = zero the gradients, bit 1 = the lean kernel): 2 no longer means "true" 
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
    bool lean = (zero_grad & 2) != 0;
    // (the lean kernel's float2 accesses touch all five arrays: every base was checked for 16-B alignment above)
    for (int i = 0; i < n_groups; ++i) lean = lean && groups_host[i].numel < (1ll << 32) - 1024;
    if (lean) {
        for (int i = 0; i < n_groups; ++i) {
            const W3dAdamGroup &g = groups_host[i];
            if (g.numel == 0) continue;
            // one workgroup per CU = the one wave per SIMD that fits under the sweep anyway, without the dispatcher's turnaround between short-lived workgroups
            // (2048 -> 256 blocks: step 6.27 -> 6.24 ms, profiles/r06at_adam_grid_ab.log; fewer than that only stretches the stream, the sweep does not notice either way)
            hipLaunchKernelGGL(adam_lean_kernel, dim3(w3d_grid((g.numel + 1) / 2, 256, 256)), dim3(256), 0, w3d_s(stream), g.param, g.grad, g.grad2, g.exp_avg, g.exp_avg_sq,
                               (uint32_t)g.numel, t.step_size[i], (float)sqrt(bc2), (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, zero_grad & 1);
        }
    } else
    hipLaunchKernelGGL(adam_kernel, dim3(w3d_grid(t.vec_start[n_groups], 256, 4096)), dim3(256), 0, w3d_s(stream), t, (float)sqrt(bc2), (float)(1.0 - beta1),
                       (float)beta2, (float)(1.0 - beta2), (float)eps, zero_grad & 1);
    W3D_CHECK_LAUNCH();
    return W3D_OK;
}
