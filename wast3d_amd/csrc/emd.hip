// a5  exact earth mover's distance between two uniform, equally sized point clouds
// (reference call site: aux_optimize_cluster_D_W_distance.py:259-270 — ot.dist + ot.emd2 with weights 1/n on 100 samples).
//
// With uniform weights and equal counts the transport polytope's optimum is a permutation (Birkhoff), so
// emd2(w, w, M) = min_sigma (1/n) sum_i M[i, sigma(i)]: a linear assignment problem.  POT solves it with a network simplex on the
// CPU; here one WAVE per problem runs the shortest-augmenting-path (Jonker-Volgenant / "Hungarian with potentials") algorithm:
//   * the n columns are spread over the 64 lanes (column j on lane j % 64), potentials and slack in f64 in LDS,
//   * every inner step is one lane-parallel relaxation + a wave arg-min (DPP shuffles, no barriers, no atomics),
//   * the cost entry M[i,j] is either read from the caller's matrix or formed on the fly as |p_i - q_j|^2 (ot.dist default).
// Work is O(n^3) scalar-equivalent, n = 100 in the reference: latency-bound and tiny; B independent problems fill B waves.
// The gradient of emd2 wrt M is the optimal plan (POT's autograd convention): 1/n at (i, sigma(i)).
#include "common.h"

namespace {

constexpr int EMD_MAX_N = 1024;
constexpr int EMD_MAX_D = 8;

struct ArgMin { double v; int j; };

__device__ __forceinline__ ArgMin wave_argmin(double v, int j) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_xor(v, o, 64);
        const int oj = __shfl_xor(j, o, 64);
        if (ov < v || (ov == v && oj < j)) { v = ov; j = oj; }
    }
    return ArgMin{v, j};
}

// One wave per problem.  MATRIX: cost read from Mat[b] (row-major, leading dimension ld); else cost = |P_i - Q_j|^2.
template <bool MATRIX>
__global__ __launch_bounds__(64) void emd_assign_kernel(const float *__restrict__ Mat, int64_t ld, int64_t mat_stride, const float *__restrict__ P,
                                                        const float *__restrict__ Q, int n, int d, float weight, float *__restrict__ loss,
                                                        float *__restrict__ gM, float *__restrict__ gP, float *__restrict__ gQ, int32_t *__restrict__ match) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    // layout: u[n+1] v[n+1] minv[n+1] (f64) | p[n+1] way[n+1] (i32) | used[n+1] (u8, padded) | pts (2*n*d f32, points mode)
    double *u = reinterpret_cast<double *>(smem);
    double *v = u + (n + 1);
    double *minv = v + (n + 1);
    int *p = reinterpret_cast<int *>(minv + (n + 1));
    int *way = p + (n + 1);
    uint8_t *used = reinterpret_cast<uint8_t *>(way + (n + 1));
    float *pp = reinterpret_cast<float *>(smem + (((size_t)(3 * 8 + 2 * 4 + 1) * (n + 1) + 15) & ~(size_t)15));
    float *qq = pp + (size_t)n * d;
    const int lane = threadIdx.x;
    const int b = blockIdx.x;
    const float *Mb = MATRIX ? Mat + (int64_t)b * mat_stride : nullptr;
    if (!MATRIX) {
        const float *Pb = P + (int64_t)b * n * d, *Qb = Q + (int64_t)b * n * d;
        for (int t = lane; t < n * d; t += 64) { pp[t] = Pb[t]; qq[t] = Qb[t]; }
    }
    for (int j = lane; j <= n; j += 64) { u[j] = 0.0; v[j] = 0.0; p[j] = 0; way[j] = 0; }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    auto cost = [&](int i, int j) -> double {          // 0-based row i, column j
        if (MATRIX) return (double)Mb[(int64_t)i * ld + j];
        float s = 0.f;
        for (int c = 0; c < d; ++c) { const float t = pp[i * d + c] - qq[j * d + c]; s = fmaf(t, t, s); }
        return (double)s;
    };

    const double INF = 1e300;
    for (int i = 1; i <= n; ++i) {
        if (lane == 0) p[0] = i;
        for (int j = lane; j <= n; j += 64) { minv[j] = INF; used[j] = 0; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        int j0 = 0;
        while (true) {
            if (lane == 0) used[j0] = 1;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int i0 = p[j0];
            const double ui0 = u[i0];
            double best = INF; int bj = 0x7fffffff;
            for (int j = 1 + lane; j <= n; j += 64) {
                if (!used[j]) {
                    const double cur = cost(i0 - 1, j - 1) - ui0 - v[j];
                    double mv = minv[j];
                    if (cur < mv) { mv = cur; minv[j] = cur; way[j] = j0; }
                    if (mv < best || (mv == best && j < bj)) { best = mv; bj = j; }
                }
            }
            const ArgMin am = wave_argmin(best, bj);
            const double delta = am.v;
            const int j1 = am.j;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (int j = lane; j <= n; j += 64) {
                if (used[j]) { u[p[j]] += delta; v[j] -= delta; }      // p[j] distinct over used columns: no write conflict
                else minv[j] -= delta;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            j0 = j1;
            if (p[j0] == 0) break;
        }
        if (lane == 0) {                                   // augment along the alternating path (short, serial)
            int jj = j0;
            while (jj) { const int jn = way[jj]; p[jj] = p[jn]; jj = jn; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }

    // results: column j is matched to row p[j]-1.  Total in f64, summed in fixed lane order.
    double tot = 0.0;
    const float wn = weight / (float)n;
    for (int j = 1 + lane; j <= n; j += 64) {
        const int i = p[j] - 1, jc = j - 1;
        tot += cost(i, jc);
        if (match) match[(int64_t)b * n + i] = jc;
        if (MATRIX) {
            if (gM) gM[(int64_t)b * mat_stride + (int64_t)i * ld + jc] += wn;
        } else {
            for (int c = 0; c < d; ++c) {
                const float g = 2.0f * wn * (pp[i * d + c] - qq[jc * d + c]);
                if (gP) gP[((int64_t)b * n + i) * d + c] += g;
                if (gQ) gQ[((int64_t)b * n + jc) * d + c] -= g;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o, 64);
    if (lane == 0 && loss) loss[b] += (float)((double)weight * tot / (double)n);
}

size_t emd_lds_bytes(int n, int d, bool matrix) {
    size_t s = (((size_t)(3 * 8 + 2 * 4 + 1) * (n + 1) + 15) & ~(size_t)15);
    if (!matrix) s += (size_t)2 * n * d * sizeof(float);
    return s;
}

}  // namespace

extern "C" int w3d_emd2_matrix_fwd_bwd(const float *M, int B, int n, int64_t ld, float weight, float *loss, float *g_M, int32_t *match,
                                       w3d_stream_t stream) {
    if (!M || B < 0 || n < 1 || ld < n) return W3D_ERR_ARG;
    if (n > EMD_MAX_N) return W3D_ERR_UNSUPPORTED;
    if (B == 0) return W3D_OK;
    const size_t lds = emd_lds_bytes(n, 0, true);
    hipLaunchKernelGGL(emd_assign_kernel<true>, dim3(B), dim3(64), lds, w3d_s(stream), M, ld, (int64_t)n * ld, nullptr, nullptr, n, 0, weight, loss, g_M,
                       nullptr, nullptr, match);
    W3D_CHECK_LAUNCH();
    return W3D_OK;
}

extern "C" int w3d_emd2_points_fwd_bwd(const float *P, const float *Q, int B, int n, int d, float weight, float *loss, float *g_P, float *g_Q,
                                       int32_t *match, w3d_stream_t stream) {
    if (!P || !Q || B < 0 || n < 1 || d < 1) return W3D_ERR_ARG;
    if (n > EMD_MAX_N || d > EMD_MAX_D) return W3D_ERR_UNSUPPORTED;
    if (B == 0) return W3D_OK;
    const size_t lds = emd_lds_bytes(n, d, false);
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&emd_assign_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return W3D_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(emd_assign_kernel<false>, dim3(B), dim3(64), lds, w3d_s(stream), nullptr, (int64_t)0, (int64_t)0, P, Q, n, d, weight, loss, nullptr, g_P,
                       g_Q, match);
    W3D_CHECK_LAUNCH();
    return W3D_OK;
}

// ------------------------------------------------------------------------------------------------------------------------------------------------------
// a5, second head: the dual of entropic optimal transport as the reference's notebook maximises it (notebooks/test_EMD_local_stats.ipynb raw lines 208 / 994:
// `ot.stochastic.loss_dual_entropic(u, v, xs, xt, reg)` inside an Adam loop over the potentials, then `ot.stochastic.plan_dual_entropic`).  POT (third party, version
// not pinned by the reference, absent here) publishes it as
//     M_ij = |xs_i - xt_j|^2 (ot.dist default),  H_ij = exp((u_i + v_j - M_ij) / reg)
//     loss = sum_i u_i ws_i + sum_j v_j wt_j - reg * sum_ij ws_i H_ij wt_j                         (ws, wt default to uniform 1/ns, 1/nt)
//     plan = ws_i H_ij wt_j
// Gradients (torch autograd of that expression): dL/du_i = ws_i (1 - sum_j H_ij wt_j), dL/dv_j = wt_j (1 - sum_i ws_i H_ij),
// dL/dxs_i = 2 ws_i sum_j wt_j H_ij (xs_i - xt_j), dL/dxt_j = -2 wt_j sum_i ws_i H_ij (xs_i - xt_j).
// Kernel: one wave per row of the side being reduced (pass 0: rows of xs -> g_u, g_xs, the loss; pass 1: rows of xt -> g_v, g_xt); sums in f64 (the notebook runs
// f64 tensors), no atomics on the gradients: bit-reproducible.  Sizes are the notebook's (100 ... a few thousand points, d = 2 or 3): latency-sized work.
namespace {
constexpr int DE_MAX_D = 8;

// side 0: a = xs, pa = u, wa = ws, b = xt, pb = v, wb = wt;  side 1: roles swapped (the kernel is symmetric up to the sign of the position gradient)
__global__ __launch_bounds__(256) void dual_entropic_kernel(const float *__restrict__ A, const float *__restrict__ pa, const float *__restrict__ wa, int na,
                                                            const float *__restrict__ B, const float *__restrict__ pb, const float *__restrict__ wb, int nb, int d,
                                                            float reg, float *__restrict__ g_pa, float *__restrict__ g_A, double *__restrict__ loss_acc, int with_loss) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= na) return;
    float ai[DE_MAX_D];
    for (int c = 0; c < DE_MAX_D; ++c) ai[c] = c < d ? A[(int64_t)i * d + c] : 0.f;
    const double ui = (double)pa[i];
    const double wi = wa ? (double)wa[i] : 1.0 / (double)na;
    const double inv_reg = 1.0 / (double)reg;
    double s = 0.0, gx[DE_MAX_D];
    for (int c = 0; c < DE_MAX_D; ++c) gx[c] = 0.0;
    for (int j = lane; j < nb; j += 64) {
        double m = 0.0, df[DE_MAX_D];
        for (int c = 0; c < DE_MAX_D; ++c) { df[c] = c < d ? (double)ai[c] - (double)B[(int64_t)j * d + c] : 0.0; m += df[c] * df[c]; }
        const double wj = wb ? (double)wb[j] : 1.0 / (double)nb;
        const double h = exp((ui + (double)pb[j] - m) * inv_reg) * wj;
        s += h;
        for (int c = 0; c < DE_MAX_D; ++c) gx[c] += h * df[c];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (g_A) {
        for (int c = 0; c < d; ++c) {
            double t = gx[c];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
            if (lane == 0) g_A[(int64_t)i * d + c] = (float)(2.0 * wi * t);
        }
    }
    if (lane == 0) {
        if (g_pa) g_pa[i] = (float)(wi * (1.0 - s));
        // loss: u.ws - reg * ws_i sum_j H_ij wt_j from the xs side; v.wt from the xt side (with_loss = 2)
        if (with_loss == 1) atomicAdd(loss_acc, ui * wi - (double)reg * wi * s);
        else if (with_loss == 2) atomicAdd(loss_acc, ui * wi);
    }
}

__global__ __launch_bounds__(256) void dual_entropic_plan_kernel(const float *__restrict__ xs, const float *__restrict__ xt, const float *__restrict__ u,
                                                                 const float *__restrict__ v, const float *__restrict__ ws, const float *__restrict__ wt, int ns, int nt,
                                                                 int d, float reg, float *__restrict__ plan) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)ns * nt) return;
    const int i = (int)(e / nt), j = (int)(e % nt);
    double m = 0.0;
    for (int c = 0; c < d; ++c) { const double df = (double)xs[(int64_t)i * d + c] - (double)xt[(int64_t)j * d + c]; m += df * df; }
    const double wi = ws ? (double)ws[i] : 1.0 / (double)ns, wj = wt ? (double)wt[j] : 1.0 / (double)nt;
    plan[e] = (float)(wi * exp(((double)u[i] + (double)v[j] - m) / (double)reg) * wj);
}

__global__ void dual_entropic_finish_kernel(const double *__restrict__ acc, float weight, float *__restrict__ loss) { loss[0] += weight * (float)acc[0]; }
}  // namespace

// loss[0] += weight * L;  g_u [ns], g_v [nt] (=), optional g_xs [ns,d], g_xt [nt,d] (=) receive dL/d. (unweighted: the caller scales);  ws / wt may be NULL (uniform).
// scratch: 8 bytes, device (f64 accumulator).
extern "C" int w3d_dual_entropic_fwd_bwd(const float *u, const float *v, const float *xs, const float *xt, const float *ws, const float *wt, int ns, int nt, int d,
                                         float reg, float weight, float *loss, float *g_u, float *g_v, float *g_xs, float *g_xt, void *scratch, w3d_stream_t stream) {
    if (!u || !v || !xs || !xt || !loss || !scratch || ns < 1 || nt < 1 || d < 1 || !(reg > 0.f)) return W3D_ERR_ARG;
    if (d > DE_MAX_D) return W3D_ERR_UNSUPPORTED;
    hipStream_t st = w3d_s(stream);
    double *acc = reinterpret_cast<double *>(scratch);
    W3D_HIP(hipMemsetAsync(acc, 0, sizeof(double), st));
    hipLaunchKernelGGL(dual_entropic_kernel, dim3((unsigned)((ns + 3) / 4)), dim3(256), 0, st, xs, u, ws, ns, xt, v, wt, nt, d, reg, g_u, g_xs, acc, 1);
    hipLaunchKernelGGL(dual_entropic_kernel, dim3((unsigned)((nt + 3) / 4)), dim3(256), 0, st, xt, v, wt, nt, xs, u, ws, ns, d, reg, g_v, g_xt, acc, 2);
    hipLaunchKernelGGL(dual_entropic_finish_kernel, dim3(1), dim3(1), 0, st, (const double *)acc, weight, loss);
    W3D_CHECK_LAUNCH();
    return W3D_OK;
}

extern "C" int w3d_dual_entropic_plan(const float *u, const float *v, const float *xs, const float *xt, const float *ws, const float *wt, int ns, int nt, int d, float reg,
                                      float *plan, w3d_stream_t stream) {
    if (!u || !v || !xs || !xt || !plan || ns < 1 || nt < 1 || d < 1 || !(reg > 0.f)) return W3D_ERR_ARG;
    hipLaunchKernelGGL(dual_entropic_plan_kernel, dim3((unsigned)(((int64_t)ns * nt + 255) / 256)), dim3(256), 0, w3d_s(stream), xs, xt, u, v, ws, wt, ns, nt, d, reg, plan);
    W3D_CHECK_LAUNCH();
    return W3D_OK;
}

// ---- ot.dist(x1, x2) with POT's default metric (aux_optimize_cluster_D_W_distance.py:268: M = ot.dist(p, q)): the squared Euclidean cost matrix, and its backward.
// POT's published formula (ot.utils.euclidean_distances, squared = True): c_ij = max(|a_i|^2 + |b_j|^2 - 2 a_i.b_j, 0); the same expression here, in f32.
namespace {
__global__ __launch_bounds__(256) void sqdist_matrix_kernel(const float *__restrict__ a, const float *__restrict__ b, int n1, int n2, int d, float *__restrict__ out) {
    const int64_t n = (int64_t)n1 * n2;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const int i = (int)(e / n2), j = (int)(e - (int64_t)i * n2);
        const float *x = a + (size_t)i * d, *y = b + (size_t)j * d;
        float x2 = 0.f, y2 = 0.f, xy = 0.f;
        for (int c = 0; c < d; ++c) { x2 = fmaf(x[c], x[c], x2); y2 = fmaf(y[c], y[c], y2); xy = fmaf(x[c], y[c], xy); }
        out[e] = fmaxf(x2 + y2 - 2.0f * xy, 0.f);
    }
}
// g_a [n1,d] = sum_j gM_ij 2 (a_i - b_j)   (transposed = 1: the roles swapped, gM read as gM_ji — the gradient of the second cloud); one thread per (row, feature),
// no atomics: bit-reproducible
__global__ __launch_bounds__(256) void sqdist_matrix_bwd_kernel(const float *__restrict__ gM, const float *__restrict__ a, const float *__restrict__ b, int n1, int n2, int d,
                                                                int transposed, float *__restrict__ g_a) {
    const int64_t n = (int64_t)n1 * d;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const int i = (int)(e / d), c = (int)(e - (int64_t)i * d);
        const float ai = a[e];
        float acc = 0.f;
        for (int j = 0; j < n2; ++j) {
            const float g = transposed ? gM[(size_t)j * n1 + i] : gM[(size_t)i * n2 + j];
            acc = fmaf(g, 2.0f * (ai - b[(size_t)j * d + c]), acc);
        }
        g_a[e] = acc;
    }
}
}  // namespace

extern "C" int w3d_sqdist_matrix_fwd(const float *x1, const float *x2, int n1, int n2, int d, float *out, w3d_stream_t stream) {
    if (!x1 || !x2 || !out || n1 < 1 || n2 < 1 || d < 1) return W3D_ERR_ARG;
    hipLaunchKernelGGL(sqdist_matrix_kernel, dim3(w3d_grid((int64_t)n1 * n2, 256, 4096)), dim3(256), 0, w3d_s(stream), x1, x2, n1, n2, d, out);
    W3D_CHECK_LAUNCH();
    return W3D_OK;
}

extern "C" int w3d_sqdist_matrix_bwd(const float *g_out, const float *x1, const float *x2, int n1, int n2, int d, float *g_x1, float *g_x2, w3d_stream_t stream) {
    if (!g_out || !x1 || !x2 || n1 < 1 || n2 < 1 || d < 1 || (!g_x1 && !g_x2)) return W3D_ERR_ARG;
    if (g_x1) hipLaunchKernelGGL(sqdist_matrix_bwd_kernel, dim3(w3d_grid((int64_t)n1 * d, 256, 4096)), dim3(256), 0, w3d_s(stream), g_out, x1, x2, n1, n2, d, 0, g_x1);
    if (g_x2) hipLaunchKernelGGL(sqdist_matrix_bwd_kernel, dim3(w3d_grid((int64_t)n2 * d, 256, 4096)), dim3(256), 0, w3d_s(stream), g_out, x2, x1, n2, n1, d, 1, g_x2);
    W3D_CHECK_LAUNCH();
    return W3D_OK;
}
