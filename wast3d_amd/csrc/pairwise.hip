// Full-matrix variant of the geometry loop (aux_optimize_cluster_D_W_distance4.py:73-84,254-283) without any N x N matrix:
// every pairwise distance — current AND frozen target — is recomputed on the fly from the [N,3]/[N,4] inputs.
//   loss_D_xyz      = mean_ij (|x_i - x_j| - |t_i - t_j|)^2
//   loss_D_rotation = mean_ij m_ij ((|a_i - x_j| + |b_i - x_j|) - (|ta_i - t_j| + |tb_i - t_j|))^2     a = rot[:3], b = rot[1:]
//   loss_D_scaling  = mean_ij m_ij (|s_i - x_j| - |ts_i - t_j|)^2        m_ij = (|t_i - t_j| <= kth_i)  (k-th smallest of target row i)
// Two O(N^2) passes over LDS tiles: rows own the row-side gradients (xyz from loss_D_xyz, rotation, scaling), columns own the
// xyz gradient of the masked terms.  Also the NBVM coverage head (nearest fitted point per domain point).
#include "common.h"

namespace {
constexpr int PT = 256;     // points per LDS tile

__device__ __forceinline__ float dist3_mm(const float *x, float x2, float yx, float yy, float yz) {   // same bits as the k-th threshold kernel
    const float y[3] = {yx, yy, yz};
    return w3d_dist_mm<3>(x, x2, y);
}
__device__ __forceinline__ float sq3(const float *v) { return w3d_sqnorm<3>(v); }

struct AuxIn {
    const float *xyz, *rot, *scale, *t_xyz, *t_rot, *t_scale, *kth_t;
    int N;
};

// pass A: thread = row i, loops over columns j
__global__ __launch_bounds__(256) void auxw_rows_kernel(AuxIn in, float w_xyz, float w_rot, float w_sc, float *__restrict__ loss /*[3]*/, float *__restrict__ g_xyz,
                                                        float *__restrict__ g_rot, float *__restrict__ g_scale) {
    __shared__ float sx[PT * 3], st[PT * 3];
    __shared__ float red[16];
    const int N = in.N;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool live = i < N;
    const int ii = live ? i : 0;
    float x[3], t[3], a[3], b[3], s[3], ta[3], tb[3], ts[3];
    for (int c = 0; c < 3; ++c) {
        x[c] = in.xyz[3 * ii + c]; t[c] = in.t_xyz[3 * ii + c];
        a[c] = in.rot[4 * ii + c]; b[c] = in.rot[4 * ii + 1 + c]; s[c] = in.scale[3 * ii + c];
        ta[c] = in.t_rot[4 * ii + c]; tb[c] = in.t_rot[4 * ii + 1 + c]; ts[c] = in.t_scale[3 * ii + c];
    }
    const float x2 = sq3(x), t2 = sq3(t), a2 = sq3(a), b2 = sq3(b), s2 = sq3(s), ta2 = sq3(ta), tb2 = sq3(tb), ts2 = sq3(ts);
    const float kth = in.kth_t[ii];
    const float inv = 1.0f / ((float)N * (float)N);
    float gx[3] = {0, 0, 0}, ga[3] = {0, 0, 0}, gb[3] = {0, 0, 0}, gs[3] = {0, 0, 0};
    float l0 = 0.f, l1 = 0.f, l2 = 0.f;
    for (int base = 0; base < N; base += PT) {
        const int n = min(PT, N - base);
        __syncthreads();
        for (int q = threadIdx.x; q < n * 3; q += 256) { sx[q] = in.xyz[3 * base + q]; st[q] = in.t_xyz[3 * base + q]; }
        __syncthreads();
        for (int j = 0; j < n; ++j) {
            const float yx = sx[3 * j], yy = sx[3 * j + 1], yz = sx[3 * j + 2];
            const float ux = st[3 * j], uy = st[3 * j + 1], uz = st[3 * j + 2];
            const float Dt = dist3_mm(t, t2, ux, uy, uz);
            const float D = dist3_mm(x, x2, yx, yy, yz);
            const float e = D - Dt;
            l0 += e * e;
            if (D > 0.f) {   // row side + the symmetric column side of the mirrored pair (D and Dt are symmetric)
                const float w = 4.0f * w_xyz * inv * e / D;
                gx[0] += w * (x[0] - yx); gx[1] += w * (x[1] - yy); gx[2] += w * (x[2] - yz);
            }
            if (Dt <= kth) {
                const float Da = dist3_mm(a, a2, yx, yy, yz), Db = dist3_mm(b, b2, yx, yy, yz);
                const float er = (Da + Db) - (dist3_mm(ta, ta2, ux, uy, uz) + dist3_mm(tb, tb2, ux, uy, uz));
                l1 += er * er;
                const float wr = 2.0f * w_rot * inv * er;
                if (Da > 0.f) { const float w = wr / Da; ga[0] += w * (a[0] - yx); ga[1] += w * (a[1] - yy); ga[2] += w * (a[2] - yz); }
                if (Db > 0.f) { const float w = wr / Db; gb[0] += w * (b[0] - yx); gb[1] += w * (b[1] - yy); gb[2] += w * (b[2] - yz); }
                const float Ds = dist3_mm(s, s2, yx, yy, yz);
                const float es = Ds - dist3_mm(ts, ts2, ux, uy, uz);
                l2 += es * es;
                if (Ds > 0.f) { const float w = 2.0f * w_sc * inv * es / Ds; gs[0] += w * (s[0] - yx); gs[1] += w * (s[1] - yy); gs[2] += w * (s[2] - yz); }
            }
        }
    }
    if (live) {
        for (int c = 0; c < 3; ++c) { g_xyz[3 * i + c] += gx[c]; g_scale[3 * i + c] += gs[c]; }
        g_rot[4 * i] += ga[0]; g_rot[4 * i + 1] += ga[1] + gb[0]; g_rot[4 * i + 2] += ga[2] + gb[1]; g_rot[4 * i + 3] += gb[2];
    }
    const float r0 = block_sum(live ? l0 : 0.f, red), r1 = block_sum(live ? l1 : 0.f, red), r2 = block_sum(live ? l2 : 0.f, red);
    if (threadIdx.x == 0) { atomicAdd(&loss[0], w_xyz * inv * r0); atomicAdd(&loss[1], w_rot * inv * r1); atomicAdd(&loss[2], w_sc * inv * r2); }
}

// pass B: thread = column j (an xyz point), loops over rows i: xyz gradient of the masked rotation / scaling terms
__global__ __launch_bounds__(256) void auxw_cols_kernel(AuxIn in, float w_rot, float w_sc, float *__restrict__ g_xyz) {
    __shared__ float sr[PT * 4], ss[PT * 3], str[PT * 4], sts[PT * 3], stt[PT * 3], sk[PT];
    const int N = in.N;
    const int j = blockIdx.x * 256 + threadIdx.x;
    const bool live = j < N;
    const int jj = live ? j : 0;
    float y[3], u[3];
    for (int c = 0; c < 3; ++c) { y[c] = in.xyz[3 * jj + c]; u[c] = in.t_xyz[3 * jj + c]; }
    const float y2 = sq3(y), u2 = sq3(u);
    const float inv = 1.0f / ((float)N * (float)N);
    float g[3] = {0, 0, 0};
    for (int base = 0; base < N; base += PT) {
        const int n = min(PT, N - base);
        __syncthreads();
        for (int q = threadIdx.x; q < n * 4; q += 256) { sr[q] = in.rot[4 * base + q]; str[q] = in.t_rot[4 * base + q]; }
        for (int q = threadIdx.x; q < n * 3; q += 256) { ss[q] = in.scale[3 * base + q]; sts[q] = in.t_scale[3 * base + q]; stt[q] = in.t_xyz[3 * base + q]; }
        for (int q = threadIdx.x; q < n; q += 256) sk[q] = in.kth_t[base + q];
        __syncthreads();
        for (int i = 0; i < n; ++i) {
            // same operand roles as pass A (row quantities first) so that every distance is bit-identical
            const float *ti = &stt[3 * i];
            const float Dt = dist3_mm(ti, sq3(ti), u[0], u[1], u[2]);
            if (Dt <= sk[i]) {
                const float *a = &sr[4 * i], *b = &sr[4 * i + 1], *s = &ss[3 * i], *ta = &str[4 * i], *tb = &str[4 * i + 1], *ts = &sts[3 * i];
                const float Da = dist3_mm(a, sq3(a), y[0], y[1], y[2]), Db = dist3_mm(b, sq3(b), y[0], y[1], y[2]);
                const float er = (Da + Db) - (dist3_mm(ta, sq3(ta), u[0], u[1], u[2]) + dist3_mm(tb, sq3(tb), u[0], u[1], u[2]));
                const float wr = 2.0f * w_rot * inv * er;
                if (Da > 0.f) { const float w = wr / Da; g[0] -= w * (a[0] - y[0]); g[1] -= w * (a[1] - y[1]); g[2] -= w * (a[2] - y[2]); }
                if (Db > 0.f) { const float w = wr / Db; g[0] -= w * (b[0] - y[0]); g[1] -= w * (b[1] - y[1]); g[2] -= w * (b[2] - y[2]); }
                const float Ds = dist3_mm(s, sq3(s), y[0], y[1], y[2]);
                const float es = Ds - dist3_mm(ts, sq3(ts), u[0], u[1], u[2]);
                if (Ds > 0.f) { const float w = 2.0f * w_sc * inv * es / Ds; g[0] -= w * (s[0] - y[0]); g[1] -= w * (s[1] - y[1]); g[2] -= w * (s[2] - y[2]); }
            }
        }
    }
    if (live) { g_xyz[3 * j] += g[0]; g_xyz[3 * j + 1] += g[1]; g_xyz[3 * j + 2] += g[2]; }
    (void)y2; (void)u2;
}

// NBVM:512-516 coverage head: loss += w * mean_j D_j, D_j = distance from domain point j to its nearest fitted point
__global__ __launch_bounds__(256) void coverage_kernel(const float *__restrict__ domain, const float *__restrict__ points, const int32_t *__restrict__ nn,
                                                       const float *__restrict__ d2, int M, int d, float w, float *__restrict__ loss, float *__restrict__ gP) {
    __shared__ float red[16];
    float acc = 0.f;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < M; j += gridDim.x * 256) {
        const float D = sqrtf(d2[j]);
        acc += D;
        if (D > 0.f && gP) {
            const int64_t p = nn[j];
            for (int c = 0; c < d; ++c) atomicAdd(&gP[p * d + c], (w / (float)M) * (points[p * d + c] - domain[(int64_t)j * d + c]) / D);
        }
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) atomicAdd(loss, w * acc / (float)M);
}
}  // namespace

extern "C" int w3d_auxw_pairwise_fwd_bwd(const float *xyz, const float *rot, const float *scale, const float *t_xyz, const float *t_rot, const float *t_scale,
                                         const float *kth_target, int N, float w_xyz, float w_rot, float w_scale, float *loss3, float *g_xyz, float *g_rot,
                                         float *g_scale, w3d_stream_t stream) {
    if (!xyz || !rot || !scale || !t_xyz || !t_rot || !t_scale || !kth_target || !loss3 || !g_xyz || !g_rot || !g_scale || N <= 0) return W3D_ERR_ARG;
    AuxIn in{xyz, rot, scale, t_xyz, t_rot, t_scale, kth_target, N};
    const dim3 grid((N + 255) / 256);
    hipLaunchKernelGGL(auxw_rows_kernel, grid, dim3(256), 0, w3d_s(stream), in, w_xyz, w_rot, w_scale, loss3, g_xyz, g_rot, g_scale);
    hipLaunchKernelGGL(auxw_cols_kernel, grid, dim3(256), 0, w3d_s(stream), in, w_rot, w_scale, g_xyz);
    W3D_CHECK_LAUNCH();
    return W3D_OK;
}

extern "C" int w3d_coverage_loss_bwd(const float *domain, const float *points, const int32_t *nn_idx, const float *nn_dist2, int M, int d, float weight,
                                     float *loss_accum, float *g_points, w3d_stream_t stream) {
    if (!domain || !points || !nn_idx || !nn_dist2 || !loss_accum || M <= 0 || d <= 0) return W3D_ERR_ARG;
    hipLaunchKernelGGL(coverage_kernel, dim3(w3d_grid(M, 256, 256)), dim3(256), 0, w3d_s(stream), domain, points, nn_idx, nn_dist2, M, d, weight, loss_accum, g_points);
    W3D_CHECK_LAUNCH();
    return W3D_OK;
}
