// SURVEY §8f row 2 — domain partitioning: one Lloyd iteration of KMeans (prepare_scene, notebooks/11.fit_cluster_to_the_full_scene.ipynb raw
// lines 171-198, calls sklearn.cluster.KMeans on the CPU).  E-step and the sums of the M-step in one pass over the points:
//   label[i] = argmin_c ( |c|^2 - 2 x_i.c )      (the form sklearn's dense Lloyd iteration ranks with; first minimum on ties)
//   sums[c]  += x_i,  counts[c] += 1              (64-bit FIXED-POINT integer atomics: associative, so the new centres do not depend on
//                                                  the order the waves run in — same labels run to run)
// Centres (+ their squared norms) and the block's private accumulators live in LDS; a block flushes its accumulators with one global
// integer atomic per touched cluster and coordinate.  HBM traffic = the points once (4d B) + labels (4 B read for the change count, 4 B written).
#include "common.h"

namespace {

constexpr int KM_THREADS = 256;
constexpr int KM_MAX_D = 16;

__device__ __forceinline__ void lds_add_u64(unsigned long long *p, unsigned long long v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// LDS layout (dynamic): float cen[C*d] | float cn[C] | (pad to 8) | u64 acc[C*d] | u32 cnt[C]      (acc/cnt only if lds_acc)
template <int D>
__global__ __launch_bounds__(KM_THREADS) void kmeans_assign_kernel(const float *__restrict__ X, int64_t N, int d_rt, const float *__restrict__ centres, int C,
                                                                   int32_t *__restrict__ labels, unsigned long long *__restrict__ sums, unsigned int *__restrict__ counts,
                                                                   double fx_scale, int lds_acc, unsigned int *__restrict__ n_changed, double *__restrict__ inertia) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int d = D > 0 ? D : d_rt;
    float *cen = reinterpret_cast<float *>(smem);
    float *cn = cen + (size_t)C * d;
    unsigned long long *acc = reinterpret_cast<unsigned long long *>(smem + (((size_t)C * (d + 1) * 4 + 7) & ~(size_t)7));
    unsigned int *cnt = reinterpret_cast<unsigned int *>(acc + (size_t)C * d);
    const int tid = threadIdx.x;
    for (int i = tid; i < C * d; i += KM_THREADS) cen[i] = centres[i];
    __syncthreads();
    for (int c = tid; c < C; c += KM_THREADS) {
        float s = 0.f;
        for (int j = 0; j < d; ++j) s = fmaf(cen[c * d + j], cen[c * d + j], s);
        cn[c] = s;
    }
    if (lds_acc) {
        for (int i = tid; i < C * d; i += KM_THREADS) acc[i] = 0ull;
        for (int i = tid; i < C; i += KM_THREADS) cnt[i] = 0u;
    }
    __syncthreads();
    unsigned int changed = 0u;
    double inert = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * KM_THREADS + tid; i < N; i += (int64_t)gridDim.x * KM_THREADS) {
        float x[D > 0 ? D : KM_MAX_D];
#pragma unroll
        for (int j = 0; j < (D > 0 ? D : KM_MAX_D); ++j) x[j] = j < d ? X[i * d + j] : 0.f;
        float best = __builtin_inff();
        int bc = 0;
        for (int c = 0; c < C; ++c) {                               // LDS broadcast reads: every lane the same centre
            float dot = 0.f;
#pragma unroll
            for (int j = 0; j < (D > 0 ? D : KM_MAX_D); ++j) if (j < d) dot = fmaf(x[j], cen[c * d + j], dot);
            const float v = fmaf(-2.0f, dot, cn[c]);
            if (v < best) { best = v; bc = c; }
        }
        if (n_changed && labels[i] != bc) ++changed;
        labels[i] = bc;
        if (inertia) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < (D > 0 ? D : KM_MAX_D); ++j) if (j < d) { const float e = x[j] - cen[bc * d + j]; s = fmaf(e, e, s); }
            inert += (double)s;
        }
        if (sums) {
#pragma unroll
            for (int j = 0; j < (D > 0 ? D : KM_MAX_D); ++j) {
                if (j < d) {
                    const unsigned long long q = (unsigned long long)__double2ll_rn((double)x[j] * fx_scale);
                    if (lds_acc) lds_add_u64(&acc[bc * d + j], q); else atomicAdd(&sums[(size_t)bc * d + j], q);
                }
            }
            if (lds_acc) atomicAdd(&cnt[bc], 1u); else atomicAdd(&counts[bc], 1u);
        }
    }
    if (n_changed) {
        for (int o = 32; o > 0; o >>= 1) changed += __shfl_xor(changed, o, 64);
        if ((tid & 63) == 0 && changed) atomicAdd(n_changed, changed);
    }
    if (inertia) {
        for (int o = 32; o > 0; o >>= 1) inert += __shfl_xor(inert, o, 64);
        if ((tid & 63) == 0) atomicAdd(inertia, inert);            // f64: order-dependent only below 1e-16 relative
    }
    if (sums && lds_acc) {
        __syncthreads();
        for (int i = tid; i < C * d; i += KM_THREADS) { const unsigned long long v = acc[i]; if (v) atomicAdd(&sums[i], v); }
        for (int i = tid; i < C; i += KM_THREADS) { const unsigned int v = cnt[i]; if (v) atomicAdd(&counts[i], v); }
    }
}

// M-step: centre = sum / count (empty clusters keep their centre; the caller relocates them as sklearn does), shift2 += |new - old|^2
__global__ __launch_bounds__(256) void kmeans_update_kernel(const unsigned long long *__restrict__ sums, const unsigned int *__restrict__ counts, int C, int d,
                                                            double inv_scale, const float *__restrict__ old_c, float *__restrict__ new_c, double *__restrict__ shift2) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    double s2 = 0.0;
    if (i < C * d) {
        const unsigned int n = counts[i / d];
        float v = old_c[i];
        if (n > 0) v = (float)(((double)(long long)sums[i] * inv_scale) / (double)n);
        new_c[i] = v;
        const double e = (double)v - (double)old_c[i];
        s2 = e * e;
    }
    for (int o = 32; o > 0; o >>= 1) s2 += __shfl_xor(s2, o, 64);
    if ((threadIdx.x & 63) == 0 && s2 != 0.0) atomicAdd(shift2, s2);
}

// k-means++ seeding (sklearn's _kmeans_plusplus, greedy variant): for each of T candidate points the potential
// pot[t] = sum_i min(closest[i], |x_i - x_cand_t|^2); the caller picks the smallest and applies it.
constexpr int KPP_MAX_T = 16;
template <bool APPLY>
__global__ __launch_bounds__(256) void kmeanspp_kernel(const float *__restrict__ X, int64_t N, int d, const int64_t *__restrict__ cand, int T, float *__restrict__ closest,
                                                       double *__restrict__ pots) {
    __shared__ float cx[KPP_MAX_T * KM_MAX_D];
    __shared__ double red[4][KPP_MAX_T];
    const int tid = threadIdx.x;
    for (int i = tid; i < T * d; i += 256) cx[i] = X[cand[i / d] * d + (i % d)];
    __syncthreads();
    double acc[KPP_MAX_T];
#pragma unroll
    for (int t = 0; t < KPP_MAX_T; ++t) acc[t] = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < N; i += (int64_t)gridDim.x * 256) {
        float x[KM_MAX_D];
#pragma unroll
        for (int j = 0; j < KM_MAX_D; ++j) x[j] = j < d ? X[i * d + j] : 0.f;
        const float cl = closest[i];
#pragma unroll
        for (int t = 0; t < KPP_MAX_T; ++t) {
            if (t < T) {
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < KM_MAX_D; ++j) if (j < d) { const float e = x[j] - cx[t * d + j]; s = fmaf(e, e, s); }
                const float m = fminf(cl, s);
                if (APPLY) closest[i] = m; else acc[t] += (double)m;
            }
        }
    }
    if (!APPLY) {
#pragma unroll
        for (int t = 0; t < KPP_MAX_T; ++t) {
            if (t < T) {
                double v = acc[t];
                for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
                if ((tid & 63) == 0) red[tid >> 6][t] = v;
            }
        }
        __syncthreads();
        if (tid < T) atomicAdd(&pots[tid], red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid]);
    }
}

}  // namespace

extern "C" int w3d_kmeanspp_pots(const float *X, int64_t N, int d, const int64_t *cand, int T, const float *closest, double *pots, w3d_stream_t stream) {
    if (!X || !cand || !closest || !pots || N < 1 || d < 1 || d > KM_MAX_D || T < 1 || T > KPP_MAX_T) return W3D_ERR_ARG;
    W3D_HIP(hipMemsetAsync(pots, 0, (size_t)T * sizeof(double), w3d_s(stream)));
    hipLaunchKernelGGL(kmeanspp_kernel<false>, dim3(w3d_grid(N, 256, 1024)), dim3(256), 0, w3d_s(stream), X, N, d, cand, T, const_cast<float *>(closest), pots);
    W3D_CHECK_LAUNCH();
    return W3D_OK;
}

extern "C" int w3d_kmeanspp_apply(const float *X, int64_t N, int d, const int64_t *centre_id, float *closest, w3d_stream_t stream) {
    if (!X || !centre_id || !closest || N < 1 || d < 1 || d > KM_MAX_D) return W3D_ERR_ARG;
    hipLaunchKernelGGL(kmeanspp_kernel<true>, dim3(w3d_grid(N, 256, 1024)), dim3(256), 0, w3d_s(stream), X, N, d, centre_id, 1, closest, (double *)nullptr);
    W3D_CHECK_LAUNCH();
    return W3D_OK;
}

extern "C" int w3d_kmeans_assign(const float *X, int64_t N, int d, const float *centres, int C, double fx_scale, int32_t *labels, uint64_t *sums, uint32_t *counts,
                                 uint32_t *n_changed, double *inertia, w3d_stream_t stream) {
    if (!X || !centres || !labels || N < 0 || d < 1 || d > KM_MAX_D || C < 1) return W3D_ERR_ARG;
    if ((sums == nullptr) != (counts == nullptr) || (sums && !(fx_scale > 0.0))) return W3D_ERR_ARG;
    if (N == 0) return W3D_OK;
    const size_t cen_bytes = (((size_t)C * (d + 1) * 4 + 7) & ~(size_t)7);
    if (cen_bytes > 128 * 1024) return W3D_ERR_UNSUPPORTED;                       // centres must fit the LDS
    const size_t acc_bytes = (size_t)C * d * 8 + (size_t)C * 4;
    const int lds_acc = sums && cen_bytes + acc_bytes <= 150 * 1024;
    const size_t lds = cen_bytes + (lds_acc ? acc_bytes : 0);
    const int grid = w3d_grid(N, KM_THREADS, 1024);
    hipStream_t st = w3d_s(stream);
    auto go = [&](auto kern) -> int {
        W3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(KM_THREADS), lds, st, X, N, d, centres, C, labels, reinterpret_cast<unsigned long long *>(sums), counts, fx_scale, lds_acc,
                           n_changed, inertia);
        return w3d_launch_status();
    };
    if (d == 3) return go(&kmeans_assign_kernel<3>);
    if (d == 2) return go(&kmeans_assign_kernel<2>);
    if (d == 4) return go(&kmeans_assign_kernel<4>);
    return go(&kmeans_assign_kernel<0>);
}

extern "C" int w3d_kmeans_update(const uint64_t *sums, const uint32_t *counts, int C, int d, double fx_scale, const float *old_centres, float *new_centres, double *shift2,
                                 w3d_stream_t stream) {
    if (!sums || !counts || !old_centres || !new_centres || !shift2 || C < 1 || d < 1 || !(fx_scale > 0.0)) return W3D_ERR_ARG;
    hipLaunchKernelGGL(kmeans_update_kernel, dim3((C * d + 255) / 256), dim3(256), 0, w3d_s(stream), reinterpret_cast<const unsigned long long *>(sums), counts, C, d,
                       1.0 / fx_scale, old_centres, new_centres, shift2);
    W3D_CHECK_LAUNCH();
    return W3D_OK;
}
