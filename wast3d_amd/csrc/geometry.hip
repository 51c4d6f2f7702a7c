// Geometry-loop kernels: descriptors (a3), their MSE head (a4), the reference-regime domain loss
// (a1+a2+a4 fused, exact tie semantics), generic low-d fp32 kNN, kNN loss/backward, CSR transpose.
// All HBM-/L2-bound gather-scatter work: coalesced index reads, wave reductions before any atomic.
#include "common.h"
#include <algorithm>
#include <rocprim/device/device_radix_sort.hpp>

// point fetch: XS = 3 is the reference's packed [n,3] layout (three dword loads, may straddle lines); XS = 4 is a padded float4
// copy made once per iteration by w3d_pack_xyz4 (one 16-B load per gather, never straddles) — used by the hot loops
struct P3 { float x, y, z; };
template <int XS>
__device__ __forceinline__ P3 ldp(const float *__restrict__ X, int64_t i) {
    if (XS == 4) { const float4 v = *reinterpret_cast<const float4 *>(X + 4 * i); return {v.x, v.y, v.z}; }
    return {X[3 * i], X[3 * i + 1], X[3 * i + 2]};
}
__global__ void pack_xyz4_kernel(const float *__restrict__ X, int64_t n, float4 *__restrict__ X4) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        X4[i] = make_float4(X[3 * i], X[3 * i + 1], X[3 * i + 2], 0.f);
}

// ------------------------------------------------------------------------------------------------
// a3  descriptors.  One wave per descriptor row: lane j handles neighbours j, j+64, ...
//     idx row read is coalesced (int32), X gathers hit L2/MALL (X is n*12 B), F row write coalesced.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint16_t f32_to_f16(float f) { const _Float16 h = (_Float16)f; uint16_t b; __builtin_memcpy(&b, &h, 2); return b; }
__device__ __forceinline__ float f16_to_f32(uint16_t b) { _Float16 h; __builtin_memcpy(&h, &b, 2); return (float)h; }

// OUT: 0 f32, 1 bf16, 2 f16
template <int OUT, int XS>
__global__ __launch_bounds__(256) void descriptor_fwd_kernel(const float *__restrict__ X, const int32_t *__restrict__ idx,
                                                             int64_t R, int K, void *__restrict__ Fout, float *__restrict__ motion2, float *__restrict__ xnorm = nullptr,
                                                             const float *__restrict__ center = nullptr) {
    // center [K-1] (optional): subtracted from every row BEFORE it is rounded to 16 bits.  Euclidean distances between rows do not change when
    // all rows are shifted by one vector, their 16-bit rounding error does: rows centred on the style set's mean carry ~10x less of it.
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t r = wave0; r < R; r += nwaves) {
        const int32_t *row = idx + r * K;
        const int32_t root = row[0];
        const P3 rp = ldp<XS>(X, root);
        const float rx = rp.x, ry = rp.y, rz = rp.z;
        float m2 = 0.f, n2 = 0.f;
        for (int j = lane; j < K - 1; j += 64) {
            const int64_t n = row[j + 1];
            const P3 q = ldp<XS>(X, n);
            const float vx = q.x - rx, vy = q.y - ry, vz = q.z - rz;
            float dist = sqrtf(vx * vx + vy * vy + vz * vz);
            if (center) dist -= center[j];
            if (OUT != 0) {
                uint16_t *f = reinterpret_cast<uint16_t *>(Fout) + r * (K - 1) + j;
                const uint16_t nb = OUT == 1 ? f32_to_bf16(dist) : f32_to_f16(dist);
                if (motion2) {                                   // |F_new - F_old|^2 of the row, before overwriting
                    const float fn = OUT == 1 ? bf16_to_f32(nb) : f16_to_f32(nb);
                    const float dd = fn - (OUT == 1 ? bf16_to_f32(*f) : f16_to_f32(*f));
                    m2 += dd * dd;
                    n2 = fmaf(fn, fn, n2);                       // |F_new|^2 of the rounded row (the kNN's |x|^2)
                }
                *f = nb;
            } else {
                reinterpret_cast<float *>(Fout)[r * (K - 1) + j] = dist;
            }
        }
        if (OUT != 0 && motion2) {
            m2 = wave_sum(m2);
            if (xnorm) n2 = wave_sum(n2);
            if (lane == 0) { motion2[r] = m2; if (xnorm) xnorm[r] = n2; }
        }
    }
}

// Backward with float atomics (index_put accumulate=True).  Root gets one atomic per wave per row.
// MSE = true: gF is replaced by 2*w/cnt*(dist - target) and the loss is reduced (fused head, F never stored).
template <bool MSE>
__global__ __launch_bounds__(256) void descriptor_bwd_kernel(const float *__restrict__ X, const int32_t *__restrict__ idx,
                                                             int64_t R, int K, const float *__restrict__ gF_or_target,
                                                             int64_t seg_rows, float weight, float *__restrict__ loss_accum,
                                                             float *__restrict__ gX) {
    __shared__ float red[16];
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int64_t rows_in_seg = seg_rows > 0 ? seg_rows : R;
    const float gscale = MSE ? 2.0f * weight / ((float)rows_in_seg * (float)(K - 1)) : 1.0f;
    const float lscale = MSE ? weight / ((float)rows_in_seg * (float)(K - 1)) : 0.f;
    // the wave keeps its loss partial in a register and flushes it only when its segment changes; rows are taken in contiguous chunks so
    // that this happens at most a few times per wave (same-address float atomics serialise at L2)
    float lacc = 0.f;
    int64_t lseg = -1;
    const int64_t chunk = (R + nwaves - 1) / nwaves;
    const int64_t r_begin = wave0 * chunk, r_end = min(R, r_begin + chunk);
    for (int64_t r = r_begin; r < r_end; ++r) {
        const int32_t *row = idx + r * K;
        const int64_t root = row[0];
        const float rx = X[3 * root], ry = X[3 * root + 1], rz = X[3 * root + 2];
        float sx = 0.f, sy = 0.f, sz = 0.f, le = 0.f;
        for (int j = lane; j < K - 1; j += 64) {
            const int64_t n = row[j + 1];
            const float vx = X[3 * n] - rx, vy = X[3 * n + 1] - ry, vz = X[3 * n + 2] - rz;
            const float dist = sqrtf(vx * vx + vy * vy + vz * vz);
            float g;
            if (MSE) {
                const float e = dist - gF_or_target[r * (K - 1) + j];
                le += e * e;
                g = gscale * e;
            } else {
                g = gF_or_target[r * (K - 1) + j];
            }
            const float w = dist > 0.f ? g / dist : 0.f;   // torch.norm backward: 0 at 0
            const float gx = w * vx, gy = w * vy, gz = w * vz;
            atomicAdd(&gX[3 * n], gx);
            atomicAdd(&gX[3 * n + 1], gy);
            atomicAdd(&gX[3 * n + 2], gz);
            sx += gx; sy += gy; sz += gz;
        }
        sx = wave_sum(sx); sy = wave_sum(sy); sz = wave_sum(sz);
        if (MSE) le = wave_sum(le);
        if (lane == 0) {
            atomicAdd(&gX[3 * root], -sx);
            atomicAdd(&gX[3 * root + 1], -sy);
            atomicAdd(&gX[3 * root + 2], -sz);
        }
        if (MSE) {
            const int64_t sg = seg_rows > 0 ? r / seg_rows : 0;
            if (sg != lseg) {
                if (lane == 0 && lseg >= 0) atomicAdd(&loss_accum[lseg], lscale * lacc);
                lacc = 0.f; lseg = sg;
            }
            lacc += le;
        }
    }
    if (MSE && lane == 0 && lseg >= 0) atomicAdd(&loss_accum[lseg], lscale * lacc);
    (void)red;
}

// ------------------------------------------------------------------------------------------------
// CSR transpose of a frozen neighbour list: entries sorted by (destination point, flat position).
// Stable radix sort keeps flat positions ascending inside each destination -> fixed summation order,
// so the CSR backward is bitwise reproducible (the atomic form is not).
// ------------------------------------------------------------------------------------------------
__global__ void iota_kernel(int32_t *v, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) v[i] = (int32_t)i;
}
__global__ void csr_rowptr_kernel(const int32_t *__restrict__ sorted_keys, int64_t nnz, int64_t n_points, int32_t *__restrict__ row_ptr) {
    // row_ptr[p] = first position whose key >= p
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= nnz; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t cur = i < nnz ? sorted_keys[i] : n_points;
        const int64_t prev = i > 0 ? sorted_keys[i - 1] : -1;
        for (int64_t p = prev + 1; p <= cur; ++p) row_ptr[p] = (int32_t)i;
    }
}

// entries2[e] = (root point of the row, flat index into gF) for neighbour roles; (-1 - r, 0) for the root role of row r
__global__ void csr_expand_kernel(const int32_t *__restrict__ pos_sorted, const int32_t *__restrict__ idx, int64_t nnz, int K, int2 *__restrict__ entries2) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += (int64_t)gridDim.x * blockDim.x) {
        const int32_t pos = pos_sorted[e];
        const int64_t r = pos / K;
        const int c = pos - (int)(r * K);
        entries2[e] = c == 0 ? make_int2(-1 - (int)r, 0) : make_int2(idx[r * K], (int)(r * (K - 1) + c - 1));
    }
}

extern "C" size_t w3d_csr_workspace_bytes(int64_t n_points, int64_t R, int K) {
    const int64_t nnz = R * K;
    size_t tmp = 0;
    rocprim::radix_sort_pairs(nullptr, tmp, (const int32_t *)nullptr, (int32_t *)nullptr, (const int32_t *)nullptr, (int32_t *)nullptr, (size_t)nnz, 0u, 32u, (hipStream_t)0);
    return (size_t)nnz * 4 * 3 + tmp + 512;   // sorted keys + iota values + sorted positions + rocprim temp
    (void)n_points;
}

extern "C" int w3d_csr_build(const int32_t *idx, int64_t R, int K, int64_t n_points, int32_t *row_ptr, int32_t *entries,
                             void *workspace, size_t workspace_bytes, w3d_stream_t stream) {
    if (!idx || !row_ptr || !entries || !workspace || R <= 0 || K <= 0) return W3D_ERR_ARG;
    const int64_t nnz = R * K;
    if (nnz >= (1ll << 31) || R * (int64_t)(K - 1) >= (1ll << 31)) return W3D_ERR_UNSUPPORTED;
    if (workspace_bytes < w3d_csr_workspace_bytes(n_points, R, K)) return W3D_ERR_WORKSPACE;
    int32_t *keys_sorted = reinterpret_cast<int32_t *>(workspace);
    int32_t *iota = keys_sorted + nnz;
    int32_t *pos_sorted = iota + nnz;
    void *tmp = reinterpret_cast<char *>(workspace) + (size_t)nnz * 12;
    tmp = reinterpret_cast<void *>((reinterpret_cast<uintptr_t>(tmp) + 255) & ~(uintptr_t)255);
    size_t tmp_bytes = 0;
    rocprim::radix_sort_pairs(nullptr, tmp_bytes, (const int32_t *)nullptr, (int32_t *)nullptr, (const int32_t *)nullptr, (int32_t *)nullptr, (size_t)nnz, 0u, 32u, (hipStream_t)0);
    hipLaunchKernelGGL(iota_kernel, dim3(w3d_grid(nnz, 256)), dim3(256), 0, w3d_s(stream), iota, nnz);
    int bits = 1;
    while ((1ll << bits) < n_points && bits < 32) ++bits;
    if (rocprim::radix_sort_pairs(tmp, tmp_bytes, idx, keys_sorted, (const int32_t *)iota, pos_sorted, (size_t)nnz, 0u, (unsigned)bits, w3d_s(stream)) != hipSuccess)
        return W3D_ERR_LAUNCH;
    hipLaunchKernelGGL(csr_rowptr_kernel, dim3(w3d_grid(nnz + 1, 256)), dim3(256), 0, w3d_s(stream), keys_sorted, nnz, n_points, row_ptr);
    hipLaunchKernelGGL(csr_expand_kernel, dim3(w3d_grid(nnz, 256)), dim3(256), 0, w3d_s(stream), pos_sorted, idx, nnz, K, reinterpret_cast<int2 *>(entries));
    W3D_CHECK_LAUNCH();
    return W3D_OK;
}

// rootsum[r] = -sum_j gvec(r,j): one wave per row (coalesced), feeds the per-destination gather below
template <int XS>
__global__ __launch_bounds__(256) void descriptor_rootsum_kernel(const float *__restrict__ X, const int32_t *__restrict__ idx, int64_t R, int K,
                                                                 const float *__restrict__ gF, float *__restrict__ rootsum /*[R,3]*/) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t r = wave0; r < R; r += nwaves) {
        const int32_t *row = idx + r * K;
        const int64_t root = row[0];
        const P3 rp = ldp<XS>(X, root);
        const float rx = rp.x, ry = rp.y, rz = rp.z;
        float sx = 0.f, sy = 0.f, sz = 0.f;
        for (int j = lane; j < K - 1; j += 64) {
            const int64_t n = row[j + 1];
            const P3 q = ldp<XS>(X, n);
            const float vx = q.x - rx, vy = q.y - ry, vz = q.z - rz;
            const float dist = sqrtf(vx * vx + vy * vy + vz * vz);
            const float w = dist > 0.f ? gF[r * (K - 1) + j] / dist : 0.f;
            sx += w * vx; sy += w * vy; sz += w * vz;
        }
        sx = wave_sum(sx); sy = wave_sum(sy); sz = wave_sum(sz);
        if (lane == 0) { rootsum[3 * r] = -sx; rootsum[3 * r + 1] = -sy; rootsum[3 * r + 2] = -sz; }
    }
}

// one wave per destination point: gathers its contributions in fixed order (entries are (root, gF slot) pairs)
template <int XS>
__global__ __launch_bounds__(256) void descriptor_bwd_csr_kernel(const float *__restrict__ X, const float *__restrict__ gF, const float *__restrict__ rootsum,
                                                                 const int32_t *__restrict__ row_ptr, const int2 *__restrict__ entries2,
                                                                 int64_t n_points, int accumulate, float *__restrict__ gX) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t p = wave0; p < n_points; p += nwaves) {
        const int beg = row_ptr[p], end = row_ptr[p + 1];
        const P3 pp = ldp<XS>(X, p);
        const float px = pp.x, py = pp.y, pz = pp.z;
        float sx = 0.f, sy = 0.f, sz = 0.f;
        for (int e = beg + lane; e < end; e += 64) {
            const int2 en = entries2[e];
            if (en.x < 0) {
                const int64_t r = -1 - en.x;
                sx += rootsum[3 * r]; sy += rootsum[3 * r + 1]; sz += rootsum[3 * r + 2];
            } else {
                const int64_t root = en.x;
                const P3 q = ldp<XS>(X, root);
                const float vx = px - q.x, vy = py - q.y, vz = pz - q.z;
                const float dist = sqrtf(vx * vx + vy * vy + vz * vz);
                const float w = dist > 0.f ? gF[en.y] / dist : 0.f;
                sx += w * vx; sy += w * vy; sz += w * vz;
            }
        }
        sx = wave_sum(sx); sy = wave_sum(sy); sz = wave_sum(sz);
        if (lane == 0) {
            if (accumulate) { gX[3 * p] += sx; gX[3 * p + 1] += sy; gX[3 * p + 2] += sz; }
            else { gX[3 * p] = sx; gX[3 * p + 1] = sy; gX[3 * p + 2] = sz; }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Batched circles sharing ONE neighbour list / target / CSR (optimize_all_by_clusters fits the SAME style patch to every
// circle, NB11:987-993): segment s works on points X[s*n .. (s+1)*n) with local indices.  The shared list, target and CSR
// (a few MB) stay L2-resident across segments; no atomics on the gradient.
//   pass 1 (row order):   rootsum[s,r] = -sum_j g(r,j),  loss_s += sum_j e^2
//   pass 2 (destination): gX[s,p] += sum over the CSR entries of p
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void desc_mse_seg_rows_kernel(const float *__restrict__ X, const int32_t *__restrict__ idx, const float *__restrict__ target,
                                                                int64_t R, int K, int64_t n, int S, float gscale, float lscale, float *__restrict__ loss_accum,
                                                                float *__restrict__ rootsum) {
    const int lane = threadIdx.x & 63;
    const int64_t wave_id = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    // every wave takes a CONTIGUOUS chunk of (segment, row) pairs: its loss partial changes segment at most a few times, so loss_accum[s]
    // sees one float atomic per wave and segment — strided rows made every row end in an atomic on one of S addresses (same-address
    // atomics serialise at L2: 266 560 of them on 64 addresses were 1.4 of this kernel's 1.6 ms at the notebook's K = 100 lists)
    const int64_t total = R * S, chunk = (total + nwaves - 1) / nwaves;
    const int64_t w_begin = wave_id * chunk, w_end = min(total, w_begin + chunk);
    float lacc = 0.f; int64_t lseg = -1;
    for (int64_t w = w_begin; w < w_end; ++w) {
        const int64_t sg = w / R, r = w - sg * R;
        const float *P = X + sg * n * 3;
        const int32_t *row = idx + r * K;
        const int64_t root = row[0];
        const float rx = P[3 * root], ry = P[3 * root + 1], rz = P[3 * root + 2];
        float sx = 0.f, sy = 0.f, sz = 0.f, le = 0.f;
        for (int j = lane; j < K - 1; j += 64) {
            const int64_t q = row[j + 1];
            const float vx = P[3 * q] - rx, vy = P[3 * q + 1] - ry, vz = P[3 * q + 2] - rz;
            const float dist = sqrtf(vx * vx + vy * vy + vz * vz);
            const float e = dist - target[r * (K - 1) + j];
            le += e * e;
            const float wgt = dist > 0.f ? gscale * e / dist : 0.f;
            sx += wgt * vx; sy += wgt * vy; sz += wgt * vz;
        }
        sx = wave_sum(sx); sy = wave_sum(sy); sz = wave_sum(sz); le = wave_sum(le);
        if (lane == 0) { float *o = rootsum + (sg * R + r) * 3; o[0] = -sx; o[1] = -sy; o[2] = -sz; }
        if (sg != lseg) {
            if (lane == 0 && lseg >= 0) atomicAdd(&loss_accum[lseg], lscale * lacc);
            lacc = 0.f; lseg = sg;
        }
        lacc += le;
    }
    if (lane == 0 && lseg >= 0) atomicAdd(&loss_accum[lseg], lscale * lacc);
}

__global__ __launch_bounds__(256) void desc_mse_seg_dest_kernel(const float *__restrict__ X, const float *__restrict__ target, const float *__restrict__ rootsum,
                                                                const int32_t *__restrict__ row_ptr, const int2 *__restrict__ entries2, int64_t R, int64_t n, int S,
                                                                float gscale, float *__restrict__ gX) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t w = wave0; w < n * S; w += nwaves) {
        const int64_t sg = w / n, p = w - sg * n;
        const float *P = X + sg * n * 3;
        const int beg = row_ptr[p], end = row_ptr[p + 1];
        if (beg == end) continue;
        const float px = P[3 * p], py = P[3 * p + 1], pz = P[3 * p + 2];
        float sx = 0.f, sy = 0.f, sz = 0.f;
        for (int e = beg + lane; e < end; e += 64) {
            const int2 en = entries2[e];
            if (en.x < 0) {
                const float *o = rootsum + (sg * R + (-1 - en.x)) * 3;
                sx += o[0]; sy += o[1]; sz += o[2];
            } else {
                const float vx = px - P[3 * (int64_t)en.x], vy = py - P[3 * (int64_t)en.x + 1], vz = pz - P[3 * (int64_t)en.x + 2];
                const float dist = sqrtf(vx * vx + vy * vy + vz * vz);
                const float wgt = dist > 0.f ? gscale * (dist - target[en.y]) / dist : 0.f;
                sx += wgt * vx; sy += wgt * vy; sz += wgt * vz;
            }
        }
        sx = wave_sum(sx); sy = wave_sum(sy); sz = wave_sum(sz);
        if (lane == 0) { float *o = gX + (sg * n + p) * 3; o[0] += sx; o[1] += sy; o[2] += sz; }
    }
}

extern "C" int w3d_descriptor_mse_seg_csr(const float *X, const int32_t *idx, const float *target, int64_t R, int K, int64_t n_points_per_seg, int S,
                                          const int32_t *row_ptr, const int32_t *entries, float weight, float *loss_accum, float *gX, float *rootsum_scratch,
                                          w3d_stream_t stream) {
    if (!X || !idx || !target || !row_ptr || !entries || !loss_accum || !gX || !rootsum_scratch || R <= 0 || K < 2 || S <= 0 || n_points_per_seg <= 0)
        return W3D_ERR_ARG;
    const float cnt = (float)R * (float)(K - 1);
    const float gscale = 2.0f * weight / cnt, lscale = weight / cnt;
    // ~16 rows per wave where the problem allows it (few loss atomics), but never fewer than min(rows/4, 2048) blocks (fill the chip)
    const int rows_grid = std::max(w3d_grid(R * S, 4, 2048), w3d_grid(R * S, 64, 16384));
    hipLaunchKernelGGL(desc_mse_seg_rows_kernel, dim3(rows_grid), dim3(256), 0, w3d_s(stream), X, idx, target, R, K, n_points_per_seg, S, gscale,
                       lscale, loss_accum, rootsum_scratch);
    hipLaunchKernelGGL(desc_mse_seg_dest_kernel, dim3(w3d_grid(n_points_per_seg * S, 4, 16384)), dim3(256), 0, w3d_s(stream), X, target, rootsum_scratch, row_ptr,
                       reinterpret_cast<const int2 *>(entries), R, n_points_per_seg, S, gscale, gX);
    W3D_CHECK_LAUNCH();
    return W3D_OK;
}

static int descriptor_fwd_impl(const float *X, int xs, const int32_t *idx, int64_t R, int K, int out_dtype, void *F, const float *center, w3d_stream_t stream) {
    if (!X || !idx || !F || R < 0 || K < 2 || (xs != 3 && xs != 4)) return W3D_ERR_ARG;
    if (xs == 4 && ((uintptr_t)X & 15)) return W3D_ERR_ARG;
    if (R == 0) return W3D_OK;
    const int grid = w3d_grid(R, 4, 8192);
    hipStream_t st = w3d_s(stream);
    if (out_dtype == W3D_DTYPE_F32) {
        if (xs == 3) hipLaunchKernelGGL((descriptor_fwd_kernel<0, 3>), dim3(grid), dim3(256), 0, st, X, idx, R, K, F, nullptr, nullptr, center);
        else hipLaunchKernelGGL((descriptor_fwd_kernel<0, 4>), dim3(grid), dim3(256), 0, st, X, idx, R, K, F, nullptr, nullptr, center);
    } else if (out_dtype == W3D_DTYPE_BF16) {
        if (xs == 3) hipLaunchKernelGGL((descriptor_fwd_kernel<1, 3>), dim3(grid), dim3(256), 0, st, X, idx, R, K, F, nullptr, nullptr, center);
        else hipLaunchKernelGGL((descriptor_fwd_kernel<1, 4>), dim3(grid), dim3(256), 0, st, X, idx, R, K, F, nullptr, nullptr, center);
    } else if (out_dtype == W3D_DTYPE_F16) {
        if (xs == 3) hipLaunchKernelGGL((descriptor_fwd_kernel<2, 3>), dim3(grid), dim3(256), 0, st, X, idx, R, K, F, nullptr, nullptr, center);
        else hipLaunchKernelGGL((descriptor_fwd_kernel<2, 4>), dim3(grid), dim3(256), 0, st, X, idx, R, K, F, nullptr, nullptr, center);
    } else {
        return W3D_ERR_ARG;
    }
    W3D_CHECK_LAUNCH();
    return W3D_OK;
}
extern "C" int w3d_descriptor_fwd(const float *X, const int32_t *idx, int64_t R, int K, int out_dtype, void *F, w3d_stream_t stream) {
    return descriptor_fwd_impl(X, 3, idx, R, K, out_dtype, F, nullptr, stream);
}
extern "C" int w3d_descriptor_fwd_x4(const float *X4, const int32_t *idx, int64_t R, int K, int out_dtype, void *F, const float *center, w3d_stream_t stream) {
    return descriptor_fwd_impl(X4, 4, idx, R, K, out_dtype, F, center, stream);
}
extern "C" int w3d_descriptor_fwd_x4_motion(const float *X4, const int32_t *idx, int64_t R, int K, int dtype, void *F_inout, float *motion2, float *xnorm,
                                            const float *center, w3d_stream_t stream) {
    if (!X4 || !idx || !F_inout || !motion2 || R <= 0 || K < 2 || ((uintptr_t)X4 & 15)) return W3D_ERR_ARG;
    if (dtype == W3D_DTYPE_BF16) hipLaunchKernelGGL((descriptor_fwd_kernel<1, 4>), dim3(w3d_grid(R, 4, 8192)), dim3(256), 0, w3d_s(stream), X4, idx, R, K, F_inout, motion2, xnorm, center);
    else if (dtype == W3D_DTYPE_F16) hipLaunchKernelGGL((descriptor_fwd_kernel<2, 4>), dim3(w3d_grid(R, 4, 8192)), dim3(256), 0, w3d_s(stream), X4, idx, R, K, F_inout, motion2, xnorm, center);
    else return W3D_ERR_ARG;
    W3D_CHECK_LAUNCH();
    return W3D_OK;
}
extern "C" int w3d_pack_xyz4(const float *X, int64_t n, float *X4, w3d_stream_t stream) {
    if (!X || !X4 || n < 0 || ((uintptr_t)X4 & 15)) return W3D_ERR_ARG;
    if (n == 0) return W3D_OK;
    hipLaunchKernelGGL(pack_xyz4_kernel, dim3(w3d_grid(n, 256)), dim3(256), 0, w3d_s(stream), X, n, reinterpret_cast<float4 *>(X4));
    W3D_CHECK_LAUNCH();
    return W3D_OK;
}

extern "C" int w3d_descriptor_bwd(const float *X, const int32_t *idx, int64_t R, int K, const float *gF, float *gX, w3d_stream_t stream) {
    if (!X || !idx || !gF || !gX || R < 0 || K < 2) return W3D_ERR_ARG;
    if (R == 0) return W3D_OK;
    hipLaunchKernelGGL(descriptor_bwd_kernel<false>, dim3(w3d_grid(R, 4, 8192)), dim3(256), 0, w3d_s(stream), X, idx, R, K, gF, (int64_t)0, 1.0f,
                       (float *)nullptr, gX);
    W3D_CHECK_LAUNCH();
    return W3D_OK;
}

// get_descriptors_vec (visualize_and_fit_cluster.ipynb cell 26): a 4-vector V[r] per list row (a rotation quaternion in the notebook); for the
// four index triples S = {012, 013, 023, 123}:  F[r, s*(K-1) + j] = | V[r,S_s] - (X[idx[r,j+1]] - X[idx[r,0]]) |.  One wave per row.
__constant__ int VEC_SUBSETS[4][3] = {{0, 1, 2}, {0, 1, 3}, {0, 2, 3}, {1, 2, 3}};

template <bool BWD>
__global__ __launch_bounds__(256) void descriptor_vec_kernel(const float *__restrict__ X, const float *__restrict__ V, const int32_t *__restrict__ idx, int64_t R, int K,
                                                             float *__restrict__ F, const float *__restrict__ gF, float *__restrict__ gX, float *__restrict__ gV) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int d = K - 1;
    for (int64_t r = wave0; r < R; r += nwaves) {
        const int32_t *row = idx + r * K;
        const int64_t root = row[0];
        const float rx = X[3 * root], ry = X[3 * root + 1], rz = X[3 * root + 2];
        const float v[4] = {V[4 * r], V[4 * r + 1], V[4 * r + 2], V[4 * r + 3]};
        float sx = 0.f, sy = 0.f, sz = 0.f, gv[4] = {0.f, 0.f, 0.f, 0.f};
        for (int j = lane; j < d; j += 64) {
            const int64_t n = row[j + 1];
            const float o[3] = {X[3 * n] - rx, X[3 * n + 1] - ry, X[3 * n + 2] - rz};
            float ox = 0.f, oy = 0.f, oz = 0.f;                              // dL/d offset of this neighbour
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float e0 = v[VEC_SUBSETS[s][0]] - o[0], e1 = v[VEC_SUBSETS[s][1]] - o[1], e2 = v[VEC_SUBSETS[s][2]] - o[2];
                const float nrm = sqrtf(e0 * e0 + e1 * e1 + e2 * e2);
                if (!BWD) {
                    F[r * (4 * (int64_t)d) + (int64_t)s * d + j] = nrm;
                } else {
                    const float g = gF[r * (4 * (int64_t)d) + (int64_t)s * d + j];
                    const float w = nrm > 0.f ? g / nrm : 0.f;                  // torch.norm backward: 0 at 0
                    gv[VEC_SUBSETS[s][0]] += w * e0; gv[VEC_SUBSETS[s][1]] += w * e1; gv[VEC_SUBSETS[s][2]] += w * e2;
                    ox -= w * e0; oy -= w * e1; oz -= w * e2;
                }
            }
            if (BWD) {
                atomicAdd(&gX[3 * n], ox); atomicAdd(&gX[3 * n + 1], oy); atomicAdd(&gX[3 * n + 2], oz);
                sx += ox; sy += oy; sz += oz;
            }
        }
        if (BWD) {
            sx = wave_sum(sx); sy = wave_sum(sy); sz = wave_sum(sz);
#pragma unroll
            for (int c = 0; c < 4; ++c) gv[c] = wave_sum(gv[c]);
            if (lane == 0) {
                atomicAdd(&gX[3 * root], -sx); atomicAdd(&gX[3 * root + 1], -sy); atomicAdd(&gX[3 * root + 2], -sz);
                gV[4 * r] += gv[0]; gV[4 * r + 1] += gv[1]; gV[4 * r + 2] += gv[2]; gV[4 * r + 3] += gv[3];     // one wave owns the row
            }
        }
    }
}

extern "C" int w3d_descriptor_vec_fwd(const float *X, const float *V, const int32_t *idx, int64_t R, int K, float *F, w3d_stream_t stream) {
    if (!X || !V || !idx || !F || R < 0 || K < 2) return W3D_ERR_ARG;
    if (R == 0) return W3D_OK;
    hipLaunchKernelGGL(descriptor_vec_kernel<false>, dim3(w3d_grid(R, 4, 8192)), dim3(256), 0, w3d_s(stream), X, V, idx, R, K, F, (const float *)nullptr,
                       (float *)nullptr, (float *)nullptr);
    W3D_CHECK_LAUNCH();
    return W3D_OK;
}

extern "C" int w3d_descriptor_vec_bwd(const float *X, const float *V, const int32_t *idx, int64_t R, int K, const float *gF, float *gX, float *gV, w3d_stream_t stream) {
    if (!X || !V || !idx || !gF || !gX || !gV || R < 0 || K < 2) return W3D_ERR_ARG;
    if (R == 0) return W3D_OK;
    hipLaunchKernelGGL(descriptor_vec_kernel<true>, dim3(w3d_grid(R, 4, 8192)), dim3(256), 0, w3d_s(stream), X, V, idx, R, K, (float *)nullptr, gF, gX, gV);
    W3D_CHECK_LAUNCH();
    return W3D_OK;
}

extern "C" int w3d_descriptor_mse_fwd_bwd(const float *X, const int32_t *idx, const float *target, int64_t R, int K, int64_t seg_rows,
                                          float weight, float *loss_accum, float *gX, w3d_stream_t stream) {
    if (!X || !idx || !target || !loss_accum || !gX || R < 0 || K < 2 || seg_rows < 0) return W3D_ERR_ARG;
    if (seg_rows > 0 && R % seg_rows != 0) return W3D_ERR_ARG;
    if (R == 0) return W3D_OK;
    hipLaunchKernelGGL(descriptor_bwd_kernel<true>, dim3(w3d_grid(R, 4, 8192)), dim3(256), 0, w3d_s(stream), X, idx, R, K, target, seg_rows, weight,
                       loss_accum, gX);
    W3D_CHECK_LAUNCH();
    return W3D_OK;
}

static int descriptor_bwd_csr_impl(const float *X, int xs, const int32_t *idx, int64_t R, int K, const float *gF, const int32_t *row_ptr, const int32_t *entries,
                                   int64_t n_points, float *rootsum, int accumulate, float *gX, w3d_stream_t stream) {
    if (!X || !idx || !gF || !row_ptr || !entries || !gX || !rootsum || R <= 0 || K < 2) return W3D_ERR_ARG;
    if (xs == 4 && ((uintptr_t)X & 15)) return W3D_ERR_ARG;
    hipStream_t st = w3d_s(stream);
    const int2 *e2 = reinterpret_cast<const int2 *>(entries);
    if (xs == 3) {
        hipLaunchKernelGGL(descriptor_rootsum_kernel<3>, dim3(w3d_grid(R, 4, 8192)), dim3(256), 0, st, X, idx, R, K, gF, rootsum);
        hipLaunchKernelGGL(descriptor_bwd_csr_kernel<3>, dim3(w3d_grid(n_points, 4, 16384)), dim3(256), 0, st, X, gF, rootsum, row_ptr, e2, n_points, accumulate, gX);
    } else {
        hipLaunchKernelGGL(descriptor_rootsum_kernel<4>, dim3(w3d_grid(R, 4, 8192)), dim3(256), 0, st, X, idx, R, K, gF, rootsum);
        hipLaunchKernelGGL(descriptor_bwd_csr_kernel<4>, dim3(w3d_grid(n_points, 4, 16384)), dim3(256), 0, st, X, gF, rootsum, row_ptr, e2, n_points, accumulate, gX);
    }
    W3D_CHECK_LAUNCH();
    return W3D_OK;
}
extern "C" int w3d_descriptor_bwd_csr(const float *X, const int32_t *idx, int64_t R, int K, const float *gF, const int32_t *row_ptr, const int32_t *entries,
                                      int64_t n_points, float *rootsum, int accumulate, float *gX, w3d_stream_t stream) {
    return descriptor_bwd_csr_impl(X, 3, idx, R, K, gF, row_ptr, entries, n_points, rootsum, accumulate, gX, stream);
}
extern "C" int w3d_descriptor_bwd_csr_x4(const float *X4, const int32_t *idx, int64_t R, int K, const float *gF, const int32_t *row_ptr, const int32_t *entries,
                                         int64_t n_points, float *rootsum, int accumulate, float *gX, w3d_stream_t stream) {
    return descriptor_bwd_csr_impl(X4, 4, idx, R, K, gF, row_ptr, entries, n_points, rootsum, accumulate, gX, stream);
}

// ------------------------------------------------------------------------------------------------
// Reference-regime domain loss (d <= 4, M up to a few thousand), batched over segments.
//   pass 1: k-th smallest distance of the row (thread-per-row bubble list for k <= 32,
//           block-per-row bitwise radix select otherwise)
//   pass 2: masked sums with D <= kth  (exactly the reference's sort -> kth -> mask -> mean)
// Y chunks are staged in LDS and read as broadcasts.
// ------------------------------------------------------------------------------------------------
template <int D>
__device__ __forceinline__ float dist_mm(const float *x, float x2, const float *y) { return w3d_dist_mm<D>(x, x2, y); }

constexpr int DL_CHUNK = 1024;

template <int D, int KMAX>
__global__ __launch_bounds__(256) void domain_loss_small_k_kernel(const float *__restrict__ X, int64_t rows_per_seg, const float *__restrict__ Y,
                                                                  const int32_t *__restrict__ yoff, int k, float weight,
                                                                  float *__restrict__ loss_accum, float *__restrict__ gX, float *__restrict__ kth_out) {
    __shared__ float ys[DL_CHUNK * D];
    __shared__ float red[16];
    const int seg = blockIdx.y;
    const int64_t rloc = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = rloc < rows_per_seg;
    const int64_t row = (int64_t)seg * rows_per_seg + (live ? rloc : 0);
    const int y0 = yoff[seg], M = yoff[seg + 1] - y0;
    float x[D], x2 = 0.f;
#pragma unroll
    for (int c = 0; c < D; ++c) { x[c] = X[row * D + c]; x2 = fmaf(x[c], x[c], x2); }
    float best[KMAX];
#pragma unroll
    for (int t = 0; t < KMAX; ++t) best[t] = __builtin_inff();
    for (int base = 0; base < M; base += DL_CHUNK) {
        const int n = min(DL_CHUNK, M - base);
        __syncthreads();
        for (int i = threadIdx.x; i < n * D; i += 256) ys[i] = Y[(int64_t)(y0 + base) * D + i];
        __syncthreads();
        for (int j = 0; j < n; ++j) {
            float v = dist_mm<D>(x, x2, &ys[j * D]);
            if (v < best[KMAX - 1]) {
#pragma unroll
                for (int t = 0; t < KMAX; ++t) { const float lo = fminf(best[t], v); v = fmaxf(best[t], v); best[t] = lo; }
            }
        }
    }
    float kth = best[0];
#pragma unroll
    for (int t = 1; t < KMAX; ++t) kth = (t == k - 1) ? best[t] : kth;
    if (M < k) kth = __builtin_inff();
    float s2 = 0.f, sy[D], cnt = 0.f;
#pragma unroll
    for (int c = 0; c < D; ++c) sy[c] = 0.f;
    for (int base = 0; base < M; base += DL_CHUNK) {
        const int n = min(DL_CHUNK, M - base);
        __syncthreads();
        for (int i = threadIdx.x; i < n * D; i += 256) ys[i] = Y[(int64_t)(y0 + base) * D + i];
        __syncthreads();
        for (int j = 0; j < n; ++j) {
            const float v = dist_mm<D>(x, x2, &ys[j * D]);
            if (v <= kth) {
                s2 += v * v;
                if (v > 0.f) {
                    cnt += 1.f;
#pragma unroll
                    for (int c = 0; c < D; ++c) sy[c] += ys[j * D + c];
                }
            }
        }
    }
    const float inv = weight / ((float)rows_per_seg * (float)M);
    if (live) {
#pragma unroll
        for (int c = 0; c < D; ++c) gX[row * D + c] += 2.0f * inv * (cnt * x[c] - sy[c]);
        if (kth_out) kth_out[row] = kth;
    }
    const float tot = block_sum(live ? s2 : 0.f, red);
    if (threadIdx.x == 0) atomicAdd(&loss_accum[seg], inv * tot);
}

// any k: one block per row, 31-step bitwise select on the (non-negative) float bit pattern of D
template <int D>
__global__ __launch_bounds__(256) void domain_loss_any_k_kernel(const float *__restrict__ X, int64_t rows_per_seg, const float *__restrict__ Y,
                                                                const int32_t *__restrict__ yoff, int k, float weight,
                                                                float *__restrict__ loss_accum, float *__restrict__ gX, float *__restrict__ kth_out) {
    __shared__ float red[16];
    __shared__ int redi[16];
    __shared__ int bc;
    const int seg = blockIdx.y;
    const int64_t row = (int64_t)seg * rows_per_seg + blockIdx.x;
    const int y0 = yoff[seg], M = yoff[seg + 1] - y0;
    float x[D], x2 = 0.f;
#pragma unroll
    for (int c = 0; c < D; ++c) { x[c] = X[row * D + c]; x2 = fmaf(x[c], x[c], x2); }
    uint32_t prefix = 0;
    int need = k;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (M >= k) {
        for (int bit = 30; bit >= 0; --bit) {
            int c0 = 0;   // values whose bits [31:bit] equal prefix's (bit itself = 0)
            for (int j = threadIdx.x; j < M; j += 256) {
                const uint32_t u = __float_as_uint(dist_mm<D>(x, x2, &Y[(int64_t)(y0 + j) * D]));
                c0 += ((u ^ prefix) >> bit) == 0u;
            }
            c0 = wave_sum_i(c0);
            if (lane == 0) redi[wid] = c0;
            __syncthreads();
            if (threadIdx.x == 0) bc = redi[0] + redi[1] + redi[2] + redi[3];
            __syncthreads();
            const int c = bc;
            if (need > c) { need -= c; prefix |= (1u << bit); }
            __syncthreads();
        }
    }
    const float kth = M >= k ? __uint_as_float(prefix) : __builtin_inff();
    float s2 = 0.f, sy[D], cnt = 0.f;
#pragma unroll
    for (int c = 0; c < D; ++c) sy[c] = 0.f;
    for (int j = threadIdx.x; j < M; j += 256) {
        const float *y = &Y[(int64_t)(y0 + j) * D];
        const float v = dist_mm<D>(x, x2, y);
        if (v <= kth) {
            s2 += v * v;
            if (v > 0.f) {
                cnt += 1.f;
#pragma unroll
                for (int c = 0; c < D; ++c) sy[c] += y[c];
            }
        }
    }
    const float inv = weight / ((float)rows_per_seg * (float)M);
    const float tot = block_sum(s2, red);
    const float cn = block_sum(cnt, red);
    float g[D];
#pragma unroll
    for (int c = 0; c < D; ++c) g[c] = block_sum(sy[c], red);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int c = 0; c < D; ++c) gX[row * D + c] += 2.0f * inv * (cn * x[c] - g[c]);
        if (kth_out) kth_out[row] = kth;
        atomicAdd(&loss_accum[seg], inv * tot);
    }
}

template <int D>
static int launch_domain_loss(const float *X, int64_t rows_per_seg, const float *Y, const int32_t *yoff, int S, int k, float weight,
                              float *loss_accum, float *gX, float *kth_out, hipStream_t st) {
    if (k <= 32) {
        dim3 grid((unsigned)((rows_per_seg + 255) / 256), (unsigned)S);
        if (k <= 8)
            hipLaunchKernelGGL((domain_loss_small_k_kernel<D, 8>), grid, dim3(256), 0, st, X, rows_per_seg, Y, yoff, k, weight, loss_accum, gX, kth_out);
        else if (k <= 20)
            hipLaunchKernelGGL((domain_loss_small_k_kernel<D, 20>), grid, dim3(256), 0, st, X, rows_per_seg, Y, yoff, k, weight, loss_accum, gX, kth_out);
        else
            hipLaunchKernelGGL((domain_loss_small_k_kernel<D, 32>), grid, dim3(256), 0, st, X, rows_per_seg, Y, yoff, k, weight, loss_accum, gX, kth_out);
    } else {
        dim3 grid((unsigned)rows_per_seg, (unsigned)S);
        hipLaunchKernelGGL((domain_loss_any_k_kernel<D>), grid, dim3(256), 0, st, X, rows_per_seg, Y, yoff, k, weight, loss_accum, gX, kth_out);
    }
    return w3d_launch_status();
}

extern "C" int w3d_domain_loss_fwd_bwd(const float *X, int64_t rows_per_seg, int d, const float *Y, const int32_t *y_seg_offsets, int S,
                                       int64_t max_M, int k, float weight, float *loss_accum, float *gX, float *kth_out, w3d_stream_t stream) {
    if (!X || !Y || !y_seg_offsets || !loss_accum || !gX || rows_per_seg <= 0 || S <= 0 || k <= 0) return W3D_ERR_ARG;
    if (S > 65535) return W3D_ERR_UNSUPPORTED;
    (void)max_M;
    hipStream_t st = w3d_s(stream);
    switch (d) {
        case 1: return launch_domain_loss<1>(X, rows_per_seg, Y, y_seg_offsets, S, k, weight, loss_accum, gX, kth_out, st);
        case 2: return launch_domain_loss<2>(X, rows_per_seg, Y, y_seg_offsets, S, k, weight, loss_accum, gX, kth_out, st);
        case 3: return launch_domain_loss<3>(X, rows_per_seg, Y, y_seg_offsets, S, k, weight, loss_accum, gX, kth_out, st);
        case 4: return launch_domain_loss<4>(X, rows_per_seg, Y, y_seg_offsets, S, k, weight, loss_accum, gX, kth_out, st);
        default: return W3D_ERR_UNSUPPORTED;
    }
}

// ------------------------------------------------------------------------------------------------
// Generic low-d fp32 kNN (d <= 16, k <= 32): thread per row, (value, index) bubble list in registers.
// ------------------------------------------------------------------------------------------------
template <int KMAX>
__global__ __launch_bounds__(256) void knn_lowd_kernel(const float *__restrict__ X, const float *__restrict__ Y, int64_t N, int64_t M, int d, int k,
                                                       int32_t *__restrict__ idx, float *__restrict__ dist2) {
    constexpr int CH = 512;
    __shared__ float ys[CH * 16];
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = row < N;
    float x[16], x2 = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c) { x[c] = (live && c < d) ? X[row * d + c] : 0.f; x2 = fmaf(x[c], x[c], x2); }
    float bv[KMAX]; int bi[KMAX];
#pragma unroll
    for (int t = 0; t < KMAX; ++t) { bv[t] = __builtin_inff(); bi[t] = 0x7fffffff; }
    for (int64_t base = 0; base < M; base += CH) {
        const int n = (int)min((int64_t)CH, M - base);
        __syncthreads();
        for (int i = threadIdx.x; i < n * d; i += 256) ys[i] = Y[base * d + i];
        __syncthreads();
        for (int j = 0; j < n; ++j) {
            float dot = 0.f, y2 = 0.f;
            for (int c = 0; c < d; ++c) { const float yv = ys[j * d + c]; dot = fmaf(x[c], yv, dot); y2 = fmaf(yv, yv, y2); }
            float v = fmaxf(x2 + y2 - 2.0f * dot, 0.f);
            int vi = (int)(base + j);
            if (v < bv[KMAX - 1]) {
#pragma unroll
                for (int t = 0; t < KMAX; ++t) {
                    const bool sw = v < bv[t];           // strict: equal values keep the earlier (lower) index first
                    const float tv = bv[t]; const int ti = bi[t];
                    bv[t] = sw ? v : tv; bi[t] = sw ? vi : ti;
                    v = sw ? tv : v; vi = sw ? ti : vi;
                }
            }
        }
    }
    if (live) {
#pragma unroll
        for (int t = 0; t < KMAX; ++t)
            if (t < k) { idx[row * k + t] = bi[t]; dist2[row * k + t] = bv[t]; }
    }
}

extern "C" int w3d_knn_lowd_fwd(const float *X, const float *Y, int64_t N, int64_t M, int d, int k, int32_t *idx, float *dist2, hipStream_t st) {
    dim3 grid((unsigned)((N + 255) / 256));
    if (k <= 8) hipLaunchKernelGGL(knn_lowd_kernel<8>, grid, dim3(256), 0, st, X, Y, N, M, d, k, idx, dist2);
    else if (k <= 20) hipLaunchKernelGGL(knn_lowd_kernel<20>, grid, dim3(256), 0, st, X, Y, N, M, d, k, idx, dist2);
    else hipLaunchKernelGGL(knn_lowd_kernel<32>, grid, dim3(256), 0, st, X, Y, N, M, d, k, idx, dist2);
    return w3d_launch_status();
}

// ------------------------------------------------------------------------------------------------
// kNN loss head and backward (any d, both dtypes): one wave per row, lanes across the feature dim.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void knn_loss_kernel(const float *__restrict__ dist2, int64_t n, float scale, float *__restrict__ loss_accum) {
    __shared__ float red[16];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) s += dist2[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) atomicAdd(loss_accum, scale * s);
}

extern "C" int w3d_knn_loss(const float *dist2, int64_t N, int k, float scale, float *loss_accum, w3d_stream_t stream) {
    if (!dist2 || !loss_accum || N < 0 || k <= 0) return W3D_ERR_ARG;
    if (N == 0) return W3D_OK;
    hipLaunchKernelGGL(knn_loss_kernel, dim3(w3d_grid(N * k, 256, 1024)), dim3(256), 0, w3d_s(stream), dist2, N * k, scale, loss_accum);
    W3D_CHECK_LAUNCH();
    return W3D_OK;
}

// DT: 0 f32, 1 bf16, 2 f16
template <int DT>
__global__ __launch_bounds__(256) void knn_bwd_kernel(const void *__restrict__ Xv, const void *__restrict__ Yv, const int32_t *__restrict__ idx, int64_t N,
                                                      int64_t M, int d, int k, float scale2, int accumulate, float *__restrict__ gX) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    auto d16 = [](uint16_t b) -> float { return DT == 1 ? bf16_to_f32(b) : f16_to_f32(b); };
    for (int64_t i = wave0; i < N; i += nwaves) {
        for (int c0 = lane * 4; c0 < d; c0 += 256) {     // 4 consecutive features per lane (8 B / 16 B loads)
            float xs[4], acc[4] = {0.f, 0.f, 0.f, 0.f};
            const int nc = min(4, d - c0);
            if (DT != 0) {
                const uint16_t *X = reinterpret_cast<const uint16_t *>(Xv);
                for (int c = 0; c < nc; ++c) xs[c] = d16(X[i * d + c0 + c]);
            } else {
                const float *X = reinterpret_cast<const float *>(Xv);
                for (int c = 0; c < nc; ++c) xs[c] = X[i * d + c0 + c];
            }
            for (int j = 0; j < k; ++j) {
                int64_t y = idx[i * k + j];
                // an unfilled list slot (the kNN kernel's sentinel: a NaN / Inf feature row collects no candidates) names no style row: nothing is
                // read out of bounds and the row's gradient is NaN, which is what the reference's cdist -> mask -> mean propagates
                if ((uint64_t)y >= (uint64_t)M) { y = 0; acc[0] = acc[1] = acc[2] = acc[3] = __builtin_nanf(""); }
                if (DT != 0) {
                    const uint16_t *Y = reinterpret_cast<const uint16_t *>(Yv);
                    if (nc == 4) {
                        const uint2 p = *reinterpret_cast<const uint2 *>(&Y[y * d + c0]);
                        acc[0] += d16((uint16_t)(p.x & 0xffffu)); acc[1] += d16((uint16_t)(p.x >> 16));
                        acc[2] += d16((uint16_t)(p.y & 0xffffu)); acc[3] += d16((uint16_t)(p.y >> 16));
                    } else {
                        for (int c = 0; c < nc; ++c) acc[c] += d16(Y[y * d + c0 + c]);
                    }
                } else {
                    const float *Y = reinterpret_cast<const float *>(Yv);
                    for (int c = 0; c < nc; ++c) acc[c] += Y[y * d + c0 + c];
                }
            }
            for (int c = 0; c < nc; ++c) {
                const float g = scale2 * ((float)k * xs[c] - acc[c]);
                if (accumulate) gX[i * d + c0 + c] += g; else gX[i * d + c0 + c] = g;
            }
        }
    }
}

// The same gradient for 16-bit rows of 256 elements, four rows per wave: a 16-lane group owns a row (32 B of it per lane), a gathered style row is two 16-B
// loads per lane, and 10 of the k rows are in flight per group (the one-row-per-wave kernel above walks the list with one dependent load per neighbour:
// 2.27 ms at C3; this form 10 GB of gathers at the rate of the warm-start bound kernel).  row_order (optional): the order the rows are TAKEN in (position -> row;
// similar rows together share their neighbours' L2 lines); results land at the row's own place.  Same terms, same order of the sum over j.
template <int DT>
__global__ __launch_bounds__(256) void knn_bwd4_kernel(const uint16_t *__restrict__ X, const uint16_t *__restrict__ Y, const int32_t *__restrict__ idx, int64_t N,
                                                       int64_t M, int k, float scale2, int accumulate, float *__restrict__ gX, const int32_t *__restrict__ row_order) {
    constexpr int D = 256, B = 10;
    const int lane = threadIdx.x & 63, sub = lane & 15, grp = lane >> 4;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    auto d16 = [](uint32_t b) -> float { return DT == 1 ? __uint_as_float(b << 16) : f16_to_f32((uint16_t)b); };
    auto lo = [&](uint32_t w) -> float { return DT == 1 ? __uint_as_float(w << 16) : d16(w & 0xffffu); };
    auto hi = [&](uint32_t w) -> float { return DT == 1 ? __uint_as_float(w & 0xffff0000u) : d16(w >> 16); };
    for (int64_t pos4 = wave0 * 4; pos4 < N; pos4 += nwaves * 4) {
        int64_t pos = pos4 + grp;
        const bool live = pos < N;
        if (!live) pos = N - 1;
        const int64_t i = row_order ? (int64_t)row_order[pos] : pos;
        const uint4 x0 = *reinterpret_cast<const uint4 *>(X + i * D + sub * 16), x1 = *reinterpret_cast<const uint4 *>(X + i * D + sub * 16 + 8);
        float acc[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) acc[c] = 0.f;
        bool bad = false;
        for (int j0 = 0; j0 < k; j0 += 16) {
            const int jm = j0 + sub;
            int32_t ym = jm < k ? idx[i * k + jm] : 0;
            // an unfilled list slot (the kNN kernel's sentinel) names no style row: row 0 is read instead and the row's gradient is NaN (as knn_bwd_kernel)
            if (jm < k && (uint32_t)ym >= (uint32_t)M) { bad = true; ym = 0; }
            const int cnt = min(16, k - j0);
            for (int u0 = 0; u0 < cnt; u0 += B) {
                uint4 y0[B], y1[B];
#pragma unroll
                for (int u = 0; u < B; ++u) {
                    const int64_t y = (int64_t)__shfl(ym, (lane & 48) | min(u0 + u, cnt - 1), 64);
                    y0[u] = *reinterpret_cast<const uint4 *>(Y + y * D + sub * 16);
                    y1[u] = *reinterpret_cast<const uint4 *>(Y + y * D + sub * 16 + 8);
                }
#pragma unroll
                for (int u = 0; u < B; ++u) {
                    if (u0 + u < cnt) {
                        const uint32_t w[8] = {y0[u].x, y0[u].y, y0[u].z, y0[u].w, y1[u].x, y1[u].y, y1[u].z, y1[u].w};
#pragma unroll
                        for (int q = 0; q < 8; ++q) { acc[2 * q] += lo(w[q]); acc[2 * q + 1] += hi(w[q]); }
                    }
                }
            }
        }
        // any lane of the group that saw a sentinel poisons the whole row
        bad = (__builtin_amdgcn_ballot_w64(bad) >> (grp * 16) & 0xffffull) != 0ull;
        if (live) {
            const uint32_t xw[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
            float g[16];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                g[2 * q] = bad ? __builtin_nanf("") : scale2 * ((float)k * lo(xw[q]) - acc[2 * q]);
                g[2 * q + 1] = bad ? __builtin_nanf("") : scale2 * ((float)k * hi(xw[q]) - acc[2 * q + 1]);
            }
            float4 *o = reinterpret_cast<float4 *>(gX + i * D + sub * 16);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                float4 t = make_float4(g[4 * v], g[4 * v + 1], g[4 * v + 2], g[4 * v + 3]);
                if (accumulate) { const float4 p = o[v]; t.x += p.x; t.y += p.y; t.z += p.z; t.w += p.w; }
                o[v] = t;
            }
        }
    }
}

extern "C" int w3d_knn_match_bwd_ordered(const void *X, const void *Y, const int32_t *idx, int64_t N, int64_t M, int d, int k, int dtype, float scale,
                                         int accumulate, const int32_t *row_order, float *grad_X, w3d_stream_t stream) {
    if (!X || !Y || !idx || !grad_X || N < 0 || M < 1 || d <= 0 || k <= 0) return W3D_ERR_ARG;
    if (d != 256 || (dtype != W3D_DTYPE_BF16 && dtype != W3D_DTYPE_F16)) return W3D_ERR_UNSUPPORTED;
    if (N == 0) return W3D_OK;
    const dim3 grid((unsigned)std::min<int64_t>((N + 15) / 16, 16384));
    if (dtype == W3D_DTYPE_BF16)
        hipLaunchKernelGGL(knn_bwd4_kernel<1>, grid, dim3(256), 0, w3d_s(stream), reinterpret_cast<const uint16_t *>(X), reinterpret_cast<const uint16_t *>(Y), idx, N, M, k,
                           2.0f * scale, accumulate, grad_X, row_order);
    else
        hipLaunchKernelGGL(knn_bwd4_kernel<2>, grid, dim3(256), 0, w3d_s(stream), reinterpret_cast<const uint16_t *>(X), reinterpret_cast<const uint16_t *>(Y), idx, N, M, k,
                           2.0f * scale, accumulate, grad_X, row_order);
    W3D_CHECK_LAUNCH();
    return W3D_OK;
}

extern "C" int w3d_knn_match_bwd(const void *X, const void *Y, const int32_t *idx, int64_t N, int64_t M, int d, int k, int dtype, float scale,
                                 int accumulate, float *grad_X, w3d_stream_t stream) {
    if (!X || !Y || !idx || !grad_X || N < 0 || M < 1 || d <= 0 || k <= 0) return W3D_ERR_ARG;
    if (N == 0) return W3D_OK;
    if (dtype == W3D_DTYPE_BF16 || dtype == W3D_DTYPE_F16) {
        if (d % 4) return W3D_ERR_UNSUPPORTED;
        if (dtype == W3D_DTYPE_BF16) hipLaunchKernelGGL(knn_bwd_kernel<1>, dim3(w3d_grid(N, 4, 16384)), dim3(256), 0, w3d_s(stream), X, Y, idx, N, M, d, k, 2.0f * scale, accumulate, grad_X);
        else hipLaunchKernelGGL(knn_bwd_kernel<2>, dim3(w3d_grid(N, 4, 16384)), dim3(256), 0, w3d_s(stream), X, Y, idx, N, M, d, k, 2.0f * scale, accumulate, grad_X);
    } else if (dtype == W3D_DTYPE_F32) {
        hipLaunchKernelGGL(knn_bwd_kernel<0>, dim3(w3d_grid(N, 4, 16384)), dim3(256), 0, w3d_s(stream), X, Y, idx, N, M, d, k, 2.0f * scale, accumulate, grad_X);
    } else {
        return W3D_ERR_ARG;
    }
    W3D_CHECK_LAUNCH();
    return W3D_OK;
}
