// ABI glue: error strings, version, kNN dispatch between the VALU (fp32, low d) and MFMA (bf16) kernels.
#include "common.h"

extern "C" int w3d_knn_lowd_fwd(const float *X, const float *Y, int64_t N, int64_t M, int d, int k, int32_t *idx, float *dist2, hipStream_t st);
extern "C" size_t w3d_knn_mfma_workspace_bytes(int64_t N, int64_t M, int d, int k);
extern "C" int w3d_knn_mfma_fwd(const void *X, const void *Y, int64_t N, int64_t M, int d, int k, const int32_t *prev_idx, int32_t *idx, float *dist2, void *ws,
                                size_t ws_bytes, int phases, int warm, hipStream_t st, const float *prev_dist2, const float *motion2);

extern "C" const char *w3d_strerror(int code) {
    switch (code) {
        case W3D_OK: return "ok";
        case W3D_ERR_ARG: return "invalid argument";
        case W3D_ERR_LAUNCH: return "HIP launch/runtime error";
        case W3D_ERR_WORKSPACE: return "workspace too small";
        case W3D_ERR_UNSUPPORTED: return "unsupported size/dtype for this build";
        default: return "unknown error";
    }
}

extern "C" int w3d_abi_version(void) { return 1; }

extern "C" size_t w3d_knn_workspace_bytes(int64_t N, int64_t M, int d, int k, int dtype) {
    if (dtype == W3D_DTYPE_BF16) return w3d_knn_mfma_workspace_bytes(N, M, d, k);
    return 256;
}

extern "C" int w3d_knn_match_fwd(const void *X, const void *Y, int64_t N, int64_t M, int d, int k, int dtype, const int32_t *prev_idx, int32_t *idx,
                                 float *dist2, void *workspace, size_t workspace_bytes, w3d_stream_t stream) {
    if (!X || !Y || !idx || !dist2 || N < 0 || M <= 0 || d <= 0 || k <= 0 || k > M) return W3D_ERR_ARG;
    if (N == 0) return W3D_OK;
    if (dtype == W3D_DTYPE_F32) {
        if (d > 16 || k > 32) return W3D_ERR_UNSUPPORTED;
        return w3d_knn_lowd_fwd((const float *)X, (const float *)Y, N, M, d, k, idx, dist2, w3d_s(stream));
    }
    if (dtype == W3D_DTYPE_BF16) return w3d_knn_mfma_fwd(X, Y, N, M, d, k, prev_idx == idx ? nullptr : prev_idx, idx, dist2, workspace, workspace_bytes, 3, 0, w3d_s(stream), nullptr, nullptr);
    return W3D_ERR_ARG;
}

// The same call in two steps (bf16 path), so that the main kernel can be timed / captured on its own:
//   prepare  = row norms (+ the warm-start bound if prev_idx != NULL) into `workspace`
//   prepared = the fused distance-GEMM + top-k kernel alone, reading a workspace filled by prepare (warm = prev_idx was given)
extern "C" int w3d_knn_prepare(const void *X, const void *Y, int64_t N, int64_t M, int d, int k, int dtype, const int32_t *prev_idx, void *workspace,
                               size_t workspace_bytes, w3d_stream_t stream) {
    if (!X || !Y || N <= 0 || M <= 0 || d <= 0 || k <= 0 || k > M) return W3D_ERR_ARG;
    if (dtype != W3D_DTYPE_BF16) return W3D_ERR_UNSUPPORTED;
    return w3d_knn_mfma_fwd(X, Y, N, M, d, k, prev_idx, nullptr, nullptr, workspace, workspace_bytes, 1, 0, w3d_s(stream), nullptr, nullptr);
}
extern "C" int w3d_knn_match_prepared(const void *X, const void *Y, int64_t N, int64_t M, int d, int k, int dtype, int warm, int32_t *idx, float *dist2,
                                      void *workspace, size_t workspace_bytes, w3d_stream_t stream) {
    if (!X || !Y || !idx || !dist2 || N <= 0 || M <= 0 || d <= 0 || k <= 0 || k > M) return W3D_ERR_ARG;
    if (dtype != W3D_DTYPE_BF16) return W3D_ERR_UNSUPPORTED;
    return w3d_knn_mfma_fwd(X, Y, N, M, d, k, nullptr, idx, dist2, workspace, workspace_bytes, 2, warm, w3d_s(stream), nullptr, nullptr);
}

// prepare variant for optimisation loops: the bound comes from the previous iteration's k-th distances and how far each row moved
// (prev_dist2 [N,k] ascending as returned by the previous call, motion2 [N] = |x_new - x_old|^2), no gather of Y rows.
extern "C" int w3d_knn_prepare_motion(const void *X, const void *Y, int64_t N, int64_t M, int d, int k, int dtype, const float *prev_dist2, const float *motion2,
                                      void *workspace, size_t workspace_bytes, w3d_stream_t stream) {
    if (!X || !Y || !prev_dist2 || !motion2 || N <= 0 || M <= 0 || d <= 0 || k <= 0 || k > M) return W3D_ERR_ARG;
    if (dtype != W3D_DTYPE_BF16) return W3D_ERR_UNSUPPORTED;
    return w3d_knn_mfma_fwd(X, Y, N, M, d, k, nullptr, nullptr, nullptr, workspace, workspace_bytes, 1, 0, w3d_s(stream), prev_dist2, motion2);
}
