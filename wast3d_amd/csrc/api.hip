// ABI glue: error strings, version, kNN dispatch between the VALU (fp32, low d) and MFMA (bf16) kernels.
#include "common.h"

extern "C" int w3d_knn_lowd_fwd(const float *X, const float *Y, int64_t N, int64_t M, int d, int k, int32_t *idx, float *dist2, hipStream_t st);
extern "C" size_t w3d_knn_mfma_workspace_bytes(int64_t N, int64_t M, int d, int k);
extern "C" int w3d_knn_mfma_fwd_fmt(const void *X, const void *Y, int64_t N, int64_t M, int d, int k, const int32_t *prev_idx, int32_t *idx, float *dist2, void *ws,
                                    size_t ws_bytes, int phases, int warm, hipStream_t st, const float *prev_dist2, const float *motion2, int fmt,
                                    const int32_t *xscale, const int32_t *yscale, const float *xnorm_in, const int32_t *row_perm);
extern "C" int w3d_knn_mfma_fwd_lb(const void *X, const void *Y, int64_t N, int64_t M, int d, int k, const int32_t *prev_idx, int32_t *idx, float *dist2, void *ws,
                                   size_t ws_bytes, int phases, int warm, hipStream_t st, const float *prev_dist2, const float *motion2, int fmt,
                                   const int32_t *xscale, const int32_t *yscale, const float *xnorm_in, const int32_t *row_perm, uint16_t *lb, const float *lb_cum,
                                   float lb_margin, float motion_ratio);
extern "C" int w3d_knn_mfma_part_fmt(int op, const void *X, const void *Y, const int32_t *xscale, const int32_t *yscale, int64_t N, int64_t M, int d, int k, int fmt, int warm,
                                     int64_t m0, int64_t m1, int64_t m_valid, int part, int parts, int32_t *part_idx, float *part_val, int32_t *idx, float *dist2, void *ws,
                                     size_t ws_bytes, hipStream_t st, const int32_t *row_perm);
// MFMA operand format of a 16-bit dtype code (-1: not an MFMA dtype)
static int mfma_fmt(int dtype) { return dtype == W3D_DTYPE_BF16 ? 0 : dtype == W3D_DTYPE_F16 ? 1 : -1; }
static int w3d_knn_mfma_fwd(const void *X, const void *Y, int64_t N, int64_t M, int d, int k, const int32_t *prev_idx, int32_t *idx, float *dist2, void *ws,
                            size_t ws_bytes, int phases, int warm, hipStream_t st, const float *prev_dist2, const float *motion2, int fmt = 0,
                            const int32_t *xs = nullptr, const int32_t *ys = nullptr, const float *xnorm_in = nullptr, const int32_t *row_perm = nullptr) {
    return w3d_knn_mfma_fwd_fmt(X, Y, N, M, d, k, prev_idx, idx, dist2, ws, ws_bytes, phases, warm, st, prev_dist2, motion2, fmt, xs, ys, xnorm_in, row_perm);
}

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

extern "C" int w3d_abi_version(void) { return 7; }

thread_local int w3d_last_hip_error_code = 0;
// Name and description of the HIP error behind the calling thread's most recent W3D_ERR_LAUNCH ("" if none): which runtime call refused what.
extern "C" const char *w3d_last_hip_error_string(void) {
    if (w3d_last_hip_error_code == 0) return "";
    return hipGetErrorString((hipError_t)w3d_last_hip_error_code);
}
// Entry points report HIP's sticky per-thread error state after their launches; an error left behind by an EARLIER runtime call of the host program
// (e.g. a benign hipErrorNotReady of an event query) must not be charged to them: callers that share the thread with other HIP users clear it first.
extern "C" void w3d_clear_hip_error(void) { (void)hipGetLastError(); }

extern "C" size_t w3d_knn_workspace_bytes(int64_t N, int64_t M, int d, int k, int dtype) {
    if (dtype == W3D_DTYPE_BF16 || dtype == W3D_DTYPE_F16 || dtype == W3D_DTYPE_FP8_E4M3) return w3d_knn_mfma_workspace_bytes(N, M, d, k);
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
    if (mfma_fmt(dtype) >= 0)
        return w3d_knn_mfma_fwd(X, Y, N, M, d, k, prev_idx == idx ? nullptr : prev_idx, idx, dist2, workspace, workspace_bytes, 3, 0, w3d_s(stream), nullptr, nullptr, mfma_fmt(dtype));
    return W3D_ERR_ARG;
}

// The same call in two steps (bf16 path), so that the main kernel can be timed / captured on its own:
//   prepare  = row norms (+ the warm-start bound if prev_idx != NULL) into `workspace`
//   prepared = the fused distance-GEMM + top-k kernel alone, reading a workspace filled by prepare (warm = prev_idx was given)
extern "C" int w3d_knn_prepare(const void *X, const void *Y, int64_t N, int64_t M, int d, int k, int dtype, const int32_t *prev_idx, const float *xnorm,
                               int style_unchanged, const int32_t *row_order, void *workspace, size_t workspace_bytes, w3d_stream_t stream) {
    if (!X || (!Y && prev_idx) || N <= 0 || M <= 0 || d <= 0 || k <= 0 || k > M) return W3D_ERR_ARG;
    if (mfma_fmt(dtype) < 0) return W3D_ERR_UNSUPPORTED;
    return w3d_knn_mfma_fwd(X, Y, N, M, d, k, prev_idx, nullptr, nullptr, workspace, workspace_bytes, style_unchanged ? 1 | 4 : 1, 0, w3d_s(stream), nullptr, nullptr, mfma_fmt(dtype),
                            nullptr, nullptr, xnorm, row_order);
}
extern "C" int w3d_knn_match_prepared(const void *X, const void *Y, int64_t N, int64_t M, int d, int k, int dtype, int warm, int32_t *idx, float *dist2,
                                      const int32_t *row_order, void *workspace, size_t workspace_bytes, w3d_stream_t stream) {
    if (!X || !Y || !idx || !dist2 || N <= 0 || M <= 0 || d <= 0 || k <= 0 || k > M) return W3D_ERR_ARG;
    if (mfma_fmt(dtype) < 0) return W3D_ERR_UNSUPPORTED;
    return w3d_knn_mfma_fwd(X, Y, N, M, d, k, nullptr, idx, dist2, workspace, workspace_bytes, 2, warm, w3d_s(stream), nullptr, nullptr, mfma_fmt(dtype), nullptr, nullptr, nullptr,
                            row_order);
}

// match_prepared with the cached lower bounds of a loop (warm must be 4: pruned sweep, EMIT selection): every swept (row block, tile) pair records its rows' exact
// minimum distances + cumulative motion in lb; the next iterations' plans (w3d_knn_prune_plan_lb) skip tiles those bounds still rule out.  lb == NULL: plain warm = 4.
extern "C" int w3d_knn_match_prepared_lb(const void *X, const void *Y, int64_t N, int64_t M, int d, int k, int dtype, int warm, int32_t *idx, float *dist2,
                                         const int32_t *row_order, uint16_t *lb, const float *lb_cum, float cnorm_max, void *workspace, size_t workspace_bytes,
                                         w3d_stream_t stream) {
    if (!X || !Y || !idx || !dist2 || N <= 0 || M <= 0 || d <= 0 || k <= 0 || k > M) return W3D_ERR_ARG;
    if (mfma_fmt(dtype) < 0) return W3D_ERR_UNSUPPORTED;
    if ((lb == nullptr) != (lb_cum == nullptr) || (lb && ((warm & 0xff) != 4 || !(cnorm_max >= 0.f))) || (warm & ~0x1ff) || ((warm & 0x100) && !lb)) return W3D_ERR_ARG;      // (0x100: the plan's straggler pairs)
    return w3d_knn_mfma_fwd_lb(X, Y, N, M, d, k, nullptr, idx, dist2, workspace, workspace_bytes, 2, warm, w3d_s(stream), nullptr, nullptr, mfma_fmt(dtype), nullptr, nullptr, nullptr,
                               row_order, lb, lb_cum, 4e-6f * 2.0f * cnorm_max, 0.f);
}

// w3d_knn_prepare(prev_idx) for loops whose rows move little: a row whose motion since the previous sweep is at most motion_ratio x its previous k-th distance takes
// the motion bound (prev_dist2 [N,k] = the previous call's dist2, motion2 [N] = |x_new - x_old|^2) and its neighbours' rows are not fetched; every other row gets the
// gathered bound.  Either bound is valid: the result is exact; motion_ratio = 0 is w3d_knn_prepare.  16-bit rows of 128 / 256 features (others: always gathered).
extern "C" int w3d_knn_prepare_hybrid(const void *X, const void *Y, int64_t N, int64_t M, int d, int k, int dtype, const int32_t *prev_idx, const float *prev_dist2,
                                      const float *motion2, float motion_ratio, const float *xnorm, int style_unchanged, const int32_t *row_order, void *workspace,
                                      size_t workspace_bytes, w3d_stream_t stream) {
    if (!X || !Y || !prev_idx || !prev_dist2 || !motion2 || N <= 0 || M <= 0 || d <= 0 || k <= 0 || k > M || !(motion_ratio >= 0.f)) return W3D_ERR_ARG;
    if (mfma_fmt(dtype) < 0) return W3D_ERR_UNSUPPORTED;
    return w3d_knn_mfma_fwd_lb(X, Y, N, M, d, k, prev_idx, nullptr, nullptr, workspace, workspace_bytes, style_unchanged ? 1 | 4 : 1, 0, w3d_s(stream), prev_dist2, motion2,
                               mfma_fmt(dtype), nullptr, nullptr, xnorm, row_order, nullptr, nullptr, 0.f, motion_ratio);
}

// prepare variant for optimisation loops: the bound comes from the previous iteration's k-th distances and how far each row moved
// (prev_dist2 [N,k] ascending as returned by the previous call, motion2 [N] = |x_new - x_old|^2), no gather of Y rows.
extern "C" int w3d_knn_prepare_motion(const void *X, const void *Y, int64_t N, int64_t M, int d, int k, int dtype, const float *prev_dist2, const float *motion2,
                                      const float *xnorm, const int32_t *row_order, void *workspace, size_t workspace_bytes, w3d_stream_t stream) {
    if (!X || !prev_dist2 || !motion2 || N <= 0 || M <= 0 || d <= 0 || k <= 0 || k > M) return W3D_ERR_ARG;      // Y == NULL: chunked sweep (w3d_knn_chunk_*), style norms per chunk
    if (mfma_fmt(dtype) < 0) return W3D_ERR_UNSUPPORTED;
    return w3d_knn_mfma_fwd(X, Y, N, M, d, k, nullptr, nullptr, nullptr, workspace, workspace_bytes, 1, 0, w3d_s(stream), prev_dist2, motion2, mfma_fmt(dtype), nullptr,
                            nullptr, xnorm, row_order);
}

// ---- chunked sweep: the style set arrives shard by shard (RCCL broadcasts over xGMI) while the distance-GEMM runs on the shards that are there.
// Sequence per iteration: w3d_knn_prepare / w3d_knn_prepare_motion with Y = NULL (content norms + bound);  per chunk c, once its rows are in place:
// w3d_knn_chunk_norms, w3d_knn_chunk_sweep (slot c);  then w3d_knn_chunk_merge.  Result = w3d_knn_match_prepared on the whole set, bit for bit.
extern "C" int w3d_knn_chunk_norms(const void *Y, int64_t N, int64_t M, int d, int k, int dtype, int64_t m0, int64_t m_valid, int64_t m1, void *workspace,
                                   size_t workspace_bytes, w3d_stream_t stream) {
    if (!Y || N <= 0 || M <= 0 || mfma_fmt(dtype) < 0) return W3D_ERR_ARG;
    return w3d_knn_mfma_part_fmt(0, Y, Y, nullptr, nullptr, N, M, d, k, mfma_fmt(dtype), 0, m0, m1, m_valid, 0, 1, nullptr, nullptr, nullptr, nullptr, workspace, workspace_bytes,
                                 w3d_s(stream), nullptr);
}
extern "C" int w3d_knn_chunk_sweep(const void *X, const void *Y, int64_t N, int64_t M, int d, int k, int dtype, int warm, int64_t m0, int64_t m1, int part, int parts,
                                   int32_t *part_idx, float *part_val, const int32_t *row_order, void *workspace, size_t workspace_bytes, w3d_stream_t stream) {
    if (!X || !Y || N <= 0 || M <= 0 || k <= 0 || mfma_fmt(dtype) < 0) return W3D_ERR_ARG;
    return w3d_knn_mfma_part_fmt(1, X, Y, nullptr, nullptr, N, M, d, k, mfma_fmt(dtype), warm, m0, m1, m1, part, parts, part_idx, part_val, nullptr, nullptr, workspace,
                                 workspace_bytes, w3d_s(stream), row_order);
}
extern "C" int w3d_knn_chunk_merge(const void *X, const void *Y, int64_t N, int64_t M, int d, int k, int dtype, int warm, int parts, const int32_t *part_idx,
                                   const float *part_val, int32_t *idx, float *dist2, const int32_t *row_order, void *workspace, size_t workspace_bytes, w3d_stream_t stream) {
    if (!X || !Y || !idx || !dist2 || N <= 0 || M <= 0 || k <= 0 || mfma_fmt(dtype) < 0) return W3D_ERR_ARG;
    return w3d_knn_mfma_part_fmt(2, X, Y, nullptr, nullptr, N, M, d, k, mfma_fmt(dtype), warm, 0, M, M, 0, parts, const_cast<int32_t *>(part_idx), const_cast<float *>(part_val),
                                 idx, dist2, workspace, workspace_bytes, w3d_s(stream), row_order);
}

// ---- fp8 (OCP e4m3) operands with one power-of-two scale per row: the distance-GEMM on v_mfma_scale_f32_32x32x64_f8f6f4 (2x the bf16 rate).
// The result is the EXACT kNN of the dequantised rows q * 2^(e-127); phases as above: prepare (1), prepared (2), or both (3).
extern "C" int w3d_knn_fp8(const void *qX, const int32_t *xscale, const void *qY, const int32_t *yscale, int64_t N, int64_t M, int d, int k, int phases, int warm,
                           const int32_t *prev_idx, const float *prev_dist2, const float *motion2, int32_t *idx, float *dist2, const int32_t *row_order,
                           void *workspace, size_t workspace_bytes, w3d_stream_t stream) {
    if (!qX || !qY || !xscale || !yscale || N <= 0 || M <= 0 || d <= 0 || k <= 0 || k > M || phases < 1 || phases > 3) return W3D_ERR_ARG;
    if ((phases & 2) && (!idx || !dist2)) return W3D_ERR_ARG;
    if ((prev_dist2 == nullptr) != (motion2 == nullptr)) return W3D_ERR_ARG;
    return w3d_knn_mfma_fwd(qX, qY, N, M, d, k, (phases & 1) ? (prev_idx == idx ? nullptr : prev_idx) : nullptr, idx, dist2, workspace, workspace_bytes, phases, warm,
                            w3d_s(stream), (phases & 1) ? prev_dist2 : nullptr, (phases & 1) ? motion2 : nullptr, 2, xscale, yscale, nullptr, row_order);
}

// the main kernel of w3d_knn_fp8 with the EMIT selection (+ cached lower bounds): see include/wast3d_hip.h
extern "C" int w3d_knn_fp8_lb(const void *qX, const int32_t *xscale, const void *qY, const int32_t *yscale, int64_t N, int64_t M, int d, int k, int warm, int32_t *idx,
                              float *dist2, const int32_t *row_order, uint16_t *lb, const float *lb_cum, float cnorm_max, void *workspace, size_t workspace_bytes,
                              w3d_stream_t stream) {
    if (!qX || !qY || !xscale || !yscale || !idx || !dist2 || N <= 0 || M <= 0 || d <= 0 || k <= 0 || k > M || warm != 4) return W3D_ERR_ARG;
    if ((lb == nullptr) != (lb_cum == nullptr) || (lb && !(cnorm_max >= 0.f))) return W3D_ERR_ARG;
    return w3d_knn_mfma_fwd_lb(qX, qY, N, M, d, k, nullptr, idx, dist2, workspace, workspace_bytes, 2, warm, w3d_s(stream), nullptr, nullptr, 2, xscale, yscale, nullptr,
                               row_order, lb, lb_cum, lb ? W3D_KNN_SCORE_ERR2_FP8 * 2.0f * cnorm_max : 0.f, 0.f);      // (the plan's allowance for the fp8 sweep's score error: the plan must have been built with it)
}
