/*
 * wast3d_hip.h — C ABI of libwast3d_hip.so: the MI355X (gfx950) implementation of the WaSt3D
 * per-iteration style-loop hot path.
 *
 * Conventions (all entry points):
 *   - plain pointers are DEVICE pointers unless the name ends in _host;
 *   - every call takes the HIP stream it must be enqueued on (`stream`, a hipStream_t passed as void*);
 *   - return value: 0 (W3D_OK) or a negative W3D_ERR_* code; nothing throws across this boundary;
 *   - the caller owns all memory, including scratch (sizes come from the *_bytes() helpers);
 *   - no global state: calls on different streams are independent.
 *
 * Each declaration cites the reference interface it replaces (paths relative to the reference repo;
 * RAST = submodules/diff-gaussian-rasterization, NB11 = notebooks/11.fit_cluster_to_the_full_scene.ipynb,
 * raw .ipynb JSON line numbers).
 */
#ifndef WAST3D_HIP_H
#define WAST3D_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define W3D_OK 0
#define W3D_ERR_ARG (-1)         /* bad argument (null pointer, unsupported size/dtype combination) */
#define W3D_ERR_LAUNCH (-2)      /* HIP launch / runtime error (hipGetLastError != success) */
#define W3D_ERR_WORKSPACE (-3)   /* scratch buffer too small */
#define W3D_ERR_UNSUPPORTED (-4) /* valid request this build has no kernel for */

#define W3D_DTYPE_F32 0
#define W3D_DTYPE_BF16 1
#define W3D_DTYPE_F16 2
#define W3D_DTYPE_FP8_E4M3 3   /* OCP e4m3fn bytes + one E8M0 scale per row (w3d_quantize_rows_fp8 / w3d_knn_fp8) */

typedef void *w3d_stream_t;

const char *w3d_strerror(int code);
int w3d_abi_version(void);
/* HIP's description of the error behind the calling thread's most recent W3D_ERR_LAUNCH ("" if none). */
const char *w3d_last_hip_error_string(void);
/* Entry points read HIP's sticky per-thread error state after their launches.  A host program that shares the thread with other HIP users (PyTorch)
 * calls this before a batch of w3d_* calls so that an error left behind by someone else's runtime call is not charged to them. */
void w3d_clear_hip_error(void);

/* ------------------------------------------------------------------------------------------------
 * a1+a2  brute-force kNN: fused pairwise squared distance + running top-k (never materialises N x M).
 * Replaces torch.cdist + torch.topk(largest=False) (NB11:1055-1056,1062) and torch.cdist + torch.sort
 * (NB11:1016-1019; aux_optimize_cluster_D_W_distance4.py:75-84,272-276) for general feature width d.
 *   X [N,d], Y [M,d] row-major, dtype W3D_DTYPE_BF16 / W3D_DTYPE_F16 (MFMA path, d in {64,128,256}) or
 *   W3D_DTYPE_F32 (VALU path, d <= 16).  d2 = |x|^2 + |y|^2 - 2 x.y (the matmul form ATen uses), fp32.
 *   idx  [N,k] int32 : the k nearest rows of Y, ascending distance (ties: lower index first)
 *   dist2[N,k] f32   : their squared distances, clamped at 0
 *   prev_idx [N,k] or NULL : any k candidate rows per content row, typically the previous iteration's result of an
 *       optimisation loop (must not alias idx).  Used only to seed the running k-th-best bound (their current
 *       distances are recomputed, so ANY content is safe); the result is identical with or without it, the
 *       selection work is ~8x smaller with a good seed.  BF16 path only.
 *   MFMA path limits: 1 <= k <= M; the k-entry lists of a wave's rows live in LDS, so k <= 24 (d = 256), 28 (d = 128), 32 (d = 64) for
 *       16-bit rows; N, M >= 1 (empty inputs are the caller's no-op); beyond that W3D_ERR_UNSUPPORTED / W3D_ERR_ARG, never a partial result.
 * ---------------------------------------------------------------------------------------------- */
size_t w3d_knn_workspace_bytes(int64_t N, int64_t M, int d, int k, int dtype);
int w3d_knn_match_fwd(const void *X, const void *Y, int64_t N, int64_t M, int d, int k, int dtype,
                      const int32_t *prev_idx, int32_t *idx, float *dist2, void *workspace,
                      size_t workspace_bytes, w3d_stream_t stream);
/* w3d_knn_match_fwd in two steps (bf16 path): `prepare` fills the workspace (row norms, warm-start bound if prev_idx != NULL),
 * `match_prepared` is the fused distance-GEMM + top-k kernel alone (warm != 0 iff prepare was given prev_idx).
 * xnorm (may be NULL): the caller's |x|^2 [N] f32 of the rows as stored (the featurisation pass produces it) — the norm pass over X is then skipped; the workspace
 *   keeps these norms in its FIRST N floats, and a caller that had them written there (xnorm == workspace) saves the copy as well.
 * style_unchanged != 0: Y holds the rows of the previous prepare on this workspace (a loop's constant style set): their norms are not recomputed. */
int w3d_knn_prepare(const void *X, const void *Y, int64_t N, int64_t M, int d, int k, int dtype, const int32_t *prev_idx, const float *xnorm, int style_unchanged,
                    const int32_t *row_order /*see w3d_knn_match_prepared; may be NULL*/, void *workspace, size_t workspace_bytes, w3d_stream_t stream);
/* Round 4: w3d_knn_prepare(prev_idx) for loops whose rows move little — a row whose motion since the previous sweep (motion2 [N] = |x_new - x_old|^2) is at most
 * motion_ratio x its previous k-th distance (prev_dist2 [N,k] = the previous call's dist2) takes the MOTION bound D + |dx| and its neighbours' rows are not fetched; every
 * other row gets the gathered bound.  Both are valid bounds: the result is exact.  motion_ratio = 0: identical to w3d_knn_prepare.  (16-bit rows of 128 / 256 features
 * and fp8 rows of 256; other shapes always gather.) */
int w3d_knn_prepare_hybrid(const void *X, const void *Y, int64_t N, int64_t M, int d, int k, int dtype, const int32_t *prev_idx, const float *prev_dist2, const float *motion2,
                           float motion_ratio, const float *xnorm, int style_unchanged, const int32_t *row_order, void *workspace, size_t workspace_bytes, w3d_stream_t stream);
/* row_order (match_prepared, knn_fp8, chunk_sweep / chunk_merge; may be NULL): a permutation [N] int32 = the ORDER in which the kernel takes the content rows
 * (position -> row).  Results are written at each row's own place, so the output is independent of it; it is a layout hint: rows with similar features
 * placed in one 32-row MFMA block share the style tiles their candidates sit in, and the selection's slow path runs in fewer 32x32 score blocks. */
int w3d_knn_match_prepared(const void *X, const void *Y, int64_t N, int64_t M, int d, int k, int dtype, int warm,
                           int32_t *idx, float *dist2, const int32_t *row_order, void *workspace, size_t workspace_bytes, w3d_stream_t stream);
/* PRUNED sweep (exact; bf16 / f16 rows): between w3d_knn_prepare* and w3d_knn_match_prepared(warm = 3).  Every 64-row style tile t is described by a centre
 * (centres [ceil(M/64)][d] in the feature format — any point, e.g. the tile's mean row), cnorm[t] = |c_t|^2 and cradius[t] >= max |y - c_t| over the tile's rows.  By the
 * triangle inequality no row of tile t can reach the bound of content row x when |x|^2/2 - max(0, |x - c_t| - r_t)^2/2 stays below it; the plan marks, per block of 512
 * content rows (in row_order), the tiles ANY of its rows may need, and the sweep takes only those.  Result identical to the unpruned sweep.
 * w3d_knn_prune_counts_offset: where in the workspace the plan's per-block tile counts lie ([ceil(N/512)] int32; statistics). */
int w3d_knn_prune_plan(const void *X, int64_t N, int64_t M, int d, int k, int dtype, const int32_t *row_order, const void *centres, const float *cnorm,
                       const float *cradius, float cnorm_max /* >= max_t cnorm[t]: scales the margin for the sweep's rounding */, void *workspace, size_t workspace_bytes,
                       w3d_stream_t stream);
size_t w3d_knn_prune_counts_offset(int64_t N, int64_t M, int d, int k);
/* A caller-given tile ORDER as the plan of the next sweep (follow with w3d_knn_match_prepared(warm = 3) after a w3d_knn_prepare* on the same workspace): every block of
 * 512 content rows takes ALL tiles in the order tile_order [ceil(N/512)][ceil(M/64)] int32 names them (each row a permutation of the tile ids); cold != 0 sets every row's
 * bound to -inf.  For a loop's FIRST sweep: with the tiles nearest to each block first the running k-th best collapses within a few tiles and the candidate storm of a
 * cold sweep over tree-ordered style rows (194 ms at C3) is gone; the result is the cold sweep's, bit for bit. */
int w3d_knn_plan_order(int64_t N, int64_t M, int d, int k, const int32_t *tile_order, int cold, void *workspace, size_t workspace_bytes, w3d_stream_t stream);
/* Round 4 — warm = 4: the pruned sweep with EMIT selection.  Under the gathered bound (w3d_knn_prepare with prev_idx) a row's bound already is its k-th distance to within
 * a few candidates, so the sweep keeps no k-lists: lanes whose score reaches their row's bound append (score, row | style index) to a per-wave queue that spills to the
 * workspace's candidate buffer (64 rows share a region of 8192 entries; a region that overflows sends its rows to the cold repair), and a second kernel selects the k best
 * per row.  Same result as warm = 3, bit for bit; use warm = 3 for loose bounds (the motion bound).
 *
 * CACHED LOWER BOUNDS of a loop (exact; Elkan's k-means bounds transplanted to kNN): state owned by the caller —
 *   lb  [ceil(N/512)][ceil(M/64)][512] uint16 = f16 bits, per (512-row block POSITION of the sweep order, style tile, row of the block): the row's exact minimum distance
 *       to the tile's rows when the block last swept the tile + the row's cumulative motion at that time, rounded down; 0xFC00 (-inf) = unknown.
 *   cum [N] f32 by ROW: cumulative motion of the row since the last reset.
 * Per iteration: w3d_knn_bounds_advance(motion2 [N] = |x_new - x_old|^2 of every row since the previous sweep — w3d_descriptor_fwd_slots_x4 / _window_x4 report it, N, cum)
 * -> w3d_knn_prepare(prev_idx) -> w3d_knn_prune_plan_lb(..., lb, cum) -> w3d_knn_match_prepared_lb(warm = 4, ..., lb, cum, cnorm_max).  The plan drops a tile the ball
 * test kept when lb - cum > D_x for all 512 rows of the block (triangle inequality: no row of the tile can have come within the row's bound); the sweep rewrites the
 * bounds of every pair it sweeps.  RESET (fill lb with 0xFC00, cum with 0) whenever row_order, the style rows or N change: the bounds are keyed by position.
 * C3 loop: the ball test keeps 0.34 of the (row block, tile) pairs, both tests 0.11; 0.09 hold a true neighbour (profiles/r04a_cache_probe.log). */
int w3d_knn_bounds_advance(const float *motion2, int64_t N, float *cum, w3d_stream_t stream);
/* The cache under a NEW row order WITHOUT a reset (a reset costs one sweep of everything the ball test keeps: 20 instead of 11.5 ms at iteration 258 of the C3 loop):
 * lb_dst (a second array of lb's size) = the bounds re-keyed to the new positions — src_pos [ceil(N/512) * 512] int32: the OLD position of the row that sits at each new
 * position (positions >= N: that of the last row); new_order: the new row order (NULL = identity).  summ / dref (both or neither): the old order's block summaries and the
 * rows' references — per (row, tile) the better of the row's own recorded bound and summary + reference (valid for each of the 64 rows the summary was formed over; what
 * covers tiles the old block never swept).  The bounds come out rebased by the rows' cum.  Afterwards: cum = 0, summ = -inf, and the next plan is a steady = 2 one (a steady
 * plan that also moves the rows' references).  Exact: test_knn_cached_bounds_follow_the_rows_into_a_new_order. */
int w3d_knn_bounds_reorder(const uint16_t *lb_src, uint16_t *lb_dst, const float *summ, const float *dref, const float *cum, const int32_t *src_pos,
                           const int32_t *new_order, int64_t N, int64_t M, w3d_stream_t stream);
/* BLOCK SUMMARIES (optional third array of the cache): summ [ceil(N/512)][ceil(M/64)][8] f32 = per 64 rows of the block the smallest bound any of them holds for the tile
 * (-inf = unknown: fill with it at a reset).  With it the plan has two forms: steady = 0 — the ball test + per-row bounds as above, and it also raises the summaries to the fresh BALL bounds (use
 * it for the first plan after every reset and every few plans after: the ball bounds of tiles never swept age with the rows' motion); steady = 1 — every other plan: no content rows, no centres, no matrix cores — a tile whose eight summaries exceed their
 * rows' largest cum + D is out, the survivors' per-row bounds are tested (and their minimum becomes the new summary), and a survivor that was never swept is swept once.  The plan sets the
 * summaries of the tiles it KEEPS to -inf; the next plan recomputes them from the bounds the sweep wrote.  The sweep does not touch the summaries. */
int w3d_knn_prune_plan_lb(const void *X, int64_t N, int64_t M, int d, int k, int dtype, const int32_t *row_order, const void *centres, const float *cnorm,
                          const float *cradius, float cnorm_max, const uint16_t *lb /*may be NULL with cum: the plain plan*/, const float *cum, float *summ /*may be NULL*/,
                          float *dref /*with summ: [N] f32 by row, the rows' cum + D at the last steady = 0 plan; the summaries are kept relative to it*/, int steady,
                          void *workspace, size_t workspace_bytes, w3d_stream_t stream);
/* (ABI 5) w3d_knn_prune_plan_lb with an explicit allowance for the sweep's score error: score_err2 = 2E relative to |x|^2 + 2 max |y|^2 (<= 0: the 16-bit sweeps' 4e-6).
 * Plans for fp8 sweeps (rows = the DEQUANTISED rows as bf16) MUST pass W3D_KNN_SCORE_ERR2_FP8: the scaled fp8 MFMA's scores are ~5x less precise than an f32 dot product
 * of the same rows (measured error up to 1.6e-5 of (|x|^2 + |y|^2) / 2), the fp8 sweeps record their lower bounds with this allowance, and a tile a plan drops is never
 * revisited — a plan built by w3d_knn_prune_plan / _lb (the 16-bit allowance) is under-margined for them and the result silently stops being exact. */
#define W3D_KNN_SCORE_ERR2_FP8 4e-5f
int w3d_knn_prune_plan_err(const void *X, int64_t N, int64_t M, int d, int k, int dtype, const int32_t *row_order, const void *centres, const float *cnorm,
                           const float *cradius, float cnorm_max, float score_err2, const uint16_t *lb, const float *cum, float *summ, float *dref, int steady,
                           void *workspace, size_t workspace_bytes, w3d_stream_t stream);
/* STRAGGLER PAIRS (round 6; exact): at C3 half of the (row block, tile) pairs a steady plan keeps are kept for at most 16 of the block's 512 rows (a quarter for ONE).
 * With bits 8-15 of `steady` = T (1 ... 255; lb and summ required) a kept tile that at most T rows of a WHOLE block still need leaves the block's list; the rows' positions
 * go to per-tile buckets in the workspace, and w3d_knn_match_prepared_lb(warm = 4 | 0x100) computes exactly those (row, tile) pairs behind the sweep on the sweep's own MFMA
 * chain (same scores bit for bit, candidates into the same regions, the rows' cached bounds for the tile refreshed).  A plan with T > 0 is consumed by exactly ONE such
 * sweep; a sweep without 0x100 (or warm = 3) on it would miss candidates.  Plans with steady = 0 accept the bits (and leave the buckets empty). */
size_t w3d_knn_straggler_counts_offset(int64_t N, int64_t M, int d, int k);   /* byte offset in the workspace of the buckets' counts [ceil(M/64)] uint32 (statistics) */
int w3d_knn_straggler_capacity(int64_t N, int64_t M);                          /* positions a tile's bucket holds (a count above it: the overflow stayed with its blocks) */
int w3d_knn_match_prepared_lb(const void *X, const void *Y, int64_t N, int64_t M, int d, int k, int dtype, int warm /*4 when lb != NULL; | 0x100: straggler pairs*/, int32_t *idx, float *dist2,
                              const int32_t *row_order, uint16_t *lb, const float *cum, float cnorm_max /*the plan's*/, void *workspace, size_t workspace_bytes,
                              w3d_stream_t stream);
/* fp8 distance-GEMM (BASELINE config C5: fp16 features, e4m3 operands with a per-row scale).  w3d_quantize_rows_fp8: F [R,d] (f32 / bf16 /
 * f16, d % 4 == 0) -> q [R,d] e4m3 bytes + scale_e8m0 [R] int32 (E8M0 exponent byte: the row's values are q * 2^(e-127); the smallest
 * power of two that brings the row into e4m3's range, round-to-nearest-even).  w3d_knn_fp8 = w3d_knn_match_fwd / prepare / prepare_motion /
 * match_prepared on such rows (d in {128,256}) through v_mfma_scale_f32_32x32x64_f8f6f4, which applies the scales in hardware at twice the
 * bf16 MFMA rate; the result is the exact kNN of the DEQUANTISED rows (dist2 = |x^|^2 + |y^|^2 - 2 x^.y^).
 * phases: 1 = prepare (norms; + bound from prev_idx, or from prev_dist2/motion2), 2 = main kernel on a prepared workspace (warm != 0 iff a
 * bound was prepared), 3 = both.  Workspace: w3d_knn_workspace_bytes(..., W3D_DTYPE_FP8_E4M3). */
int w3d_quantize_rows_fp8(const void *F, int dtype, int64_t R, int d, void *q, int32_t *scale_e8m0,
                          float *motion2 /*NULL, or [R]: |x^_new - x^_old|^2 of the dequantised rows, q/scale holding the old values on entry*/,
                          w3d_stream_t stream);
int w3d_knn_fp8(const void *qX, const int32_t *xscale, const void *qY, const int32_t *yscale, int64_t N, int64_t M, int d, int k,
                int phases, int warm, const int32_t *prev_idx, const float *prev_dist2, const float *motion2, int32_t *idx,
                float *dist2, const int32_t *row_order, void *workspace, size_t workspace_bytes, w3d_stream_t stream);
/* (ABI 5, round 5) The main kernel of w3d_knn_fp8 (phase 2) with the EMIT selection and the cached lower bounds of w3d_knn_match_prepared_lb: warm = 4 — a pruned sweep
 * (plan in the workspace: w3d_knn_prune_plan_err(score_err2 = W3D_KNN_SCORE_ERR2_FP8) on the DEQUANTISED rows, after w3d_knn_fp8(phases = 1, prev_idx); NOT w3d_knn_prune_plan /
 * _lb, whose allowance is the 16-bit sweeps') that keeps no lists; lb / cum both NULL: no bounds
 * are recorded.  cnorm_max = the plan's.  Same result as w3d_knn_fp8(phases = 2, warm = 3): the exact kNN of the dequantised rows. */
int w3d_knn_fp8_lb(const void *qX, const int32_t *xscale, const void *qY, const int32_t *yscale, int64_t N, int64_t M, int d, int k, int warm /*4*/,
                   int32_t *idx, float *dist2, const int32_t *row_order, uint16_t *lb, const float *cum, float cnorm_max, void *workspace,
                   size_t workspace_bytes, w3d_stream_t stream);
/* Number of launches of the fused distance-GEMM + top-k kernel one w3d_knn_match_prepared call issues (1, or 2 when the last,
 * partially filled round of row blocks is launched separately with the style range split): lets a profiler-side per-launch average
 * be compared with a timed call. */
int w3d_knn_mfma_launches(int64_t N, int64_t M);
/* Warm-started calls (prev_idx, or prev_dist2 + motion2) seed the selection with a bound computed in ordinary f32 arithmetic, minus a slack of a few
 * sigma of the rounding difference to the matrix cores (a worst-case slack lets through several times more candidates when the neighbours are much
 * closer than the rows are long).  A row whose bound still came out too tight ends the sweep with fewer than k entries: it is detected and redone
 * with a cold bound inside the same call (the result stays exact); this reports how many rows the last warm call on `workspace` redid and the
 * capacity of that repair (count > capacity = rows left unrepaired: treat as an error).  Blocks on `stream`. */
int w3d_knn_mfma_repair_count(const void *workspace, int64_t N, int64_t M, int k, int32_t *count_host, int32_t *capacity_host, w3d_stream_t stream);
/* The same for loops, without blocking: enqueues a copy of {rows the last call redid, the largest such count of ANY call since the workspace was
 * zero-filled} into host_pair[2] (pinned host memory) and returns; read it after a later event on `stream`.  host_pair[1] > *capacity_host = some
 * sweep left rows with fewer than k neighbours (their unfilled slots hold the sentinel index 0x3ffffff / distance +inf, which w3d_knn_match_bwd and
 * w3d_knn_descriptor_bwd_fused_x4 turn into a NaN gradient instead of reading out of bounds).  The kNN workspace must be ZERO-FILLED once before its
 * first use for this word to be meaningful. */
int w3d_knn_mfma_repair_poll(const void *workspace, int64_t N, int64_t M, int k, int32_t *host_pair, int32_t *capacity_host, w3d_stream_t stream);
/* Diagnostic knobs of the fused kNN (process-global; tests and A/B scripts): mode bits 16 = no split tail round, 32 = 32x32x16 MFMA shape (default: 16x16x32),
 * 8 = 32-row tiles (in-kernel bits need a -DW3D_KNN_DEBUG build), 0x1000000 = round 4's emit selection instead of the round-5 emit kernel, 0x8000000 = sweeps
 * without a bound (warm = 0) on the list kernel instead of the cold emit sweep; slack_rel <= 0 = the built-in slack of the warm-start bounds.  The environment
 * (W3D_KNN_MODE, W3D_KNN_SLACK) is read once at library load, never in the launch path. */
int w3d_knn_mfma_set_tuning(int mode, float slack_rel);
/* CUs (0 ... 128; process-wide, one process per GPU) that the persistent launch of a pruned sweep leaves free for the kernels of other streams: its workgroups
 * hold every VGPR and all LDS of their CU for the whole launch.  0 (default): every CU.  A rank of a sharded scene sets 32: its image band's ~250 tile workgroups
 * are latency-bound and run next to the sweep on 32 CUs as fast as on 256. */
int w3d_knn_mfma_set_cu_reserve(int cus);
/* Diagnostics of the pruned sweep (process-global; scripts only): clk = device buffer of 8 int64, plan_clk of 4 int64 per row block, or NULL (off, the default).
 * Every row block of the following pruned sweeps writes {start, end (100-MHz wall clock), XCC id << 32 | HW_ID register, style tiles it swept, shader clock at start,
 * shader clock at end} at clk[8 * block];
 * every block of the following prune plans {block start, centre loop start, centre loop end, block end} at plan_clk[4 * block]. */
int w3d_knn_mfma_debug_clocks(int64_t *clk, int64_t *plan_clk);
/* (ABI 7) Two HIP events (hipEvent_t with timing, the caller's) that the following pruned emit sweeps record on their stream right before and right behind the emit kernel:
 * the dominant kernel's own duration inside the launch group (counter reset, emit, straggler pairs, select).  NULL, NULL: off (the default).  Process-global; bench.py. */
int w3d_knn_mfma_debug_events(void *before, void *after);
/* Cheaper seed for optimisation loops (no gather of Y rows): prev_dist2 [N,k] = the previous call's dist2 for the same rows,
 * motion2 [N] = |x_new - x_old|^2 (w3d_descriptor_fwd_x4_motion produces it while overwriting the feature rows).  Triangle
 * inequality: new k-th distance <= old k-th distance + |dx|.  Follow with w3d_knn_match_prepared(warm = 1); result is exact.
 * xnorm: NULL, or [N] = |x|^2 of the rows as stored (w3d_descriptor_fwd_x4_motion can produce it in the same pass): saves the norm pass over X.
 * (Round 2's "hot tiles first" order and hybrid half-gathered bound lost their A/B against the gathered bound + prune plan and were removed in ABI 3.) */
int w3d_knn_prepare_motion(const void *X, const void *Y, int64_t N, int64_t M, int d, int k, int dtype, const float *prev_dist2,
                           const float *motion2, const float *xnorm, const int32_t *row_order, void *workspace, size_t workspace_bytes, w3d_stream_t stream);
/* Chunked sweep for a style set that is SHARDED over GPUs (BASELINE config C4; no reference counterpart: the reference is single-GPU): the shards
 * arrive one by one (RCCL broadcast / all-gather over xGMI) while the distance-GEMM already runs on those that are there.  Per iteration:
 *   w3d_knn_prepare / w3d_knn_prepare_motion with Y = NULL      content-row norms + warm-start bound (no style rows touched)
 *   per chunk c, once rows [m0, m1) of Y are in place:
 *     w3d_knn_chunk_norms(Y, .., m0, m_valid, m1)               -0.5|y|^2 of [m0, m_valid); [m_valid, m1) is padding of the shard layout: never selected
 *     w3d_knn_chunk_sweep(X, Y, .., m0, m1, part = c, parts)    all content rows against style tiles [m0/64, ceil(m1/64)) (m0 % 64 == 0); partial k-lists
 *                                                               (raw scores, score order) into slot c of part_idx / part_val [N, parts, k]; parts <= 8
 *   w3d_knn_chunk_merge(..)                                     idx / dist2 exactly as w3d_knn_match_prepared on the whole set returns them (+ the cold repair)
 * 64-row-tile kernel only (k <= 20 at d = 256).  Workspace: w3d_knn_workspace_bytes(N, M, ...), the same one for all calls of an iteration. */
int w3d_knn_chunk_norms(const void *Y, int64_t N, int64_t M, int d, int k, int dtype, int64_t m0, int64_t m_valid, int64_t m1, void *workspace,
                        size_t workspace_bytes, w3d_stream_t stream);
int w3d_knn_chunk_sweep(const void *X, const void *Y, int64_t N, int64_t M, int d, int k, int dtype, int warm, int64_t m0, int64_t m1, int part, int parts,
                        int32_t *part_idx, float *part_val, const int32_t *row_order, void *workspace, size_t workspace_bytes, w3d_stream_t stream);
int w3d_knn_chunk_merge(const void *X, const void *Y, int64_t N, int64_t M, int d, int k, int dtype, int warm, int parts, const int32_t *part_idx,
                        const float *part_val, int32_t *idx, float *dist2, const int32_t *row_order, void *workspace, size_t workspace_bytes, w3d_stream_t stream);
/* loss_accum[0] += scale * sum(dist2)  — the head of get_loss_domain_adaptation (NB11:1021) given the
 * k selected entries per row (scale = weight / (N*M)). */
int w3d_knn_loss(const float *dist2, int64_t N, int k, float scale, float *loss_accum, w3d_stream_t stream);
/* grad_X[i,:] (=|+=) 2*scale * sum_j (x_i - y_idx[i,j])   (cdist backward with the mask constant,
 * NB11:1020: the mask is cast to float, so it is not differentiated).  grad_X is f32 [N,d]. */
int w3d_knn_match_bwd(const void *X, const void *Y, const int32_t *idx, int64_t N, int64_t M, int d, int k,
                      int dtype, float scale, int accumulate, float *grad_X, w3d_stream_t stream);
/* The same gradient for bf16 / f16 rows of d = 256 by the four-rows-per-wave kernel (10 of the k gathered rows in flight per row).  row_order (optional, [N]):
 * the order the rows are taken in — position -> row, e.g. the sweep's descriptor-tree order, in which neighbouring rows share their neighbours' cache lines;
 * grad_X rows land at their own places.  Other shapes: W3D_ERR_UNSUPPORTED (use w3d_knn_match_bwd). */
int w3d_knn_match_bwd_ordered(const void *X, const void *Y, const int32_t *idx, int64_t N, int64_t M, int d, int k, int dtype, float scale, int accumulate,
                              const int32_t *row_order, float *grad_X, w3d_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * a1+a2+a4 fused, reference regime (d = 3, M of a few hundred): get_loss_domain_adaptation
 * (NB11:1011-1023) forward AND backward in one call, batched over independent problems ("circles",
 * optimize_all_by_clusters NB11:987-993).  Exact reference semantics including ties:
 *   kth_i = k-th smallest distance of row i;  mask_ij = (D_ij <= kth_i);
 *   loss_s += weight * sum_{i in s, j} mask_ij D_ij^2 / (rows_per_seg * M_s);   gX_i += d loss / d x_i.
 *   X [S*rows_per_seg, d] f32; Y [sum M_s, d] f32; y_seg_offsets [S+1] int32 DEVICE array (prefix sums of M_s).
 *   loss_accum [S] f32 (+=), gX [S*rows_per_seg, d] f32 (+=), kth_out optional [rows] (distance, not squared).
 * ---------------------------------------------------------------------------------------------- */
int w3d_domain_loss_fwd_bwd(const float *X, int64_t rows_per_seg, int d, const float *Y,
                            const int32_t *y_seg_offsets /* device [S+1] */, int S, int64_t max_M, int k,
                            float weight, float *loss_accum, float *gX, float *kth_out, w3d_stream_t stream);

/* Script variant on full matrices (aux_optimize_cluster_D_W_distance4.py:73-84,254-283), fused and matrix-free:
 * loss3[0..2] += w * (loss_D_xyz, loss_D_rotation, loss_D_scaling); g_* += w * gradients.  kth_target[i] = k-th smallest
 * distance of target row i (k = 100 in the script), e.g. the kth_out of w3d_domain_loss_fwd_bwd(t_xyz, t_xyz).
 * (loss_D_to_shape of the same script is w3d_domain_loss_fwd_bwd with domain = shape_xyz / 5, k = 20.) */
int w3d_auxw_pairwise_fwd_bwd(const float *xyz, const float *rot, const float *scale, const float *t_xyz,
                              const float *t_rot, const float *t_scale, const float *kth_target, int N,
                              float w_xyz, float w_rot, float w_scale, float *loss3, float *g_xyz, float *g_rot,
                              float *g_scale, w3d_stream_t stream);
/* loss_domain_coverage (visualize_and_fit_multiple_clusters.ipynb cells 13-20): mean over domain points of the distance to
 * the nearest fitted point; nn_idx/nn_dist2 [M] come from w3d_knn_match_fwd(domain, points, k = 1).  g_points += (may be NULL). */
int w3d_coverage_loss_bwd(const float *domain, const float *points, const int32_t *nn_idx, const float *nn_dist2, int M,
                          int d, float weight, float *loss_accum, float *g_points, w3d_stream_t stream);

/* Sliced-Wasserstein head (BASELINE config C1; NOT in the reference — parity unpinned, known answer = 1-D optimal transport):
 * px [L,N] content projections, py_sorted [L,N] style projections sorted ascending per row;
 * loss_accum[0] += weight * mean((sort(px) - py_sorted)^2); g_px [L,N] = weight * d loss / d px (may be NULL). */
size_t w3d_sw_workspace_bytes(int L, int N);
int w3d_sw_loss_fwd_bwd(const float *px, const float *py_sorted, int L, int N, float weight, float *loss_accum,
                        float *g_px, void *workspace, size_t workspace_bytes, w3d_stream_t stream);
/* The same head hand-written end to end on FEATURE rows (ABI 6; clouds of at most w3d_sw_max_points() = 16 384 points, else W3D_ERR_UNSUPPORTED: use the form above):
 *   w3d_sw_project_sort      out_sorted [L,N] = every projection <F_i, proj_l> sorted ascending per l, out_perm [L,N] (may be NULL) the row each sorted value came from
 *                            (ties by row number = a stable argsort).  One workgroup per projection: projections, (key, index) pairs and the sort stay in LDS.
 *                            The constant style side: py_sorted = w3d_sw_project_sort(F_Y) once per style set.
 *   w3d_sw_features_fwd_bwd  loss_accum[0] += weight * mean_{l,r}((sort_r <FX, proj_l> - py_sorted[l,r])^2);  g_FX [N,d] = weight * d loss / d FX (written, not
 *                            added; may be NULL).  Two launches (project + sort + head per projection; dL/dp x proj with the loss parts folded in a fixed order:
 *                            bit-reproducible).  F, proj f32 row-major; workspace >= w3d_sw_features_workspace_bytes(L, N). */
int w3d_sw_max_points(void);
int w3d_sw_project_sort(const float *F, const float *proj, int N, int d, int L, float *out_sorted, int32_t *out_perm,
                        w3d_stream_t stream);
size_t w3d_sw_features_workspace_bytes(int L, int N);
int w3d_sw_features_fwd_bwd(const float *FX, const float *proj, const float *py_sorted, int N, int d, int L, float weight,
                            float *loss_accum, float *g_FX, void *workspace, size_t workspace_bytes, w3d_stream_t stream);

/* a5  exact earth mover's distance between two uniform clouds of EQUAL size n (aux_optimize_cluster_D_W_distance.py:259-270:
 * M = ot.dist(p, q) [squared Euclidean]; ot.emd2(1/n, 1/n, M)).  With uniform weights the optimal plan is a permutation, so the
 * value is min_sigma (1/n) sum_i M[i, sigma(i)], solved exactly (shortest augmenting paths, f64 potentials), one wave per problem.
 * POT itself is absent from the reference tree and unpinned: parity is anchored on scipy.optimize.linear_sum_assignment (oracle).
 *   matrix form : M [B, n, ld] f32 (ld >= n); loss[b] += weight * emd2; g_M [same layout] += weight * plan (1/n at matched
 *                 entries, POT's autograd convention); match [B, n] = sigma (may be NULL); n <= 1024.
 *   points form : P, Q [B, n, d] f32 (d <= 8), cost formed on the fly; g_P += weight * 2 (p_i - q_sigma(i)) / n, g_Q the negative. */
int w3d_emd2_matrix_fwd_bwd(const float *M, int B, int n, int64_t ld, float weight, float *loss, float *g_M, int32_t *match,
                            w3d_stream_t stream);
int w3d_emd2_points_fwd_bwd(const float *P, const float *Q, int B, int n, int d, float weight, float *loss, float *g_P,
                            float *g_Q, int32_t *match, w3d_stream_t stream);
/* (ABI 6) ot.dist(x1, x2) itself (aux_optimize_cluster_D_W_distance.py:268, default metric 'sqeuclidean'): out [n1,n2] = max(|a_i|^2 + |b_j|^2 - 2 a_i.b_j, 0) — POT's
 * published euclidean_distances(squared = True) — and its backward: g_x1 [n1,d] = sum_j g_out_ij 2 (a_i - b_j), g_x2 [n2,d] = sum_i g_out_ij 2 (b_j - a_i) (either may be
 * NULL; written, not added; no atomics).  x1, x2 f32 row-major, any d. */
int w3d_sqdist_matrix_fwd(const float *x1, const float *x2, int n1, int n2, int d, float *out, w3d_stream_t stream);
int w3d_sqdist_matrix_bwd(const float *g_out, const float *x1, const float *x2, int n1, int n2, int d, float *g_x1, float *g_x2,
                          w3d_stream_t stream);
/* a5, second head: the dual of entropic OT as notebooks/test_EMD_local_stats.ipynb maximises it (raw lines 208 / 994: ot.stochastic.loss_dual_entropic(u, v, xs, xt, reg)
 * in an Adam loop over the potentials, then ot.stochastic.plan_dual_entropic).  POT's published definition (third party, unpinned, absent from the image):
 *   M_ij = |xs_i - xt_j|^2,  H_ij = exp((u_i + v_j - M_ij) / reg),  L = u.ws + v.wt - reg * sum_ij ws_i H_ij wt_j,  plan_ij = ws_i H_ij wt_j   (ws, wt NULL = uniform).
 * loss[0] += weight * L;  g_u [ns], g_v [nt], and (optional) g_xs [ns,d], g_xt [nt,d] receive the UNWEIGHTED gradients (=, not +=).  d <= 8; sums in f64;
 * scratch = 8 bytes of device memory.  No atomics on the gradients (bit-reproducible). */
int w3d_dual_entropic_fwd_bwd(const float *u, const float *v, const float *xs, const float *xt, const float *ws, const float *wt, int ns, int nt, int d, float reg,
                              float weight, float *loss, float *g_u, float *g_v, float *g_xs, float *g_xt, void *scratch, w3d_stream_t stream);
int w3d_dual_entropic_plan(const float *u, const float *v, const float *xs, const float *xt, const float *ws, const float *wt, int ns, int nt, int d, float reg,
                           float *plan /*[ns,nt]*/, w3d_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Setup stage around the loop, point clouds with d = 3 (4 for the list builder):
 * w3d_knn_lists   = torch.cdist + torch.topk(largest=False, sorted) with large K (NB11:1055-1063: K = 2000 / 100 on the style
 *                   patch; NB11:154-169: the distance to the 30th neighbour).  idx [N,K] int32 ascending by (distance, index),
 *                   dist [N,K] f32 (cdist's value, may be NULL).  self_first != 0 (requires X == Y): row i's own point leads its
 *                   list whatever its rounding (column 0 = the root).  K <= 8192, K <= M.  Exact for any order of
 *                   the points; chunk-coherent orders (space-filling curve) skip far chunks by their bounding boxes.
 * w3d_radius_count= (cdist(X, Y[::y_stride]) <= radius).sum(1)      (NB11:119-152 sample_points_by_density; M = strided count)
 * w3d_dist_row    = cdist(q[None], Y)[0];  w3d_rowmax_dist = cdist(X, Y).max(1)   (NB11:731-785 get_cluster_radius, sample_circles)
 * ---------------------------------------------------------------------------------------------- */
size_t w3d_knn_lists_workspace_bytes(int64_t M, int d);
int w3d_knn_lists(const float *X, const float *Y, int64_t N, int64_t M, int d, int K, int self_first, int32_t *idx, float *dist,
                  void *workspace, size_t workspace_bytes, w3d_stream_t stream);
int w3d_radius_count(const float *X, const float *Y, int64_t N, int64_t M, int d, int64_t y_stride, float radius, int32_t *counts,
                     w3d_stream_t stream);
int w3d_dist_row(const float *q, const float *Y, int64_t M, int d, float *out, w3d_stream_t stream);
int w3d_rowmax_dist(const float *X, const float *Y, int64_t N, int64_t M, int d, float *out, w3d_stream_t stream);

/* Domain partitioning: one Lloyd iteration of the KMeans that prepare_scene runs through sklearn on the CPU
 * (notebooks/11.fit_cluster_to_the_full_scene.ipynb raw lines 171-198).
 * w3d_kmeans_assign: labels[i] = argmin_c (|c|^2 - 2 x_i.c) (sklearn's dense-Lloyd ranking, first minimum on ties), X [N,d] f32, d <= 16,
 *   centres [C,d] f32 (must fit the LDS: C*(d+1)*4 <= 128 KB).  Optional outputs (NULL = skip):
 *   sums [C,d] u64 / counts [C] u32 (+=; caller zeroes): per-cluster sums of rint(x * fx_scale) as two's-complement 64-bit integers — integer
 *     atomics are associative, so the new centres do not depend on scheduling; choose fx_scale with max|x| * N * fx_scale < 2^62;
 *   n_changed (+=): number of labels that differ from the previous content of `labels`;  inertia (+=): sum |x_i - c_label|^2 in f64.
 * w3d_kmeans_update: new_centres = sums / (fx_scale * counts) (clusters with count 0 keep the old centre), shift2 += sum |new - old|^2. */
int w3d_kmeans_assign(const float *X, int64_t N, int d, const float *centres, int C, double fx_scale, int32_t *labels, uint64_t *sums,
                      uint32_t *counts, uint32_t *n_changed, double *inertia, w3d_stream_t stream);
/* k-means++ seeding steps (sklearn's greedy _kmeans_plusplus): pots[t] = sum_i min(closest[i], |x_i - x_cand[t]|^2) for T <= 16 candidate
 * point ids (device int64), f64 sums; apply: closest[i] = min(closest[i], |x_i - x_centre_id[0]|^2) (closest = +inf before the first centre). */
int w3d_kmeanspp_pots(const float *X, int64_t N, int d, const int64_t *cand, int T, const float *closest, double *pots,
                      w3d_stream_t stream);
int w3d_kmeanspp_apply(const float *X, int64_t N, int d, const int64_t *centre_id, float *closest, w3d_stream_t stream);
int w3d_kmeans_update(const uint64_t *sums, const uint32_t *counts, int C, int d, double fx_scale, const float *old_centres,
                      float *new_centres, double *shift2, w3d_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * a3  get_descriptors (NB11:994-1009): F[r,j] = | X[idx[r,j+1]] - X[idx[r,0]] |, X [n,3] f32,
 * idx [R,K] int32.  out_dtype F32 or BF16 (feature rows feed the kNN).
 * Backward = scatter-add of index_put(accumulate=True); gX is accumulated into (+=).
 * ---------------------------------------------------------------------------------------------- */
int w3d_descriptor_fwd(const float *X, const int32_t *idx, int64_t R, int K, int out_dtype, void *F,
                       w3d_stream_t stream);
int w3d_descriptor_bwd(const float *X, const int32_t *idx, int64_t R, int K, const float *gF, float *gX,
                       w3d_stream_t stream);
/* get_descriptors_vec (notebooks/visualize_and_fit_cluster.ipynb cell 26, raw lines 771-808): V [R,4] f32 is a 4-vector per list row; for the
 * index triples S = {012, 013, 023, 123}:  F[r, s*(K-1)+j] = | V[r,S_s] - (X[idx[r,j+1]] - X[idx[r,0]]) |,  F [R, 4*(K-1)] f32.
 * Backward accumulates (+=) into gX [n,3] and gV [R,4]. */
int w3d_descriptor_vec_fwd(const float *X, const float *V, const int32_t *idx, int64_t R, int K, float *F, w3d_stream_t stream);
int w3d_descriptor_vec_bwd(const float *X, const float *V, const int32_t *idx, int64_t R, int K, const float *gF, float *gX,
                           float *gV, w3d_stream_t stream);
/* Hot-loop variants on a float4-padded copy of the points (one 16-B load per gather instead of three dword loads that may straddle a
 * cache line): w3d_pack_xyz4 once per iteration, then the *_x4 twins of w3d_descriptor_fwd / w3d_descriptor_bwd_csr.  gX stays [n,3]. */
int w3d_pack_xyz4(const float *X, int64_t n, float *X4 /*[n,4], 16-B aligned*/, w3d_stream_t stream);
/* center (the *_x4 and window forms; may be NULL): a vector [K-1] f32 (16-B aligned) subtracted from every feature row BEFORE it is rounded to
 * 16 bits, typically the mean row of the style set, whose rows are stored with the same shift.  Pairwise Euclidean distances — all the match
 * uses — are unchanged by a common shift, but the 16-bit rounding error scales with the magnitude of the stored values: centred rows keep the
 * bf16 match within the fp32 reference's tolerance.  Backward passes are unaffected (d row / d X does not depend on the shift). */
int w3d_descriptor_fwd_x4(const float *X4, const int32_t *idx, int64_t R, int K, int out_dtype, void *F, const float *center, w3d_stream_t stream);
/* 16-bit rows updated in place; motion2[r] = |F_new[r] - F_old[r]|^2 (F must hold the previous iteration's rows); xnorm (may be NULL):
 * xnorm[r] = |F_new[r]|^2 of the ROUNDED row, for w3d_knn_prepare_motion */
int w3d_descriptor_fwd_x4_motion(const float *X4, const int32_t *idx, int64_t R, int K, int dtype /*BF16 or F16*/, void *F_inout, float *motion2, float *xnorm,
                                 const float *center, w3d_stream_t stream);
/* The featurisation with the block's window of positions staged in LDS (144 KB): neighbour fetches inside the window are LDS reads, the rest
 * global gathers (any list is handled); 16-bit rows, motion2 / xnorm optional (NULL).  X4 [n_points,4]. */
int w3d_descriptor_fwd_window_x4(const float *X4, int64_t n_points, const int32_t *idx, int64_t R, int K, int dtype /*BF16 or F16*/, void *F,
                                 float *motion2, float *xnorm, int window_points, const float *center, w3d_stream_t stream);
/* The same featurisation over the backward's slot plan (w3d_knn_descriptor_bwd_fused_slots: rows_per_block, max_set, nbr_slot [R][K-1], dest, dest_off): a row block
 * stages the positions of the set of points its rows name; every neighbour fetch is an LDS read at the entry's slot.  (K-1) % 4 == 0, K-1 <= 256. */
int w3d_descriptor_fwd_slots_x4(const float *X4, int64_t n_points, const int32_t *idx, int64_t R, int K, int dtype, void *F, float *motion2, float *xnorm,
                                const float *center, int rows_per_block, int max_set, const uint16_t *nbr_slot, const int32_t *dest, const int32_t *dest_off,
                                const int32_t *root_idx /*NULL, or [R] = idx[:,0] when NO entry has slot 0xffff: the index rows are then not read*/, w3d_stream_t stream);
int w3d_descriptor_bwd_csr_x4(const float *X4, const int32_t *idx, int64_t R, int K, const float *gF, const int32_t *row_ptr,
                              const int32_t *entries, int64_t n_points, float *rootsum_scratch, int accumulate, float *gX,
                              w3d_stream_t stream);
/* Row-order backward with an LDS accumulation window (scenes stored along a space-filling curve: the neighbours of consecutive rows
 * fall into a narrow index range).  Destinations inside a block's window are 32-bit fixed-point LDS adds (gfx950 retires integer LDS
 * atomics 30x faster than float ones), the rest global float atomics, so any list is handled; gX [n_points,3] += dL/dX.
 * max_fanin: an upper bound of the number of list entries (columns >= 1) naming any one point (w3d_list_max_fanin; the lists are
 * frozen, so once) — it sizes the fixed-point headroom: the caller must not pass less than the true fan-in (the 32-bit sums would wrap).
 * Resolution per contribution: max|gF| of the block's rows / 2^(30 - ceil(log2 max_fanin)).
 * window_points <= 0 selects the default window (12288 points = 144 KB of LDS, leaving room for a co-resident block of another kernel;
 * at most 13312). */
int w3d_list_max_fanin(const int32_t *idx, int64_t R, int K, int64_t n_points, int32_t *counts_scratch /*[n_points]*/,
                       int32_t *out_max /*device*/, w3d_stream_t stream);
int w3d_absmax_16(const void *v, int dtype /*BF16 or F16*/, int64_t n, float *out /*device*/, w3d_stream_t stream);
int w3d_descriptor_bwd_window_x4(const float *X4, const int32_t *idx, int64_t R, int K, const float *gF, int64_t n_points,
                                 int window_points, int max_fanin, float *gX, w3d_stream_t stream);
/* The same pass fused with w3d_knn_match_bwd for descriptor features (F_X = descriptors(X4, desc_idx), d = K-1, d % 8 == 0):
 * dL/dF_X[r,:] = 2*scale*(k*F_X[r,:] - sum_j F_Y[nn_idx[r,j],:]) is formed in registers and scattered at once;
 * gF_out (may be NULL) receives it as [R,d] f32.  fy_absmax = max |F_Y| (w3d_absmax_16; F_Y is constant) bounds the
 * contributions together with the block's own max |F_X|.  F_X, F_Y: 16-bit rows of `dtype`. */
int w3d_knn_descriptor_bwd_fused_x4(const float *X4, const int32_t *desc_idx, int64_t R, int K, int dtype /*BF16 or F16*/, const void *FX_bf16,
                                    const void *FY_bf16, int64_t M, float fy_absmax, const int32_t *nn_idx, int k, float scale,
                                    int64_t n_points, int window_points, int max_fanin, float *gF_out, float *gX,
                                    const float *nn_dist2 /*NULL, or [R,k]: then loss_accum[0] += scale * sum(nn_dist2), the w3d_knn_loss head*/,
                                    float *loss_accum, w3d_stream_t stream);
/* The fused pass over a SLOT PLAN of frozen lists: the rows are cut into blocks of rows_per_block; block b's window is not an index range but the set of points
 * its rows name, dest[dest_off[b] .. dest_off[b+1]) (at most max_set <= 13312 of them: 12 B of LDS each), and nbr_slot [R][K-1] (8-B aligned rows, (K-1) % 8 == 0,
 * K-1 <= 256) holds each entry's place in its block's set, 0xffff = not in the set (a global float atomic).  For lists that are local in space but not along the
 * storage curve — the exact nearest neighbours of a Z-ordered scene leave 16 % of their entries outside a 12 288-point index window — every contribution then stays in
 * LDS.  The plan is host-side work on the frozen lists (wast3d_amd.ops.SlotPlan: per block, the sorted distinct points and each entry's rank among them). */
int w3d_knn_descriptor_bwd_fused_slots(const float *X4, const int32_t *desc_idx, int64_t R, int K, int dtype, const void *FX_bf16, const void *FY_bf16, int64_t M,
                                       float fy_absmax, const int32_t *nn_idx, int k, float scale, int64_t n_points, int max_fanin, float *gX,
                                       const float *nn_dist2, float *loss_accum, int rows_per_block, int max_set, const uint16_t *nbr_slot,
                                       const int32_t *dest, const int32_t *dest_off,
                                       const int32_t *root_idx /*NULL, or [R] = desc_idx[:,0] when NO entry has slot 0xffff and max_set <= 6826 (positions staged too):
                                                                  the index rows are then not read*/, w3d_stream_t stream);
/* Round 4 — the same pass with the NEIGHBOUR-SUM CACHE of a loop: sum_cache [R][K-1] f32 (16-B aligned) and sum_idx [R][k] int32 (k <= 64) are the caller's and live from
 * call to call; fill sum_idx with -1 before the first call and whenever the style rows change.  A row whose k neighbours are the same SET (exact comparison) as at the call
 * that filled its cache entry takes sum_j FY[nn_j] from the cache — one coalesced 1-KB read instead of k gathered rows; 75-96 % of the rows of the C3 loop — every other
 * row gathers and refreshes its entry.  Gradients equal w3d_knn_descriptor_bwd_fused_slots' up to the f32 addition order of the k rows. */
int w3d_knn_descriptor_bwd_fused_slots_cached(const float *X4, const int32_t *desc_idx, int64_t R, int K, int dtype, const void *FX_bf16, const void *FY_bf16, int64_t M,
                                              float fy_absmax, const int32_t *nn_idx, int k, float scale, int64_t n_points, int max_fanin, float *gX,
                                              const float *nn_dist2, float *loss_accum, int rows_per_block, int max_set, const uint16_t *nbr_slot,
                                              const int32_t *dest, const int32_t *dest_off, const int32_t *root_idx, float *sum_cache, int32_t *sum_idx,
                                              w3d_stream_t stream);
/* (Round 2's far-entry plan — staging of out-of-window contributions, w3d_far_plan / w3d_descriptor_far_gather — was superseded by the slot plan and removed in ABI 3.) */
/* loss_points_local (NB11:1076-1080) fused: loss_accum[0] += weight*mean((F - target)^2); gX += weight*dL/dX.
 * F is not materialised.  `seg_rows` > 0 splits rows into segments of that many rows, each writing
 * loss_accum[segment] and normalised by its own count (batched circles); 0 = one segment. */
int w3d_descriptor_mse_fwd_bwd(const float *X, const int32_t *idx, const float *target, int64_t R, int K,
                               int64_t seg_rows, float weight, float *loss_accum, float *gX,
                               w3d_stream_t stream);
/* Deterministic, atomic-free backward through a transposed (CSR) index built once per neighbour list:
 * w3d_csr_* replaces index_put(accumulate=True) when the lists are frozen (they are: NB11:1055-1063). */
size_t w3d_csr_workspace_bytes(int64_t n_points, int64_t R, int K);
int w3d_csr_build(const int32_t *idx, int64_t R, int K, int64_t n_points, int32_t *row_ptr /*[n+1]*/,
                  int32_t *entries /*[2*R*K]: (root point, gF slot) pairs in destination order*/, void *workspace, size_t workspace_bytes, w3d_stream_t stream);
int w3d_descriptor_bwd_csr(const float *X, const int32_t *idx, int64_t R, int K, const float *gF,
                           const int32_t *row_ptr, const int32_t *entries, int64_t n_points,
                           float *rootsum_scratch /*[R,3]*/, int accumulate, float *gX, w3d_stream_t stream);

/* loss_points_local for S problems that share one neighbour list / target / CSR (optimize_all_by_clusters, NB11:987-993: the
 * same style patch is fitted to every circle): X [S*n,3], idx/target/row_ptr/entries refer to LOCAL point indices 0..n-1.
 * loss_accum[s] += weight * mean((F_s - target)^2);  gX += weight * dL/dX.  Atomic-free on the gradient (bit-reproducible).
 * rootsum_scratch: [S*R,3] floats. */
int w3d_descriptor_mse_seg_csr(const float *X, const int32_t *idx, const float *target, int64_t R, int K,
                               int64_t n_points_per_seg, int S, const int32_t *row_ptr, const int32_t *entries,
                               float weight, float *loss_accum, float *gX, float *rootsum_scratch,
                               w3d_stream_t stream);


/* ------------------------------------------------------------------------------------------------
 * a9  fused multi-group Adam: torch.optim.Adam as configured at NB11:1067 (one group, eps 1e-8) and
 * scene/gaussian_model.py:154-163 (six groups, eps 1e-15, per-group lr).  Same operation order as
 * torch (lerp, addcmul, sqrt/bias2 + eps, addcdiv).  zero_grad bit 0: the gradient is zeroed in the
 * same pass (optimizer.zero_grad(), NB11:1089).  zero_grad bit 1 (round 5): the LEAN form of the kernel — the
 * same arithmetic bit for bit, one launch per group, 32 VGPRs, no LDS, non-temporal accesses: a wave of it fits on
 * a SIMD next to the persistent emit sweep's two waves, so the step can run on another stream UNDER the sweep.
 * zero_grad is a BIT FIELD (values 0..3; anything else: W3D_ERR_ARG — a caller that passed 2 for "true" before ABI 5 would otherwise silently get no zeroing).
 * All five arrays of a group must be 16-byte aligned (W3D_ERR_ARG otherwise).
 * ---------------------------------------------------------------------------------------------- */
#define W3D_ADAM_MAX_GROUPS 8
typedef struct {
    float *param;
    float *grad;
    float *exp_avg;
    float *exp_avg_sq;
    int64_t numel;
    double lr;   /* doubles: torch computes lr/(1-beta1^t), 1-beta1, 1-beta2 in double before rounding to f32 */
    float *grad2; /* NULL, or a second gradient buffer of the same shape: the step uses grad + grad2 (two producers running on different
                   * streams need no extra add pass); zeroed together with grad */
} W3dAdamGroup;
int w3d_adam_step(const W3dAdamGroup *groups_host, int n_groups, int step, double beta1, double beta2,
                  double eps, int zero_grad, w3d_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * a8  image-space losses on the render: tv_loss (utils/loss_utils.py:213-215) and MSE
 * (utils/loss_utils.py:23-24, train_st_depth_hotdog.py:218-231), forward + gradient in one pass.
 *   loss_accum[0] += weight * loss ;  grad (=|+=) weight * d loss / d img
 * ---------------------------------------------------------------------------------------------- */
int w3d_tv_loss_fwd_bwd(const float *img, int C, int H, int W, float weight, int accumulate,
                        float *loss_accum, float *grad, w3d_stream_t stream);
int w3d_mse_loss_fwd_bwd(const float *a, const float *b, int64_t n, float weight, int accumulate,
                         float *loss_accum, float *grad_a, w3d_stream_t stream);
/* ssim (utils/loss_utils.py:34-64; the D-SSIM term of train.py:93): 11x11 Gaussian window (sigma 1.5), zero padding, one window per [H,W] plane (groups = channel).
 * planes = product of the leading dimensions.  w3d_ssim_fwd: plane_sums [planes] += sum of each plane's SSIM map (zero it first; ssim = sum / element count — over
 * everything, or per batch item for size_average = False); dmaps [3][planes*H*W] (or NULL: value only) receives the three derivative maps the backward filters.
 * w3d_ssim_bwd: grad1 (=|+=) plane_w[plane] * d(plane sum)/d img1. */
int w3d_ssim_fwd(const float *img1, const float *img2, int planes, int H, int W, float *plane_sums, float *dmaps, w3d_stream_t stream);
int w3d_ssim_bwd(const float *img1, const float *img2, const float *dmaps, int planes, int H, int W, const float *plane_w, int accumulate, float *grad1,
                 w3d_stream_t stream);

/* a8  Gram style loss of one feature layer (utils/loss_utils.py:178-205 calc_style_loss): F [C, HW] f32 (a [1,C,H,W] feature map),
 * G = F F^T, loss = mean((G - A)^2) with A [C,C] the style image's Gram matrix (w3d_gram_matrix of its features).
 * loss_accum[0] += weight * loss; g_F [C,HW] += weight * dloss/dF (may be NULL).  f32 matrix cores, split-K with a fixed-order
 * reduction (bit-reproducible).  content_loss (:113-131) is w3d_mse_loss_fwd_bwd per layer. */
size_t w3d_gram_workspace_bytes(int C, int HW);
int w3d_gram_matrix(const float *F, int C, int HW, float *G, void *workspace, size_t workspace_bytes, w3d_stream_t stream);
int w3d_gram_style_loss_fwd_bwd(const float *F, const float *A, int C, int HW, float weight, float *loss_accum, float *g_F,
                                void *workspace, size_t workspace_bytes, w3d_stream_t stream);

/* a8 / SURVEY 8f row 4: the VGG-19 feature stack of the image-style leg (utils/loss_utils.py:66-109: torchvision vgg19.features[:29] on the render
 * resized to 112 x 112 with F.interpolate's default 'nearest'; taps at layers 0, 5, 10, 19, 28 — the reference appends the conv output tensor and
 * torchvision's inplace ReLU then rewrites it, so taps 0 / 5 / 10 / 19 hold relu(conv) and only tap 28 is a pre-activation).  The reference
 * runs cuDNN; these are the f32 building blocks, single image, NCHW, all tensors contiguous:
 *   w3d_conv3x3          out [Cout,H,W] = bias + conv(act(in [Cin,H,W]), weight [Cout,Cin,3,3]), pad 1, stride 1; act = ReLU if relu_input (activations
 *                        are stored pre-ReLU); bias may be NULL.  Backward-data of a layer = the same call on its flipped / transposed weights.
 *   w3d_relu             out [n] = max(z [n], 0): the tapped maps the reference sees post-ReLU.
 *   w3d_maxpool2         2x2 / stride 2 on even H, W.
 *   w3d_relu_pool_bwd    gz [C,H,W] (=|+=) gradient wrt the pre-activation z from g_act = gradient wrt relu(z) [C,H,W], or wrt relu(pool(z))
 *                        [C,H/2,W/2] if pooled (arg-max = first maximum in row-major order, as torch's max_pool2d backward).
 *   w3d_nearest_resize / _bwd   F.interpolate(mode='nearest') and its adjoint (g_in +=). */
int w3d_conv3x3(const float *in, const float *weight, const float *bias, int Cin, int Cout, int H, int W, int relu_input, float *out, w3d_stream_t stream);
int w3d_relu(const float *z, int64_t n, float *out, w3d_stream_t stream);
int w3d_maxpool2(const float *in, int C, int H, int W, float *out, w3d_stream_t stream);
int w3d_relu_pool_bwd(const float *z, const float *g_act, int C, int H, int W, int pooled, int accumulate, float *gz, w3d_stream_t stream);
int w3d_nearest_resize(const float *in, int C, int H, int W, int Ho, int Wo, float *out, w3d_stream_t stream);
int w3d_nearest_resize_bwd(const float *g_out, int C, int H, int W, int Ho, int Wo, float *g_in, w3d_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * a6/a7  differentiable Gaussian rasteriser (fork with depth output + per-pixel sampling offsets).
 * Replaces RasterizeGaussiansCUDA / RasterizeGaussiansBackwardCUDA / markVisible
 * (RAST/rasterize_points.h:18-70, RAST/ext.cpp:15-19) and CudaRasterizer::Rasterizer::{forward,backward}
 * (RAST/cuda_rasterizer/rasterizer.h).  All tensors f32 contiguous; "absent" optionals are NULL
 * (the reference passes the data pointer of a 0-element tensor, RAST/diff_gaussian_rasterization/__init__.py:213-223).
 * Matrices are the TRANSPOSED (row-vector) 4x4s the reference passes (scene/cameras.py:54-57).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int P;              /* number of Gaussians */
    int sh_degree;      /* active SH degree D (0..3) */
    int sh_coeffs;      /* M = coefficients per Gaussian in `shs` (e.g. 16), 0 if colors_precomp */
    int H, W;
    const float *background;      /* [3] */
    const float *means3D;         /* [P,3] */
    const float *shs;             /* [P,M,3] or NULL */
    const float *colors_precomp;  /* [P,3] or NULL */
    const float *opacities;       /* [P,1] */
    const float *scales;          /* [P,3] or NULL */
    const float *rotations;       /* [P,4] or NULL */
    const float *cov3D_precomp;   /* [P,6] or NULL */
    const float *viewmatrix;      /* [16] */
    const float *projmatrix;      /* [16] */
    const float *campos;          /* [3] */
    const float *sampling_offsets;/* [H,W,2] */
    float scale_modifier;
    float tan_fovx, tan_fovy;
    int prefiltered;
    const float *shs_rest;        /* NULL, or [P,M-1,3]: then `shs` is the DC coefficient alone [P,1,3] and the higher orders live here (the
                                   * reference's _features_dc / _features_rest parameter tensors, used without concatenating them) */
} W3dRasterIn;

/* Splat record: what one projected Gaussian contributes to the blend, and the unit that crosses GPUs when Gaussians are
 * sharded (all-gather of records, reduce-scatter of the equally sized gradient accumulators):
 *   [0] px [1] py [2] view depth [3] radius (int32 bits, 0 = culled) [4..6] conic xx,xy,yy [7] opacity [8..10] rgb [11] 0
 * The records of a forward are the first W3D_SPLAT_FLOATS*P floats of its geom_buffer. */
#define W3D_SPLAT_FLOATS 12

/* Scratch: geometry state [P], image state [H*W], binning state [R] — same three opaque buffers the
 * reference hands back to Python (RAST/rasterize_points.cu:27-33, __init__.py:104). */
size_t w3d_raster_geom_bytes(int P);
size_t w3d_raster_image_bytes(int H, int W);
size_t w3d_raster_binning_bytes(int64_t num_rendered);

/* Buffer-growing callback, the C form of std::function<char*(size_t)> (rasterizer.h:25-27):
 * must return a device pointer to at least `bytes` bytes that stays valid until backward has run. */
typedef void *(*w3d_alloc_fn)(void *user, size_t bytes);

/* Forward.  Blocks once on `stream` to read the number of tile instances (the reference's cudaMemcpy,
 * rasterizer_impl.cu:283), then asks `binning_alloc` for the binning buffer.
 * Outputs: out_color [3,H,W], out_depth [H,W], radii [P] int32; *num_rendered_host = R. */
int w3d_raster_forward(const W3dRasterIn *in, void *geom_buffer, void *image_buffer,
                       w3d_alloc_fn binning_alloc, void *alloc_user, float *out_color, float *out_depth,
                       int32_t *radii, int *num_rendered_host, w3d_stream_t stream);

typedef struct {
    const float *dL_dout_color; /* [3,H,W] */
    const float *dL_dout_depth; /* [H,W] */
} W3dRasterGradIn;
typedef struct {              /* all must be zero-initialised by the caller (the reference allocates zeros) */
    float *dL_dmeans2D;   /* [P,3], or NULL (ABI 6): not wanted */
    float *dL_dcolors;    /* [P,3], or NULL (ABI 6): not wanted */
    float *dL_dopacity;   /* [P,1] */
    float *dL_dmeans3D;   /* [P,3] */
    float *dL_dcov3D;     /* [P,6], or NULL (ABI 6) when scales / rotations are given: not wanted (the chain to dL_dscales / dL_drotations needs no stored copy) */
    float *dL_dsh;        /* [P,M,3] or NULL */
    float *dL_dscales;    /* [P,3] or NULL */
    float *dL_drotations; /* [P,4] or NULL */
    float *accum;         /* [P,12] scratch: packed per-Gaussian accumulators of the blend backward (colour 0-2, view depth 3,
                           * mean2D 4-5, conic 6-8, opacity 9); the view-depth slot is folded into dL_dmeans3D (backward.cu:395-397) */
    float *dL_dsh_rest;   /* with W3dRasterIn.shs_rest: [P,M-1,3], dL_dsh is then [P,1,3]; else NULL */
} W3dRasterGradOut;
int w3d_raster_backward(const W3dRasterIn *in, const int32_t *radii, const W3dRasterGradIn *gin,
                        W3dRasterGradOut *gout, const void *geom_buffer, const void *binning_buffer,
                        const void *image_buffer, int num_rendered, w3d_stream_t stream);
/* Round 4: the same backward ADDING to every per-Gaussian output (dL_dmeans2D ... dL_dsh_rest) instead of overwriting it — the gradients of several views of one
 * iteration (BASELINE C5: 4) accumulate in place.  gout->accum (the blend backward's scratch) must still be zero-filled by the caller before each call; Gaussians culled in
 * this view are not touched. */
int w3d_raster_backward_accumulate(const W3dRasterIn *in, const int32_t *radii, const W3dRasterGradIn *gin,
                                   W3dRasterGradOut *gout, const void *geom_buffer, const void *binning_buffer,
                                   const void *image_buffer, int num_rendered, w3d_stream_t stream);
/* (ABI 7) The two calls above for a loop's PERSISTENT buffers, without a fill per iteration: gout->accum must be all zero on entry (zero-fill it once, at allocation) and is all zero
 * again on return (the per-Gaussian kernel zeroes the rows it consumed; rows of culled Gaussians are never written); with accumulate = 0 the outputs dL_dopacity, dL_dscales and
 * dL_drotations are written for EVERY Gaussian (zeros for the culled ones) — the outputs a caller reads for all of them (the activation backward) — so they need no fill either.  The
 * other outputs are written for visible Gaussians only, as before (a loop whose optimiser zeroes the gradients it consumed keeps them clean).  Same results. */
int w3d_raster_backward_selfclean(const W3dRasterIn *in, const int32_t *radii, const W3dRasterGradIn *gin,
                                  W3dRasterGradOut *gout, const void *geom_buffer, const void *binning_buffer,
                                  const void *image_buffer, int num_rendered, int accumulate, w3d_stream_t stream);
/* ---- the same pipeline in four stages, for Gaussian-sharded multi-GPU runs (no counterpart in the reference, which is
 * single-GPU): A project local Gaussians -> splat records; B bin+sort+blend the tile rows [tile_row0, tile_row1) from ANY
 * set of records (e.g. all-gathered from every rank) — writes only those rows of out_color/out_depth; C blend backward of
 * the same rows into accum[P,12] (+=; reduce-scatter to the owners afterwards); D per-Gaussian chain on the owner.
 * w3d_raster_forward == A + B(all rows); w3d_raster_backward == C(all rows) + D. */
int w3d_raster_preprocess(const W3dRasterIn *in, void *geom_buffer, int32_t *radii, w3d_stream_t stream);
size_t w3d_raster_band_scratch_bytes(int P);
int w3d_raster_blend_forward(const float *splats, int P, int H, int W, int tile_row0, int tile_row1, const float *background,
                             const float *sampling_offsets, void *band_scratch, void *image_buffer,
                             w3d_alloc_fn binning_alloc, void *alloc_user, float *out_color, float *out_depth,
                             int *num_rendered_host, w3d_stream_t stream);
int w3d_raster_blend_backward(const float *splats, int P, int H, int W, int tile_row0, int tile_row1, const float *background,
                              const float *sampling_offsets, const void *image_buffer, const void *binning_buffer,
                              int num_rendered, const float *dL_dout_color, const float *dL_dout_depth, float *accum,
                              w3d_stream_t stream);
int w3d_raster_preprocess_backward(const W3dRasterIn *in, const int32_t *radii, const void *geom_buffer,
                                   W3dRasterGradOut *gout, w3d_stream_t stream);

/* GaussianRasterizer.markVisible (RAST/diff_gaussian_rasterization/__init__.py:192-201) */
int w3d_mark_visible(const float *means3D, const float *viewmatrix, const float *projmatrix, int P,
                     uint8_t *present, w3d_stream_t stream);

/* a10  parameter activations of scene/gaussian_model.py:95-115 fused fwd / bwd:
 * scales = exp(_scaling), rot = normalize(_rotation), opacity = sigmoid(_opacity). */
int w3d_activations_fwd(const float *scaling_raw, const float *rotation_raw, const float *opacity_raw, int P,
                        float *scales, float *rotations, float *opacities, w3d_stream_t stream);
int w3d_activations_bwd(const float *scaling_raw, const float *rotation_raw, const float *opacity_raw, int P,
                        const float *g_scales, const float *g_rotations, const float *g_opacities,
                        float *g_scaling_raw, float *g_rotation_raw, float *g_opacity_raw, w3d_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* WAST3D_HIP_H */
