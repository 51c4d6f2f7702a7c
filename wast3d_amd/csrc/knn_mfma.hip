// a1+a2  fused distance-GEMM + running top-k on the gfx950 matrix cores (bf16 / f16 / fp8-e4m3 + E8M0 row scales in, fp32 accumulate).
//
// Layout of the work ("flash-attention without the PV product"):
//   block  = 512 threads = 8 waves, owns 512 content rows of X; streams ALL style rows of Y once.
//   wave   = 64 X rows (RB = 2 blocks of 32), kept as MFMA B-operand fragments in VGPRs for the whole kernel.
//   Y tile = 64 (or 32) style rows x D, double-buffered in LDS by global_load_lds (XOR-swizzled 16-B chunks: conflict-free
//            ds_read_b128), shared by the 8 waves; one s_barrier per tile.
//   MFMA   = v_mfma_f32_32x32x16_bf16, operands swapped (A = Y rows, B = X^T) so that the X row sits on the
//            lane (col = lane & 31) and the 16 accumulator registers of a lane are 16 different candidates of
//            ONE row -> the running-threshold test is a per-lane v_max3 tree + one compare.
//   acc    = -0.5|y|^2 (C-in) + x.y   ("val", bigger = closer);  d2 = |x|^2 - 2 val.
// Selection: candidates that beat the row's current k-th best are appended to a per-wave LDS queue
// (scalar slot counter, no atomics); when a queue passes half (voted block-wide) the wave counting-sorts it by row and every lane
// merges its own row's entries into its k-entry list (LDS, lane-strided) and refreshes the threshold.  Optimisation loops seed the
// thresholds from the previous iteration (exactly: neighbours' current scores, or k-th distance + row motion).
// The hot loop is issue-slot- and power-sensitive: every scalar / vector instruction saved per k-step shows up in the time.
// Algorithmic work: 2*N*M*D flop; HBM bytes (N+M)*D*2 + N*k*8 (negligible) -> MFMA-bound.
#include "common.h"
#include <cstdlib>
#include <cmath>
#include <algorithm>
#include <type_traits>
#include <atomic>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Operand formats of the distance-GEMM (FMT template parameter):
//   0  bf16  v_mfma_f32_32x32x16_bf16 (or 16x16x32)                       2 B / element
//   1  f16   v_mfma_f32_32x32x16_f16  (or 16x16x32)                       2 B / element
//   2  fp8 e4m3 (OCP) with one power-of-two scale per ROW (E8M0, the MX block-scale format with all blocks of a row equal):
//      v_mfma_scale_f32_32x32x64_f8f6f4 — the only fp8 form that runs at twice the bf16 rate on gfx950; the hardware applies
//      2^(sx-127) * 2^(sy-127) to the products, so the accumulator still starts at -0.5|y|^2 and holds the ranking score.
//      Lane map (checked with exact integer data, scripts/micro/mfma_scale_layout.hip): lane l holds A[row l&31][k = 32(l>>5)+j],
//      B[k = 32(l>>5)+j][col l&31] in byte j of its 8 VGPRs; byte 0 of the scale VGPR is the scale of that lane's row / column.
constexpr int FMT_BF16 = 0, FMT_F16 = 1, FMT_FP8 = 2;
#ifndef W3D_EMIT_SH
#define W3D_EMIT_SH 1
#endif
template <int FMT> struct Fmt { static constexpr int EB = FMT == FMT_FP8 ? 1 : 2; /* bytes per element */ static constexpr int FB = FMT == FMT_FP8 ? 32 : 16; /* operand bytes per lane per MFMA */ };

template <int FMT>
__device__ __forceinline__ float decode16(uint16_t v) {
    if (FMT == FMT_F16) { _Float16 h; __builtin_memcpy(&h, &v, 2); return (float)h; }
    return __uint_as_float(((uint32_t)v) << 16);
}
// sum of squares of the 16 bytes a lane holds (8 x 16-bit or 16 x fp8, unscaled)
template <int FMT>
__device__ __forceinline__ float sumsq16(const uint4 v) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (FMT == FMT_FP8) {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                float f;
                if (b == 0) f = __builtin_amdgcn_cvt_f32_fp8((int)w[i], 0); else if (b == 1) f = __builtin_amdgcn_cvt_f32_fp8((int)w[i], 1);
                else if (b == 2) f = __builtin_amdgcn_cvt_f32_fp8((int)w[i], 2); else f = __builtin_amdgcn_cvt_f32_fp8((int)w[i], 3);
                s = fmaf(f, f, s);
            }
        } else {
            const float lo = decode16<FMT>((uint16_t)(w[i] & 0xffffu)), hi = decode16<FMT>((uint16_t)(w[i] >> 16));
            s = fmaf(lo, lo, s); s = fmaf(hi, hi, s);
        }
    }
    return s;
}
__device__ __forceinline__ float e8m0_to_f32(int e) { return __uint_as_float(((uint32_t)(e & 0xff)) << 23); }     // 2^(e-127), e in [1,254]

// BUILD PARTITIONING (wast3d_amd/build.py).  This file is compiled NINE times: once without W3D_KNN_FMT — the "main" object: every extern "C" entry point, argument
// checks, the workspace layout, the format-independent kernels — and once per (operand format, feature width) with -DW3D_KNN_FMT=f -DW3D_KNN_D=d — a "leaf" object
// that instantiates the sweep / plan kernels of that one shape and exports three plain functions the main object dispatches to.  One translation unit with all
// 8 shapes x 7 kernels x 4 tile / MFMA variants took six minutes of hipcc; the leaves compile in parallel in about one.
#ifdef W3D_KNN_FMT
#define W3D_KNN_LEAF 1
extern int w3d_g_knn_mode; extern float w3d_g_knn_slack; extern int w3d_g_knn_reserve; extern int64_t *w3d_g_knn_clk; extern int64_t *w3d_g_knn_plan_clk; extern void *w3d_g_knn_ev[2];
#else
#define W3D_KNN_LEAF 0
// Diagnostic knobs: read from the environment ONCE at library load (W3D_KNN_MODE, W3D_KNN_SLACK) and changeable only through
// w3d_knn_mfma_set_tuning — never re-read in the launch path, so a running loop cannot be retuned from outside.
int w3d_g_knn_mode = getenv("W3D_KNN_MODE") ? atoi(getenv("W3D_KNN_MODE")) : 0;
float w3d_g_knn_slack = getenv("W3D_KNN_SLACK") ? (float)atof(getenv("W3D_KNN_SLACK")) : 0.f;      // 0 = KNN_SLACK_REL
int w3d_g_knn_reserve = 0;                    // CUs a persistent pruned sweep leaves free for the kernels of other streams (w3d_knn_mfma_set_cu_reserve)
// Diagnostics (w3d_knn_mfma_debug_clocks): when set, every block of a pruned sweep records {start, end (100-MHz wall clock), XCC | HW_ID, tiles swept} at
// clk[8 * block ..] (+ the shader clock at start / end in words 4, 5, the wall clock at the start / end of the tile loop in words 6, 7) — what the launch looks like from the inside (scripts/prune_diag.py).  NULL (default): one scalar branch per block.
int64_t *w3d_g_knn_clk = nullptr;
int64_t *w3d_g_knn_plan_clk = nullptr;        // the same for the blocks of the prune plan kernel: {block start, loop start, loop end, block end}
void *w3d_g_knn_ev[2] = {nullptr, nullptr};  // (w3d_knn_mfma_debug_events) HIP events recorded right before / behind the emit kernel of a pruned emit sweep: the kernel's own duration
#endif
#define g_knn_mode w3d_g_knn_mode
#define g_knn_slack w3d_g_knn_slack
#define g_knn_reserve w3d_g_knn_reserve
#define g_knn_clk w3d_g_knn_clk
#define g_knn_plan_clk w3d_g_knn_plan_clk
#define g_knn_ev w3d_g_knn_ev

namespace {

constexpr int W3D_MAX_DEVICES = 64;
constexpr int YPAD = 64;       // the -0.5|y|^2 array is padded with -inf to a multiple of this
constexpr int NWAVES = 8;
constexpr int RB = 2;         // 32-row blocks per wave
constexpr int ROWS_W = RB * 32;
constexpr int ROWS_B = NWAVES * ROWS_W;
constexpr int KNN_NQ = 8;                  // work queues of the persistent sweep = XCDs of the chip
constexpr int KNN_Q_INTS = 16;            // ints after the plan in the workspace: the queue cursors [KNN_NQ], then one word: 1 = the plan's tile order is not ascending
// regions of the emit sweep's candidate buffer: one per wave of every row block — the last block's waves past the end of the rows sweep copies of the last row and emit like any other
inline int64_t knn_regions(int64_t N) { return ((N + ROWS_B - 1) / ROWS_B) * NWAVES; }
// STRAGGLER rows (round 6): per style tile a bucket of content-row POSITIONS (of the sweep order) whose cached bounds alone keep the tile in their block's plan — see
// knn_straggler_kernel.  Capacity per tile: 16x the average number of rows per tile, 256 ... 8192 positions (C3: 8192 = 32 MB for the 1024 tiles)
constexpr int KNN_STRAG_CH = 256;         // bucket positions a workgroup of the straggler kernel takes at a time (16 MFMA groups of 16 rows)
constexpr int KNN_STRAG_WG = 8;           // workgroups per tile (a bucket's chunks are dealt to them round-robin)
inline int knn_strag_cap(int64_t N, int64_t M) {
    const int64_t nt = (M + 63) / 64;
    int64_t c = 16 * ((N + nt - 1) / nt);
    c = c < 256 ? 256 : (c > 8192 ? 8192 : c);
    return (int)((c + KNN_STRAG_CH - 1) / KNN_STRAG_CH * KNN_STRAG_CH);
}
// Slack of the warm-start bounds relative to S = (|x|^2 + |y|^2) / 2 >= sum |x_i y_i|.  Products of 16-bit (or fp8) operands are exact in f32; the
// matrix cores round once per k-step (d / 16 <= 16 roundings of partial sums <= S: <= 1e-6 S), the threshold kernels a dozen times more (<= 6e-7 S):
// 4e-6 S covers both with a margin.  (The first version used a blanket 4e-5: with neighbours 50x closer than the rows are long that is 25 % of the
// k-th distance and lets ~5x more candidates through.)  Whatever the slack misses — and seeds that are not k distinct rows — shows as a list with
// fewer than k entries: detected at write-out and redone with a cold bound (repair launch), so the result is exact regardless.
constexpr float KNN_SLACK_REL = 4e-6f;
// fp8 rows (round 5): the scaled 32x32x64 MFMA's scores differ from an f32 dot product of the same dequantised rows by up to ~1.5e-5 S (measured: under the 16-bit slack
// 4.5 % of 420 000 clustered rows ended one neighbour short and went through the repair, 0.14 % at 8e-6, 1 row at 1.6e-5, none at 4e-5: scripts/_ab/fp8_dbg.py) — the
// slack of the fp8 bounds and the score-error allowance of the plans built for fp8 sweeps (KNN_SCORE_ERR2_FP8 = 2E relative to |x|^2 + 2 max |y|^2) are 5x the 16-bit ones
constexpr float KNN_SLACK_REL_FP8 = 2e-5f;
// (round 6, ADVICE r05: the PLAN's allowance — unlike the warm-start slack, a tile the plan drops is never revisited and no repair sees it — is twice the slack: 4e-5, where
// zero rows failed in the measurement above; W3D_KNN_SCORE_ERR2_FP8 of the header, which plans for fp8 sweeps must be built with: w3d_knn_prune_plan_err)
constexpr float KNN_SCORE_ERR2 = 4e-6f, KNN_SCORE_ERR2_FP8 = W3D_KNN_SCORE_ERR2_FP8;
constexpr int QCAP64 = 128, QCAP32 = 256;   // per-wave candidate queue (entries) of the 64-row-tile / 32-row-tile kernels
constexpr int KNN_FLAG_CAP = 16384;   // rows one repair launch can take
// flags: [0] rows flagged by the last call, [1] largest [0] of any call since the workspace was zeroed (sticky: > KNN_FLAG_CAP = some call left rows
// unrepaired; read asynchronously by w3d_knn_mfma_repair_poll), [2..] row ids
constexpr int KNN_FLAG_ROW0 = 2;
}  // namespace
struct __attribute__((aligned(8))) Entry {
    float val;
    uint32_t packed;   // (local row << 26) | style index     (queue)   /   style index (lists)
};
namespace {

// "a is a better neighbour than b": larger val, ties -> lower style index (torch.topk / stable sort order)
__device__ __forceinline__ bool better(float va, uint32_t ia, float vb, uint32_t ib) { return va > vb || (va == vb && ia < ib); }

__device__ __forceinline__ uint32_t lds_addr(const void *p) { return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)p; }

// XOR term of a row's 16-B chunk index; ROWB = row bytes (>= 128: at most two rows share a 256-B bank row)
template <int ROWB>
__device__ __forceinline__ int swz(int r) { return ROWB >= 256 ? (r & 15) : ((r >> 1) & 7); }

// out[r] = scale * sum_c A[r,c]^2, one lane per 16 B of the row; fp8: times the row's own 2^(2(e-127))
template <int D, int FMT>
__global__ __launch_bounds__(256) void row_sqnorm_kernel(const uint8_t *__restrict__ A, const int32_t *__restrict__ rscale, int64_t rows, int64_t rows_padded, float scale,
                                                         float *__restrict__ out) {
    constexpr int ROWB = D * Fmt<FMT>::EB;
    constexpr int LPR = ROWB / 16;                // lanes per row (8, 16, 32)
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t r = gid / LPR;
    const int c = (int)(gid % LPR);
    float s = 0.f;
    if (r < rows) s = sumsq16<FMT>(*reinterpret_cast<const uint4 *>(A + r * ROWB + c * 16));
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (c == 0) {
        if (r < rows) { if (FMT == FMT_FP8) { const float f = e8m0_to_f32(rscale[r]); s *= f * f; } out[r] = scale * s; }
        else if (r < rows_padded) out[r] = -__builtin_inff();
    }
}

// Warm start for iterative loops: the k neighbours found in the previous iteration are k real candidates, so the worst of
// their CURRENT scores is a valid lower bound of this iteration's k-th best score.  One wave per row, lanes over features.
// thr_val[i] = min_j (x_i.y_j - 0.5|y_j|^2) - slack   (slack covers fp32 summation-order differences vs the MFMA path)
// The same bound, four rows per wave (16-bit rows of 256 or 128 elements): a 16-lane group owns a row — each lane holds 32 (16) bytes of it, a gathered style row
// is two (one) 16-byte loads per lane, a dot product closes with 4 cross-lane steps inside the group instead of 6 across the wave.  The one-row-per-wave form above is
// bound by its gather latency (10 rows of 512 B in flight per wave, 1.26 ms at C3 for 5 GB of Infinity-Cache reads); here a wave keeps 4 x 5 rows in flight and the
// groups' index / norm reads are one vector instruction each.  Same arguments, same arithmetic per term; the sum of a dot product is taken in another order
// (covered by the slack, which exists for the MFMA's own order).
template <int D, int FMT>
__global__ __launch_bounds__(256) void knn_prev_threshold4_kernel(const uint8_t *__restrict__ X, const uint8_t *__restrict__ Y, const int32_t *__restrict__ xsc,
                                                                  const int32_t *__restrict__ ysc, const float *__restrict__ xnorm,
                                                                  const float *__restrict__ ynh, const int32_t *__restrict__ prev_idx, int64_t N, int64_t M, int k,
                                                                  float *__restrict__ thr_val, float slack_rel, const int32_t *__restrict__ row_perm,
                                                                  const float *__restrict__ prev_dist2, const float *__restrict__ motion2, float motion_ratio) {
    // Round 4 — rows that hardly moved skip the gathers: with prev_dist2 (the previous sweep's distances, [N,k] ascending) and motion2 (|x_new - x_old|^2), a row whose
    // motion is below motion_ratio x its previous k-th distance takes the MOTION bound D + |dx| (triangle inequality on the k old neighbours: valid, the result stays
    // exact) instead of fetching its 20 neighbours' rows.  In a loop's steady state |dx| / D ~ 1 %: the bound is 1 % looser (~(1.01)^16 = 1.17x the candidates) and the
    // kernel's 10 GB of L2 gathers are gone for almost every row; in a run's first iterations (|dx| ~ D) every row gathers as before.
    // (Round 3 measured a HYBRID bound — only the G farthest previous neighbours fetched, the nearer ones covered by their old distance + the row's motion — and dropped
    // it: at G = 10 of 20 the bound equals this one for 98 % of the rows and the kernel takes 0.46 instead of 0.74 ms, but the sweep that follows takes 9.13 instead of
    // 8.91 ms; G = 8: 9.22, G = 6: 9.44 — same device, same run; the step time came out equal or worse.  scripts/hybrid_bound_probe.py)
    static_assert(D * Fmt<FMT>::EB == 512 || D * Fmt<FMT>::EB == 256, "rows of 512 or 256 bytes (16-bit rows of 256 / 128 elements, fp8 rows of 256)");
    constexpr int V = D * Fmt<FMT>::EB / 16 / 16;          // 16-byte loads per lane and row (2 or 1)
#ifndef W3D_BOUND_B
#define W3D_BOUND_B 4
#endif
    // gathered rows in flight per group.  Round 6: 4 instead of 5 — a round covers 16 neighbours (k = 20: 16 + 4), and a batch that runs past the round's end re-fetches its
    // last row: batches of 5 issued 25 row loads for 20 neighbours, on a kernel that runs at the L2 -> L1 gather rate (MI355X_MICROARCH 'Indexed rows')
    constexpr int B = W3D_BOUND_B;
    const int lane = threadIdx.x & 63, sub = lane & 15, grp = lane >> 4;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    auto dot = [&](const uint4 (&a)[V], const uint4 (&b)[V]) -> float {
        float acc = 0.f;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const uint32_t aw[4] = {a[v].x, a[v].y, a[v].z, a[v].w}, bw[4] = {b[v].x, b[v].y, b[v].z, b[v].w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if constexpr (FMT == FMT_FP8) {                  // e4m3 bytes: four products per dword (the rows' power-of-two scales multiply the sum)
                    const auto a0 = __builtin_amdgcn_cvt_pk_f32_fp8((int)aw[q], false), a1 = __builtin_amdgcn_cvt_pk_f32_fp8((int)aw[q], true);
                    const auto b0 = __builtin_amdgcn_cvt_pk_f32_fp8((int)bw[q], false), b1 = __builtin_amdgcn_cvt_pk_f32_fp8((int)bw[q], true);
                    acc = fmaf(a0[0], b0[0], acc); acc = fmaf(a0[1], b0[1], acc); acc = fmaf(a1[0], b1[0], acc); acc = fmaf(a1[1], b1[1], acc);
                } else if constexpr (FMT == FMT_BF16) {          // two products per instruction (v_dot2c_f32_bf16): the kernel was VALU-bound on unpack + fma (75 % VALU-busy, PMC)
                    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
                    bf16x2_t av, bv; __builtin_memcpy(&av, &aw[q], 4); __builtin_memcpy(&bv, &bw[q], 4);
                    acc = __builtin_amdgcn_fdot2_f32_bf16(av, bv, acc, false);
                } else {
                    typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
                    f16x2_t av, bv; __builtin_memcpy(&av, &aw[q], 4); __builtin_memcpy(&bv, &bw[q], 4);
                    acc = __builtin_amdgcn_fdot2(av, bv, acc, false);
                }
            }
        }
        return acc;
    };
    for (int64_t pos4 = wave0 * 4; pos4 < N; pos4 += nwaves * 4) {
        int64_t pos = pos4 + grp;
        const bool live = pos < N;
        if (!live) pos = N - 1;
        const int64_t i = row_perm ? (int64_t)row_perm[pos] : pos;
        uint4 xv[V];
#pragma unroll
        // (lane `sub` takes the 16-byte pieces sub, sub + 16 of a row — round 6; it was 2 sub, 2 sub + 1: with two pieces per lane every load instruction of a group then touched
        // all four 128-byte lines of its row instead of two)
        for (int v = 0; v < V; ++v) xv[v] = *reinterpret_cast<const uint4 *>(X + i * (D * Fmt<FMT>::EB) + (v * 16 + sub) * 16);
        const float xn = xnorm[i];
        const float xf = FMT == FMT_FP8 ? e8m0_to_f32(xsc[i]) : 1.0f;
        float worst = __builtin_inff();
        bool still = false;                                  // this group's row takes the motion bound: no gathers
        if (prev_dist2) {
            const float dk = sqrtf(fmaxf(prev_dist2[i * k + (k - 1)], 0.f)), mv = sqrtf(fmaxf(motion2[i], 0.f));
            still = live && mv <= motion_ratio * dk && dk < 3.0e38f;
            if (still) {                                     // as knn_motion_threshold_kernel: thr = (|x|^2 - (D + |dx|)^2) / 2 - slack
                const float Db = dk * (1.0f + 1e-5f) + mv * (1.0f + 1e-5f);
                const float sx = sqrtf(xn) + Db;
                worst = 0.5f * (xn - Db * Db) - (2.5f * slack_rel * (xn + sx * sx) + 1e-30f);
            }
        }
        if (__builtin_amdgcn_ballot_w64(!still) == 0ull) {   // all four rows of the wave: nothing to fetch
            if (live && sub == 0) thr_val[i] = worst;
            continue;
        }
        for (int j0 = 0; j0 < k && !still; j0 += 16) {       // 16 neighbours per round: lane `sub` of the group keeps neighbour j0 + sub (index, norm, score)
            const int jm = j0 + sub;
            int32_t ym = jm < k ? prev_idx[i * k + jm] : 0;
            if (ym < 0 || (int64_t)ym >= M) ym = 0;              // a stale / foreign list only loosens the bound
            const float yh = ynh[ym];
            const float yf = FMT == FMT_FP8 ? xf * e8m0_to_f32(ysc[ym]) : 1.0f;
            float mine = 0.f;
            const int cnt = min(16, k - j0);
            for (int u0 = 0; u0 < cnt; u0 += B) {
                uint4 yv[B][V];
#pragma unroll
                for (int u = 0; u < B; ++u) {
                    const int64_t y = (int64_t)__shfl(ym, (lane & 48) | min(u0 + u, cnt - 1), 64);
#pragma unroll
                    for (int v = 0; v < V; ++v) yv[u][v] = *reinterpret_cast<const uint4 *>(Y + y * (D * Fmt<FMT>::EB) + (v * 16 + sub) * 16);
                }
#pragma unroll
                for (int u = 0; u < B; ++u) {
                    float dt = dot(xv, yv[u]);
                    dt += __shfl_xor(dt, 8, 64); dt += __shfl_xor(dt, 4, 64); dt += __shfl_xor(dt, 2, 64); dt += __shfl_xor(dt, 1, 64);
                    if (sub == u0 + u) mine = FMT == FMT_FP8 ? dt * yf : dt;
                }
            }
            if (jm < k) worst = fminf(worst, mine + yh - slack_rel * (0.5f * xn - yh));
        }
        worst = fminf(worst, __shfl_xor(worst, 8, 64)); worst = fminf(worst, __shfl_xor(worst, 4, 64));
        worst = fminf(worst, __shfl_xor(worst, 2, 64)); worst = fminf(worst, __shfl_xor(worst, 1, 64));
        if (live && sub == 0) thr_val[i] = worst;
    }
}

template <int D, int FMT>
__global__ __launch_bounds__(256) void knn_prev_threshold_kernel(const uint8_t *__restrict__ X, const uint8_t *__restrict__ Y, const int32_t *__restrict__ xsc,
                                                                 const int32_t *__restrict__ ysc, const float *__restrict__ xnorm, const float *__restrict__ ynh,
                                                                 const int32_t *__restrict__ prev_idx, int64_t N, int64_t M, int k, float *__restrict__ thr_val, float slack_rel,
                                                                 const int32_t *__restrict__ row_perm) {
    constexpr int EPL = D / 64;          // elements per lane (1, 2 or 4)
    constexpr int EB = Fmt<FMT>::EB;
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    auto elem = [&](const uint8_t *base, int64_t row, int e) -> float {
        const uint8_t *p = base + (row * D + lane * EPL + e) * EB;
        if (FMT == FMT_FP8) return __builtin_amdgcn_cvt_f32_fp8((int)*p, 0);
        return decode16<FMT>(*reinterpret_cast<const uint16_t *>(p));
    };
    // 16-bit rows: the lane's EPL elements of a row are ONE load (8 / 4 / 2 bytes), and the k gathered rows are fetched five at a time (the compiler
    // otherwise issues one 2-byte load per element and waits for each row before asking for the next)
    auto ldrow = [&](const uint8_t *base, int64_t row) -> uint2 {
        uint2 v = make_uint2(0u, 0u);
        const uint8_t *p = base + (row * D + lane * EPL) * EB;
        if (EPL * EB == 8) v = *reinterpret_cast<const uint2 *>(p);
        else if (EPL * EB == 4) v.x = *reinterpret_cast<const uint32_t *>(p);
        else if (EPL * EB == 2) v.x = *reinterpret_cast<const uint16_t *>(p);
        else v.x = *p;
        return v;
    };
    auto dot16 = [&](uint2 a, uint2 b) -> float {
        float s = decode16<FMT>((uint16_t)(a.x & 0xffffu)) * decode16<FMT>((uint16_t)(b.x & 0xffffu));
        if (EPL >= 2) s = fmaf(decode16<FMT>((uint16_t)(a.x >> 16)), decode16<FMT>((uint16_t)(b.x >> 16)), s);
        if (EPL == 4) { s = fmaf(decode16<FMT>((uint16_t)(a.y & 0xffffu)), decode16<FMT>((uint16_t)(b.y & 0xffffu)), s); s = fmaf(decode16<FMT>((uint16_t)(a.y >> 16)), decode16<FMT>((uint16_t)(b.y >> 16)), s); }
        return s;
    };
    for (int64_t pos = wave0; pos < N; pos += nwaves) {
        // row_perm: rows are taken in the order the sweep takes them (similar rows together: their gathered style rows share L2 lines)
        const int64_t i = row_perm ? (int64_t)row_perm[pos] : pos;
        float xs[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) xs[e] = elem(X, i, e);
        const float xn = xnorm[i];
        const float xf = FMT == FMT_FP8 ? e8m0_to_f32(xsc[i]) : 1.0f;
        float worst = __builtin_inff();
        if (FMT != FMT_FP8) {
            const uint2 xv = ldrow(X, i);
            for (int j0 = 0; j0 < k; j0 += 5) {
                uint2 yv[5]; int64_t yi[5];
#pragma unroll
                for (int u = 0; u < 5; ++u) { int64_t y = j0 + u < k ? prev_idx[i * k + j0 + u] : 0; if (y < 0 || y >= M) y = 0; yi[u] = y; yv[u] = ldrow(Y, y); }
#pragma unroll
                for (int u = 0; u < 5; ++u) {
                    const float dt = wave_sum(dot16(xv, yv[u]));
                    const float yh = ynh[yi[u]];
                    if (j0 + u < k) worst = fminf(worst, dt + yh - slack_rel * (0.5f * xn - yh));
                }
            }
            if (lane == 0) thr_val[i] = worst;
            continue;
        }
        for (int j = 0; j < k; ++j) {
            int64_t y = prev_idx[i * k + j];
            if (y < 0 || y >= M) y = 0;                       // a stale / foreign list only loosens the bound
            float dot = 0.f;
#pragma unroll
            for (int e = 0; e < EPL; ++e) dot = fmaf(xs[e], elem(Y, y, e), dot);
            dot = wave_sum(dot);
            if (FMT == FMT_FP8) dot *= xf * e8m0_to_f32(ysc[y]);
            const float yh = ynh[y];
            const float val = dot + yh;
            const float slack = slack_rel * (0.5f * xn - yh);
            worst = fminf(worst, val - slack);
        }
        if (lane == 0) thr_val[i] = worst;
    }
}

// Warm start without touching Y: if row i moved by |dx_i| since the iteration that found its neighbours at k-th distance D_i, its new
// k-th distance is at most D_i + |dx_i| (triangle inequality on the k old neighbours).  thr_val = (|x|^2 - (D + |dx|)^2) / 2 - slack.
__global__ __launch_bounds__(256) void knn_motion_threshold_kernel(const float *__restrict__ xnorm, const float *__restrict__ prev_dist2, const float *__restrict__ motion2,
                                                                   int64_t N, int k, float *__restrict__ thr_val, float slack_rel) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const float xn = xnorm[i];
    const float Db = sqrtf(fmaxf(prev_dist2[i * k + (k - 1)], 0.f)) * (1.0f + 1e-5f) + sqrtf(motion2[i]) * (1.0f + 1e-5f);
    const float sx = sqrtf(xn) + Db;
    const float slack = slack_rel * (xn + sx * sx) + 1e-30f;
    thr_val[i] = 0.5f * (xn - Db * Db) - slack;
}

// minimum over the 16 lanes of a DPP row (all lanes get it)
__device__ __forceinline__ float row16_min(float v) {
    auto dpp = [](float x, auto ctrl) -> float { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(ctrl)::value, 0xf, 0xf, true)); };
    v = fminf(v, dpp(v, std::integral_constant<int, 0xB1>{}));          // quad_perm [1,0,3,2]
    v = fminf(v, dpp(v, std::integral_constant<int, 0x4E>{}));          // quad_perm [2,3,0,1]
    v = fminf(v, dpp(v, std::integral_constant<int, 0x141>{}));         // row_half_mirror
    v = fminf(v, dpp(v, std::integral_constant<int, 0x140>{}));         // row_mirror
    return v;
}
// minimum over aligned groups of 8 lanes (all lanes get it)
__device__ __forceinline__ float row8_min(float v) {
    auto dpp = [](float x, auto ctrl) -> float { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(ctrl)::value, 0xf, 0xf, true)); };
    v = fminf(v, dpp(v, std::integral_constant<int, 0xB1>{}));
    v = fminf(v, dpp(v, std::integral_constant<int, 0x4E>{}));
    v = fminf(v, dpp(v, std::integral_constant<int, 0x141>{}));         // row_half_mirror: lane i <-> 7 - i inside each group of 8
    return v;
}
__device__ __forceinline__ float wave64_min(float v) {
    v = row16_min(v);
    const int b = __float_as_int(v);
    return fminf(fminf(__int_as_float(__builtin_amdgcn_readlane(b, 0)), __int_as_float(__builtin_amdgcn_readlane(b, 16))),
                 fminf(__int_as_float(__builtin_amdgcn_readlane(b, 32)), __int_as_float(__builtin_amdgcn_readlane(b, 48))));
}

// Write-out of a wave's 64 k-entry lists (LDS, lane-strided [k][64]; lane L owns local row L): rank sort by score (ties: lower index), conversion to squared
// distances with exact distance ties ordered by index, rows with fewer than k entries flagged for the cold repair, entry-wise coalesced stores.  Shared by the sweep
// proper (lists kept during the sweep) and by knn_select_kernel (lists built from a sweep's emitted candidates).
__device__ __forceinline__ void knn_write_out(Entry *lists, uint32_t *rcnt, const int lane, const int lcount, const int k, const int64_t N, const int64_t wave_row0,
                                              const int64_t rowblk0, const int parts, const int part, const bool split, const float *__restrict__ xnorm,
                                              const int32_t *__restrict__ row_perm, int32_t *__restrict__ flags, int32_t *__restrict__ out_idx, float *__restrict__ out_d2,
                                              const int nin_arg = -1) {
    // nin (knn_select_kernel): the lane's list holds nin >= k candidates in slots 0 ... nin - 1 (nin <= 32): the rank sort keeps the k best of them
    const int nin = nin_arg < 0 ? k : nin_arg;
    // ---- write out: each lane selection-sorts its k entries (best score first), converts to squared distances and orders exact
    // distance ties by index (two scores can round to the same d2): the documented order is (d2 ascending, index ascending)
    const int64_t lrow = wave_row0 + lane;                               // position in the processing order
    if (lrow < N) {
        const int64_t row = row_perm ? (int64_t)row_perm[lrow] : lrow;  // the content row itself
        // a warm-start bound that was too tight (rounding beyond the slack, duplicate seeds) shows as a list with fewer than k entries:
        // the row is queued for the repair launch (cold bound); rows with k entries are exact (everything above the bound was seen)
        if (flags && !split && lcount < k) { const int slot = atomicAdd(&flags[0], 1); if (slot < KNN_FLAG_CAP) flags[KNN_FLAG_ROW0 + slot] = (int32_t)row; }
        const float xn = xnorm[row];
        // RANK sort of the lane's k entries, in registers: one 64-bit key per entry — score bits mapped to an unsigned order | inverted style index | slot (unique,
        // so the ranks are a permutation even among unfilled slots) — and rank(t) = number of larger keys.  The k^2 selection sort through the LDS this replaces
        // took 16 of the 20 us a row block spent after its last tile (scripts/rank_sweep_diag.py).  Order: better() — score descending, ties: lower index.
        constexpr int KMAX = 32;                                         // the lists live in LDS: k <= 24 ... 32 for every format this kernel takes (launch())
        if (k <= KMAX) {
            uint64_t key[KMAX];
#pragma unroll
            for (int u = 0; u < KMAX; ++u) {
                key[u] = 0ull;                                           // slots past the last entry: below every real key (those carry 63 - u >= 31 in their low bits)
                if (u < nin) {
                    const Entry e = lists[u * ROWS_W + lane];
                    uint32_t b = __float_as_uint(e.val);
                    if (b == 0x80000000u) b = 0u;                    // -0 == +0 in better()
                    const uint32_t ord = b ^ ((uint32_t)((int32_t)b >> 31) | 0x80000000u);          // float order -> unsigned order
                    const uint32_t ix = e.packed < 0x3ffffffu ? e.packed : 0x3ffffffu;
                    key[u] = ((uint64_t)ord << 32) | (uint64_t)(((0x3ffffffu - ix) << 6) | (uint32_t)(63 - u));
                }
            }
#pragma unroll
            for (int t = 0; t < KMAX; ++t) {
                if (t < nin) {
                    int rank = 0;
#pragma unroll
                    for (int u = 0; u < KMAX; ++u) rank += key[u] > key[t] ? 1 : 0;
                    const uint32_t ord = (uint32_t)(key[t] >> 32);
                    const uint32_t b = ord ^ ((ord >> 31) ? 0x80000000u : 0xffffffffu);
                    if (rank < k) lists[rank * ROWS_W + lane] = Entry{__uint_as_float(b), 0x3ffffffu - ((uint32_t)key[t] >> 6)};
                }
            }
        } else {
            for (int t = 0; t < k; ++t) {                                // (no caller reaches this: selection sort through the LDS)
                float bv = -__builtin_inff(); uint32_t bi = 0xffffffffu; int bp = t;
                for (int u = t; u < k; ++u) {
                    const Entry le = lists[u * ROWS_W + lane];
                    if (better(le.val, le.packed, bv, bi)) { bv = le.val; bi = le.packed; bp = u; }
                }
                const Entry tmp = lists[t * ROWS_W + lane];
                lists[bp * ROWS_W + lane] = tmp;
                lists[t * ROWS_W + lane] = Entry{bv, bi};
            }
        }
        // split sweeps (parts > 1) hand their lists over in SCORE order with raw scores: the merge must select by score exactly as a single sweep does
        // (two scores can round to one d2: selecting by (d2, index) there would keep a different k-th neighbour than the unsplit sweep)
        // an unfilled slot (sentinel index: fewer than k candidates, e.g. a NaN row) reads distance +inf — tested on the INDEX, not on the value
        // (this file is compiled without NaN semantics)
        if (!split) {
            for (int t = 0; t < k; ++t) {
                const Entry e = lists[t * ROWS_W + lane];
                const uint32_t bi = e.packed;
                const float dcur = bi >= 0x3ffffffu ? __builtin_inff() : fmaxf(xn - 2.0f * e.val, 0.f);
                int pos = t;
                while (pos > 0) {                            // rare: bubble past entries with the same rounded distance and a larger index
                    const Entry pe = lists[(pos - 1) * ROWS_W + lane];
                    if (pe.val == dcur && pe.packed > bi) { lists[pos * ROWS_W + lane] = pe; --pos; } else break;
                }
                lists[pos * ROWS_W + lane] = Entry{dcur, bi};        // slots [0, t] now hold (d2, index)
            }
        }
        rcnt[lane] = (uint32_t)row;                                      // (rows < 2^32: the index arrays are int32)
    }
    // The lists leave ENTRY by entry, not lane by lane: lane l of pass p writes entry (p * 64 + l) of the wave's 64 x k block, i.e. consecutive lanes write
    // consecutive words of a row's k-entry segment (80 B at k = 20) instead of 64 different cache lines per store instruction.
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    {
        const int64_t nrow = N - wave_row0 < (int64_t)ROWS_W ? N - wave_row0 : (int64_t)ROWS_W;        // rows of this wave that exist (<= 0: none)
        const int dq = ROWS_W / k, dr = ROWS_W % k;                      // 64 entries further on: dq rows and dr entries
        int rr = lane / k, t = lane % k;
        for (int p = 0; p < k; ++p) {                                    // 64 k entries = k passes of 64
            if (rr < nrow) {
                const Entry e = lists[t * ROWS_W + rr];
                const int64_t o = split ? ((wave_row0 + rr - rowblk0 * ROWS_B) * parts + part) * k + t : (int64_t)rcnt[rr] * k + t;   // partial lists: by position; results: by row
                out_idx[o] = (int32_t)e.packed;
                out_d2[o] = e.val;
            }
            rr += dq; t += dr;
            if (t >= k) { t -= k; ++rr; }
        }
    }
}

// SEL (selection scheme of the epilogue):
//   0  running k-lists in LDS (direct fill / queue + merge, threshold tightened at merges): every sweep that has to FIND its bound — cold sweeps, the motion bound,
//      the repair — and the chunked / split sweeps that hand partial lists to a merge.
//   1  EMIT (round 4): under the gathered warm-start bound the threshold is already the k-th distance to within a few candidates per row (~25 pass for k = 20), so
//      the sweep keeps no lists at all: a lane whose score reaches its row's bound appends (score, row | style index) to a per-wave LDS queue (ballot + mbcnt, no
//      atomics, no merge, no block-wide vote); a full queue is copied to the wave's region of a global candidate buffer; knn_select_kernel picks the k best per row
//      afterwards (same order, same tie rule, same repair of rows that end short: results bit-equal to SEL 0).  What it buys: the candidate-bearing unit costs a
//      queue append instead of an LDS-atomic round trip + list store, the 10 KB of lists per wave and five per-lane state registers are gone from the tile loop, and
//      the per-block write-out (rank sort, ~20 us with the matrix pipe idle) runs as its own bandwidth-trivial kernel.
//      With `lb` (cached lower bounds, see knn_tile_prune_kernel): every swept (row block, tile) pair also records per content row the exact minimum distance to the
//      tile's rows (the unit maxima the pretest forms anyway, max-reduced over the tile in LDS) + the row's cumulative motion, as f16 rounded down.
constexpr int QCAP_EMIT = 1024;          // per-wave candidate queue of the emit sweep (entries; 8 KB of the 10 KB the lists used to take)
constexpr int KNN_REGION_CAP = 8192;     // candidates a wave's 64 rows can emit per sweep (entries of its region of the candidate buffer; beyond: rows go to the cold repair)
template <int D, int BN, int QCAP, int SH, int FMT, bool DF = true, int SEL = 0>
__device__ __forceinline__ void knn_mfma_body(const uint8_t *__restrict__ X, const uint8_t *__restrict__ Y, const int32_t *__restrict__ xsc,
                                                          const int32_t *__restrict__ ysc, const float *__restrict__ xnorm,
                                                          const float *__restrict__ ynh, const float *__restrict__ thr_init, int64_t N_arg, int64_t M, int k,
                                                          int32_t *__restrict__ out_idx, float *__restrict__ out_d2, int mode_arg, int64_t rowblk0, int parts,
                                                          int32_t *__restrict__ flags, const int32_t *__restrict__ n_dev, int64_t tile0 = -1, int64_t tile1 = -1,
                                                          int part_fixed = -1, const int32_t *__restrict__ row_perm = nullptr,
                                                          const int32_t *__restrict__ tile_order = nullptr, const int32_t *__restrict__ tile_count = nullptr,
                                                          const int32_t *__restrict__ block_order = nullptr, int64_t *__restrict__ clk = nullptr,
                                                          int64_t rblk_given = -1, int part_given = -1, Entry *__restrict__ cand_buf = nullptr,
                                                          int32_t *__restrict__ cand_cnt = nullptr, uint16_t *__restrict__ lb = nullptr,
                                                          const float *__restrict__ lb_cum = nullptr, float lb_margin = 0.f, bool tiles_ascending = true) {
    const int64_t clk_t0 = clk ? (int64_t)wall_clock64() : 0;
    const int64_t clk_c0 = clk ? (int64_t)__builtin_amdgcn_s_memtime() : 0;
    // in-kernel diagnostic switches (W3D_KNN_MODE bits 1,2,4,64,128,256,512,1024,2048,4096) cost a branch per k-step: compiled in only with -DW3D_KNN_DEBUG
#ifdef W3D_KNN_DEBUG
    const int mode = mode_arg;
#else
    constexpr int mode = 0; (void)mode_arg;
#endif
    // repair launch (see launch()): the number of rows is a device value; blocks past it leave at once
    const int64_t N = n_dev ? (int64_t)min((int64_t)*n_dev, N_arg) : N_arg;
    if (n_dev && (rowblk0 + (int64_t)(parts > 1 ? blockIdx.x / parts : blockIdx.x)) * (NWAVES * 64) >= N) return;
    static_assert(FMT != FMT_FP8 || SH == 0, "the scaled fp8 MFMA is used in its 32x32x64 shape");
    constexpr int EB = Fmt<FMT>::EB;      // bytes per element
    constexpr int FB = Fmt<FMT>::FB;      // operand bytes per lane per MFMA (16: 8 x 16-bit; 32: 32 x fp8)
    constexpr int NCH = FB / 16;          // ds_read_b128 per fragment
    constexpr int XBR = SH ? 16 : 32;     // content rows per MFMA (columns of its output block)
    constexpr int XBN = ROWS_W / XBR;     // such blocks per wave
    constexpr int HB = 64 / XBR;          // k-groups across the lanes of one operand fragment
    constexpr int ROWB = D * EB;
    constexpr int KS = ROWB / (FB * HB);  // MFMA k-steps per tile
    constexpr int NR = SH ? 8 : 16;       // candidates per lane per (32 style rows, x block)
    constexpr int PF = 2;                 // fragment reads run PF (1 or 2) k-steps ahead of the MFMAs: LDS latency under 8-wave load exceeds one k-step
    static_assert(PF == 2, "the s_waitcnt literals below are written for PF = 2");
    constexpr int CPR = ROWB / 16;        // 16-B chunks per row
    constexpr int TILE_BYTES = BN * ROWB;
    constexpr int NT = NWAVES * 64;
    constexpr int CHUNKS_PER_THREAD = (BN * CPR + NT - 1) / NT;
    constexpr int QE = SEL ? QCAP_EMIT : QCAP;   // entries of the per-wave queue
    constexpr int EPL = QCAP / 64;        // queue entries per lane during a flush
    constexpr int SCB = FMT == FMT_FP8 ? 2 * BN * 4 : 0;                  // per-row scales of the style tile (fp8)
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *ytile = smem;                                               // [2][TILE_BYTES]
    float *ynh_t = reinterpret_cast<float *>(smem + 2 * TILE_BYTES);     // [2][BN]
    typedef volatile __attribute__((address_space(3))) int *lds_vint_t;   // explicit LDS pointer: ds_read/ds_write, not flat
    int *ysc_t = reinterpret_cast<int *>(smem + 2 * TILE_BYTES + 2 * BN * 4);   // [2][BN] (fp8 only)
    lds_vint_t vote_ring = (lds_vint_t)(smem + 2 * TILE_BYTES + 2 * BN * 4 + SCB);   // [4]
    uint8_t *wave_base = smem + 2 * TILE_BYTES + 2 * BN * 4 + SCB + 16;
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // wave-uniform: lives in an SGPR (LDS-DMA destinations, per-wave LDS bases and the row base become scalar)
    const int kr = (k + 3) & ~3;                                         // list rows, padded to a multiple of 4 (pad = +inf, never the worst)
    // SEL 1: queue [QE] | rowmax [ROWS_W] f32 (per content row the best score of the current tile: cached lower bounds) — no lists, no counters
    const int per_wave = SEL ? QE * (int)sizeof(Entry) + ROWS_W * 4 : (QCAP + kr * ROWS_W) * (int)sizeof(Entry) + 3 * ROWS_W * 4;
    Entry *queue = reinterpret_cast<Entry *>(wave_base + wid * per_wave);
    Entry *sorted = queue;                                               // the counting sort is IN PLACE: every lane holds its entries in registers while the queue is rewritten
    Entry *lists = queue + QCAP;                                         // [kr][ROWS_W], lane-strided
    float *rowmax = reinterpret_cast<float *>(queue + QE);              // (SEL 1)
    uint32_t *rcnt = reinterpret_cast<uint32_t *>(lists + kr * ROWS_W);  // [ROWS_W] per-row counters of a flush
    uint32_t *roff = rcnt + ROWS_W;                                      // [ROWS_W] per-row segment offsets
    uint32_t *rowcnt = roff + ROWS_W;                                    // [ROWS_W] candidates seen per row: the first k of a row go STRAIGHT into its list (slot = old count)

    const int xr = lane & (XBR - 1), h = lane / XBR;
    // parts > 1 (tail launch): `parts` consecutive blocks share a row block and each sweeps 1/parts of the style tiles; they write
    // partial k-lists to [local row][part][k] which a merge kernel combines (keeps the last, partially filled block round short)
    // part_fixed >= 0 (chunked sweep, w3d_knn_mfma_part): EVERY block sweeps the style tiles [tile0, tile1) — e.g. the shard of the style set that has
    // just arrived from another GPU — and writes its partial lists into slot `part_fixed` of `parts`; the merge runs once all chunks are through
    const int part = part_fixed >= 0 ? part_fixed : part_given >= 0 ? part_given : (parts > 1 ? (int)(blockIdx.x % parts) : 0);   // part_given: persistent sliced sweep
    // block_order (optional, pruned sweeps): launch position -> row block, longest tile list first (the blocks of a pruned sweep differ in length by up to the
    // whole sweep; started in that order the last ones to finish are the short ones)
    const int64_t bid = (parts > 1 && part_fixed < 0) ? blockIdx.x / parts : blockIdx.x;
    // rblk_given >= 0 (persistent pruned sweep): the workgroup took this row block from its XCD's work queue
    const int64_t rblk = rblk_given >= 0 ? rblk_given : rowblk0 + (block_order ? (int64_t)block_order[bid] : bid);
    const int64_t wave_row0 = rblk * ROWS_B + wid * ROWS_W;

    // ---- X fragments (B operand): lane holds the FB bytes X[row = rb*XBR + xr][(HB s + h) FB .. +FB] for every k-step s
    typedef typename std::conditional<FMT == FMT_FP8, i32x8, typename std::conditional<FMT == FMT_F16, f16x8, bf16x8>::type>::type frag_t;
    frag_t xf[XBN][KS];
    int xscale[XBN];                      // fp8: E8M0 scale of the lane's content row (scale operand of the B side)
#pragma unroll
    for (int rb = 0; rb < XBN; ++rb) {
        int64_t row = wave_row0 + rb * XBR + xr;
        if (row >= N) row = N - 1;                                       // clamp; never written out
        // row_perm (optional): the ORDER the content rows are processed in — position -> row.  Rows with similar features placed in one 32-row MFMA
        // block share the style tiles their candidates sit in, so the selection's slow path runs in fewer 32x32 score blocks; results land at the row's
        // own place, so the output does not depend on the order
        if (row_perm) row = row_perm[row];
#pragma unroll
        for (int s = 0; s < KS; ++s) xf[rb][s] = *reinterpret_cast<const frag_t *>(X + row * ROWB + (HB * s + h) * FB);
        xscale[rb] = FMT == FMT_FP8 ? xsc[row] : 0;
    }
    if constexpr (SEL == 0) {
        for (int t = 0; t < k; ++t) lists[t * ROWS_W + lane] = Entry{-__builtin_inff(), 0x3ffffffu};   // unfilled slots sort last (a split sweep may see < k candidates)
        for (int t = k; t < kr; ++t) lists[t * ROWS_W + lane] = Entry{__builtin_inff(), 0u};
        rowcnt[lane] = 0u;
    } else {
        rowmax[lane] = -__builtin_inff();
    }
    // cached lower bounds (SEL 1, lb != NULL): constants of the lane's own row (local row `lane`): m^2 = lbA - 2 (best score of the tile), s = m + lbC
    float lbA = 0.f, lbC = 0.f;
    if constexpr (SEL == 1) {
        if (lb) {
            int64_t r0 = wave_row0 + lane; r0 = r0 < N ? r0 : N - 1; if (row_perm) r0 = row_perm[r0];
            lbA = xnorm[r0] * (1.0f - (FMT == FMT_FP8 ? KNN_SCORE_ERR2_FP8 : KNN_SCORE_ERR2)) - lb_margin;      // |x|^2 - 2E, E = the plan's bound on a computed score's error (knn_tile_prune_kernel)
            lbC = lb_cum[r0];
        }
    }

    // ---- per-lane list state: lane L owns local row L of this wave
    int lcount = 0, lminpos = 0;
    float lmin_v = __builtin_inff();
    uint32_t lmin_i = 0;
    float thr[XBN];
    float thr0 = -__builtin_inff();                                      // warm-start bound of lane's own row (row wave_row0 + lane)
    if (thr_init) { int64_t r0 = wave_row0 + lane; r0 = r0 < N ? r0 : N - 1; if (row_perm) r0 = row_perm[r0]; thr0 = thr_init[r0]; }
#pragma unroll
    for (int rb = 0; rb < XBN; ++rb) thr[rb] = __shfl(thr0, rb * XBR + xr, 64);
    int qcnt = 0;                                                        // wave-uniform (lives in an SGPR)
    // DF (direct list fill: the first k candidates of a row skip the queue) pays under a warm-start bound, where a row sees few candidates beyond its first k.
    // A cold sweep's candidate storm (every score of the first tiles) is cheaper through the queue alone with the plain merge: C2 (100k x 100k x 128, cold) 4.7 ms
    // against 5.6 with the direct-fill code — so cold sweeps and the repair launch are compiled with DF = false.
    // queue fill at which a wave asks the whole block to merge (tuning: bits 16.. of the mode word; default half the queue)
    const int vote_at = ((mode_arg >> 16) & 0xff) > 0 ? min((mode_arg >> 16) & 0xff, QCAP - 1) : QCAP / 2;

    // Merge the queue into the per-row lists.  Counting sort by row with LDS atomics (one ds_add_rtn per entry,
    // one wave scan), then every lane folds its own row's segment into its list.
    // SEL 1: the wave's queue goes to its region of the candidate buffer (region = the wave's 64 rows; slices of a sliced sweep share it: one atomic per spill)
    const int64_t region = wave_row0 / ROWS_W;
    auto flush = [&](bool tile_end = false) {
        if constexpr (SEL == 1) {
            const int n = __builtin_amdgcn_readfirstlane(qcnt);
            if (n > 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                int base = 0;
                if (lane == 0) base = atomicAdd(&cand_cnt[region], n);
                base = __builtin_amdgcn_readfirstlane(base);
                Entry *dst = cand_buf + region * KNN_REGION_CAP;
                for (int e = lane; e < n; e += 64)
                    if (base + e < KNN_REGION_CAP) dst[base + e] = queue[e];     // past the capacity: dropped — the count says so, knn_select_kernel sends the rows to the repair
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            qcnt = 0;
            return;
        }
        if (mode & 2) { qcnt = 0; return; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        rcnt[lane] = 0u;
        Entry mine[EPL]; uint32_t pos[EPL];
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            const int e = lane + 64 * j;
            mine[j] = queue[e < qcnt ? e : 0];
        }
#pragma unroll
        for (int j = 0; j < EPL; ++j)
            if (lane + 64 * j < qcnt) pos[j] = atomicAdd(&rcnt[mine[j].packed >> 26], 1u);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int c = (int)rcnt[lane];
        int incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int tt = __shfl_up(incl, o, 64); if (lane >= o) incl += tt; }
        const int off = incl - c;
        roff[lane] = (uint32_t)off;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < EPL; ++j)
            if (lane + 64 * j < qcnt) sorted[roff[mine[j].packed >> 26] + pos[j]] = mine[j];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // the first k candidates of a row were written straight into its list by the epilogue (slot = the row's candidate count at that time); a list
        // completed since the last merge gets its worst entry located now — from then on the row's threshold is its exact k-th best.  Queue entries are
        // candidates that found their row's list full: each is compared with the list's worst entry and, if better, replaces it.  ONE rescan site
        // (the loop runs one extra turn for a rescan that is still due): the kernel has no registers to spare for several inlined copies.
        if constexpr (!DF) {                     // plain merge: every candidate came through the queue
            for (int i = 0; i < c; ++i) {
                const Entry en = sorted[off + i];
                const uint32_t yi = en.packed & 0x3ffffffu;
                bool rescan = false;
                if (lcount < k) {
                    lists[lcount * ROWS_W + lane] = Entry{en.val, yi};
                    ++lcount;
                    rescan = (lcount == k);
                } else if (better(en.val, yi, lmin_v, lmin_i)) {
                    lists[lminpos * ROWS_W + lane] = Entry{en.val, yi};
                    rescan = true;
                }
                if (rescan) {   // worst entry of the list = current k-th best; loads issued 4 at a time
                    float wv = __builtin_inff(); uint32_t wi = 0; int wp = 0;
                    for (int t = 0; t < kr; t += 4) {
                        Entry le[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) le[u] = lists[(t + u) * ROWS_W + lane];
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (better(wv, wi, le[u].val, le[u].packed) || (t + u) == 0) { wv = le[u].val; wi = le[u].packed; wp = t + u; }
                    }
                    lmin_v = wv; lmin_i = wi; lminpos = wp;
                }
            }
        } else {
            const int seen = (int)rowcnt[lane];
            const int filled = seen < k ? seen : k;
            bool due = filled == k && lcount < k;
            lcount = filled;
            for (int i = 0; i <= c; ++i) {
                if (due) {   // worst entry of the list = current k-th best; loads issued 4 at a time
                    float wv = __builtin_inff(); uint32_t wi = 0; int wp = 0;
                    for (int t = 0; t < kr; t += 4) {
                        Entry le[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) le[u] = lists[(t + u) * ROWS_W + lane];
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (better(wv, wi, le[u].val, le[u].packed) || (t + u) == 0) { wv = le[u].val; wi = le[u].packed; wp = t + u; }
                    }
                    lmin_v = wv; lmin_i = wi; lminpos = wp;
                    due = false;
                }
                if (i == c) break;
                const Entry en = sorted[off + i];
                const uint32_t yi = en.packed & 0x3ffffffu;
                if (lcount < k) {                  // (only candidates retried after a full queue come here with a list that is not full)
                    lists[lcount * ROWS_W + lane] = Entry{en.val, yi};
                    ++lcount;
                    due = lcount == k;
                } else if (better(en.val, yi, lmin_v, lmin_i)) {
                    lists[lminpos * ROWS_W + lane] = Entry{en.val, yi};
                    due = true;
                }
            }
            if (lcount > seen) rowcnt[lane] = (uint32_t)lcount;
        }
        // Threshold of a FULL list.  A candidate that only ties the k-th best score replaces it only with a lower style index (better()); the tiles of a sweep are
        // taken in ascending order, so after a merge at a tile's END every later candidate has a higher index than every entry of the list: ties can no longer
        // win, and the threshold is the NEXT float above the k-th best.  On quantised rows (fp8: few distinct scores) ties at the k-th place are common and every
        // one of them used to walk the whole candidate path only to be rejected at the next merge.  A merge in mid-tile (queue overflow) keeps the inclusive
        // threshold: the rest of that tile may still hold a tie with a lower index.
        float kth = lmin_v;
        if (tile_end && lcount == k && tiles_ascending) {                // (a caller-given tile order, w3d_knn_plan_order, is not ascending: a later tile may hold a tie with a LOWER index)
            const uint32_t b = __float_as_uint(lmin_v);
            kth = lmin_v == 0.f ? 1.4e-45f : __uint_as_float(lmin_v > 0.f ? b + 1u : b - 1u);
        }
        const float mythr = (lcount == k) ? fmaxf(kth, thr0) : thr0;
#pragma unroll
        for (int rb = 0; rb < XBN; ++rb) thr[rb] = __shfl(mythr, rb * XBR + xr, 64);
        qcnt = 0;
        __builtin_amdgcn_wave_barrier();
    };

    // ---- Y tile staging: global -> LDS directly (global_load_lds, 16 B per lane, no staging registers).
    // The LDS image is lane-linear, so the bank swizzle is applied to the SOURCE chunk and again on the read:
    // slot (r, c') holds logical chunk c' ^ swz(r).
    const int64_t ntiles_all = (M + BN - 1) / BN;
    // tile_count (optional, with tile_order: a PRUNED sweep, w3d_knn_prune_plan): only the first tile_count[row block] positions of the block's order are swept —
    // the other tiles provably hold no row that reaches any of the block's bounds
    const int64_t ntiles_blk = tile_count ? (int64_t)tile_count[rblk] : ntiles_all;
    const int64_t t_begin = part_fixed >= 0 ? tile0 : ntiles_blk * part / parts, ntiles = part_fixed >= 0 ? (tile1 < ntiles_all ? tile1 : ntiles_all) : ntiles_blk * (part + 1) / parts;
    typedef const __attribute__((address_space(1))) void *gptr_t;
    typedef __attribute__((address_space(3))) void *lptr_t;
    constexpr int NPIECE = CHUNKS_PER_THREAD + 1 + (FMT == FMT_FP8 ? 1 : 0);   // DMA instructions per wave per tile (then -0.5|y|^2, then the fp8 row scales)
    // per piece: rows advance by NT/CPR (a multiple of 16, so the swizzle term is the same for every piece): the source address is a
    // wave-uniform tile base (SGPR pair) + one 32-bit lane offset -> 2 VALU per piece instead of a 64-bit address chain
    constexpr int ROWS_PER_PIECE = NT / CPR;
    int dma_r0 = (wid * 64 + lane) / CPR;
    uint32_t dma_choff = (uint32_t)((((wid * 64 + lane) % CPR) ^ swz<ROWB>(dma_r0)) << 4);
    asm volatile("" : "+v"(dma_r0), "+v"(dma_choff));
    auto issue_piece = [&](int64_t t, int buf, int i) {
        if (i < CHUNKS_PER_THREAD) {
            const int q0 = wid * 64 + NT * i;                     // wave-uniform first chunk of this instruction
            if (q0 < BN * CPR) {
                const char *tbase = reinterpret_cast<const char *>(Y) + t * (int64_t)TILE_BYTES;
                const int64_t left = M - t * BN;                  // rows of Y in this tile (rows past the end re-read the last one; their
                const int last = (int)(left < BN ? left : BN) - 1; // -0.5|y|^2 is -inf, so they are never selected)
                // off = min(dma_r0 + ROWS_PER_PIECE*i, last) * ROWB + dma_choff, formed HERE in three VALU ops: as plain C++ the compiler
                // hoists the four per-piece offsets (or 64-bit addresses) to the top of the tile and keeps them in VGPRs this kernel does not have
                uint32_t off;
                asm volatile("v_add_u32 %0, %1, %2\n\tv_min_i32 %0, %0, %3\n\tv_mad_u32_u24 %0, %0, %4, %5"
                             : "=&v"(off) : "v"(dma_r0), "s"(ROWS_PER_PIECE * i), "s"(last), "s"((int)ROWB), "v"(dma_choff));
                __builtin_amdgcn_global_load_lds((gptr_t)(tbase + off), (lptr_t)(ytile + buf * TILE_BYTES + q0 * 16), 16, 0, 0);
            }
        } else if (i == CHUNKS_PER_THREAD) {
            if (wid == 0 && lane < BN)                            // -0.5|y|^2 of the tile rows (padded rows hold -inf: never selected)
                __builtin_amdgcn_global_load_lds((gptr_t)(ynh + t * BN + lane), (lptr_t)(ynh_t + buf * BN), 4, 0, 0);
        } else if (FMT == FMT_FP8 && wid == 1 && lane < BN) {     // E8M0 scales of the tile rows (rows past the end: the last row's, never selected)
            int64_t yr = t * BN + lane;
            if (yr >= M) yr = M - 1;
            __builtin_amdgcn_global_load_lds((gptr_t)(ysc + yr), (lptr_t)(ysc_t + buf * BN), 4, 0, 0);
        }
    };
    auto issue_tile = [&](int64_t t, int buf) {
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) issue_piece(t, buf, i);
    };

    if (threadIdx.x < 4) vote_ring[threadIdx.x] = 0;
    // tile_order (optional, [row blocks][all tiles]): the ORDER in which this row block takes the style tiles — position -> tile.  The loop runs over
    // POSITIONS (LDS buffer parity, vote ring, the split of a sweep into parts); a position's tile id is needed one tile ahead for the LDS-DMA, so it is
    // fetched two positions ahead by an ordinary global load (uniform address) that has a whole tile's time to land.
    const int32_t *tord = tile_order ? tile_order + rblk * ntiles_all : nullptr;
    auto tile_of = [&](int64_t pos) -> int64_t { return tord ? (int64_t)__builtin_amdgcn_readfirstlane(tord[pos < ntiles_all ? pos : ntiles_all - 1]) : pos; };
    int64_t t_cur = tile_of(t_begin), t_nxt = tile_of(t_begin + 1);
    int pref = tord ? tord[t_begin + 2 < ntiles_all ? t_begin + 2 : ntiles_all - 1] : 0;
    if (t_begin < ntiles) issue_tile(t_cur, (int)(t_begin & 1));        // (a slice of a pruned block's list can be empty: nothing to fetch, nothing left in flight at exit)

    // Round 3 — the two waves of a SIMD run OUT OF PHASE.  The matrix pipe of a SIMD is shared by two waves of this block (wave w and wave w + 4); per tile each of
    // them has two MFMA phases (the tile's 32-row halves, ~1000 pipe cycles each) and two selection epilogues (VALU / LDS, 300-600 cycles each, more in tiles that
    // hold candidates).  With the tile barrier at the top of the iteration both waves start every phase together: they share the pipe through both MFMA phases and
    // then sit in their epilogues TOGETHER while the pipe idles — per tile 4 MFMA phases + 2 exposed epilogues (measured: 2.57 us for a tile without candidates,
    // 3.10 us for one with, against 1.7 us of MFMA issue; profiles/r03a_prune_diag.log).  Now:
    //   * priorities order the four MFMA phases of a SIMD and tile:  A.half0 (3) > B.half0 (2) > A.half1 (1) > B.half1 (0)   (A = waves 0-3, B = waves 4-7), so
    //     that each wave's epilogue falls under the OTHER wave's MFMA phase:  A.M0 | B.M0 + A.E0 | A.M1 + B.E0 | B.M1 + A.E1 | (next tile) A.M0 + B.E1 ...
    //   * the tile barrier (tile t's buffer is free; tile t + 1 has landed) only needs every wave's MFMA phases of tile t: it sits AFTER a wave's last MFMA phase —
    //     for the A waves after their last epilogue as well (they are ahead), for the B waves BEFORE theirs, which then runs under the A waves' first MFMA phase of
    //     the next tile.
    // Tuning bit 0x4000000: all waves as A with one priority (round 2's lock step, barrier at the tile's end instead of its top).
    const bool skew = !(mode_arg & 0x4000000);
    const bool grpB = skew && wid >= NWAVES / 2;                         // wave-uniform
    auto tile_barrier = [&](int vs) {
        // block-wide vote: if any wave's queue is past half, ALL waves merge at this tile's end — a lone wave flushing would stall the other seven at the next
        // barrier (a 3-slot flag ring in LDS: slot t % 3 is raised before barrier t, read after it, cleared two iterations ahead by one thread — one s_barrier per tile)
        if (SEL == 0 && qcnt > vote_at && lane == 0) vote_ring[vs] = 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // every wave drains its own LDS-DMA before the barrier (explicit: see knn_tile_prune_kernel)
        if (!(mode & 256)) __syncthreads();                              // tile t + 1 has landed
        if (threadIdx.x == 0) vote_ring[vs == 0 ? 2 : vs - 1] = 0;       // slot (t + 2) % 3
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                                     // the first tile has landed (and the vote ring is initialised)
    const int64_t clk_l0 = clk ? (int64_t)wall_clock64() : 0;            // diagnostics: the prologue ends here
    int vslot = (int)(t_begin % 3);                                      // t % 3 kept as a counter (a 64-bit modulo is ~30 scalar instructions per tile)
    for (int64_t pos = t_begin; pos < ntiles; ++pos, vslot = vslot == 2 ? 0 : vslot + 1) {
        const int cur = (int)(pos & 1);
        const int64_t t = t_cur;                                         // tile id of this position (== pos without a tile order)
        const int64_t t1 = tord ? t_nxt : pos + 1;                       // tile id of the next position (its DMA is issued during this one)
        // the DMA of tile t+1 is issued piecewise between the MFMAs of the first sub-block (hidden under the matrix pipe),
        // the vote is read at the END of the tile (no LDS round trip between the barrier and the first MFMA)
        const bool dma_next = (pos + 1 < ntiles) && !(mode & 64);

      for (int yb = 0; yb < BN / 32; ++yb) {       // 32-row sub-blocks of the tile
        // the wave that is in its MFMA phase wins the issue arbitration against the SIMD's other wave (which is in its selection epilogue or waiting at the
        // barrier): -0.5 ms per sweep at C3 (same-box A/B x3, profiles/README r02_d); tuning bit 0x4000 leaves the priority alone
        if (!(mode_arg & 0x4000)) {
            if (!skew) __builtin_amdgcn_s_setprio(3);
            else if (yb == 0) { if (grpB) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(3); }
            else { if (grpB) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(1); }
        }
        // ---- distance block: acc = -0.5|y|^2 + x.y for 32 style rows x the wave's 64 content rows.
        // SH = 0: 2 x v_mfma_f32_32x32x16 per k-step (lane: 16 candidates of row xb*32+xr, y = (j&3)+8(j>>2)+4h)
        // SH = 1: 8 x v_mfma_f32_16x16x32 per k-step (lane: 2x4 candidates of row xb*16+xr, y = 16(j>>2)+(j&3)+4h)
        //         same LDS bytes per flop; the chip holds a higher clock on this shape (MI355X_MICROARCH "DVFS give-back" (7))
        f32x16 acc0[SH ? 1 : XBN];
        f32x4 acc1[2][SH ? XBN : 1];
        // A fragments: slot of logical chunk c = HB*s+h of a row is c ^ swz.  Rows are 2*D bytes and tile-aligned, so the XOR
        // acts inside the row: ONE lane-constant byte offset, XORed with the compile-time (HB*s << 4) — no per-step address registers.
        //
        // Every LDS read of this block is an explicit ds_read_b128 with hand-placed s_waitcnt lgkmcnt(n) (LDS returns in order and no
        // scalar load is in flight inside the tile loop, so "all but the n youngest" is exact):
        //   * a compiler-visible LDS load makes the waitcnt pass assume it may alias the LDS-DMA of tile t+1 issued above and emit
        //     s_waitcnt vmcnt(0) — waiting for the NEXT tile's global loads before computing THIS one (buffers (cur^1) vs cur are
        //     disjoint and ordered by the barrier);
        //   * the pass also drains to lgkmcnt(0) at every other k-step, which caps the effective prefetch distance at one k-step;
        //     here fragment reads run PF k-steps ahead of the MFMAs that consume them.
        const uint32_t tbase = lds_addr(ytile + cur * TILE_BYTES + (yb * 32 + xr) * ROWB);
        uint32_t lane_off = (uint32_t)(((h * NCH) ^ swz<ROWB>(xr)) << 4);   // chunk NCH*(HB s + h) + p sits in slot (that ^ swz): XOR-linear in s, h, p
        asm volatile("" : "+v"(lane_off));                       // keep it one register (do not expand into KS precomputed addresses)
        const uint32_t ya = lds_addr(&ynh_t[cur * BN + yb * 32 + 4 * h]);      // -0.5|y|^2 of the sub-block rows this lane accumulates
        if constexpr (SH == 0 && FMT == FMT_FP8) {
            // fp8: a fragment is 32 B per lane = two ds_read_b128 (slots differ by XOR 16), one scaled 32x32x64 MFMA per content block
            f32x4 yn[4];
            int ysc_v;                                           // E8M0 scale of the style row this lane feeds (row xr of the sub-block)
            asm volatile("ds_read_b32 %0, %1" : "=v"(ysc_v) : "v"(lds_addr(&ysc_t[cur * BN + yb * 32 + xr])));
            asm volatile("ds_read_b128 %0, %1" : "=v"(yn[0]) : "v"(ya));
            asm volatile("ds_read_b128 %0, %1 offset:32" : "=v"(yn[1]) : "v"(ya));
            asm volatile("ds_read_b128 %0, %1 offset:64" : "=v"(yn[2]) : "v"(ya));
            asm volatile("ds_read_b128 %0, %1 offset:96" : "=v"(yn[3]) : "v"(ya));
            i32x4 afrag[PF + 1][2];
#pragma unroll
            for (int s = 0; s < PF; ++s) {
                const uint32_t ad = lane_off ^ (uint32_t)((HB * NCH * s) << 4);
                asm volatile("ds_read_b128 %0, %1" : "=v"(afrag[s][0]) : "v"(tbase + ad));
                asm volatile("ds_read_b128 %0, %1" : "=v"(afrag[s][1]) : "v"(tbase + (ad ^ 16u)));
            }
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if (dma_next && yb == 0) {
#pragma unroll
                    for (int i = 0; i < NPIECE; ++i) if ((i * KS) / NPIECE == s) issue_piece(t1, cur ^ 1, i);
                }
                if (s + PF < KS) {
                    const uint32_t ad = lane_off ^ (uint32_t)((HB * NCH * (s + PF)) << 4);
                    asm volatile("ds_read_b128 %0, %1" : "=v"(afrag[(s + PF) % (PF + 1)][0]) : "v"(tbase + ad));
                    asm volatile("ds_read_b128 %0, %1" : "=v"(afrag[(s + PF) % (PF + 1)][1]) : "v"(tbase + (ad ^ 16u)));
                }
                const int left = (KS - 1 - s) < PF ? (KS - 1 - s) : PF;      // fragments (2 reads each) younger than step s's
                if (s == 0) {
                    if (left == 2) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(yn[0]), "+v"(yn[1]), "+v"(yn[2]), "+v"(yn[3]), "+v"(ysc_v), "+v"(afrag[0][0]), "+v"(afrag[0][1]));
                    else if (left == 1) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(yn[0]), "+v"(yn[1]), "+v"(yn[2]), "+v"(yn[3]), "+v"(ysc_v), "+v"(afrag[0][0]), "+v"(afrag[0][1]));
                    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(yn[0]), "+v"(yn[1]), "+v"(yn[2]), "+v"(yn[3]), "+v"(ysc_v), "+v"(afrag[0][0]), "+v"(afrag[0][1]));
#pragma unroll
                    for (int rb = 0; rb < XBN; ++rb) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            acc0[rb][4 * g + 0] = yn[g][0]; acc0[rb][4 * g + 1] = yn[g][1]; acc0[rb][4 * g + 2] = yn[g][2]; acc0[rb][4 * g + 3] = yn[g][3];
                        }
                    }
                } else {
                    if (left == 2) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(afrag[s % (PF + 1)][0]), "+v"(afrag[s % (PF + 1)][1]));
                    else if (left == 1) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(afrag[s % (PF + 1)][0]), "+v"(afrag[s % (PF + 1)][1]));
                    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(afrag[s % (PF + 1)][0]), "+v"(afrag[s % (PF + 1)][1]));
                }
                const i32x4 lo = afrag[s % (PF + 1)][0], hi = afrag[s % (PF + 1)][1];
                const i32x8 a8 = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                for (int rb = 0; rb < XBN; ++rb)
                    acc0[rb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, xf[rb][s], acc0[rb], 0, 0, 0, ysc_v, 0, xscale[rb]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if constexpr (SH == 0) {
            f32x4 yn[4];
            asm volatile("ds_read_b128 %0, %1" : "=v"(yn[0]) : "v"(ya));
            asm volatile("ds_read_b128 %0, %1 offset:32" : "=v"(yn[1]) : "v"(ya));
            asm volatile("ds_read_b128 %0, %1 offset:64" : "=v"(yn[2]) : "v"(ya));
            asm volatile("ds_read_b128 %0, %1 offset:96" : "=v"(yn[3]) : "v"(ya));
            frag_t afrag[PF + 1];
            // 512-B rows: k-steps are taken in the order 0,8,1,9,...: steps s and s+8 differ only in address bit 8 (above the swizzled bits), so
            // the second read of a pair reuses the first one's address register with offset:256 — one address computation per two k-steps
            constexpr bool PAIR = ROWB == 512;
            auto kmap = [](int si) -> int { return PAIR ? ((si >> 1) | ((si & 1) << 3)) : si; };
            uint32_t pair_addr = 0;
            auto frag_read = [&](int si, frag_t &dst) {
                if (PAIR && (si & 1)) {
                    asm volatile("ds_read_b128 %0, %1 offset:256" : "=v"(dst) : "v"(pair_addr));
                } else {
                    pair_addr = tbase + (lane_off ^ (uint32_t)((HB * kmap(si)) << 4));
                    asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(pair_addr));
                }
            };
#pragma unroll
            for (int si = 0; si < PF; ++si) frag_read(si, afrag[si]);
#pragma unroll
            for (int si = 0; si < KS; ++si) {
                if (dma_next && yb == 0) {
#pragma unroll
                    for (int i = 0; i < NPIECE; ++i) if ((i * KS) / NPIECE == si) issue_piece(t1, cur ^ 1, i);
                }
                // order per k-step: wait for fragment si (the PF-1 younger reads may stay in flight) -> issue the read of fragment si+PF -> MFMAs.
                // The read between the wait and the MFMAs also keeps the compiler from padding "asm defines the fragment" -> MFMA with s_nop.
                const int left = (KS - 1 - si) < (PF - 1) ? (KS - 1 - si) : (PF - 1);      // reads younger than step si's fragment at the wait
                if (si == 0) {
                    if (left == 1) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(yn[0]), "+v"(yn[1]), "+v"(yn[2]), "+v"(yn[3]), "+v"(afrag[0]));
                    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(yn[0]), "+v"(yn[1]), "+v"(yn[2]), "+v"(yn[3]), "+v"(afrag[0]));
#pragma unroll
                    for (int rb = 0; rb < XBN; ++rb) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            acc0[rb][4 * g + 0] = yn[g][0]; acc0[rb][4 * g + 1] = yn[g][1]; acc0[rb][4 * g + 2] = yn[g][2]; acc0[rb][4 * g + 3] = yn[g][3];
                        }
                    }
                } else {
                    if (left == 1) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(afrag[si % (PF + 1)]));
                    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(afrag[si % (PF + 1)]));
                }
                if (si + PF < KS && !(mode & 512)) frag_read(si + PF, afrag[(si + PF) % (PF + 1)]);
#pragma unroll
                for (int rb = 0; rb < XBN; ++rb) {
                    if constexpr (FMT == FMT_F16) acc0[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afrag[si % (PF + 1)], xf[rb][kmap(si)], acc0[rb], 0, 0, 0);
                    else acc0[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag[si % (PF + 1)], xf[rb][kmap(si)], acc0[rb], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);           // keep read(si+PF) / wait(si) / MFMAs(si) in this order
            }
        } else {
            constexpr int UOFF = 16 * ROWB;                  // second 16-row half of the sub-block
            f32x4 yn[2];
            asm volatile("ds_read_b128 %0, %1" : "=v"(yn[0]) : "v"(ya));
            asm volatile("ds_read_b128 %0, %1 offset:64" : "=v"(yn[1]) : "v"(ya));
            frag_t afrag[PF + 1][2];
#pragma unroll
            for (int s = 0; s < PF; ++s) {
                const uint32_t ad = tbase + (lane_off ^ (uint32_t)((HB * s) << 4));
                asm volatile("ds_read_b128 %0, %1" : "=v"(afrag[s][0]) : "v"(ad));
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(afrag[s][1]) : "v"(ad), "n"(UOFF));
            }
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if (dma_next && yb == 0) {
#pragma unroll
                    for (int i = 0; i < NPIECE; ++i) if ((i * KS) / NPIECE == s) issue_piece(t1, cur ^ 1, i);
                }
                if (s + PF < KS) {
                    const uint32_t ad = tbase + (lane_off ^ (uint32_t)((HB * (s + PF)) << 4));
                    asm volatile("ds_read_b128 %0, %1" : "=v"(afrag[(s + PF) % (PF + 1)][0]) : "v"(ad));
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(afrag[(s + PF) % (PF + 1)][1]) : "v"(ad), "n"(UOFF));
                }
                const int left = (KS - 1 - s) < PF ? (KS - 1 - s) : PF;
                if (s == 0) {
                    if (left == 2) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(yn[0]), "+v"(yn[1]), "+v"(afrag[0][0]), "+v"(afrag[0][1]));
                    else if (left == 1) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(yn[0]), "+v"(yn[1]), "+v"(afrag[0][0]), "+v"(afrag[0][1]));
                    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(yn[0]), "+v"(yn[1]), "+v"(afrag[0][0]), "+v"(afrag[0][1]));
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
#pragma unroll
                        for (int xb = 0; xb < XBN; ++xb) acc1[u][xb] = yn[u];
                    }
                } else {
                    if (left == 2) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(afrag[s % (PF + 1)][0]), "+v"(afrag[s % (PF + 1)][1]));
                    else if (left == 1) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(afrag[s % (PF + 1)][0]), "+v"(afrag[s % (PF + 1)][1]));
                    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(afrag[s % (PF + 1)][0]), "+v"(afrag[s % (PF + 1)][1]));
                }
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int xb = 0; xb < XBN; ++xb) {
                        if constexpr (FMT == FMT_F16) acc1[u][xb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(afrag[s % (PF + 1)][u], xf[xb][s], acc1[u][xb], 0, 0, 0);
                        else if constexpr (FMT == FMT_BF16) acc1[u][xb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag[s % (PF + 1)][u], xf[xb][s], acc1[u][xb], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if constexpr (SH == 0) {                                 // all accumulators are complete HERE (keeps the per-block MFMA chains from being sunk into their epilogues)
#pragma unroll
            for (int rb = 0; rb < XBN; ++rb) asm volatile("" : "+v"(acc0[rb]));
        }
        if (!(mode_arg & 0x4000)) __builtin_amdgcn_s_setprio(0);
        if (yb == BN / 32 - 1 && grpB) tile_barrier(vslot);               // B waves: barrier before their last epilogue (it runs under the A waves' next MFMA phase)
        auto cand = [&](int xb, int j) -> float { if constexpr (SH == 0) return acc0[xb][j]; else return acc1[j >> 2][xb][j & 3]; };
        auto yoff = [](int j) -> uint32_t { return SH == 0 ? (uint32_t)((j & 3) + 8 * (j >> 2)) : (uint32_t)(16 * (j >> 2) + (j & 3)); };

        // ---- selection epilogue: one compare + one scalar branch per accumulator register; appends only where a lane beats its row's threshold
        const uint32_t ybase = (uint32_t)(t * BN) + yb * 32 + 4 * h;
        // One accumulator register's candidates (lanes whose score reaches their row's threshold).  The first k candidates a row ever sees go straight
        // into its list: slot = the row's candidate count so far (one LDS atomic), one 8-B store — no queue, no sort, no merge; only candidates that find
        // the list full (a few per row and sweep under a warm-start bound) take the queue to be compared with the list's worst entry at the next merge.
        // Returns false when the queue could not take the register's candidates if all of them overflowed (decided BEFORE any count is touched, so the
        // caller can merge and retry).
        auto take = [&](int rb, int r, uint32_t rowtag, uint32_t yidx) __attribute__((always_inline)) -> bool {
            const float v = cand(rb, r);
            const bool hit = v >= thr[rb];
            const unsigned long long mask = __builtin_amdgcn_ballot_w64(hit);
            if (__builtin_expect(mask == 0ull, 1)) return true;
            const int q0 = __builtin_amdgcn_readfirstlane(qcnt);                    // wave-uniform by construction: keep the bookkeeping scalar
            if (q0 + __builtin_popcountll(mask) > QE) return false;
            if constexpr (!DF || SEL == 1) {                                         // every candidate takes the queue
                const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
                if (hit) queue[q0 + rank] = Entry{v, rowtag | yidx};
                qcnt = q0 + __builtin_popcountll(mask);
                return true;
            }
            uint32_t rowl = rowtag >> 26;
            asm volatile("" : "+v"(rowl));             // formed HERE (rare path): hoisted out of the tile loop it would cost registers the kernel does not have
            uint32_t slot = 0u;
            if (hit) slot = atomicAdd(&rowcnt[rowl], 1u);
            const bool direct = hit && slot < (uint32_t)k;
            if (direct) lists[slot * ROWS_W + rowl] = Entry{v, yidx};
            const unsigned long long over = __builtin_amdgcn_ballot_w64(hit && !direct);
            if (__builtin_expect(over != 0ull, 0)) {
                const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(over >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)over, 0));
                if (hit && !direct) queue[q0 + rank] = Entry{v, rowtag | yidx};
                qcnt = q0 + __builtin_popcountll(over);
            }
            return true;
        };
#pragma unroll
        for (int rb = 0; rb < XBN; ++rb) {
            if (mode & 1) continue;
            uint32_t rowtag = (uint32_t)(rb * XBR + xr) << 26;
            asm volatile("" : "+v"(rowtag));                     // one register; the per-register tags are formed only on the (rare) append path
            uint32_t pending = 0u;                   // wave-uniform bitmask of registers that did not fit into the queue
            // most blocks have no candidate at all once the bounds are tight: a max3 tree + 1 compare instead of NR compare+branch
            float m4[4];                              // four independent chains (not one serial max chain)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                m4[q] = cand(rb, q);
#pragma unroll
                for (int j = q + 4; j < NR; j += 4) m4[q] = fmaxf(m4[q], cand(rb, j));
            }
            {
                const float mx = fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));
                if constexpr (SEL == 1) {
                    // cached lower bounds: the best score of this unit per content row, max-reduced over the lanes that share the row (h) and over the tile's units in LDS
                    // (ds_max_f32, no return value: nothing waits for it)
                    if (lb) asm volatile("ds_max_f32 %0, %1" :: "v"(lds_addr(rowmax + rb * XBR + xr)), "v"(mx) : "memory");
                }
                if (!(mode & 1024) && __builtin_amdgcn_ballot_w64(mx >= thr[rb]) == 0ull) continue;        // debug bit 1024: always scan the registers
            }
            if constexpr (SEL == 1) {
                // EMIT, straight line: under the cached lower bounds nearly every unit the sweep still visits holds candidates (1.4 per unit at C3, often several), so the
                // quarter tests and per-register branches of the list sweep — seven VALU -> ballot -> scalar-branch round trips for one candidate — are replaced by one
                // compare + ballot + mbcnt + exec-masked queue store per register, no branch: ~8 issue slots per register that co-issue with the partner wave's MFMAs.
                // The store is asm: a compiler-visible LDS store makes the waitcnt pass drain vmcnt(0) first (it cannot tell the queue from the tile ring the LDS-DMA
                // of the next tile is writing), i.e. wait for the next tile in the FIRST sub-block's epilogue instead of at the tile barrier.
                if (__builtin_amdgcn_readfirstlane(qcnt) + NR * 64 > QE) flush();      // (rare; wave-uniform) room for a unit's worst case
                int q = __builtin_amdgcn_readfirstlane(qcnt);
                const uint32_t qbase = lds_addr(queue);
                // (the chain maxima of the pretest say which quarter of the registers — r mod 4 — holds a candidate: a unit has 1.4 on average, so 4 quarter tests + the
                // registers of ~1.3 quarters instead of all NR register steps, each of which costs a VALU -> exec -> scalar-branch round trip even when nobody hits)
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    if (__builtin_amdgcn_ballot_w64(m4[qq] >= thr[rb]) == 0ull) continue;
#pragma unroll
                    for (int r = qq; r < NR; r += 4) {
                        const float v = cand(rb, r);
                        const bool hit = v >= thr[rb];
                        const unsigned long long mask = __builtin_amdgcn_ballot_w64(hit);
                        const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
                        if (hit) {
                            const uint64_t ent = (uint64_t)__float_as_uint(v) | ((uint64_t)(rowtag | (ybase + yoff(r))) << 32);
                            asm volatile("ds_write_b64 %0, %1" :: "v"(qbase + (uint32_t)(q + rank) * 8u), "v"(ent) : "memory");
                        }
                        q += __builtin_popcountll(mask);
                    }
                }
                qcnt = q;
                continue;
            }
            // a block with candidates usually has ONE: the chain maxima say which quarter of the registers (r mod 4) holds it — 4 + NR/4 tests
            // instead of NR (with neighbours spread over the sweep ~1/4 of the blocks come here, and every wave waits for the slowest at the tile barrier)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (!(mode & 1024) && __builtin_amdgcn_ballot_w64(m4[q] >= thr[rb]) == 0ull) continue;
#pragma unroll
                for (int r = q; r < NR; r += 4) {
                    if (!take(rb, r, rowtag, ybase + yoff(r))) pending |= (1u << r);
                }
            }
            while (__builtin_expect(pending != 0u, 0)) {      // rare: merge the queue, then retry the registers that did not fit
                flush();
                uint32_t again = 0u;
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    if ((pending >> r) & 1u) {               // retry through the queue alone (the merge files a candidate whose list is not full yet)
                        const bool hit = cand(rb, r) >= thr[rb];
                        const unsigned long long mask = __builtin_amdgcn_ballot_w64(hit);
                        const int n = __builtin_popcountll(mask);
                        if (qcnt + n > QE) {
                            again |= (1u << r);
                        } else if (n) {
                            const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
                            if (hit) queue[qcnt + rank] = Entry{cand(rb, r), rowtag | (ybase + yoff(r))};
                            qcnt += n;
                        }
                    }
                }
                pending = again;
            }
        }

        if (yb == BN / 32 - 1 && !grpB) tile_barrier(vslot);              // A waves: after their last epilogue
        if constexpr (SEL == 1) {
            // the tile's lower bounds leave AFTER the wave's tile barrier (A waves: just passed; B waves: before their last epilogue), so that the store has a whole
            // tile's time before the next barrier's vmcnt(0) waits for it
            if (yb == BN / 32 - 1 && lb) {
                float mv;
                const uint32_t ra = lds_addr(rowmax + lane);
                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(mv) : "v"(ra) : "memory");
                asm volatile("ds_write_b32 %0, %1" :: "v"(ra), "v"(-__builtin_inff()) : "memory");
                const float m = __builtin_sqrtf(fmaxf(lbA - 2.0f * mv, 0.f));
                const _Float16 hv = (_Float16)fminf((m + lbC) * 0.9990234375f, 60000.f);       // rounded DOWN: the conversion's half ulp (2^-11) is inside the factor 1 - 2^-10
                uint16_t hb; __builtin_memcpy(&hb, &hv, 2);
                lb[((rblk * ntiles_all + t) * ROWS_B) + wid * ROWS_W + lane] = hb;
            }
        }
      }
        // block-wide vote (raised before this tile's barrier): if any wave's queue was past half, ALL waves merge now
        if constexpr (SEL == 0) { if (!(mode & 128) && vote_ring[vslot] && !(mode & 4)) flush(true); }
        if (tord) {                                                      // advance: next tile becomes current, the prefetched id the next one; fetch the id after that
            t_cur = t_nxt;
            t_nxt = (int64_t)__builtin_amdgcn_readfirstlane(pref);
            pref = tord[pos + 3 < ntiles_all ? pos + 3 : ntiles_all - 1];
        } else {
            t_cur = pos + 1;
        }
    }
    flush();
    const int64_t clk_l1 = clk ? (int64_t)wall_clock64() : 0;            // diagnostics: the tile loop (and the last merge) ends here

    if constexpr (SEL == 0) knn_write_out(lists, rcnt, lane, lcount, k, N, wave_row0, rowblk0, parts, part, parts > 1 || part_fixed >= 0, xnorm, row_perm, flags, out_idx, out_d2);
    if (clk && threadIdx.x == 0) {
        uint32_t hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        const int64_t ci = rblk_given >= 0 ? rblk_given * parts + part : (int64_t)blockIdx.x;
        clk[8 * ci + 0] = clk_t0;
        clk[8 * ci + 1] = (int64_t)wall_clock64();
        clk[8 * ci + 2] = ((int64_t)(xcc & 0xf) << 32) | (int64_t)hwid;
        clk[8 * ci + 3] = (int64_t)(ntiles - t_begin);
        clk[8 * ci + 4] = clk_c0;                                        // shader clock (s_memtime) at block start / end: in-kernel clock = d(memtime) / d(wall clock) x 100 MHz
        clk[8 * ci + 5] = (int64_t)__builtin_amdgcn_s_memtime();
        clk[8 * ci + 6] = clk_l0;                                        // prologue | tile loop | write-out
        clk[8 * ci + 7] = clk_l1;
    }
}

// Two entry points on the same body: the sweep proper, and the repair launch (row count on the device) under its own name so that profiler
// statistics of the sweep are not diluted by launches that normally leave at once.
#define W3D_KNN_ARGS const uint8_t *__restrict__ X, const uint8_t *__restrict__ Y, const int32_t *__restrict__ xsc, const int32_t *__restrict__ ysc, \
    const float *__restrict__ xnorm, const float *__restrict__ ynh, const float *__restrict__ thr_init, int64_t N, int64_t M, int k,                 \
    int32_t *__restrict__ out_idx, float *__restrict__ out_d2, int mode_arg, int64_t rowblk0, int parts, int32_t *__restrict__ flags
template <int D, int BN, int QCAP, int SH, int FMT>
__global__ __launch_bounds__(NWAVES * 64, 2) void knn_mfma_kernel(W3D_KNN_ARGS, const int32_t *__restrict__ row_perm, const int32_t *__restrict__ guard) {
    if (guard && guard[0] <= KNN_FLAG_CAP) return;
    knn_mfma_body<D, BN, QCAP, SH, FMT>(X, Y, xsc, ysc, xnorm, ynh, thr_init, N, M, k, out_idx, out_d2, mode_arg, rowblk0, parts, flags, nullptr, -1, -1, -1, row_perm);
}
// the sweep without a warm-start bound (no direct list fill: see DF in the body)
template <int D, int BN, int QCAP, int SH, int FMT>
// guard != NULL: the launch is a FALLBACK — it runs only if guard[0] (rows the call before it flagged for the repair) exceeds what the repair can take (cold emit sweep)
__global__ __launch_bounds__(NWAVES * 64, 2) void knn_mfma_cold_kernel(W3D_KNN_ARGS, const int32_t *__restrict__ row_perm, const int32_t *__restrict__ guard) {
    if (guard && guard[0] <= KNN_FLAG_CAP) return;
    knn_mfma_body<D, BN, QCAP, SH, FMT, false>(X, Y, xsc, ysc, xnorm, ynh, thr_init, N, M, k, out_idx, out_d2, mode_arg, rowblk0, parts, flags, nullptr, -1, -1, -1, row_perm);
}
// the same sweep with a per-row-block tile order ("hot tiles first"); its own entry so that the plain sweep compiles without the indirection
template <int D, int BN, int QCAP, int SH, int FMT>
__global__ __launch_bounds__(NWAVES * 64, 2) void knn_mfma_hot_kernel(W3D_KNN_ARGS, const int32_t *__restrict__ row_perm, const int32_t *__restrict__ tile_order,
                                                                      const int32_t *__restrict__ tile_count, const int32_t *__restrict__ block_order, int64_t *__restrict__ clk) {
    const int64_t nblk_all = (N + ROWS_B - 1) / ROWS_B;
    knn_mfma_body<D, BN, QCAP, SH, FMT>(X, Y, xsc, ysc, xnorm, ynh, thr_init, N, M, k, out_idx, out_d2, mode_arg, rowblk0, parts, flags, nullptr, -1, -1, -1, row_perm, tile_order,
                                        tile_count, block_order, clk, -1, -1, nullptr, nullptr, nullptr, nullptr, 0.f, tile_count[2 * nblk_all + KNN_NQ] == 0);
}
// PERSISTENT pruned sweep: one workgroup per CU for the whole launch; its items — row blocks, or (row block, slice) pairs of a short sweep — come from eight work queues,
// queue q = positions q, q + 8, ... of the plan's longest-first block order (knn_block_order_kernel), a workgroup's home queue = its XCD.  Why: no dispatcher turnaround between blocks (a 155-KB-LDS workgroup is relaunched only after the previous one has fully
// retired), and the balance is dynamic — a workgroup whose queue has run dry steals from the other XCDs' queues.  8.45 vs 8.60 ms per pruned sweep at C3 against one
// workgroup per row block (same device, profiles/r03b_prune_diag.log).  Same body, same arithmetic, same results.
template <int D, int BN, int QCAP, int SH, int FMT, bool DF = true, int SEL = 0>
__global__ __launch_bounds__(NWAVES * 64, 2) void knn_mfma_persistent_kernel(W3D_KNN_ARGS, const int32_t *__restrict__ row_perm, const int32_t *__restrict__ tile_order,
                                                                             const int32_t *__restrict__ tile_count, const int32_t *__restrict__ block_order,
                                                                             int32_t *__restrict__ qcursor, int nitems, int64_t *__restrict__ clk,
                                                                             Entry *__restrict__ cand_buf, int32_t *__restrict__ cand_cnt, uint16_t *__restrict__ lb,
                                                                             const float *__restrict__ lb_cum, float lb_margin) {
    // queue q = positions q, q + 8, q + 16, ... of the longest-first block order: every XCD gets the same mix of long and short blocks
    __shared__ int s_item;
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const int home = (int)(xcc & (KNN_NQ - 1));
    for (;;) {
        if (threadIdx.x == 0) {
            int item = -1;
            for (int dq = 0; dq < KNN_NQ && item < 0; ++dq) {          // own queue first, then the others' (work stealing at the end of the launch)
                const int q = (home + dq) & (KNN_NQ - 1);
                const int len = (nitems - q + KNN_NQ - 1) / KNN_NQ;
                if (__hip_atomic_load(&qcursor[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= len) continue;
                const int j = __hip_atomic_fetch_add(&qcursor[q], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (j < len) item = j * KNN_NQ + q;
            }
            s_item = item;
        }
        __syncthreads();
        const int item = s_item;
        if (item < 0) return;
        // an item is (row block, slice): parts == 1 — a whole row block; parts > 1 (a short sweep, e.g. a rank's share: launch()) — slice item % parts of block item / parts
        const int ob = parts > 1 ? item / parts : item;
        knn_mfma_body<D, BN, QCAP, SH, FMT, DF, SEL>(X, Y, xsc, ysc, xnorm, ynh, thr_init, N, M, k, out_idx, out_d2, mode_arg, rowblk0, parts, flags, nullptr, -1, -1, -1, row_perm,
                                                     tile_order, tile_count, nullptr, clk, (int64_t)(block_order ? block_order[ob] : ob), parts > 1 ? item % parts : -1,
                                                     cand_buf, cand_cnt, lb, lb_cum, lb_margin, qcursor[KNN_NQ] == 0);      // the word after the cursors: 1 = the plan's tile order is not ascending
        __syncthreads();                                                 // the block's LDS (tile ring, lists, s_item) is reused by the next row block
    }
}
// ===================================================================================================================================================================
// Round 5 — the EMIT sweep as a kernel of its own (16-bit rows, 16x16x32 MFMAs, 64-row tiles; the fp8 rows keep the SEL = 1 instantiation of the body above).
// Same work item, same tile ring, same MFMA phase, same A / B wave roles, same candidate-buffer regions and the same knn_select_kernel behind it — but written for
// what the pruned loop's tiles look like since the cached lower bounds: nearly EVERY 16 x 32 unit a block still sweeps holds a candidate (1.4-1.9 per unit), so the
// "rare" path of the body above is the common one here, and that path was a chain of VALU -> SGPR -> scalar-branch round trips (per candidate register: compare,
// saveexec + branch, two mbcnt, address, store, restore, popcount, add: ~14 instructions and three dependent hops; measured per wave and tile 193 vector + 176 scalar
// instructions beside the 128 MFMAs, the wave's selection epilogue ~2 000 cycles against the partner wave's 1 024-cycle MFMA phase it is meant to hide under:
// profiles/r04t_inst_mix.log, r05a anatomy).  What changed:
//   * per-LANE candidate queues ([slot][lane] in LDS, 8 B entries): a lane appends its own candidates, no ballot / mbcnt rank, no scalar queue cursor.  The append is
//     "store always, advance if hit": ds_write2_b32 (score, tag) to the lane's next free slot and cnt += (score >= bound) through the carry of v_addc — 5 vector
//     instructions per register, no branch, nothing scalar; a miss only scribbles on a free slot.  Lanes outside the quarter's hit mask are switched off (one
//     s_and_saveexec per quarter), so the stores cost no LDS bandwidth to speak of.
//   * the pretests of a sub-block's four units — max tree, four quarter compares into SGPR pairs (ballots), the unit maximum for the cached bounds — are issued back
//     to back BEFORE any scalar looks at them; the scan that follows branches on masks that are long complete (one exposed VALU -> scalar hop per sub-block, not ten).
//   * nothing in the tile loop is decided at run time that is known at compile time: no mode bits, no vote ring, no parts / chunk / repair cases, 32-bit positions,
//     the two tile buffers and the two sub-blocks as four unrolled bodies (every A-fragment address is a lane constant + an instruction offset: no address VALU at
//     all), full tiles take their LDS-DMA source offsets from one lane constant.
// A queue that could overflow in the next unit (any lane beyond EMQ_SLOTS - 8 entries) is copied to the wave's region first (one atomic per spill, as before).
// Results: candidates reach knn_select_kernel in another order — it buckets them by row and rank-sorts, so the lists are bit-equal to the list sweep's
// (test_knn_emit_sweep_equals_the_list_sweep and the loop tests hold that).
constexpr int EMQ_SLOTS = 22;            // per-lane queue slots (LDS per wave: EMQ_SLOTS * 512 B; 158.5 KB per block at d = 256)
// REGISTER BUDGET: the loop launches the render leg's Adam step (adam_lean_kernel, 32 VGPRs, no LDS) on a second stream while this kernel holds two waves on every SIMD; it is
// co-resident only while 2 x alloc(VGPRs of this kernel) + 32 <= 512, i.e. <= 240 VGPRs (allocation granule 8).  Round 6 found the LB instantiation at 241 — the Adam then ran AFTER
// the sweep, next to the select kernel — and moved the two per-row constants of the cached lower bounds (lbA, lbC) to the wave's LDS area, read once per tile: 240 again
// (tests/test_kernel_resources_cpu.py reads the built code objects and holds it there; amdgpu_num_vgpr / waves_per_eu attributes do not: the compiler ignores a cap above its own need).
// compile-time loop: f(std::integral_constant<int, I>) for I = I0 ... N - 1 (indices that end up in "n" asm operands / instruction offsets)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
// SH: the MFMA shape, as in knn_mfma_body — 1 = v_mfma_f32_16x16x32 (a unit = 16 content rows x 32 style rows, 8 scores per lane, 4 units per sub-block),
//                                           0 = v_mfma_f32_32x32x16 (a unit = 32 x 32, 16 scores per lane, 2 units per sub-block)
// COLD (round 5): the sweep WITHOUT a bound, on the same machinery — see "cold emit sweep" below the kernel's flush.
template <int D, int FMT, bool LB, int SH, bool COLD = false>
__global__ __launch_bounds__(NWAVES * 64, 2) void knn_emit_kernel(const uint8_t *__restrict__ X, const uint8_t *__restrict__ Y, const int32_t *__restrict__ xsc,
                                                                  const int32_t *__restrict__ ysc, const float *__restrict__ xnorm,
                                                                  const float *__restrict__ ynh, const float *__restrict__ thr_init, int N, int M,
                                                                  const int32_t *__restrict__ row_perm, const int32_t *__restrict__ tile_order,
                                                                  const int32_t *__restrict__ tile_count, const int32_t *__restrict__ block_order,
                                                                  int32_t *__restrict__ qcursor, int nitems, int parts, Entry *__restrict__ cand_buf,
                                                                  int32_t *__restrict__ cand_cnt, uint16_t *__restrict__ lb, const float *__restrict__ lb_cum, float lb_margin,
                                                                  int64_t *__restrict__ clk, int cold_tm, int cold_t0) {
    static_assert(!COLD || (SH == 1 && !LB && FMT != FMT_FP8), "the cold emit sweep: 16-bit rows on 16x16x32, no cached bounds");
    // FMT_FP8 (round 5): e4m3 rows with one E8M0 scale per row on v_mfma_scale_f32_32x32x64_f8f6f4 (twice the bf16 rate): the 32 x 32 unit layout (SH = 0), a fragment =
    // 32 bytes per lane = two ds_read_b128 whose slots differ by XOR 16, the tile rows' scales as one more LDS-DMA piece, xsc / ysc the rows' scale bytes
    constexpr bool FP8 = FMT == FMT_FP8;
    static_assert(!FP8 || SH == 0, "the scaled fp8 MFMA is used in its 32x32x64 shape");
    constexpr int ROWB = D * Fmt<FMT>::EB, CPR = ROWB / 16, TILE_BYTES = 64 * ROWB, NT = NWAVES * 64;
    constexpr int FB = Fmt<FMT>::FB, NCH = FB / 16;                      // operand bytes per lane per MFMA (16 / fp8: 32), ds_read_b128 per fragment half
    constexpr int XBR = SH ? 16 : 32;                                    // content rows per MFMA (columns of its output block) = rows of a unit
    constexpr int XBN = 64 / XBR;                                        // units per wave and sub-block (4 / 2)
    constexpr int HB = 64 / XBR;                                         // k-groups across the lanes of one operand fragment (4 / 2)
    constexpr int KS = ROWB / (FB * HB);                                 // MFMA k-steps per tile (d = 256: 8 / 16; fp8: 4)
    constexpr int NRQ = SH ? 2 : 4;                                      // a lane's scores per unit and quarter (style rows q + 4 h + {0, 16} / + {0, 8, 16, 24})
    constexpr int PIECES = (64 * CPR + NT - 1) / NT;                     // LDS-DMA instructions per wave and tile (4 / 2 / 1)
    constexpr int RPP = NT / CPR;                                        // tile rows one DMA instruction of the block covers (16 / 32 / 64)
#ifndef W3D_EMIT_PF
#define W3D_EMIT_PF 2
#endif
    constexpr int PF = W3D_EMIT_PF < KS ? W3D_EMIT_PF : KS - 1;          // fragment reads run PF k-steps ahead of the MFMAs that consume them
    constexpr int SHL = (SH || FP8) ? 6 : 5;                             // a k-step advances the fragment's byte offset by FB HB
    constexpr int FLD = ROWB >= 256 ? 8 : 7;                             // the swizzle XORs address bits 4 ... FLD - 1
    constexpr int NJ = KS < (1 << (FLD - SHL)) ? KS : (1 << (FLD - SHL));   // distinct swizzled fragment addresses; k-steps s and s + NJ differ by an instruction offset
    typedef typename std::conditional<FMT == FMT_F16, f16x8, bf16x8>::type frag16_t;
    typedef typename std::conditional<FP8, i32x8, frag16_t>::type frag_t;        // B operand (content rows, in registers)
    typedef typename std::conditional<FP8, i32x4, frag16_t>::type afrag_t;       // one ds_read_b128 of the A operand
    typedef const __attribute__((address_space(1))) void *gptr_t;
    typedef __attribute__((address_space(3))) void *lptr_t;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ int s_item;
    uint8_t *ytile = smem;                                               // [2][TILE_BYTES]
    float *ynh_t = reinterpret_cast<float *>(smem + 2 * TILE_BYTES);     // [2][64]
    int *ysc_t = reinterpret_cast<int *>(smem + 2 * TILE_BYTES + 2 * 64 * 4);      // [2][64] (fp8: the tile rows' E8M0 scales)
    constexpr int SCB = FP8 ? 2 * 64 * 4 : 0;
    constexpr int PER_WAVE = EMQ_SLOTS * 512 + 3 * ROWS_W * 4;
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint8_t *wbase = smem + 2 * TILE_BYTES + 2 * 64 * 4 + SCB + 16 + wid * PER_WAVE;
    Entry *queue = reinterpret_cast<Entry *>(wbase);                     // [EMQ_SLOTS][64]
    float *rowmax = reinterpret_cast<float *>(wbase + EMQ_SLOTS * 512);  // [3][64]: per content row the best score of the current tile (cached lower bounds), then the row's two constants lbA, lbC (in LDS, not in registers: REGISTER BUDGET above)
    const int xr = lane & (XBR - 1), h = lane / XBR;
    const bool grpB = wid >= NWAVES / 2;                                 // wave-uniform: see "the two waves of a SIMD run OUT OF PHASE" in knn_mfma_body
    const int ntiles_all = (M + 63) >> 6;
    // ---- lane constants of the whole launch
    uint32_t abase[NJ];                                                  // A fragments: slot of logical chunk HB s + h of tile row (.. + xr), buffer 0, sub-block 0, first half
#pragma unroll
    for (int j = 0; j < NJ; ++j) abase[j] = lds_addr(ytile + xr * ROWB) + ((uint32_t)(((h * NCH) ^ swz<ROWB>(xr)) << 4) ^ (uint32_t)(j << SHL));
    uint32_t abase2[FP8 ? NJ : 1];                                       // fp8: the fragment's second 16 bytes (slot XOR 1)
#pragma unroll
    for (int j = 0; j < (FP8 ? NJ : 1); ++j)                             // (the XOR acts on the OFFSET inside the row, not on the sum with the ring's base address)
        abase2[j] = lds_addr(ytile + xr * ROWB) + (((uint32_t)(((h * NCH) ^ swz<ROWB>(xr)) << 4) ^ (uint32_t)(j << SHL)) ^ 16u);
    const uint32_t ysc_lane = lds_addr(ysc_t + xr);                      // fp8: scale of the style row this lane feeds (row xr of the sub-block)
    const uint32_t ya_lane = lds_addr(ynh_t + 4 * h);                    // -0.5 |y|^2 of tile rows 4 h ... 4 h + 3 (+ the unit's / sub-block's / buffer's rows as instruction offsets)
    const uint32_t rm_lane = lds_addr(rowmax + xr);                      // (+ unit * XBR * 4 bytes)
    const uint32_t q_lane = lds_addr(queue) + (uint32_t)lane * 8u;       // the lane's slot 0
    // LDS-DMA source of the lane's chunk in piece i of a tile: row min(dma_r0 + i RPP, last row of the tile)
    int dma_r0 = (wid * 64 + lane) / CPR;
    uint32_t dma_choff = (uint32_t)((((wid * 64 + lane) % CPR) ^ swz<ROWB>(dma_r0)) << 4);
    asm volatile("" : "+v"(dma_r0), "+v"(dma_choff));
    const uint32_t lane_tag = ((uint32_t)xr << 26) | (uint32_t)(4 * h);
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const int home = (int)(xcc & (KNN_NQ - 1));
    for (;;) {
        if (threadIdx.x == 0) {
            int item = -1;
            for (int dq = 0; dq < KNN_NQ && item < 0; ++dq) {          // own queue first, then the others' (work stealing at the end of the launch)
                const int q = (home + dq) & (KNN_NQ - 1);
                const int len = (nitems - q + KNN_NQ - 1) / KNN_NQ;
                if (__hip_atomic_load(&qcursor[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= len) continue;
                const int j = __hip_atomic_fetch_add(&qcursor[q], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (j < len) item = j * KNN_NQ + q;
            }
            s_item = item;
        }
        __syncthreads();
        const int item = __builtin_amdgcn_readfirstlane(s_item);
        if (item < 0) return;
        const int64_t clk_t0 = clk ? (int64_t)wall_clock64() : 0;
        const int64_t clk_c0 = clk ? (int64_t)__builtin_amdgcn_s_memtime() : 0;
        const int ob = parts > 1 ? item / parts : item, part = parts > 1 ? item % parts : 0;
        const int rblk = __builtin_amdgcn_readfirstlane(block_order ? block_order[ob] : ob);
        const int ntb = __builtin_amdgcn_readfirstlane(!tile_count ? ntiles_all : tile_count[rblk]);
        const int t_begin = (int)((int64_t)ntb * part / parts), t_end = (int)((int64_t)ntb * (part + 1) / parts);
        const int wave_row0 = rblk * ROWS_B + wid * ROWS_W;
        const int region = rblk * NWAVES + wid;
        // ---- X fragments (B operand): lane holds X[row = XBR xb + xr][(HB s + h) 16 B .. + 16 B] for every k-step s
        frag_t xf[XBN][KS];
        int xscale[XBN];                                                 // fp8: E8M0 scale of the lane's content row (scale operand of the B side)
#pragma unroll
        for (int xb = 0; xb < XBN; ++xb) xscale[xb] = 0;
#pragma unroll
        for (int xb = 0; xb < XBN; ++xb) {
            int row = wave_row0 + xb * XBR + xr;
            if (row >= N) row = N - 1;                                   // positions past the end: copies of the last row (their region's rows past N are never written out)
            if (row_perm) row = row_perm[row];
#pragma unroll
            for (int s = 0; s < KS; ++s) xf[xb][s] = *reinterpret_cast<const frag_t *>(X + (int64_t)row * ROWB + (HB * s + h) * FB);
            if constexpr (FP8) xscale[xb] = xsc[row];
        }
        float thr[XBN];
        {
            int r0 = wave_row0 + lane; r0 = r0 < N ? r0 : N - 1; if (row_perm) r0 = row_perm[r0];
            const float thr0 = COLD ? -__FLT_MAX__ : thr_init[r0];      // (cold: finite, so that the -inf scores of rows past the end of Y never reach it)
#pragma unroll
            for (int xb = 0; xb < XBN; ++xb) thr[xb] = __shfl(thr0, xb * XBR + xr, 64);
            if constexpr (LB) {
                rowmax[64 + lane] = xnorm[r0] * (1.0f - (FP8 ? KNN_SCORE_ERR2_FP8 : KNN_SCORE_ERR2)) - lb_margin;      // lbA = |x|^2 - 2E, E = the plan's bound on a computed score's error (knn_tile_prune_kernel)
                rowmax[128 + lane] = lb_cum[r0];                                                                         // lbC
                rowmax[lane] = -__builtin_inff();
            }
        }
#ifdef W3D_KNN_STAMP                                                      // diagnostic build (scripts/emit_stamp.py): shader cycles per wave in the phases of its tiles
        int64_t st_acc[6] = {0, 0, 0, 0, 0, 0};                          // M0, E0, M1, E1 (without the barrier), barrier wait, spills
        int64_t st_t = 0, st_tile = 0, st_pos[4] = {0, 0, 0, 0};         // whole tiles: the item's first, second, last-but-one, last (cycles)
        const int64_t st_item0 = (int64_t)__builtin_amdgcn_s_memtime();
#define W3D_STAMP_BEGIN() st_t = (int64_t)__builtin_amdgcn_s_memtime()
#define W3D_STAMP(i) do { const int64_t n_ = (int64_t)__builtin_amdgcn_s_memtime(); st_acc[i] += n_ - st_t; st_t = n_; } while (0)
#else
#define W3D_STAMP_BEGIN() do {} while (0)
#define W3D_STAMP(i) do {} while (0)
#endif
        uint32_t qcnt = 0u;                                              // entries in the lane's queue
        int roff = 0;                                                    // (parts == 1: the wave owns its region) entries the wave has written to it so far — wave-uniform
        const uint32_t q_lane_f = q_lane;
        // ---- cold emit sweep (COLD): no bound comes with the rows, so the sweep builds one as it goes.  A lane sees a fixed quarter of every tile's style rows (its h) for
        // each of its XBN content rows; per (lane, content row) it TRACKS the TM best scores of that quarter in registers.  Quarters are disjoint, so the smallest of the four
        // lanes' tm-th best (tm = ceil(k / 4) <= TM) is reached by at least k different style rows: a valid bound, refreshed from the trackers at every spill.  The trackers
        // are fed from the QUEUE (a spill walks the lanes' entries anyway, all lanes busy), not from the epilogue's masked blocks: an entry was emitted because it reached
        // the row's bound, which is never above the lane's own tm-th best — whatever was not emitted could not have entered a tracker, so the trackers are exact.
        // Three phases over the item's tiles: steps [0, T0) only build the trackers — no queue, no scan: every lane feeds the best score of each unit straight in (see the
        // epilogue) — so the storm of the first tiles, where every tile beats the last, never reaches the queues or the candidate buffer; steps [T0, n) track and emit;
        // then the first T0 tiles once more, emitting, with trackers that start over (see the phase change in the tile loop).
        // At C2 (1563 tiles, T0 = 128): 76 entries per row reach the candidate buffer; without the first phase 237 (any exact streaming selection admits
        // ~k (1 + ln(tiles * 64 / k)) = 200), at ~100 cycles of masked appends each.
        constexpr int TM = 5;
        float trk[COLD ? XBN : 1][TM];
#pragma unroll
        for (int xb = 0; xb < (COLD ? XBN : 1); ++xb)
#pragma unroll
            for (int i = 0; i < TM; ++i) trk[xb][i] = -__FLT_MAX__;
        float thr_base[COLD ? XBN : 1];                                  // (third phase) the bounds the second phase ended with
#pragma unroll
        for (int xb = 0; xb < (COLD ? XBN : 1); ++xb) thr_base[xb] = -__FLT_MAX__;
        bool emit_on = true, track_on = COLD;                            // wave-uniform
        auto flush = [&]() __attribute__((always_inline)) {
            // the wave's queues -> its region of the candidate buffer, slot level by slot level (level s = the lanes that hold more than s entries, compacted by
            // ballot + mbcnt: no cross-lane shuffles, stores of a level are contiguous).  A wave that owns its region keeps the write offset in a register and
            // stores the count once at the end of the item; the slices of a sliced sweep share a region and reserve their room with one atomic per spill.
            // The LDS reads are asm: a compiler-visible LDS load would first wait for the next tile's LDS-DMA (vmcnt(0)).
            Entry *dst = cand_buf + (int64_t)region * KNN_REGION_CAP;
            int base = roff;
            if constexpr (COLD) {
                // cold sweep (parts == 1): first the trackers see every queued entry and the rows' bounds are refreshed, then the entries that still reach their row's
                // (new) bound go to the region — in a storm of ever better tiles most of a queue is obsolete by the time it spills
                if (track_on) {
#pragma unroll 1
                    for (int sl = 0; sl < EMQ_SLOTS; ++sl) {
                        const bool has = qcnt > (uint32_t)sl;
                        if (__builtin_amdgcn_ballot_w64(has) == 0ull) break;
                        uint64_t e;
                        asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(e) : "v"(q_lane_f + (uint32_t)sl * 512u) : "memory");
                        const float ev = has ? __uint_as_float((uint32_t)e) : -__FLT_MAX__;
                        const uint32_t exb = (uint32_t)(e >> 62);       // the entry's content row = 16 xb + xr (bits 26 ... 31 of the tag): xb
#pragma unroll
                        for (int xb = 0; xb < XBN; ++xb) {
                            float x = exb == (uint32_t)xb ? ev : -__FLT_MAX__;
#pragma unroll
                            for (int i = 0; i < TM; ++i) { const float hi = fmaxf(trk[COLD ? xb : 0][i], x); x = fminf(trk[COLD ? xb : 0][i], x); trk[COLD ? xb : 0][i] = hi; }
                        }
                    }
                    // the rows' bounds from the trackers: min over the four lanes (h) that share the row
                    const uint32_t a16 = (uint32_t)(lane ^ 16) << 2, a32 = (uint32_t)(lane ^ 32) << 2;
#pragma unroll
                    for (int xb = 0; xb < XBN; ++xb) {
                        static_assert(TM == 5, "the tm-th best by explicit selects (a loop over the tracker turns into a dynamic index: scratch)");
                        const float *tk = trk[COLD ? xb : 0];
                        const float v01 = cold_tm == 1 ? tk[0] : tk[1], v23 = cold_tm == 3 ? tk[2] : tk[3];
                        float v = cold_tm >= 5 ? tk[4] : (cold_tm <= 2 ? v01 : v23);
                        float o;
                        asm volatile("ds_bpermute_b32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(o) : "v"(a16), "v"(v));
                        v = fminf(v, o);
                        asm volatile("ds_bpermute_b32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(o) : "v"(a32), "v"(v));
                        thr[xb] = fmaxf(fminf(v, o), thr_base[COLD ? xb : 0]);
                    }
                }
                if (emit_on) {
#pragma unroll 1
                    for (int sl = 0; sl < EMQ_SLOTS; ++sl) {
                        const bool has = qcnt > (uint32_t)sl;
                        if (__builtin_amdgcn_ballot_w64(has) == 0ull) break;
                        uint64_t e;
                        asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(e) : "v"(q_lane_f + (uint32_t)sl * 512u) : "memory");
                        const uint32_t exb = (uint32_t)(e >> 62);
                        const float t01 = exb == 0u ? thr[0] : thr[1 % XBN], t23 = exb == 2u ? thr[2 % XBN] : thr[3 % XBN];
                        const bool keep = has && __uint_as_float((uint32_t)e) >= (exb < 2u ? t01 : t23);
                        const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
                        const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
                        if (keep && base + rank < KNN_REGION_CAP) reinterpret_cast<uint64_t *>(dst)[base + rank] = e;
                        base += __builtin_popcountll(m);
#ifdef W3D_KNN_STAMP
                        st_acc[5] += __builtin_popcountll(m);            // (cold sweeps: entries written to the region, compacted or not)
#endif
                    }
                    roff = base;
                }
                qcnt = 0u;
                return;
            }
            if (parts > 1) {
                int tot = 0;
#pragma unroll 1
                for (int sl = 0; sl < EMQ_SLOTS; ++sl) { const unsigned long long m = __builtin_amdgcn_ballot_w64(qcnt > (uint32_t)sl); if (m == 0ull) break; tot += __builtin_popcountll(m); }
                if (tot > 0) {
                    if (lane == 0) base = atomicAdd(&cand_cnt[region], tot);
                    base = __builtin_amdgcn_readfirstlane(base);
                }
            }
#pragma unroll 1
            for (int sl = 0; sl < EMQ_SLOTS; ++sl) {
                const bool has = qcnt > (uint32_t)sl;
                const unsigned long long m = __builtin_amdgcn_ballot_w64(has);
                if (m == 0ull) break;
                const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
                uint64_t e;
                asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(e) : "v"(q_lane_f + (uint32_t)sl * 512u) : "memory");
                if (has && base + rank < KNN_REGION_CAP) reinterpret_cast<uint64_t *>(dst)[base + rank] = e;   // past the capacity: dropped — the count says so, knn_select_kernel sends the rows to the repair
                base += __builtin_popcountll(m);
            }
            if (parts == 1) roff = base;
            qcnt = 0u;
        };
        // cold sweep: the wave's region is COMPACTED in place when it is half full (checked once per tile: a tile adds at most 4096 entries) — bounds only rise, so an entry
        // under its row's current bound is dead; what survives is the ~30 best per row so far.  Entries are re-read with device-scope loads (the wave's own earlier stores
        // went through to the L2; its L1 may hold older lines).  Exact ties in their hundreds can keep a region full: it overflows, and its rows go to the repair as ever.
        auto compact = [&]() {
            uint64_t *dst = reinterpret_cast<uint64_t *>(cand_buf + (int64_t)region * KNN_REGION_CAP);
            {                                                            // the rows' bounds by row: lane (h, xr) holds row 16 h + xr's in thr[h]
                const float t01 = h == 0 ? thr[0] : thr[1 % XBN], t23 = h == 2 ? thr[2 % XBN] : thr[3 % XBN];
                const float tv = h < 2 ? t01 : t23;
                const uint32_t ra = lds_addr(rowmax + lane);
                asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" :: "v"(ra), "v"(tv) : "memory");
            }
            const int n = roff < KNN_REGION_CAP ? roff : KNN_REGION_CAP;
            int w = 0;
#pragma unroll 1
            for (int r0 = 0; r0 < n; r0 += 256) {
                uint64_t e[4]; bool keep[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int i = r0 + u * 64 + lane; e[u] = __hip_atomic_load(dst + (i < n ? i : n - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float tv;
                    const uint32_t ta = lds_addr(rowmax) + (uint32_t)(e[u] >> 58) * 4u;
                    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(tv) : "v"(ta) : "memory");
                    keep[u] = r0 + u * 64 + lane < n && __uint_as_float((uint32_t)e[u]) >= tv;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {                            // (every entry of the chunk is in registers: the writes land at or below the chunk's start)
                    const unsigned long long m = __builtin_amdgcn_ballot_w64(keep[u]);
                    const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
                    if (keep[u]) dst[w + rank] = e[u];
                    w += __builtin_popcountll(m);
                }
            }
            roff = w;
        };
        // ---- tile positions of this item: [t_begin, t_end) of the block's order; a position's tile id is needed one tile ahead (its LDS-DMA): fetched two ahead
        // (cold: the item's steps i = 0 ... n_it + T0 - 1 visit positions t_begin + i, then the first T0 once more; tile_order == NULL: every tile in natural order)
        const int n_it = t_end - t_begin;
        const int T0 = COLD ? min(cold_t0, n_it >> 1) : 0;
        const int n_tot = n_it + T0;
        const bool ident = !tile_order;                                 // (a sweep without a plan: cold, or warm = 1 on the emit kernel)
        const int32_t *tord = ident ? nullptr : tile_order + (int64_t)rblk * ntiles_all;
        auto pm = [&](int i) -> int { const int p = t_begin + ((COLD && i >= n_it) ? i - n_it : i); return p < ntiles_all ? p : ntiles_all - 1; };
        auto tile_at = [&](int i) -> int { const int p = pm(i); return ident ? p : __builtin_amdgcn_readfirstlane(tord[p]); };
        int t_cur = tile_at(0), t_nxt = tile_at(1);
        // (a VECTOR load on purpose: as a scalar load the compiler waits for it on the spot — lgkmcnt(0) in the middle of the tile — while a vector load is consumed by the
        // v_readfirstlane one tile later; the zero offset in a VGPR keeps the address from being recognised as uniform)
        int vzero = 0;
        asm volatile("" : "+v"(vzero));
        int pref = ident ? pm(2) : tord[pm(2) + vzero];
        static_assert(64 * CPR == NT * PIECES, "a tile is PIECES whole instructions of the block");
        auto issue_piece = [&](const char *tb, int last, int t, int buf, int i) {   // i < PIECES: a 1-KB piece of tile t (tb = its first byte, last = its last row of Y); i == PIECES: the tile rows' -0.5 |y|^2 (wave 0)
            if (i < PIECES) {
                // rows past the end of Y re-read the tile's last row (their -0.5 |y|^2 is -inf: never selected)
                const uint32_t off = ((uint32_t)min(dma_r0 + i * RPP, last) * ROWB) | dma_choff;
                __builtin_amdgcn_global_load_lds((gptr_t)(tb + off), (lptr_t)(ytile + buf * TILE_BYTES + (wid * 64 + NT * i) * 16), 16, 0, 0);
            } else if (wid == 0 && lane < 64) {
                __builtin_amdgcn_global_load_lds((gptr_t)(ynh + (int64_t)t * 64 + lane), (lptr_t)(ynh_t + buf * 64), 4, 0, 0);
            } else if (FP8 && wid == 1 && lane < 64) {                   // E8M0 scales of the tile rows (rows past the end: the last row's, never selected)
                __builtin_amdgcn_global_load_lds((gptr_t)(ysc + min(t * 64 + lane, M - 1)), (lptr_t)(ysc_t + buf * 64), 4, 0, 0);
            }
        };
        auto tile_base = [&](int t) -> const char * { return reinterpret_cast<const char *>(Y) + (int64_t)t * TILE_BYTES; };
        auto tile_last = [&](int t) -> int { return min(M - t * 64, 64) - 1; };
        if (t_begin < t_end) {
#pragma unroll
            for (int i = 0; i <= PIECES; ++i) issue_piece(tile_base(t_cur), tile_last(t_cur), t_cur, 0, i);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                                 // the first tile has landed
        const int64_t clk_l0 = clk ? (int64_t)wall_clock64() : 0;
        uint16_t *lbrow = LB ? lb + ((int64_t)rblk * ntiles_all * ROWS_B + wid * ROWS_W) : nullptr;      // (wave-uniform; + tile * 512 + lane)

        // Advance of the tile ids (next becomes current, the prefetched id the next one, the id after that is fetched).  Called right BEHIND the wave's tile barrier,
        // whose vmcnt(0) has just drained everything — the v_readfirstlane of the prefetched id then waits for nothing.  At the tile's end, where round 4 had it, it sat
        // behind the store of the tile's lower bounds: vector-memory operations complete in order, so reading the (long arrived) id waited for that store's round trip,
        // ~900 cycles per tile (scripts/emit_stamp.py: a wave's item = its tiles + 106 000 cycles at 119 tiles).
        auto advance = [&](const int i) {
            t_cur = t_nxt;
            t_nxt = __builtin_amdgcn_readfirstlane(pref);
            pref = ident ? pm(i + 3) : tord[pm(i + 3) + vzero];
        };
        // one step of the item; CUR = the LDS buffer its tile sits in (compile time: every LDS address of the MFMA phase is a lane constant + an instruction offset)
#ifndef W3D_COLD_R
#define W3D_COLD_R 8
#endif
#ifndef W3D_COLD_RS
#define W3D_COLD_RS 3
#endif
#ifndef W3D_COLD_COMP
#define W3D_COLD_COMP 512
#endif
        int next_refresh = 0, comp_next = 1024;
        auto tile = [&](auto CURC, const int pos) {
            constexpr int CUR = decltype(CURC)::value;
            if constexpr (COLD) {
                // phase changes spill under the OLD phase's rules (what the trackers have not seen yet is fed, nothing of phase 1 is emitted); between them the
                // bounds are refreshed W3D_COLD_R tiles after the last refresh, later an eighth of the tiles swept so far (the k-th best of t tiles moves like 1 / t:
                // a bound that is t / 8 tiles old admits 6 % more; a full queue spills — and refreshes — by itself)
                if (pos == T0 || pos == n_it || pos >= next_refresh) { flush(); next_refresh = pos + max(W3D_COLD_R, pos >> W3D_COLD_RS); }
                emit_on = pos >= T0;
                if (pos == n_it) {
                    // third phase: the trackers start over and see what THIS phase emits (rows of the first tiles — feeding them to the old trackers could count a row
                    // twice: the first phase put its unit maxima there); a row's bound = the larger of what the second phase ended with and what these trackers say,
                    // either of which k rows reach.  Without it the bounds stand still here, and first tiles that hold a row's own cluster flood the region.
                    next_refresh = pos + 1;
#pragma unroll
                    for (int xb = 0; xb < XBN; ++xb) {
                        thr_base[COLD ? xb : 0] = thr[xb];
#pragma unroll
                        for (int i = 0; i < TM; ++i) trk[COLD ? xb : 0][i] = -__FLT_MAX__;
                    }
                }
                // compaction: when 512 entries have come in since the last one left `comp_next - 512` alive (the queue is spilled first: a tile then adds at most its
                // own 4096; past the capacity entries are lost already and the count must keep saying so)
                if (roff > comp_next && roff <= KNN_REGION_CAP) { flush(); compact(); comp_next = max(roff, 512) + W3D_COLD_COMP; }
            }
            const int t = t_cur, t1 = t_nxt;
            const bool dma_next = pos + 1 < n_tot;
            const char *tb1 = tile_base(t1);
            const int last1 = tile_last(t1);
            W3D_STAMP_BEGIN();
#ifdef W3D_KNN_STAMP
            st_tile = st_t;
#endif
            static_for<0, 2>([&](auto YBC) {
                constexpr int yb = decltype(YBC)::value;
                // priorities order the four MFMA phases of a SIMD and tile: A.half0 (3) > B.half0 (2) > A.half1 (1) > B.half1 (0)
#ifndef W3D_EMIT_PRIO
#define W3D_EMIT_PRIO 0
#endif
                if constexpr (W3D_EMIT_PRIO == 0) {
                    if (yb == 0) { if (grpB) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(3); }
                    else { if (grpB) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(1); }
                } else if constexpr (W3D_EMIT_PRIO == 2) {              // (A/B variants, profiles/r05l_prio_ab.log) MFMA phase 1, epilogue 0
                    __builtin_amdgcn_s_setprio(1);
                } else if constexpr (W3D_EMIT_PRIO == 4) {              // MFMA phase 0, epilogue 1
                    __builtin_amdgcn_s_setprio(0);
                }
                // scores of the sub-block: SH = 1: acc4[u][xb][i] = style row 32 yb + 16 u + 4 h + i of content row 16 xb + xr;
                //                          SH = 0: acc16[xb][j]   = style row 32 yb + (j & 3) + 8 (j >> 2) + 4 h of content row 32 xb + xr
                f32x4 acc4[SH ? 2 : 1][SH ? XBN : 1];
                f32x16 acc16[SH ? 1 : XBN];
                {
                    constexpr int OFFB = CUR * TILE_BYTES + yb * 32 * ROWB;      // tile buffer + sub-block, as an instruction offset
                    constexpr int YOFF = (CUR * 64 + yb * 32) * 4;
                    constexpr int NYN = SH ? 2 : 4;                               // -0.5 |y|^2 quads: rows 16 u + 4 h + (0..3) / 8 g + 4 h + (0..3)
                    f32x4 yn[NYN];
                    const uint32_t ya = ya_lane;
                    constexpr int YST = SH ? 64 : 32;
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(yn[0]) : "v"(ya), "n"(YOFF));
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(yn[1]) : "v"(ya), "n"(YOFF + YST));
                    if constexpr (NYN == 4) {
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(yn[2 % NYN]) : "v"(ya), "n"(YOFF + 2 * YST));
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(yn[3 % NYN]) : "v"(ya), "n"(YOFF + 3 * YST));
                    }
                    int ysc_v = 0;
                    if constexpr (FP8) { const uint32_t ysa = ysc_lane; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(ysc_v) : "v"(ysa), "n"(YOFF)); }
                    constexpr int NAF = (SH || FP8) ? 2 : 1;                      // ds_read_b128 per k-step (SH = 1: the two 16-row halves of the sub-block; fp8: the fragment's halves)
                    afrag_t af[PF + 1][NAF];
                    auto frag_read = [&](auto SC, afrag_t (&dst)[NAF]) {
                        constexpr int s = decltype(SC)::value;
                        constexpr int o0 = OFFB + ((s / NJ) << 8);
                        const uint32_t aa = abase[s % NJ];               // (a plain use: asm operands alone do not make a nested generic lambda capture the variable)
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[0]) : "v"(aa), "n"(o0));
                        if constexpr (SH) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[NAF - 1]) : "v"(aa), "n"(o0 + 16 * ROWB));
                        if constexpr (FP8) { const uint32_t ab = abase2[FP8 ? s % NJ : 0]; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[NAF - 1]) : "v"(ab), "n"(o0)); }
                    };
                    static_for<0, PF>([&](auto SC) { frag_read(SC, af[decltype(SC)::value]); });
                    static_for<0, KS>([&](auto SC) {
                        constexpr int s = decltype(SC)::value;
                        if (dma_next && yb == 0) {
#pragma unroll
                            for (int i = 0; i <= PIECES; ++i) if ((i * KS) / (PIECES + 1) == s) issue_piece(tb1, last1, t1, CUR ^ 1, i);
                        }
                        if constexpr (s + PF < KS) frag_read(std::integral_constant<int, (s + PF < KS ? s + PF : 0)>{}, af[(s + PF) % (PF + 1)]);
                        constexpr int left = ((KS - 1 - s) < PF ? (KS - 1 - s) : PF) * NAF;      // ds_reads younger than step s's fragment
                        constexpr int cs = s % (PF + 1);
                        static_assert(left <= 8, "lgkmcnt literal");
                        if constexpr (s == 0) {
                            if constexpr (SH) {
                                asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(yn[0]), "+v"(yn[1]), "+v"(af[0][0]), "+v"(af[0][NAF - 1]) : "n"(left));
#pragma unroll
                                for (int u = 0; u < 2; ++u)
#pragma unroll
                                    for (int xb = 0; xb < XBN; ++xb) acc4[u][SH ? xb : 0] = yn[u];
                            } else {
                                // (every operand once: the same variable twice in one asm makes the compiler copy it — BEFORE the wait, i.e. before the data has landed)
                                if constexpr (FP8) asm volatile("s_waitcnt lgkmcnt(%7)" : "+v"(yn[0]), "+v"(yn[1]), "+v"(yn[2 % NYN]), "+v"(yn[3 % NYN]), "+v"(ysc_v), "+v"(af[0][0]), "+v"(af[0][NAF - 1]) : "n"(left));
                                else asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(yn[0]), "+v"(yn[1]), "+v"(yn[2 % NYN]), "+v"(yn[3 % NYN]), "+v"(af[0][0]) : "n"(left));
#pragma unroll
                                for (int xb = 0; xb < XBN; ++xb)
#pragma unroll
                                    for (int g = 0; g < 4; ++g) {
                                        acc16[SH ? 0 : xb][4 * g + 0] = yn[g % NYN][0]; acc16[SH ? 0 : xb][4 * g + 1] = yn[g % NYN][1];
                                        acc16[SH ? 0 : xb][4 * g + 2] = yn[g % NYN][2]; acc16[SH ? 0 : xb][4 * g + 3] = yn[g % NYN][3];
                                    }
                            }
                        } else if constexpr (SH || FP8) {
                            asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(af[cs][0]), "+v"(af[cs][NAF - 1]) : "n"(left));
                        } else {
                            asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(af[cs][0]) : "n"(left));
                        }
                        if constexpr (SH) {
#pragma unroll
                            for (int u = 0; u < 2; ++u)
#pragma unroll
                                for (int xb = 0; xb < XBN; ++xb) {
                                    if constexpr (FMT == FMT_F16) acc4[u][SH ? xb : 0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[cs][u % NAF], xf[xb][s], acc4[u][SH ? xb : 0], 0, 0, 0);
                                    else acc4[u][SH ? xb : 0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[cs][u % NAF], xf[xb][s], acc4[u][SH ? xb : 0], 0, 0, 0);
                                }
                        } else if constexpr (FP8) {
                            const i32x8 a8 = __builtin_shufflevector(af[cs][0], af[cs][NAF - 1], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                            for (int xb = 0; xb < XBN; ++xb)
                                acc16[SH ? 0 : xb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, xf[xb][s], acc16[SH ? 0 : xb], 0, 0, 0, ysc_v, 0, xscale[xb]);
                        } else {
#pragma unroll
                            for (int xb = 0; xb < XBN; ++xb) {
                                if constexpr (FMT == FMT_F16) acc16[SH ? 0 : xb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[cs][0], xf[xb][s], acc16[SH ? 0 : xb], 0, 0, 0);
                                else acc16[SH ? 0 : xb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[cs][0], xf[xb][s], acc16[SH ? 0 : xb], 0, 0, 0);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    });
                }
                if constexpr (!SH) {                                     // all accumulators are complete HERE (keeps the per-block MFMA chains from being sunk into their epilogues)
#pragma unroll
                    for (int xb = 0; xb < XBN; ++xb) asm volatile("" : "+v"(acc16[SH ? 0 : xb]));
                }
                if constexpr (W3D_EMIT_PRIO == 0 || W3D_EMIT_PRIO == 2) __builtin_amdgcn_s_setprio(0);
                else if constexpr (W3D_EMIT_PRIO == 4) __builtin_amdgcn_s_setprio(1);
                W3D_STAMP(2 * yb);
                // B waves: the tile barrier (tile t's buffer is free; tile t + 1 has landed) BEFORE their last epilogue, which then runs under the A waves' next MFMA phase
                if (yb == 1 && grpB) { asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory"); W3D_STAMP(4); advance(pos); }

                // ---- selection epilogue.  A unit's scores in a lane: NRQ per quarter q (the lane's style rows q + 4 h + step m, m < NRQ).  Per unit a vector-only PRETEST
                // (four quarter maxima, their compares against the row's bound as four wave masks, the unit maximum for the cached bounds) and a scalar SCAN of its masks
                // with one straight-line block per hit quarter; the pretest of unit u + 1 is issued before the scan of unit u, so the scan's masks are long complete.
                auto score = [&](auto XBC, auto QC, auto MC) -> float {
                    constexpr int xb = decltype(XBC)::value, q = decltype(QC)::value, m = decltype(MC)::value;
                    if constexpr (SH) return acc4[m % 2][SH ? xb : 0][q]; else return acc16[SH ? 0 : xb][q + 4 * m];
                };
                unsigned long long qm[XBN][4];
                auto pretest = [&](auto XBC) {
                    constexpr int xb = decltype(XBC)::value;
                    float m4[4];
                    static_for<0, 4>([&](auto QC) {
                        constexpr int q = decltype(QC)::value;
                        float v = fmaxf(score(XBC, QC, std::integral_constant<int, 0>{}), score(XBC, QC, std::integral_constant<int, 1>{}));
                        if constexpr (NRQ == 4) v = fmaxf(v, fmaxf(score(XBC, QC, std::integral_constant<int, 2>{}), score(XBC, QC, std::integral_constant<int, 3>{})));
                        m4[q] = v;
                    });
#pragma unroll
                    for (int q = 0; q < 4; ++q) qm[xb][q] = __builtin_amdgcn_ballot_w64(m4[q] >= thr[xb]);
                    if constexpr (LB) {
                        // cached lower bounds: the best score of this unit per content row, max-reduced over the lanes that share the row (h) and over the tile's units in LDS
                        const float mx = fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));
                        const uint32_t rml = rm_lane;
                        asm volatile("ds_max_f32 %0, %1 offset:%2" :: "v"(rml), "v"(mx), "n"(xb * XBR * 4) : "memory");
                    }
                };
                const uint32_t tag_t = (uint32_t)(t * 64 + yb * 32);
                // one quarter of a unit: lanes of the quarter's mask only; per score: store (score, tag) to the lane's next free slot, advance the lane's count if the score
                // reaches the bound (the carry of v_addc)
                auto quarter = [&](auto XBC, auto QC, uint32_t tag) {
                    constexpr int xb = decltype(XBC)::value, q = decltype(QC)::value;
                    unsigned long long sv; uint32_t qa, tg;
                    const unsigned long long mk = qm[xb][q];
                    const float th = thr[xb];
                    const uint32_t ql = q_lane, tagv = tag;
                    uint32_t cnt = qcnt;                                // (plain uses: asm operands alone do not make the nested generic lambda capture these)
                    const float a0 = score(XBC, QC, std::integral_constant<int, 0>{}), a1 = score(XBC, QC, std::integral_constant<int, 1>{});
#define W3D_EMIT_ONE(A, K)                                          \
                    "v_cmp_ge_f32_e32 vcc, %[" A "], %[th]\n\t"          \
                    "v_lshl_add_u32 %[qa], %[cnt], 9, %[ql]\n\t"         \
                    "v_add_u32_e32 %[tg], %[" K "], %[tag]\n\t"          \
                    "ds_write2_b32 %[qa], %[" A "], %[tg] offset1:1\n\t" \
                    "v_addc_co_u32_e32 %[cnt], vcc, 0, %[cnt], vcc\n\t"
                    if constexpr (NRQ == 2) {
                        asm volatile("s_and_saveexec_b64 %[sv], %[mk]\n\t" W3D_EMIT_ONE("a0", "k0") W3D_EMIT_ONE("a1", "k1") "s_mov_b64 exec, %[sv]"
                                     : [sv] "=&s"(sv), [qa] "=&v"(qa), [tg] "=&v"(tg), [cnt] "+v"(cnt)
                                     : [mk] "s"(mk), [a0] "v"(a0), [a1] "v"(a1), [th] "v"(th), [ql] "v"(ql), [tag] "v"(tagv), [k0] "n"(q), [k1] "n"(16 + q)
                                     : "vcc", "scc", "memory");
                    } else {
                        const float a2 = score(XBC, QC, std::integral_constant<int, 2 % NRQ>{}), a3 = score(XBC, QC, std::integral_constant<int, 3 % NRQ>{});
                        asm volatile("s_and_saveexec_b64 %[sv], %[mk]\n\t" W3D_EMIT_ONE("a0", "k0") W3D_EMIT_ONE("a1", "k1") W3D_EMIT_ONE("a2", "k2") W3D_EMIT_ONE("a3", "k3")
                                     "s_mov_b64 exec, %[sv]"
                                     : [sv] "=&s"(sv), [qa] "=&v"(qa), [tg] "=&v"(tg), [cnt] "+v"(cnt)
                                     : [mk] "s"(mk), [a0] "v"(a0), [a1] "v"(a1), [a2] "v"(a2), [a3] "v"(a3), [th] "v"(th), [ql] "v"(ql), [tag] "v"(tagv),
                                       [k0] "n"(q), [k1] "n"(8 + q), [k2] "n"(16 + q), [k3] "n"(24 + q)
                                     : "vcc", "scc", "memory");
                    }
#undef W3D_EMIT_ONE
                    qcnt = cnt;
                };
                // a lane takes at most 8 entries per step (a unit of 8 scores, or half a unit of 16): a queue with fewer than 8 free slots is spilled before the step (the
                // check rides behind the step before: its compare is long complete when the branch reads it)
                unsigned long long full = __builtin_amdgcn_ballot_w64(qcnt > (uint32_t)(EMQ_SLOTS - 8));
                auto scan = [&](auto XBC) {
                    constexpr int xb = decltype(XBC)::value;
                    if ((qm[xb][0] | qm[xb][1] | qm[xb][2] | qm[xb][3]) == 0ull) return;
                    const uint32_t tag = lane_tag + (tag_t + ((uint32_t)(xb * XBR) << 26));
                    static_for<0, (NRQ == 2 ? 1 : 2)>([&](auto HC) {       // steps of at most 8 scores per lane
                        constexpr int hq = decltype(HC)::value;
                        constexpr int q0 = NRQ == 2 ? 0 : 2 * hq, q1 = NRQ == 2 ? 4 : 2 * hq + 2;
                        bool any = false;
#pragma unroll
                        for (int q = q0; q < q1; ++q) any = any || qm[xb][q] != 0ull;
                        if (!any) return;
                        if (full != 0ull) flush();
                        static_for<q0, q1>([&](auto QC) {
                            constexpr int q = decltype(QC)::value;
                            if (qm[xb][q] == 0ull) return;
                            quarter(XBC, QC, tag);
                        });
                        full = __builtin_amdgcn_ballot_w64(qcnt > (uint32_t)(EMQ_SLOTS - 8));
                    });
                };
                if (COLD && pos < T0) {
                    // cold sweep, first phase: no queue, no scan — every lane feeds the best of its 8 scores of each unit straight to the row's tracker (all lanes busy:
                    // 7 + 10 vector instructions per unit).  The top scores of a quarter almost never share a unit, so the trackers come out as good as exact; any
                    // real score in them keeps the bound valid.
                    static_for<0, XBN>([&](auto XBC) {
                        constexpr int xb = decltype(XBC)::value;
                        float x = -__FLT_MAX__;
                        static_for<0, 4>([&](auto QC) {
                            x = fmaxf(x, fmaxf(score(XBC, QC, std::integral_constant<int, 0>{}), score(XBC, QC, std::integral_constant<int, 1>{})));
                            if constexpr (NRQ == 4) x = fmaxf(x, fmaxf(score(XBC, QC, std::integral_constant<int, 2>{}), score(XBC, QC, std::integral_constant<int, 3>{})));
                        });
#pragma unroll
                        for (int i = 0; i < TM; ++i) { const float hi = fmaxf(trk[COLD ? xb : 0][i], x); x = fminf(trk[COLD ? xb : 0][i], x); trk[COLD ? xb : 0][i] = hi; }
                    });
                } else {
                pretest(std::integral_constant<int, 0>{});
                pretest(std::integral_constant<int, 1>{});
                __builtin_amdgcn_sched_barrier(0);
                scan(std::integral_constant<int, 0>{});
                if constexpr (XBN == 4) {
                    __builtin_amdgcn_sched_barrier(0);
                    pretest(std::integral_constant<int, 2 % XBN>{});
                    __builtin_amdgcn_sched_barrier(0);
                    scan(std::integral_constant<int, 1>{});
                    __builtin_amdgcn_sched_barrier(0);
                    pretest(std::integral_constant<int, 3 % XBN>{});
                    __builtin_amdgcn_sched_barrier(0);
                    scan(std::integral_constant<int, 2 % XBN>{});
                    scan(std::integral_constant<int, 3 % XBN>{});
                } else {
                    scan(std::integral_constant<int, 1>{});
                }
                }
                W3D_STAMP(2 * yb + 1);
                // A waves: the tile barrier after their last epilogue
                if (yb == 1 && !grpB) { asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory"); W3D_STAMP(4); advance(pos); }
                if constexpr (LB) {
                    // the tile's lower bounds leave AFTER the wave's tile barrier, so that the store has a whole tile's time before the next barrier's vmcnt(0) waits for it
                    if (yb == 1) {
                        float mv, lbA, lbC;
                        const uint32_t ra = lds_addr(rowmax + lane);
                        asm volatile("ds_read_b32 %0, %3\n\tds_read_b32 %1, %3 offset:256\n\tds_read_b32 %2, %3 offset:512\n\ts_waitcnt lgkmcnt(0)" : "=&v"(mv), "=&v"(lbA), "=&v"(lbC) : "v"(ra) : "memory");
                        asm volatile("ds_write_b32 %0, %1" :: "v"(ra), "v"(-__builtin_inff()) : "memory");
                        const float m = __builtin_amdgcn_sqrtf(fmaxf(lbA - 2.0f * mv, 0.f));           // v_sqrt_f32 (1 ulp; a denormal argument may read 0: a smaller bound is still a bound)
                        const _Float16 hv = (_Float16)fminf((m + lbC) * 0.9990234375f, 60000.f);       // rounded DOWN: the conversion's half ulp (2^-11) and the root's ulp are inside the factor 1 - 2^-10
                        uint16_t hb; __builtin_memcpy(&hb, &hv, 2);
                        lbrow[(int64_t)t * ROWS_B + (uint32_t)lane] = hb;
                    }
                }
            });
#ifdef W3D_KNN_STAMP
            {
                const int64_t n_ = (int64_t)__builtin_amdgcn_s_memtime() - st_tile;
                if (pos == 0) st_pos[0] = n_;
                if (pos == 1) st_pos[1] = n_;
                if (pos == n_tot - 2) st_pos[2] = n_;
                if (pos == n_tot - 1) st_pos[3] = n_;
            }
#endif
        };
        if constexpr (W3D_EMIT_PRIO == 3) { if (grpB) __builtin_amdgcn_s_setprio(1); }      // static: the younger half ahead, no per-phase flips
        // positions alternate between the two LDS buffers, the item's first one in buffer 0
        int pos = 0;                                                     // (steps of the item: position t_begin + pos of the block's tile order)
        while (pos < n_tot) {
            tile(std::integral_constant<int, 0>{}, pos); ++pos;
            if (pos >= n_tot) break;
            tile(std::integral_constant<int, 1>{}, pos); ++pos;
        }
        flush();
        // (cold sweep) the item's end: the region is compacted with the final bounds — knn_select_kernel reads ~35 entries per row instead of everything that was
        // alive at some time since the last compaction (0.12 -> 0.09 ms at C2)
        if constexpr (COLD) { if (roff > 0 && roff <= KNN_REGION_CAP) compact(); }
        if (parts == 1 && lane == 0) cand_cnt[region] = roff;          // (the counts were zeroed before the launch; sliced sweeps added theirs atomically)
#ifdef W3D_KNN_STAMP
        if (clk && lane == 0) {                                          // behind the block records (8 words per item): 8 words per (item, wave)
            int64_t *sp = clk + 8 * (int64_t)nitems + ((int64_t)rblk * parts + part) * 64 + wid * 8;
            for (int i = 0; i < 6; ++i) sp[i] = st_acc[i];
            sp[6] = (int64_t)(t_end - t_begin); sp[7] = (int64_t)roff;
            int64_t *sq = clk + 8 * (int64_t)nitems + 64 * (int64_t)nitems + ((int64_t)rblk * parts + part) * 64 + wid * 8;      // second record: whole tiles by position, item span
            for (int i = 0; i < 4; ++i) sq[i] = st_pos[i];
            sq[4] = (int64_t)__builtin_amdgcn_s_memtime() - st_item0;   // the wave's whole item (prologue + tile loop + spills), cycles
        }
#endif
#undef W3D_STAMP_BEGIN
#undef W3D_STAMP
        if (clk && threadIdx.x == 0) {
            uint32_t hwid;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            const int64_t ci = (int64_t)rblk * parts + part;
            const int64_t now = (int64_t)wall_clock64();
            clk[8 * ci + 0] = clk_t0; clk[8 * ci + 1] = now;
            clk[8 * ci + 2] = ((int64_t)(xcc & 0xf) << 32) | (int64_t)hwid;
            clk[8 * ci + 3] = (int64_t)(t_end - t_begin);
            clk[8 * ci + 4] = clk_c0; clk[8 * ci + 5] = (int64_t)__builtin_amdgcn_s_memtime();
            clk[8 * ci + 6] = clk_l0; clk[8 * ci + 7] = now;
        }
        __syncthreads();                                                 // the block's LDS (tile ring, queues, s_item) is reused by the next item
    }
}

template <int D, int BN, int QCAP, int SH, int FMT>
__global__ __launch_bounds__(NWAVES * 64, 2) void knn_mfma_repair_kernel(W3D_KNN_ARGS, const int32_t *__restrict__ n_dev) {
    knn_mfma_body<D, BN, QCAP, SH, FMT, false>(X, Y, xsc, ysc, xnorm, ynh, thr_init, N, M, k, out_idx, out_d2, mode_arg, rowblk0, parts, flags, n_dev);
}
template <int D, int BN, int QCAP, int SH, int FMT>
__global__ __launch_bounds__(NWAVES * 64, 2) void knn_mfma_part_kernel(W3D_KNN_ARGS, int64_t tile0, int64_t tile1, int part_fixed, const int32_t *__restrict__ row_perm) {
    knn_mfma_body<D, BN, QCAP, SH, FMT>(X, Y, xsc, ysc, xnorm, ynh, thr_init, N, M, k, out_idx, out_d2, mode_arg, rowblk0, parts, flags, nullptr, tile0, tile1, part_fixed, row_perm);
}
#undef W3D_KNN_ARGS

// Second half of the EMIT sweep (SEL 1): one wave per region (64 consecutive positions of the sweep order) turns the candidates its rows emitted into their k-entry
// lists.  Fast path, ONE pass over the region: every candidate takes a slot of its row's bucket (an LDS atomic on the row's count; 32 slots per row, lane-strided) and
// the shared write-out's rank sort — 32 keys per lane in registers — keeps the k best of a bucket (score descending, ties: lower style index: the sweep's own order).
// A region in which some row received more candidates than its bucket holds (a loose bound, many ties) is redone by the general path: chunks of SELECT_CHUNK candidates
// counting-sorted by row in LDS (one LDS atomic per entry, one wave scan), every lane folding its own row's segment into its list with the list sweep's merge (append
// until k, then replace the worst if better()).  A region whose count exceeds its capacity lost candidates: its rows are flagged for the cold repair like any row that
// ends short.  (The general path alone — two passes over every chunk, a dependent chain of LDS atomics — took 0.43-0.47 ms for 25 M candidates at C3; the bucket path 0.21.)
constexpr int SELECT_BUCKET = 32;        // slots per row = the rank sort's width (k <= 32 on every MFMA path)
constexpr int SELECT_CHUNK = 512;
// STRAGGLER pairs of a pruned emit sweep (round 6; see plan_bounds_phase): the (content row, style tile) pairs a steady plan took OUT of the blocks' tile lists because only
// a handful of the block's 512 rows still needed the tile.  One workgroup per (tile, KNN_STRAG_CH bucket positions): the tile's 64 rows and their -0.5 |y|^2 in LDS (rows
// padded by 16 B: the 16 lanes of a fragment read land on 64 different banks), then per wave groups of 16 bucket rows as the B operand of the sweep's own
// v_mfma_f32_16x16x32 chain — accumulators started at -0.5 |y|^2, the k-steps in the sweep's order over the same 32 features each: every score is the sweep's bit for bit
// (an MFMA result element depends on its A row, its B column and the accumulator only).  Scores that reach the row's bound are appended to the row's region of the
// candidate buffer (one atomic each: a row has ~k of them in the whole sweep) — AFTER the emit kernel, whose waves own their regions and store the counts at the end of
// their items; knn_select_kernel behind both sees one set of candidates.  The rows' cached bounds for the tile are refreshed as the sweep does it for the tiles it runs
// (the other rows of the block keep their older, still valid, bounds).
template <int D, int FMT>
__global__ __launch_bounds__(256) void knn_straggler_kernel(const uint8_t *__restrict__ X, const uint8_t *__restrict__ Y, const float *__restrict__ xnorm, const float *__restrict__ ynh,
                                                            const float *__restrict__ thr, int M, const int32_t *__restrict__ row_perm, const uint32_t *__restrict__ strag_cnt,
                                                            const uint32_t *__restrict__ strag, int cap, int ntiles, Entry *__restrict__ cand_buf, int32_t *__restrict__ cand_cnt,
                                                            uint16_t *__restrict__ lb, const float *__restrict__ lb_cum, float lb_margin) {
    static_assert(FMT != FMT_FP8, "16-bit rows");
    constexpr int ROWB = D * 2, KS = D / 32, LROW = ROWB + 16, CPR = ROWB / 16;
    typedef typename std::conditional<FMT == FMT_F16, f16x8, bf16x8>::type frag_t;
    __shared__ __attribute__((aligned(16))) uint8_t ytile[64 * LROW];
    __shared__ __attribute__((aligned(16))) float ynh_s[64];
    // grid = tiles x KNN_STRAG_WG: workgroup j of a tile takes the bucket's chunks j, j + KNN_STRAG_WG, ... (a grid of tiles x cap / KNN_STRAG_CH workgroups, most of
    // them empty, cost 0.16 ms of dispatch alone at C3: 32 768 workgroups)
    const int t = (int)(blockIdx.x % (unsigned)ntiles), wg = (int)(blockIdx.x / (unsigned)ntiles);
    const uint32_t cnt = min(strag_cnt[t], (uint32_t)cap);
    if ((uint32_t)wg * KNN_STRAG_CH >= cnt) return;
    for (int i = threadIdx.x; i < 64 * CPR; i += 256) {
        const int r = i / CPR, c = i % CPR;
        const int64_t row = min((int64_t)t * 64 + r, (int64_t)M - 1);       // rows past the end of Y: the last row's (their -0.5 |y|^2 is -inf: never selected)
        *reinterpret_cast<uint4 *>(ytile + r * LROW + c * 16) = *reinterpret_cast<const uint4 *>(Y + row * ROWB + c * 16);
    }
    if (threadIdx.x < 64) ynh_s[threadIdx.x] = ynh[(int64_t)t * 64 + threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, m = lane & 15, h = lane >> 4;
    constexpr int GPW = KNN_STRAG_CH / 16 / 4;                              // groups of 16 bucket rows per wave and chunk
    for (uint32_t e0 = (uint32_t)wg * KNN_STRAG_CH; e0 < cnt; e0 += (uint32_t)KNN_STRAG_WG * KNN_STRAG_CH) {
        // the chunk's dependent loads (bucket position -> row of the order -> the row's bound and features) are issued for all of the wave's groups at once: one latency
        // chain per chunk instead of per group (a group at a time: 2 us each, 0.18 ms for the 1.4 M pairs of a C3 plan); the feature rows are double-buffered
        uint32_t posv[GPW]; bool val[GPW]; uint32_t rowv[GPW]; float thv[GPW];
#pragma unroll
        for (int q = 0; q < GPW; ++q) {
            const uint32_t e = e0 + (uint32_t)(wid + 4 * q) * 16u + (uint32_t)m;
            posv[q] = e < cnt ? strag[(int64_t)t * cap + e] : 0xffffffffu;
        }
#pragma unroll
        for (int q = 0; q < GPW; ++q) {
            val[q] = posv[q] != 0xffffffffu;                               // (voided slots of a bucket that overflowed: their tile stayed in the block's list)
            if (!val[q]) posv[q] = 0u;
            rowv[q] = row_perm ? (uint32_t)row_perm[posv[q]] : posv[q];
        }
#pragma unroll
        for (int q = 0; q < GPW; ++q) thv[q] = thr[rowv[q]];
        frag_t xf[2][KS];
#pragma unroll
        for (int s2 = 0; s2 < KS; ++s2) xf[0][s2] = *reinterpret_cast<const frag_t *>(X + (int64_t)rowv[0] * ROWB + (4 * s2 + h) * 16);
#pragma unroll
        for (int q = 0; q < GPW; ++q) {
            if (q + 1 < GPW) {
#pragma unroll
                for (int s2 = 0; s2 < KS; ++s2) xf[(q + 1) & 1][s2] = *reinterpret_cast<const frag_t *>(X + (int64_t)rowv[(q + 1) % GPW] * ROWB + (4 * s2 + h) * 16);
            }
            if (__builtin_amdgcn_ballot_w64(val[q]) == 0ull) continue;
            const uint32_t pos = posv[q];
            const int64_t row = (int64_t)rowv[q];
            const float th = thv[q];
            f32x4 acc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = *reinterpret_cast<const f32x4 *>(ynh_s + 16 * u + 4 * h);
#pragma unroll
            for (int s2 = 0; s2 < KS; ++s2) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {                               // (four independent chains; each accumulates its k-steps in the sweep's order)
                    const frag_t a = *reinterpret_cast<const frag_t *>(ytile + (16 * u + m) * LROW + (4 * s2 + h) * 16);
                    if constexpr (FMT == FMT_F16) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, xf[q & 1][s2], acc[u], 0, 0, 0);
                    else acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, xf[q & 1][s2], acc[u], 0, 0, 0);
                }
            }
            float best = -__builtin_inff();
            uint32_t hit = 0u;                                              // bit 4 u + i: style row 16 u + 4 h + i reaches the row's bound
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) { best = fmaxf(best, acc[u][i]); hit |= (acc[u][i] >= th ? 1u : 0u) << (4 * u + i); }
            if (!val[q]) hit = 0u;
            if (hit) {                                                      // one reservation per lane (a row has ~k candidates in the whole sweep)
                const int64_t region = (int64_t)(pos >> 6);
                Entry *dst = cand_buf + region * KNN_REGION_CAP;
                const uint32_t tag0 = ((pos & 63u) << 26) | (uint32_t)(t * 64 + 4 * h);
                int slot = atomicAdd(&cand_cnt[region], __builtin_popcount(hit));
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if ((hit >> (4 * u + i)) & 1u) {
                            if (slot < KNN_REGION_CAP) dst[slot] = Entry{acc[u][i], tag0 + (uint32_t)(16 * u + i)};      // past the capacity: dropped — the count says so, knn_select_kernel sends the rows to the repair
                            ++slot;
                        }
            }
            best = fmaxf(best, __shfl_xor(best, 16, 64));
            best = fmaxf(best, __shfl_xor(best, 32, 64));
            if (val[q] && h == 0) {
                // the row's bound for this tile, exactly as the sweep records it (knn_emit_kernel, LB)
                const float lbA = xnorm[row] * (1.0f - KNN_SCORE_ERR2) - lb_margin;
                const float md = __builtin_amdgcn_sqrtf(fmaxf(lbA - 2.0f * best, 0.f));
                const _Float16 hv = (_Float16)fminf((md + lb_cum[row]) * 0.9990234375f, 60000.f);
                uint16_t hb; __builtin_memcpy(&hb, &hv, 2);
                lb[((int64_t)(pos >> 9) * ntiles + t) * ROWS_B + (pos & 511u)] = hb;
            }
        }
    }
}

// the words an emit sweep counts in, zeroed by ONE launch (three hipMemsetAsync until round 6: three ~5-us fill kernels between the plan and the sweep, on an idle chip):
// the flagged-row count, the work queues' cursors, the regions' candidate counts
__global__ __launch_bounds__(256) void knn_emit_reset_kernel(int32_t *__restrict__ flags, int32_t *__restrict__ qcursor, int32_t *__restrict__ cand_cnt, int64_t nreg) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < nreg) cand_cnt[i] = 0;
    if (i < KNN_NQ) qcursor[i] = 0;
    if (i == 0) flags[0] = 0;
}
__global__ __launch_bounds__(64) void knn_select_kernel(const Entry *__restrict__ cand_buf, const int32_t *__restrict__ cand_cnt, int64_t N, int k,
                                                        const float *__restrict__ xnorm, const int32_t *__restrict__ row_perm, int32_t *__restrict__ out_idx,
                                                        float *__restrict__ out_d2, int32_t *__restrict__ flags) {
    __shared__ __attribute__((aligned(16))) Entry lists[SELECT_BUCKET * ROWS_W];        // [slot][row], lane-strided
    // (general path) the chunk's sorted entries live in the bucket rows the k-entry lists do not use (k <= 24 of 32 rows: 512 entries) — 16.5 instead of 20.5 KB of LDS
    // per workgroup: 9 instead of 7 regions in flight per CU of a kernel that waits for its loads and LDS atomics
    static_assert(SELECT_CHUNK * sizeof(Entry) <= (SELECT_BUCKET - 24) * ROWS_W * sizeof(Entry), "the sorted chunk fits behind 24 list rows");
    Entry *sorted = lists + 24 * ROWS_W;
    __shared__ uint32_t rcnt[ROWS_W], roff[ROWS_W];
    const int lane = threadIdx.x;
    const int kr = (k + 3) & ~3;
    const int64_t region = blockIdx.x;
    const int64_t wave_row0 = region * ROWS_W;
    const int total = cand_cnt[region];
    const bool lost = total > KNN_REGION_CAP;
    const int n = lost ? 0 : total;
    const Entry *src = cand_buf + region * KNN_REGION_CAP;
    for (int t = 0; t < SELECT_BUCKET; ++t) lists[t * ROWS_W + lane] = Entry{-__builtin_inff(), 0x3ffffffu};       // unfilled slots sort last
    rcnt[lane] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const bool try_buckets = n <= SELECT_BUCKET * ROWS_W;               // (more entries than all buckets hold — a cold emit sweep: straight to the general path)
    constexpr int SEL_FLIGHT = 8;                                        // candidates per lane in flight (round 6: 8 instead of 4 — the kernel waits for these loads at 9 waves per CU)
    for (int e0 = 0; try_buckets && e0 < n; e0 += SEL_FLIGHT * 64) {
        Entry en[SEL_FLIGHT];
#pragma unroll
        for (int u = 0; u < SEL_FLIGHT; ++u) { const int e = e0 + u * 64 + lane; en[u] = src[e < n ? e : n - 1]; }
#pragma unroll
        for (int u = 0; u < SEL_FLIGHT; ++u) {
            if (e0 + u * 64 + lane < n) {
                const uint32_t row = en[u].packed >> 26;
                const uint32_t slot = atomicAdd(&rcnt[row], 1u);
                if (slot < (uint32_t)SELECT_BUCKET) lists[slot * ROWS_W + row] = Entry{en[u].val, en[u].packed & 0x3ffffffu};
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int seen = (int)rcnt[lane];
    if (try_buckets && __builtin_amdgcn_ballot_w64(seen > SELECT_BUCKET) == 0ull) {
        int nin = seen < k ? k : seen;                                   // the sentinels of a short row take part in the sort, as in the list sweep
        knn_write_out(lists, rcnt, lane, lost ? 0 : (seen < k ? seen : k), k, N, wave_row0, 0, 1, 0, false, xnorm, row_perm, flags, out_idx, out_d2, nin);
        return;
    }
    // ---- general path (some row overflowed its bucket): from scratch
    __builtin_amdgcn_wave_barrier();
    for (int t = 0; t < k; ++t) lists[t * ROWS_W + lane] = Entry{-__builtin_inff(), 0x3ffffffu};
    for (int t = k; t < kr; ++t) lists[t * ROWS_W + lane] = Entry{__builtin_inff(), 0u};
    int lcount = 0, lminpos = 0;
    float lmin_v = __builtin_inff();
    uint32_t lmin_i = 0;
    // (k > 24: the lists reach into the rows the sorted chunk borrows — such a region, rare twice over, goes to the cold repair whole: its rows end with 0 entries)
    const int ng = kr > 24 ? 0 : n;
    for (int c0 = 0; c0 < ng; c0 += SELECT_CHUNK) {
        const int cn = min(SELECT_CHUNK, ng - c0);
        rcnt[lane] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int e = lane; e < cn; e += 64) atomicAdd(&rcnt[src[c0 + e].packed >> 26], 1u);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int c = (int)rcnt[lane];
        int incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int tt = __shfl_up(incl, o, 64); if (lane >= o) incl += tt; }
        const int off = incl - c;
        roff[lane] = (uint32_t)off;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int e = lane; e < cn; e += 64) {                            // second pass over the chunk (L2-resident): slot = running offset of the entry's row
            const Entry en = src[c0 + e];
            sorted[atomicAdd(&roff[en.packed >> 26], 1u)] = en;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int i = 0; i < c; ++i) {
            const Entry en = sorted[off + i];
            const uint32_t yi = en.packed & 0x3ffffffu;
            bool rescan = false;
            if (lcount < k) {
                lists[lcount * ROWS_W + lane] = Entry{en.val, yi};
                ++lcount;
                rescan = (lcount == k);
            } else if (better(en.val, yi, lmin_v, lmin_i)) {
                lists[lminpos * ROWS_W + lane] = Entry{en.val, yi};
                rescan = true;
            }
            if (rescan) {   // worst entry of the list = current k-th best
                float wv = __builtin_inff(); uint32_t wi = 0; int wp = 0;
                for (int t = 0; t < kr; t += 4) {
                    Entry le[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) le[u] = lists[(t + u) * ROWS_W + lane];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (better(wv, wi, le[u].val, le[u].packed) || (t + u) == 0) { wv = le[u].val; wi = le[u].packed; wp = t + u; }
                }
                lmin_v = wv; lmin_i = wi; lminpos = wp;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    knn_write_out(lists, rcnt, lane, lost ? 0 : lcount, k, N, wave_row0, 0, 1, 0, false, xnorm, row_perm, flags, out_idx, out_d2);
}

// p-way merge of a split sweep's partial lists, thread per row.  Each part lists its best k in score order (raw scores, unfilled slots last); the k
// best scores over the parts are selected with the sweep's own order (score descending, ties: lower index) and only then converted to squared
// distances, with the sweep's fix-up for scores that round to the same distance (index ascending) — bit for bit what an unsplit sweep returns.
__global__ __launch_bounds__(256) void knn_merge_parts_kernel(const int32_t *__restrict__ pidx, const float *__restrict__ pval, int64_t rows, int64_t row0, int64_t N, int parts,
                                                              int k, const float *__restrict__ xnorm, int32_t *__restrict__ out_idx, float *__restrict__ out_d2,
                                                              int32_t *__restrict__ flags, const int32_t *__restrict__ row_perm, const int32_t *__restrict__ n_dev) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n_dev) rows = min(rows, (int64_t)*n_dev);
    if (r >= rows || row0 + r >= N) return;
    const int64_t row = row_perm ? (int64_t)row_perm[row0 + r] : row0 + r;           // partial lists are stored by position, results by row
    int head[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int32_t *pi = pidx + r * parts * k; const float *pv = pval + r * parts * k;
    int32_t *oi = out_idx + row * k; float *od = out_d2 + row * k;
    int cnt = 0;
    float hv[8]; uint32_t hx[8];                                                    // the parts' current heads, kept in registers: one dependent load pair per output
#pragma unroll
    for (int p = 0; p < 8; ++p) { hv[p] = 0.f; hx[p] = 0x3ffffffu; if (p < parts) { hv[p] = pv[p * k]; hx[p] = (uint32_t)pi[p * k]; } }
    for (int t = 0; t < k; ++t) {
        int best = -1; float bv = 0.f; uint32_t bi = 0;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            if (p >= parts || hx[p] >= 0x3ffffffu) continue;                        // part exhausted / unfilled slot of a short part
            if (best < 0 || better(hv[p], hx[p], bv, bi)) { best = p; bv = hv[p]; bi = hx[p]; }
        }
        if (best < 0) break;
        oi[t] = (int32_t)bi; od[t] = bv; ++cnt;
        int h = 0;
#pragma unroll
        for (int p = 0; p < 8; ++p) if (p == best) h = ++head[p];                  // unrolled selects: no dynamically indexed register array
        const float nv = h < k ? pv[best * k + h] : 0.f; const uint32_t nx = h < k ? (uint32_t)pi[best * k + h] : 0x3ffffffu;
#pragma unroll
        for (int p = 0; p < 8; ++p) if (p == best) { hv[p] = nv; hx[p] = nx; }
    }
    if (cnt < k && flags) {                                                         // fewer than k entries over all parts: too tight a warm-start bound
        const int slot = atomicAdd(&flags[0], 1);
        if (slot < KNN_FLAG_CAP) flags[KNN_FLAG_ROW0 + slot] = (int32_t)row;
    }
    for (int t = cnt; t < k; ++t) { oi[t] = 0x3ffffff; od[t] = __builtin_inff(); }  // unfilled slots read like an unsplit sweep's: sentinel index, +inf distance
    const float xn = xnorm[row];
    for (int t = 0; t < cnt; ++t) {
        const uint32_t bi = (uint32_t)oi[t];
        const float dcur = fmaxf(xn - 2.0f * od[t], 0.f);
        int pos = t;
        while (pos > 0 && od[pos - 1] == dcur && (uint32_t)oi[pos - 1] > bi) { od[pos] = od[pos - 1]; oi[pos] = oi[pos - 1]; --pos; }
        od[pos] = dcur; oi[pos] = (int32_t)bi;
    }
}

// The same merge with the 64 rows' partial lists staged in LDS: one wave per 64 consecutive positions reads their parts*k entries as ONE contiguous block
// (coalesced), every lane merges its row out of LDS (row stride padded to an odd number of words: no bank conflicts), and the results leave as one 80-byte run per
// row.  The thread-per-row kernel above walks global memory with one dependent, uncoalesced load pair per output: 0.28 ms for 125 k rows x 5 parts — a sixth of a
// rank's sweep at 8 GPUs.  Same selection, same tie rule, same flags.
__global__ __launch_bounds__(64) void knn_merge_parts_lds_kernel(const int32_t *__restrict__ pidx, const float *__restrict__ pval, int64_t rows, int64_t row0, int64_t N, int parts,
                                                                  int k, const float *__restrict__ xnorm, int32_t *__restrict__ out_idx, float *__restrict__ out_d2,
                                                                  int32_t *__restrict__ flags, const int32_t *__restrict__ row_perm, const int32_t *__restrict__ n_dev) {
    extern __shared__ __attribute__((aligned(16))) uint32_t merge_lds[];
    if (n_dev) rows = min(rows, (int64_t)*n_dev);                        // the repair's merge: the row count is a device value (normally 0: every block leaves at once)
    const int E = parts * k, ES = E | 1;                                 // entries per row, padded stride
    float *lv = reinterpret_cast<float *>(merge_lds);                    // [64][ES] scores
    uint32_t *li = merge_lds + 64 * ES;                                  // [64][ES] indices
    float *ov = reinterpret_cast<float *>(li + 64 * ES);                 // [64][k] merged distances
    uint32_t *oi = reinterpret_cast<uint32_t *>(ov + 64 * k);            // [64][k] merged indices
    const int lane = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * 64;
    const int64_t nrow = min((int64_t)64, min(rows - r0, N - row0 - r0));
    if (nrow <= 0) return;
    const int64_t total = nrow * E;
    for (int64_t e = lane; e < total; e += 64) {                         // coalesced: the block's entries are contiguous in the partial arrays
        const int rr = (int)(e / E), cc = (int)(e % E);
        lv[rr * ES + cc] = pval[r0 * E + e];
        li[rr * ES + cc] = (uint32_t)pidx[r0 * E + e];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const bool live = lane < nrow;
    int64_t row = 0;
    if (live) {
        row = row_perm ? (int64_t)row_perm[row0 + r0 + lane] : row0 + r0 + lane;
        const float *pv = lv + lane * ES; const uint32_t *pi = li + lane * ES;
        int head[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        float hv[8]; uint32_t hx[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) { hv[p] = 0.f; hx[p] = 0x3ffffffu; if (p < parts) { hv[p] = pv[p * k]; hx[p] = pi[p * k]; } }
        int cnt = 0;
        for (int t = 0; t < k; ++t) {
            int best = -1; float bv = 0.f; uint32_t bi = 0;
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                if (p >= parts || hx[p] >= 0x3ffffffu) continue;
                if (best < 0 || better(hv[p], hx[p], bv, bi)) { best = p; bv = hv[p]; bi = hx[p]; }
            }
            if (best < 0) break;
            oi[lane * k + t] = bi; ov[lane * k + t] = bv; ++cnt;
            int h = 0;
#pragma unroll
            for (int p = 0; p < 8; ++p) if (p == best) h = ++head[p];
            const float nv = h < k ? pv[best * k + h] : 0.f; const uint32_t nx = h < k ? pi[best * k + h] : 0x3ffffffu;
#pragma unroll
            for (int p = 0; p < 8; ++p) if (p == best) { hv[p] = nv; hx[p] = nx; }
        }
        if (cnt < k && flags) {
            const int slot = atomicAdd(&flags[0], 1);
            if (slot < KNN_FLAG_CAP) flags[KNN_FLAG_ROW0 + slot] = (int32_t)row;
        }
        for (int t = cnt; t < k; ++t) { oi[lane * k + t] = 0x3ffffffu; ov[lane * k + t] = __builtin_inff(); }
        const float xn = xnorm[row];
        for (int t = 0; t < cnt; ++t) {                                  // scores -> distances, exact distance ties ordered by index (as the unsplit sweep's write-out)
            const uint32_t bi = oi[lane * k + t];
            const float dcur = fmaxf(xn - 2.0f * ov[lane * k + t], 0.f);
            int pos = t;
            while (pos > 0 && ov[lane * k + pos - 1] == dcur && oi[lane * k + pos - 1] > bi) { ov[lane * k + pos] = ov[lane * k + pos - 1]; oi[lane * k + pos] = oi[lane * k + pos - 1]; --pos; }
            ov[lane * k + pos] = dcur; oi[lane * k + pos] = bi;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // results leave row by row: k consecutive entries per row (rows land at their own places: row_perm)
    for (int rr = 0; rr < (int)nrow; ++rr) {
        const int64_t orow = __shfl(row, rr, 64);
        if (lane < k) { out_idx[orow * k + lane] = (int32_t)oi[rr * k + lane]; out_d2[orow * k + lane] = ov[rr * k + lane]; }
    }
}


// picks the LDS-staged merge when its staging fits (k <= 64 and parts * k entries per row: 64 rows x (2 x padded entries + 2 x k) words)
inline void launch_merge_parts(hipStream_t st, const int32_t *pidx, const float *pval, int64_t rows, int64_t row0, int64_t N, int parts, int k, const float *xnorm, int32_t *idx,
                               float *d2, int32_t *flags, const int32_t *row_perm, const int32_t *n_dev = nullptr) {
    const size_t lds = (size_t)64 * (2 * ((size_t)(parts * k) | 1) + 2 * (size_t)k) * 4;
    if (k <= 64 && parts <= 8 && lds <= 96 * 1024) {
        static std::atomic<size_t> lds_set{0};
        if (lds_set.load(std::memory_order_acquire) < lds) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(knn_merge_parts_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            lds_set.store(96 * 1024, std::memory_order_release);
        }
        hipLaunchKernelGGL(knn_merge_parts_lds_kernel, dim3((unsigned)((rows + 63) / 64)), dim3(64), lds, st, pidx, pval, rows, row0, N, parts, k, xnorm, idx, d2, flags, row_perm, n_dev);
    } else {
        hipLaunchKernelGGL(knn_merge_parts_kernel, dim3((unsigned)((rows + 63) / 64)), dim3(64), 0, st, pidx, pval, rows, row0, N, parts, k, xnorm, idx, d2, flags, row_perm, n_dev);
    }
}

// Repair of rows whose warm-start bound turned out too tight: their feature rows are gathered into a compact matrix, the kNN kernel sweeps it with
// a cold bound (row count read on the device), the results are scattered back.  Normally zero rows: three launches that leave at once.
template <int ROWB>
__global__ __launch_bounds__(256) void knn_gather_flagged_kernel(const uint8_t *__restrict__ X, const int32_t *__restrict__ xsc, const float *__restrict__ xnorm,
                                                                 const int32_t *__restrict__ flags, uint8_t *__restrict__ Xc, int32_t *__restrict__ xscc,
                                                                 float *__restrict__ xnormc) {
    const int n = min(flags[0], KNN_FLAG_CAP);
    constexpr int LPR = ROWB / 16;
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < (int64_t)n * LPR; g += (int64_t)gridDim.x * 256) {
        const int slot = (int)(g / LPR), c = (int)(g % LPR);
        const int64_t row = flags[KNN_FLAG_ROW0 + slot];
        reinterpret_cast<uint4 *>(Xc + (size_t)slot * ROWB)[c] = reinterpret_cast<const uint4 *>(X + row * ROWB)[c];
        if (c == 0) { xnormc[slot] = xnorm[row]; if (xsc) xscc[slot] = xsc[row]; }
    }
}
__global__ __launch_bounds__(256) void knn_scatter_flagged_kernel(int32_t *__restrict__ flags, const int32_t *__restrict__ idxc, const float *__restrict__ d2c, int k,
                                                                  int32_t *__restrict__ out_idx, float *__restrict__ out_d2, int fallback_follows = 0) {
    const int n = min(flags[0], KNN_FLAG_CAP);
    // (fallback_follows: a count beyond the repair's capacity is not an unrepaired sweep — the guarded list sweep launched next redoes every row; flags[0] keeps saying so)
    if (blockIdx.x == 0 && threadIdx.x == 0 && flags[0] > flags[1] && !(fallback_follows && flags[0] > KNN_FLAG_CAP)) flags[1] = flags[0];       // sticky maximum (one writer: launches on a workspace are stream-ordered)
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < (int64_t)n * k; g += (int64_t)gridDim.x * 256) {
        const int slot = (int)(g / k), t = (int)(g % k);
        const int64_t row = flags[KNN_FLAG_ROW0 + slot];
        out_idx[row * k + t] = idxc[(size_t)slot * k + t];
        out_d2[row * k + t] = d2c[(size_t)slot * k + t];
    }
}

// PRUNED sweeps (exact): which style tiles can a row block skip?  A tile t is given by a centre c_t (any point: here the tile's mean row rounded to the feature
// format) and the radius r_t = max |y - c_t| over its 64 rows.  For a content row x with bound thr_x (a candidate needs score x.y - |y|^2/2 >= thr_x):
//     x.y - |y|^2/2 = |x|^2/2 - |x - y|^2/2 <= |x|^2/2 - max(0, |x - c_t| - r_t)^2 / 2        for every y of the tile (triangle inequality),
// so no row of tile t can reach thr_x when the right-hand side stays below it.  The kernel evaluates that for the 512 rows of a row block against all tile centres
// (x.c_t on the matrix cores: 1/64 of the sweep's flops), marks the tiles ANY of its rows may need and writes the block's tile order (needed tiles first, ascending)
// and their number; knn_mfma_hot_kernel then sweeps only those.  In the kernel the test reads x.c_t >= (|x|^2 + |c_t|^2 - (r_t + D_x)^2) / 2 with D_x^2 = |x|^2 - 2 thr_x
// (one compare per (row, centre) after four flops); margins: (r + D)^2 grown by 2^-9 of itself, the threshold lowered by 2e-5 (|x|^2 + |c|^2) — orders of magnitude
// above the f32 accumulation error of x.c and of the sweep's own scores (~1e-6 relative), and a thousandth of the distances that decide anything.  Measured on the C3 loop (scripts/prune_probe.py): 46 % of the (row block, tile) pairs are skipped on the
// density-matched style patch, 99.5 % on the sparse one.
// ---- plan helpers shared by the ball-test kernel (leaf objects) and the steady kernel (main object)
// q_x = cum_x + D_x (+ margins) of the block's 512 rows into qlds; D_x exactly as the ball test takes it
// dlds / dref (optional): the row's REFERENCE value — its q at the last ball-test plan.  The summaries are kept relative to it (min over rows of bound - dref), and
// level 1 compares them with max over rows of q - dref: what enters the comparison is how much each row's reach has CHANGED since the reference, not the spread of the
// reaches over the rows of a sub-block (which made level 1 keep 4 % more tiles than the per-row ball test at C3, 25 % more at M = 1e6).
__device__ __forceinline__ float plan_fill_q(float *qlds, const int32_t *__restrict__ row_perm, const float *__restrict__ xnorm, const float *__restrict__ thr,
                                             const float *__restrict__ lb_cum, int64_t N, float cnorm_max, float mrg, float *dlds = nullptr, float *__restrict__ dref = nullptr,
                                             bool write_ref = false) {
    int64_t pos = (int64_t)blockIdx.x * ROWS_B + threadIdx.x;
    const bool live = pos < N;
    if (!live) pos = N - 1;                                              // positions past the end: the sweep computes them as copies of the last row, and so are their bounds
    const int64_t row = row_perm ? (int64_t)row_perm[pos] : pos;
    const float x2 = xnorm[row];
    const float dx = sqrtf(fmaxf(x2 - 2.0f * thr[row], 0.f) + mrg * (x2 + 2.0f * cnorm_max));      // mrg = 2E / (|x|^2 + 2 max |y|^2): the sweep's score error (4e-6; plans for fp8 sweeps 2e-5)
    const float q = (lb_cum[row] + dx) * (1.0f + 1e-6f) + 1e-7f;
    qlds[threadIdx.x] = q;
    if (dlds) {
        float r = q;                                                     // a ball-test plan: the reference becomes this plan's q
        if (!write_ref) r = dref[row];
        else if (live) dref[row] = q;
        dlds[threadIdx.x] = r;
        return q - r;
    }
    return q;
}
// per-row cached bounds of the tiles marked in `need`: a tile stays only if some row's bound - cum_now <= D_x.  Wave w owns the bitmap words w, w + 8, ...
// STRAGGLERS (round 6; strag_T > 0, steady plans of whole blocks only): half of the (block, tile) pairs the plan keeps at C3 are kept for at most 16 of the block's 512 rows —
// a quarter for ONE row (profiles/r06av_row_need.log) — and the sweep then runs the tile for all 512.  A kept tile that at most strag_T rows still need leaves the block's
// list; the rows' positions (pos0 + row of the block) go to the tile's bucket (strag [tiles][cap], counts strag_cnt), and knn_straggler_kernel computes exactly those
// (row, tile) pairs behind the sweep — same scores bit for bit, candidates into the same regions, the rows' cached bounds refreshed.  A bucket that is full: the tile stays.
__device__ __forceinline__ void plan_bounds_phase(uint32_t *need, int words, const float *qlds, const uint16_t *__restrict__ lbb, int lane, int wid, float *summ_row = nullptr,
                                                  const float *dlds = nullptr, uint32_t *__restrict__ strag_cnt = nullptr, uint32_t *__restrict__ strag = nullptr, int strag_cap = 0,
                                                  int strag_T = 0, uint32_t pos0 = 0, uint8_t *mjl = nullptr /*the wave's [32][64] bytes of LDS*/) {
    float q8[8], d8[8];                                                  // lane l of every wave: rows 8 l ... 8 l + 7 of the block (one 16-byte load of a tile's bounds)
#pragma unroll
    for (int j = 0; j < 8; ++j) { q8[j] = qlds[8 * lane + j]; d8[j] = dlds ? dlds[8 * lane + j] : 0.f; }
    constexpr int LBB = 8;                                               // tiles in flight per wave
    for (int wi = wid; wi < words; wi += NWAVES) {
        uint32_t bits = need[wi];
        uint32_t keep = bits;
        // stragglers: the row masks of the word's straggler tiles wait in the wave's LDS area (mjl [32 tiles][64 lanes] bytes) until the word's tiles are all tested; lane t
        // then reserves room in the bucket of tile t of the word — ONE atomic round trip per word of 32 tiles (per tile: 0.1 ms of the plan at C3; per batch of 8: the same)
        uint32_t smask = 0u;                                             // tiles of this word that at most strag_T rows still need
        int my_tot = 0;                                                  // lane t: that number for tile t of the word
        while (bits) {
            int tt[LBB]; uint4 v[LBB];
#pragma unroll
            for (int u = 0; u < LBB; ++u) {
                tt[u] = -1;
                if (bits) { tt[u] = __builtin_ctz(bits); bits &= bits - 1u; }
            }
#pragma unroll
            for (int u = 0; u < LBB; ++u)
                if (tt[u] >= 0) v[u] = *reinterpret_cast<const uint4 *>(lbb + ((int64_t)wi * 32 + tt[u]) * ROWS_B + 8 * lane);
#pragma unroll
            for (int u = 0; u < LBB; ++u) {
                if (tt[u] < 0) continue;
                const uint32_t wv[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
                uint32_t mj = 0u;                                        // bit r: row 8 lane + r of the block may still have a candidate in the tile
                float lo = __builtin_inff();
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    _Float16 h0, h1; const uint16_t b0 = (uint16_t)(wv[j] & 0xffffu), b1 = (uint16_t)(wv[j] >> 16);
                    __builtin_memcpy(&h0, &b0, 2); __builtin_memcpy(&h1, &b1, 2);
                    mj |= ((float)h0 <= q8[2 * j] ? 1u : 0u) << (2 * j) | ((float)h1 <= q8[2 * j + 1] ? 1u : 0u) << (2 * j + 1);      // bound - cum_now <= D_x
                    lo = fminf(lo, fminf((float)h0 - d8[2 * j], (float)h1 - d8[2 * j + 1]));
                }
                if (__builtin_amdgcn_ballot_w64(mj != 0u) == 0ull) keep &= ~(1u << tt[u]);
                else if (strag_T > 0) {
                    int total = 0;
#pragma unroll
                    for (int r = 0; r < 8; ++r) total += __builtin_popcountll(__builtin_amdgcn_ballot_w64(((mj >> r) & 1u) != 0u));
                    if (total <= strag_T) {
                        smask |= 1u << tt[u];
                        if (lane == tt[u]) my_tot = total;
                        mjl[tt[u] * 64 + lane] = (uint8_t)mj;
                    }
                }
                if (summ_row) {                                          // the summaries of a tile whose per-row bounds were just read: their minimum per 64 rows (kept tiles: see plan_compact)
                    lo = row8_min(lo);                                  // lanes 8 s ... 8 s + 7 hold rows 64 s ... 64 s + 63
                    if ((lane & 7) == 0) summ_row[((int64_t)wi * 32 + tt[u]) * NWAVES + (lane >> 3)] = lo;
                }
            }
        }
        if (smask) {
            uint32_t my_base = 0u;
            if (lane < 32 && ((smask >> lane) & 1u)) my_base = atomicAdd(&strag_cnt[(int64_t)wi * 32 + lane], (uint32_t)my_tot);
            const unsigned long long below = (1ull << lane) - 1ull;
            while (smask) {
                const int tw = __builtin_ctz(smask); smask &= smask - 1u;
                const int64_t tile = (int64_t)wi * 32 + tw;
                const uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)my_base, tw);
                const uint32_t total = (uint32_t)__builtin_amdgcn_readlane(my_tot, tw);
                const bool fits = base + total <= (uint32_t)strag_cap;
                const uint32_t mj = mjl[tw * 64 + lane];
                uint32_t at = base;
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const unsigned long long bmr = __builtin_amdgcn_ballot_w64(((mj >> r) & 1u) != 0u);
                    if (((mj >> r) & 1u) != 0u) {
                        const uint32_t slot = at + (uint32_t)__builtin_popcountll(bmr & below);
                        if (slot < (uint32_t)strag_cap) strag[tile * strag_cap + slot] = fits ? pos0 + 8u * (uint32_t)lane + (uint32_t)r : 0xffffffffu;      // (a full bucket: voided slots, the tile stays)
                    }
                    at += (uint32_t)__builtin_popcountll(bmr);
                }
                if (fits) keep &= ~(1u << tw);
            }
        }
        if (lane == 0) need[wi] = keep;
    }
}
// needed tiles first (ascending), then the others; the count of the first group; summ_row (optional): the block summaries of the KEPT tiles are set to -inf
// ("unknown"): the sweep is about to rewrite their per-row bounds, and the next plan's level 2 — which such a tile always reaches — recomputes the summary from them
__device__ __forceinline__ void plan_compact(const uint32_t *need, int64_t ntiles, int32_t *out, int32_t *tile_count_b, int *wave_tot, int lane, int wid, float *summ_row) {
    int base = 0;
    for (int pass = 0; pass < 2; ++pass) {
        for (int64_t t0 = 0; t0 < ntiles; t0 += 512) {
            const int64_t t = t0 + threadIdx.x;
            const bool set = t < ntiles && (((need[t >> 5] >> (t & 31)) & 1u) != 0u);
            const bool take = t < ntiles && (pass == 0 ? set : !set);
            if (pass == 0 && set && summ_row) {
                const float4 ni = make_float4(-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff());
                reinterpret_cast<float4 *>(summ_row + t * NWAVES)[0] = ni; reinterpret_cast<float4 *>(summ_row + t * NWAVES)[1] = ni;
            }
            const unsigned long long m = __builtin_amdgcn_ballot_w64(take);
            const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
            if (lane == 0) wave_tot[wid] = __builtin_popcountll(m);
            __syncthreads();
            int before = 0, total = 0;
#pragma unroll
            for (int w = 0; w < 8; ++w) { const int c = wave_tot[w]; if (w < wid) before += c; total += c; }
            if (take) out[base + before + rank] = (int32_t)t;
            base += total;
            __syncthreads();
        }
        if (pass == 0 && threadIdx.x == 0) *tile_count_b = base;
    }
}

constexpr int PRUNE_CG = 128;            // centre groups (of 64) whose constants the plan kernel keeps in LDS at a time (64 KB)
template <int D, int FMT, bool SUMM = false>
__global__ __launch_bounds__(512) void knn_tile_prune_kernel(const uint8_t *__restrict__ X, const int32_t *__restrict__ row_perm, const float *__restrict__ xnorm,
                                                             const float *__restrict__ thr, const uint8_t *__restrict__ C, const float *__restrict__ cnorm,
                                                             const float *__restrict__ cradius, int64_t N, int64_t ntiles, int32_t *__restrict__ tile_order,
                                                             int32_t *__restrict__ tile_count, float cnorm_max, int64_t *__restrict__ clk,
                                                             const uint16_t *__restrict__ lb, const float *__restrict__ lb_cum, float *__restrict__ summ, float *__restrict__ dref,
                                                             float mrg) {
    const int64_t clk0 = clk ? (int64_t)wall_clock64() : 0;
    // Round 3: the centres stream through a 2-deep LDS ring by global_load_lds (XOR-swizzled 16-B chunks, as the sweep's style tiles) — group g + 1 lands while group g
    // is on the matrix cores; fragment and per-centre constant reads are explicit ds_reads with hand-placed lgkmcnt waits (a compiler-visible LDS load after an LDS-DMA
    // makes the waitcnt pass drain vmcnt(0), i.e. wait for the NEXT group before computing this one).  Round 2 staged every group synchronously through registers:
    // 0.84 ms at C3, a third of it the exposed staging latency.
    static_assert(FMT != FMT_FP8, "16-bit rows");
    constexpr int KS = D / 32, ROWB = D * 2, CPR = ROWB / 16, TILE_BYTES = 64 * ROWB, NT = 512;
    constexpr int PIECES = (64 * CPR + NT - 1) / NT;                   // LDS-DMA instructions per wave and centre group (4 at d = 256)
    typedef typename std::conditional<FMT == FMT_F16, f16x8, bf16x8>::type frag_t;
    typedef const __attribute__((address_space(1))) void *gptr_t;
    typedef __attribute__((address_space(3))) void *lptr_t;
    extern __shared__ __attribute__((aligned(16))) uint8_t prune_lds[];
    uint8_t *ctile = prune_lds;                                       // [2][64][ROWB], swizzled
    const int64_t ncent_tiles = (ntiles + 63) / 64;                      // centres are taken 64 at a time
    // [cg][128]: B_c and (1 + 2e) r of the centres of up to PRUNE_CG groups (8 KB at 1024 tiles: ALL of them, no global load inside the loop; a style set of more
    // than 8192 tiles — M = 1e6: 15 625 — refills the table every PRUNE_CG groups)
    const int cg = (int)(ncent_tiles < (int64_t)PRUNE_CG ? ncent_tiles : (int64_t)PRUNE_CG);
    float *cinfo = reinterpret_cast<float *>(prune_lds + 2 * TILE_BYTES);
    uint32_t *need = reinterpret_cast<uint32_t *>(prune_lds + 2 * TILE_BYTES + (size_t)cg * 128 * 4);   // bitmap over the tiles
    __shared__ int wave_tot[8];
    const int lane = threadIdx.x & 63, xr = lane & 15, h = lane >> 4;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int words = (int)((ntiles + 31) / 32);
    for (int i = threadIdx.x; i < words; i += 512) need[i] = 0u;
    // content rows of this wave: B operand fragments in registers, as in the sweep
    frag_t xf[4][KS];
    float xn[4], th[4];
    float x2s[SUMM ? 4 : 1], cxs[SUMM ? 4 : 1];                          // SUMM: |x|^2 and cumulative motion of the lane's rows (block summaries of the ball bounds)
#pragma unroll
    for (int xb = 0; xb < 4; ++xb) {
        int64_t pos = (int64_t)blockIdx.x * ROWS_B + wid * ROWS_W + xb * 16 + xr;
        const bool live = pos < N;
        if (!live) pos = N - 1;
        const int64_t row = row_perm ? (int64_t)row_perm[pos] : pos;
#pragma unroll
        for (int s = 0; s < KS; ++s) xf[xb][s] = *reinterpret_cast<const frag_t *>(X + row * ROWB + (4 * s + h) * 16);
        xn[xb] = xnorm[row];
        // D_x = the distance within which a candidate of row x must lie: score >= thr  <=>  |x - y|^2 <= |x|^2 - 2 thr.
        // The test x.c >= (|x|^2 + |c|^2)/2 - (1/2 + e)(r + D_x)^2 - m (|x|^2 + |c|^2)   [e = 2^-10, m = 2e-5: the margins]
        // splits into a row part A_x, a centre part B_c and one product:  x.c - A_x + (1 + 2e) r D_x >= B_c  — the accumulators START at -A_x, so the test is one
        // fma and one compare per (row, centre).  Rows past the end start at -inf: no tile passes.
        // D_x is taken from D_x^2 + 2E with E = 2e-6 (|x|^2 + 2 max|c|^2) >= the sweep's score error (f32 accumulation of x.y, |y|^2/2, the norm pass over x): a row
        // whose COMPUTED score reaches the bound lies within that enlarged distance, also when D_x itself is ~0 (duplicates), where an error in the score is a large error
        // in the distance
        if constexpr (SUMM) { x2s[xb] = xn[xb]; cxs[xb] = lb_cum[row]; }
        th[xb] = live ? sqrtf(fmaxf(xn[xb] - 2.0f * thr[row], 0.f) + mrg * (xn[xb] + 2.0f * cnorm_max)) : 0.f;
        if constexpr (SUMM) {
            // cum - reference of the row, the reference being THIS plan's q (plan_fill_q writes the same expression to dref after the loop): the summaries are relative to it
            const float dxl = sqrtf(fmaxf(x2s[xb] - 2.0f * thr[row], 0.f) + mrg * (x2s[xb] + 2.0f * cnorm_max));
            cxs[xb] = cxs[xb] - ((cxs[xb] + dxl) * (1.0f + 1e-6f) + 1e-7f);
        }
        xn[xb] = live ? -((0.5f - 2e-5f) * xn[xb] - (0.5f + 0.0009765625f) * th[xb] * th[xb]) : -__builtin_inff();      // from here on: -A_x
    }
    // centre chunk (row r, slot c') of piece i: thread chunk id q = tid + NT i, r = q / CPR, c' = q % CPR; the slot holds logical chunk c' ^ swz(r)
    auto issue_group = [&](int64_t ct, int buf) {
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const int q0 = wid * 64 + NT * i;                          // wave-uniform first chunk of this instruction
            if (q0 < 64 * CPR) {
                const int q = q0 + lane, r = q / CPR, cs = q % CPR;
                int64_t t = ct * 64 + r; if (t >= ntiles) t = ntiles - 1;      // rows past the end: the last centre again (never marked: `valid`)
                __builtin_amdgcn_global_load_lds((gptr_t)(C + t * ROWB + ((cs ^ swz<ROWB>(r)) << 4)), (lptr_t)(ctile + buf * TILE_BYTES + q0 * 16), 16, 0, 0);
            }
        }
    };
    auto fill_cinfo = [&](int64_t g0) {                                  // B_c (0..63) and (1 + 2e) r (64..127) of groups g0 ... g0 + cg - 1
        for (int i = threadIdx.x; i < cg * 128; i += 512) {
            int64_t t = (g0 + (i >> 7)) * 64 + (i & 63); if (t >= ntiles) t = ntiles - 1;
            const float cr = cradius[t];
            cinfo[i] = (i & 64) == 0 ? (0.5f - 2e-5f) * cnorm[t] - (0.5f + 0.0009765625f) * cr * cr : (1.0f + 0.001953125f) * cr;
        }
    };
    fill_cinfo(0);
    issue_group(0, 0);
    uint32_t lane_off = (uint32_t)((((swz<ROWB>(xr) >> 2) << 2 | (h ^ (swz<ROWB>(xr) & 3))) << 4));      // slot of logical chunk 4 s + h of row (.. + xr): XOR-linear in s
    asm volatile("" : "+v"(lane_off));
    const int64_t clk1 = clk ? (int64_t)wall_clock64() : 0;
    for (int64_t ct = 0; ct < ncent_tiles; ++ct) {
        const int cur = (int)(ct & 1);
        // every wave drains its OWN LDS-DMA before the barrier — explicitly: the compiler does not order global_load_lds against s_barrier by itself (round 3 shipped this
        // loop for a few hours without the wait: alone on the chip the centres had always landed in time; inside the loop, next to the render leg's memory traffic, a
        // wave now and then read a group before it was complete, a needed tile went unmarked, and the sweep lost a neighbour — found by scripts/repair_trace.py,
        // held by tests/test_gpu_geometry.py::test_prune_plan_is_exact_under_memory_contention)
#ifndef W3D_TEST_NO_DMA_WAIT                                              // (scripts/_ab: a library WITHOUT the wait, to show that the contention test has teeth)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        __syncthreads();                                                 // group ct has landed; group ct - 1 is consumed
        const int cslot = (int)(ct % cg);
        if (cslot == 0 && ct > 0) {                                      // (style sets of more than PRUNE_CG x 64 tiles) the next PRUNE_CG groups' constants; no LDS-DMA is in flight here
            fill_cinfo(ct);
            __syncthreads();
        }
        if (ct + 1 < ncent_tiles) issue_group(ct + 1, cur ^ 1);
        unsigned long long tilemask = 0ull;                              // centres of this group of 64 that some row of the wave may need
#pragma unroll
        for (int yb = 0; yb < 2; ++yb) {
            f32x4 acc[2][4];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int xb = 0; xb < 4; ++xb) acc[u][xb] = f32x4{xn[xb], xn[xb], xn[xb], xn[xb]};
            // rows yb*32 + u*16 + xr: (row & 15) == xr (and ((row >> 1) & 7) == (xr >> 1) for 128-B rows... see swz): one lane-constant slot offset
            const uint32_t tb0 = lds_addr(ctile + cur * TILE_BYTES + (yb * 32 + xr) * ROWB);
            frag_t a[2][2];
            auto frag_read = [&](int s, frag_t (&dst)[2]) {
                const uint32_t ad = tb0 + (lane_off ^ (uint32_t)(s << 6));
                asm volatile("ds_read_b128 %0, %1" : "=v"(dst[0]) : "v"(ad));
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[1]) : "v"(ad), "n"(16 * ROWB));
            };
            frag_read(0, a[0]);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if (s + 1 < KS) {
                    frag_read(s + 1, a[(s + 1) & 1]);
                    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a[s & 1][0]), "+v"(a[s & 1][1]));
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[s & 1][0]), "+v"(a[s & 1][1]));
                }
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int xb = 0; xb < 4; ++xb) {
                        if constexpr (FMT == FMT_F16) acc[u][xb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[s & 1][u], xf[xb][s], acc[u][xb], 0, 0, 0);
                        else acc[u][xb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s & 1][u], xf[xb][s], acc[u][xb], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            // per-centre constants of the 2 x 4 centres this lane tests: B_c and (1 + 2e) r of centres yb*32 + u*16 + 4h + (0..3)
            f32x4 bcv[2], crv[2];
            const uint32_t ci0 = lds_addr(cinfo + cslot * 128 + yb * 32 + 4 * h);
            asm volatile("ds_read_b128 %0, %1" : "=v"(bcv[0]) : "v"(ci0));
            asm volatile("ds_read_b128 %0, %1 offset:64" : "=v"(bcv[1]) : "v"(ci0));
            asm volatile("ds_read_b128 %0, %1 offset:256" : "=v"(crv[0]) : "v"(ci0));
            asm volatile("ds_read_b128 %0, %1 offset:320" : "=v"(crv[1]) : "v"(ci0));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bcv[0]), "+v"(bcv[1]), "+v"(crv[0]), "+v"(crv[1]));
            // element q of acc[u][xb] in lane (xr, h): content row xb*16 + xr, centre ct*64 + yb*32 + u*16 + 4h + q
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int ci = yb * 32 + u * 16 + 4 * h + q;
                    const bool valid = ct * 64 + ci < ntiles;
                    const float bc = bcv[u][q], cr = crv[u][q];
                    // tile needed by row x  <=>  |x - c| <= r + D_x (with the margins)  <=>  (x.c - A_x) + (1 + 2e) r D_x >= B_c
                    bool flag = false;
#pragma unroll
                    for (int xb = 0; xb < 4; ++xb) flag |= fmaf(cr, th[xb], acc[u][xb][q]) >= bc;
                    flag = flag && valid;
                    if constexpr (SUMM) {
                        // Summary of the BALL bound per 64 rows (the plan right after a reset of the cached bounds): min over the wave's rows of max(|x - c| - r, 0) + cum_x — a
                        // lower bound of every row's distance to every row of the tile + its motion so far, in the units of the per-row bounds.  x.c = acc + A_x (the
                        // accumulators started at -A_x = xn[]); margins as in the test above, on the safe side.  The steady plans (knn_plan_steady_kernel) drop a tile whose
                        // summary exceeds every row's cum + D without touching its 512 per-row bounds — and without this kernel's x.c pass.
                        const int64_t tc = min(ct * 64 + ci, ntiles - 1);
                        const float cn = cnorm[tc];
                        float best = __builtin_inff();
#pragma unroll
                        for (int xb = 0; xb < 4; ++xb) {
                            const float xc = acc[u][xb][q] - xn[xb];
                            const float d2 = x2s[xb] + cn - 2.0f * xc - 4e-5f * (x2s[xb] + cn);
                            best = fminf(best, fmaxf(sqrtf(fmaxf(d2, 0.f)) * (1.0f - 1e-6f) - cr, 0.f) + cxs[xb]);      // cr = (1 + 2^-9) r: on the safe side
                        }
                        best = row16_min(best);                             // the 16 lanes that hold this centre (same h): with the four xb, all 64 rows of the wave
                        // wave w = rows 64 w ... 64 w + 63 of the block.  Overwritten, not merged: this plan also moves the rows' references (dref), and a summary is only
                        // meaningful relative to the references it was formed with
                        if (xr == 0 && valid) summ[((int64_t)blockIdx.x * ntiles + ct * 64 + ci) * NWAVES + wid] = best - 0.0009765625f * fabsf(best);
                    }
                    const unsigned long long m = __builtin_amdgcn_ballot_w64(flag);       // scalar: the 16 lanes of group hh hold centre yb*32 + u*16 + 4hh + q
#pragma unroll
                    for (int hh = 0; hh < 4; ++hh)
                        if ((m >> (16 * hh)) & 0xffffull) tilemask |= 1ull << (yb * 32 + u * 16 + 4 * hh + q);
                }
        }
        // one pair of LDS atomics per wave and 64 centres (a returning atomic per marked centre was the kernel's time)
        if (lane == 0 && tilemask != 0ull) {
            if ((uint32_t)tilemask) __hip_atomic_fetch_or(&need[2 * ct], (uint32_t)tilemask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if ((uint32_t)(tilemask >> 32)) __hip_atomic_fetch_or(&need[2 * ct + 1], (uint32_t)(tilemask >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    const int64_t clk2 = clk ? (int64_t)wall_clock64() : 0;
    __syncthreads();
    // ---- Round 4: CACHED LOWER BOUNDS (exact, Elkan-style).  When this block last swept tile t (iteration tau) the sweep recorded, per row x of the block, the exact
    // minimum distance m_t(x) to the tile's rows plus the row's cumulative motion at that time: lb = m + cum_tau (f16, rounded down; -inf = never swept).  Rows move by
    // |dx| per iteration and cum sums those, so by the triangle inequality every row y of the tile is still at least lb - cum_now away from x: the tile can hold no
    // candidate of x while lb - cum_now > D_x (D_x = this iteration's bound, the same enlarged distance the ball test uses).  A tile the ball test kept is dropped when
    // that holds for ALL 512 rows.  The ball test alone keeps 0.34 of the (row block, tile) pairs of the C3 loop, both tests 0.11, and 0.09 hold a true neighbour
    // (scripts/cache_bound_probe.py, profiles/r04a_cache_probe.log).  The bounds are keyed by POSITION in the sweep order: the caller resets them (all -inf, cum = 0)
    // whenever the order or the style set changes — the iteration after a reset sweeps what the ball test keeps, and the one after that is back at 0.1.
    if (lb) {
        float *qlds = reinterpret_cast<float *>(ctile);                 // the centre ring is free now: q_x = cum_x + D_x of the block's 512 rows
        float *dlds = qlds + ROWS_B;
        plan_fill_q(qlds, row_perm, xnorm, thr, lb_cum, N, cnorm_max, mrg, summ ? dlds : nullptr, dref, true);
        __syncthreads();
        plan_bounds_phase(need, words, qlds, lb + (int64_t)blockIdx.x * ntiles * ROWS_B, lane, wid, summ ? summ + (int64_t)blockIdx.x * ntiles * NWAVES : nullptr, summ ? dlds : nullptr);
        __syncthreads();
    }
    plan_compact(need, ntiles, tile_order + (int64_t)blockIdx.x * ntiles, tile_count + blockIdx.x, wave_tot, lane, wid, summ ? summ + (int64_t)blockIdx.x * ntiles * NWAVES : nullptr);
    if (clk && threadIdx.x == 0) {                                       // diagnostics: block start, loop start, loop end, block end
        clk[4 * (int64_t)blockIdx.x + 0] = clk0; clk[4 * (int64_t)blockIdx.x + 1] = clk1; clk[4 * (int64_t)blockIdx.x + 2] = clk2;
        clk[4 * (int64_t)blockIdx.x + 3] = (int64_t)wall_clock64();
    }
}

// Split of the last, partially filled round of blocks: 256 CUs each hold one block; nblk % 256 leftover row blocks would run a full
// sweep on a fraction of the chip.  They are launched separately with the style range cut in `parts` (chosen to minimise
// ceil(rem*parts/256)/parts) and merged.  Returns parts (1 = single launch over all row blocks).
int tail_split(int64_t nblk, int64_t M, size_t tail_rows_cap, int mode, int64_t *full_out, int64_t *rem_out, bool warm = true) {
    constexpr int CUS = 256;
    int64_t full = (nblk / CUS) * CUS, rem = nblk - full;
    int parts = 1;
    // a lone partial round (full == 0, e.g. 100k rows = 196 blocks on 256 CUs) is split only under a warm-start bound: with cold bounds every part
    // rediscovers its own top-k and the candidate stream grows with the number of parts (measured: 5.1 -> 8.7 ms at 100k x 100k x 128)
    if (rem > 0 && (full > 0 || warm) && !(mode & 16)) {
        double best = 1.0;
        for (int p = 2; p <= 8; ++p) {
            const double c = (double)((rem * p + CUS - 1) / CUS) / p;
            if (c < best - 1e-9 && (size_t)(rem * ROWS_B) <= tail_rows_cap && ((M + 63) / 64) / p >= 8) { best = c; parts = p; }
        }
    }
    if (parts == 1) { full = nblk; rem = 0; }
    *full_out = full; *rem_out = rem;
    return parts;
}

template <int D, int FMT>
int launch(const uint8_t *X, const uint8_t *Y, const int32_t *xsc, const int32_t *ysc, int64_t N, int64_t M, int k, int32_t *idx, float *d2, float *xnorm, float *ynh,
           const int32_t *prev_idx, float *thr, int phases, int warm, hipStream_t st, const float *prev_dist2, const float *motion2, int32_t *tail_idx, float *tail_d2,
           size_t tail_rows_cap, const float *xnorm_in, int32_t *flags, uint8_t *Xc, int32_t *xscc, float *xnormc, int32_t *idxc, float *d2c, const int32_t *row_perm,
           int32_t *tord, Entry *cand_buf, int32_t *cand_cnt, uint16_t *lb, const float *lb_cum, float lb_margin, float motion_ratio = 0.f, uint32_t *strag_cnt = nullptr,
           uint32_t *strag = nullptr, int strag_cap = 0) {
    constexpr int ROWB = D * Fmt<FMT>::EB;
    constexpr int LPR = ROWB / 16;
    // warm bit 0x100 (with warm = 4 and lb): the plan took straggler pairs out of the tile lists (w3d_knn_prune_plan_err, steady bits 8-15) — knn_straggler_kernel runs them
    const bool strag_on = (warm & 0x100) != 0;
    warm &= 0xff;
    if (strag_on && (warm != 4 || !lb || !strag_cnt || !strag || FMT == FMT_FP8)) return W3D_ERR_ARG;
    const float slack_rel = g_knn_slack > 0.f ? g_knn_slack : (FMT == FMT_FP8 ? KNN_SLACK_REL_FP8 : KNN_SLACK_REL);
    if (phases & 1) {   // prepare: row norms and (optionally) the warm-start bound
        // caller's |x|^2 (e.g. from the featurisation pass); written straight into the workspace's own slot (its first N floats) it is not even copied
        if (xnorm_in) { if (xnorm_in != xnorm) W3D_HIP(hipMemcpyAsync(xnorm, xnorm_in, (size_t)N * sizeof(float), hipMemcpyDeviceToDevice, st)); }
        else hipLaunchKernelGGL((row_sqnorm_kernel<D, FMT>), dim3((unsigned)((N * LPR + 255) / 256)), dim3(256), 0, st, X, xsc, N, N, 1.0f, xnorm);
        const int64_t Mp = (M + YPAD - 1) / YPAD * YPAD;
        if (Y && !(phases & 4)) hipLaunchKernelGGL((row_sqnorm_kernel<D, FMT>), dim3((unsigned)((Mp * LPR + 255) / 256)), dim3(256), 0, st, Y, ysc, M, Mp, -0.5f, ynh);    // Y == NULL: chunked sweep, norms per chunk; phase bit 4: the style rows are the ones of the last prepare on this workspace
        if (prev_idx && Y) {
            if constexpr (D * Fmt<FMT>::EB == 512 || D * Fmt<FMT>::EB == 256) {
                if (!(g_knn_mode & 0x800))                          // tuning bit 0x800: the one-row-per-wave form
                    hipLaunchKernelGGL((knn_prev_threshold4_kernel<D, FMT>), dim3((unsigned)std::min<int64_t>((N + 15) / 16, 16384)), dim3(256), 0, st, X, Y, xsc, ysc, xnorm, ynh,
                                       prev_idx, N, M, k, thr, slack_rel, row_perm, (prev_dist2 && motion2 && motion_ratio > 0.f) ? prev_dist2 : (const float *)nullptr, motion2, motion_ratio);
                else
                    hipLaunchKernelGGL((knn_prev_threshold_kernel<D, FMT>), dim3((unsigned)std::min<int64_t>((N + 3) / 4, 16384)), dim3(256), 0, st, X, Y, xsc, ysc, xnorm, ynh,
                                       prev_idx, N, M, k, thr, slack_rel, row_perm);
            } else {
                hipLaunchKernelGGL((knn_prev_threshold_kernel<D, FMT>), dim3((unsigned)std::min<int64_t>((N + 3) / 4, 16384)), dim3(256), 0, st, X, Y, xsc, ysc, xnorm, ynh,
                                   prev_idx, N, M, k, thr, slack_rel, row_perm);
            }
        }
        if (prev_dist2 && motion2 && !(prev_idx && Y)) hipLaunchKernelGGL(knn_motion_threshold_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, xnorm, prev_dist2, motion2, N, k, thr, 2.5f * slack_rel);
        if (!(phases & 2)) return w3d_launch_status();
    }
    const int mode = g_knn_mode;
    const float *thr_arg = (prev_idx || warm) ? thr : nullptr;
    // (Round 3 measured a SEEDED cold sweep — a pass over a 1/f row sample of the style set gives every row the score of its j-th best sampled row as a bound, the sweep
    // proper then runs warm, rows whose bound came out too tight go through the repair — and dropped it: 6.9 ms against 4.9 ms unseeded at C2 for f = 4 ... 32
    // (profiles/r03e_prune_diag.log).  On structureless rows any exact streaming selection admits ~k (1 + ln(M / k)) candidates per row; a sample bound only moves them around.)
    auto lds_bytes = [&](int bn, int qcap) {
        return 2 * (size_t)bn * ROWB + 2 * bn * 4 + (FMT == FMT_FP8 ? 2 * bn * 4 : 0) + 16 +
               (size_t)NWAVES * ((qcap + (size_t)((k + 3) & ~3) * ROWS_W) * sizeof(Entry) + 3 * ROWS_W * 4);
    };
    const dim3 block(NWAVES * 64);
    const int64_t nblk = (N + ROWS_B - 1) / ROWS_B;
    int64_t full, rem;
    // a pruned sweep (warm == 3) is ONE launch: its blocks differ in length (each sweeps its own tile list), the hardware's block scheduler balances them better than
    // a split tail round planned for equal blocks (12.85 vs 13.22 ms at C3)
    const bool emit = warm == 4;                                       // pruned sweep, EMIT selection (SEL 1): candidates to the workspace's candidate buffer + knn_select_kernel
    if (emit) warm = 3;                                                 // (everything else about it is the pruned sweep's)
    if (lb && !emit) return W3D_ERR_ARG;                               // the cached lower bounds are recorded by the emit sweep only
    int parts = tail_split(nblk, M, tail_rows_cap, mode | (warm == 3 ? 16 : 0), &full, &rem, thr_arg != nullptr);
    // ... unless the whole sweep is under three rounds of blocks (a rank's share of a sharded scene: 245 blocks at 8 GPUs): one round of UNEQUAL blocks lasts as long
    // as its longest one.  Every block is then cut into `parts` slices of its own tile list (the split-tail path: partial lists + merge), longest
    // first through the persistent kernel's work queues — there is something to balance again.  Tuning bit 0x1000: keep one block per row block.
    // (the list-based sliced sweep parks its partial lists in the split-tail scratch: at most 255 row blocks; the emit sweep's slices share their regions of the candidate buffer)
    if (warm == 3 && nblk < 3 * 256 && (emit || (size_t)(nblk * ROWS_B) <= tail_rows_cap) && ((M + 63) / 64) >= 64 && !(mode & 0x1000)) {
        // >= 2 (row block, slice) items per CU for the work queues to balance, and a slice count that fills whole rounds: 196 row blocks (C2) in 3 slices are 2.3 items per
        // CU — a third of the chip idles in the last round (warm emit sweep 3.57 ms) — in 5 slices 3.83 (2.91 ms; profiles/r04o_c2_slices.log).  A rank's 245 blocks: 2 or 3
        // slices fill their rounds alike (0.96).  Few blocks: as many slices as there are (8).
        {
            double best_u = -1.0; parts = 8;
            for (int p = 2; p <= 8; ++p) {
                const double r = (double)(nblk * p) / 256.0;
                if (r < 2.0) continue;
                const double u = r / std::ceil(r);
                if (u > best_u + 0.02) { best_u = u; parts = p; }
            }
        }
        if ((mode >> 28) & 7) parts = (mode >> 28) & 7;                  // tuning bits 28-30: slices per row block of a short pruned sweep (A/B)
        full = 0; rem = nblk;
    }
    const bool hot = warm == 3;                                         // pruned sweep: the plan's tile lists, counts and block order are in the workspace
    const int32_t *tcnt = warm == 3 ? tord + ((N + ROWS_B - 1) / ROWS_B) * ((M + 63) / 64) : nullptr;     // warm == 3: pruned sweep (w3d_knn_prune_plan filled order + counts)
    const int32_t *bord = (warm == 3 && !(mode & 0x2000)) ? tcnt + (N + ROWS_B - 1) / ROWS_B : nullptr;     // ... and the blocks' launch order (tuning bit 0x2000: natural order)
    // persistent launch of an unsliced pruned sweep: eight work queues over the plan's longest-first block order; tuning bit 0x8000: one workgroup per row block instead
    int32_t *qcursor = warm == 3 ? tord + nblk * ((M + 63) / 64) + 2 * nblk : nullptr;
    auto run = [&](auto kern_warm, auto kern_cold, auto kern_hot, auto kern_persist, auto kern_persist_q, auto kern_repair, auto kern_emit, size_t lds, size_t lds_emit) -> int {
        auto kern = thr_arg ? kern_warm : kern_cold;                   // same signature: a sweep without a bound takes the queue-only instantiation
        // the dynamic-LDS attribute is per (device, kernel): set once per device for this instantiation and size (one static record per `run` instantiation)
        static std::atomic<size_t> lds_set[W3D_MAX_DEVICES];
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= W3D_MAX_DEVICES) return W3D_ERR_LAUNCH;
        if (lds_set[dev].load(std::memory_order_acquire) != lds) {
            W3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern_warm), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            W3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern_cold), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            W3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern_hot), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            W3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern_persist), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            W3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern_persist_q), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            W3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern_repair), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            W3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern_emit), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_emit));
            lds_set[dev].store(lds, std::memory_order_release);
        }
        // round 5: a sweep WITHOUT a bound on the emit kernel (COLD: the bound is tracked in registers as the sweep goes; candidates + knn_select_kernel + the
        // repair for regions that overflowed).  16-bit rows, k <= 20 (the trackers hold 5 scores per lane and row); tuning bit 0x8000000 keeps the list kernel (A/B).
        // The tracker-only first phase: a twelfth of the item's tiles, at most 128 (build macros W3D_COLD_T0_DIV / W3D_COLD_T0_MAX)
        const int64_t ntiles_c = (M + 63) / 64;
        const bool cold_emit = !thr_arg && !hot && !emit && FMT != FMT_FP8 && k <= 20 && cand_buf && cand_cnt && tord && !(mode & 0x8000000) && N < (1ll << 31) - ROWS_B &&
                               M < (1ll << 26);
        // ... and so does a WARM sweep without a plan (warm = 1: the bound of w3d_knn_prepare*, every tile in natural order; 16-bit rows, k <= 24): the emit kernel with the
        // sliced launch of short sweeps, knn_select_kernel, the repair, and the same guarded fallback should a loose bound (a motion bound far from the k-th distance)
        // overflow more regions than the repair takes.  C2 warm: 2.87 -> 2.0 ms; C3 unpruned: 22.6 -> 20.8 ms.  Tuning bit 0x8000000 keeps the list kernel here too.
        const bool warm_emit = thr_arg && warm == 1 && !hot && !emit && FMT != FMT_FP8 && k <= 24 && cand_buf && cand_cnt && tord && !(mode & 0x8000000) && !(mode & 0x1000000) &&
                               N < (1ll << 31) - ROWS_B && M < (1ll << 26) && ntiles_c >= 2;
        int32_t *fl = (thr_arg || cold_emit) ? flags : nullptr;        // only a warm-start bound can be too tight (cold emit: a region can overflow)
        if (!(cold_emit || warm_emit || (hot && emit)))                // (the emit sweeps zero it with their other counters: knn_emit_reset_kernel)
            W3D_HIP(hipMemsetAsync(flags, 0, sizeof(int32_t), st));   // also for cold sweeps: w3d_knn_mfma_repair_count reads it
        if (cold_emit) {
            if constexpr (FMT != FMT_FP8) {
                static std::atomic<int> n_cu4[W3D_MAX_DEVICES];
                int cus = n_cu4[dev].load(std::memory_order_acquire);
                if (cus <= 0) { W3D_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)); n_cu4[dev].store(cus, std::memory_order_release); }
                constexpr size_t lds_lean = 2 * (size_t)64 * ROWB + 2 * 64 * 4 + 16 + (size_t)NWAVES * ((size_t)EMQ_SLOTS * 512 + 3 * ROWS_W * 4);
                auto kc = &knn_emit_kernel<D, FMT, false, 1, true>;
                static std::atomic<int> cold_set[W3D_MAX_DEVICES];
                if (!cold_set[dev].load(std::memory_order_acquire)) {
                    W3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kc), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_lean));
                    cold_set[dev].store(1, std::memory_order_release);
                }
                const int64_t nreg = knn_regions(N);
                int32_t *qc = tord + nblk * ntiles_c + 2 * nblk;
                hipLaunchKernelGGL(knn_emit_reset_kernel, dim3((unsigned)((std::max<int64_t>(nreg, KNN_NQ) + 255) / 256)), dim3(256), 0, st, flags, qc, cand_cnt, nreg);
#ifndef W3D_COLD_T0_DIV
#define W3D_COLD_T0_DIV 12
#endif
#ifndef W3D_COLD_T0_MAX
#define W3D_COLD_T0_MAX 128
#endif
                const int t0_cfg = (int)std::min<int64_t>(std::max<int64_t>(ntiles_c / W3D_COLD_T0_DIV, 1), W3D_COLD_T0_MAX);
                hipLaunchKernelGGL(kc, dim3((unsigned)std::min<int64_t>(nblk, std::max(cus - g_knn_reserve, 8))), block, lds_lean, st, X, Y, xsc, ysc, (const float *)xnorm, (const float *)ynh,
                                   (const float *)nullptr, (int)N, (int)M, row_perm, (const int32_t *)nullptr, (const int32_t *)nullptr, (const int32_t *)nullptr, qc, (int)nblk, 1,
                                   cand_buf, cand_cnt, (uint16_t *)nullptr, (const float *)nullptr, 0.f, g_knn_clk, (k + 3) / 4, t0_cfg);
                hipLaunchKernelGGL(knn_select_kernel, dim3((unsigned)nreg), dim3(64), 0, st, (const Entry *)cand_buf, (const int32_t *)cand_cnt, N, k, (const float *)xnorm, row_perm, idx, d2, fl);
            }
        } else if (warm_emit) {
            if constexpr (FMT != FMT_FP8) {
                static std::atomic<int> n_cu5[W3D_MAX_DEVICES];
                int cus = n_cu5[dev].load(std::memory_order_acquire);
                if (cus <= 0) { W3D_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)); n_cu5[dev].store(cus, std::memory_order_release); }
                constexpr size_t lds_lean = 2 * (size_t)64 * ROWB + 2 * 64 * 4 + 16 + (size_t)NWAVES * ((size_t)EMQ_SLOTS * 512 + 3 * ROWS_W * 4);
                constexpr int ESH = W3D_EMIT_SH;
                auto kw = &knn_emit_kernel<D, FMT, false, ESH>;
                static std::atomic<int> warm_set[W3D_MAX_DEVICES];
                if (!warm_set[dev].load(std::memory_order_acquire)) {
                    W3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kw), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_lean));
                    warm_set[dev].store(1, std::memory_order_release);
                }
                int p = 1;                                               // slices of a short sweep: the rule of the pruned sweeps (whole rounds of the chip filled best)
                if (nblk < 3 * 256 && ntiles_c >= 64 && !(mode & 0x1000)) {
                    double best_u = -1.0; p = 8;
                    for (int q = 2; q <= 8; ++q) {
                        const double r = (double)(nblk * q) / 256.0;
                        if (r < 2.0) continue;
                        const double u = r / std::ceil(r);
                        if (u > best_u + 0.02) { best_u = u; p = q; }
                    }
                }
                const int64_t items = nblk * p, nreg = knn_regions(N);
                int32_t *qc = tord + nblk * ntiles_c + 2 * nblk;
                hipLaunchKernelGGL(knn_emit_reset_kernel, dim3((unsigned)((std::max<int64_t>(nreg, KNN_NQ) + 255) / 256)), dim3(256), 0, st, flags, qc, cand_cnt, nreg);
                hipLaunchKernelGGL(kw, dim3((unsigned)std::min<int64_t>(items, std::max(cus - g_knn_reserve, 8))), block, lds_lean, st, X, Y, xsc, ysc, (const float *)xnorm, (const float *)ynh,
                                   thr_arg, (int)N, (int)M, row_perm, (const int32_t *)nullptr, (const int32_t *)nullptr, (const int32_t *)nullptr, qc, (int)items, p, cand_buf, cand_cnt,
                                   (uint16_t *)nullptr, (const float *)nullptr, 0.f, g_knn_clk, 0, 0);
                hipLaunchKernelGGL(knn_select_kernel, dim3((unsigned)nreg), dim3(64), 0, st, (const Entry *)cand_buf, (const int32_t *)cand_cnt, N, k, (const float *)xnorm, row_perm, idx, d2, fl);
            }
        } else if (hot && emit) {
            // ONE persistent launch for whole row blocks (parts = 1) and for the slices of a short sweep alike: the slices of a row block append to the same regions of the
            // candidate buffer, so there are no partial lists and no merge — knn_select_kernel reads each region once
            static std::atomic<int> n_cu3[W3D_MAX_DEVICES];
            int cus = n_cu3[dev].load(std::memory_order_acquire);
            if (cus <= 0) { W3D_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)); n_cu3[dev].store(cus, std::memory_order_release); }
            const int p = full > 0 ? 1 : parts;
            const int64_t items = nblk * p;
            const int64_t nreg = knn_regions(N);
            hipLaunchKernelGGL(knn_emit_reset_kernel, dim3((unsigned)((std::max<int64_t>(nreg, KNN_NQ) + 255) / 256)), dim3(256), 0, st, flags, qcursor, cand_cnt, nreg);
            bool lean = false;
            {
                // round 5: the emit sweep's own kernel (tuning bit 0x1000000 keeps the SEL = 1 instantiation of the general body: A/B)
                if (!(mode & 0x1000000) && N < (1ll << 31) - ROWS_B && M < (1ll << 26) && thr_arg) {
                    constexpr size_t lds_lean = 2 * (size_t)64 * ROWB + 2 * 64 * 4 + (FMT == FMT_FP8 ? 2 * 64 * 4 : 0) + 16 + (size_t)NWAVES * ((size_t)EMQ_SLOTS * 512 + 3 * ROWS_W * 4);
                    static std::atomic<int> lean_set[W3D_MAX_DEVICES];
                    constexpr int ESH = FMT == FMT_FP8 ? 0 : W3D_EMIT_SH;    // the emit sweep's MFMA shape (1 = 16x16x32, 0 = 32x32x16; fp8: the scaled 32x32x64)
                    auto kl1 = &knn_emit_kernel<D, FMT, true, ESH>; auto kl0 = &knn_emit_kernel<D, FMT, false, ESH>;
                    if (!lean_set[dev].load(std::memory_order_acquire)) {
                        W3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kl1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_lean));
                        W3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kl0), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_lean));
                        lean_set[dev].store(1, std::memory_order_release);
                    }
                    const dim3 grid((unsigned)std::min<int64_t>(items, std::max(cus - g_knn_reserve, 8)));
                    if (g_knn_ev[0]) W3D_HIP(hipEventRecord((hipEvent_t)g_knn_ev[0], st));
                    if (lb) hipLaunchKernelGGL(kl1, grid, block, lds_lean, st, X, Y, xsc, ysc, (const float *)xnorm, (const float *)ynh, thr_arg, (int)N, (int)M, row_perm, (const int32_t *)tord, tcnt, bord,
                                               qcursor, (int)items, p, cand_buf, cand_cnt, lb, lb_cum, lb_margin, g_knn_clk, 0, 0);
                    else hipLaunchKernelGGL(kl0, grid, block, lds_lean, st, X, Y, xsc, ysc, (const float *)xnorm, (const float *)ynh, thr_arg, (int)N, (int)M, row_perm, (const int32_t *)tord, tcnt, bord,
                                            qcursor, (int)items, p, cand_buf, cand_cnt, (uint16_t *)nullptr, (const float *)nullptr, 0.f, g_knn_clk, 0, 0);
                    lean = true;
                    if (g_knn_ev[1]) W3D_HIP(hipEventRecord((hipEvent_t)g_knn_ev[1], st));
                    if (strag_on) {
                        if constexpr (FMT != FMT_FP8) {
                            static_assert(ESH == 1, "knn_straggler_kernel repeats the emit sweep's 16x16x32 MFMA chain: its scores are the sweep's bit for bit only for that shape");
                            if (!lb || !thr_arg) return W3D_ERR_ARG;
                            const int64_t ntl = (M + 63) / 64;
                            hipLaunchKernelGGL((knn_straggler_kernel<D, FMT>), dim3((unsigned)(ntl * KNN_STRAG_WG)), dim3(256), 0, st, X, Y, (const float *)xnorm, (const float *)ynh,
                                               thr_arg, (int)M, row_perm, (const uint32_t *)strag_cnt, (const uint32_t *)strag, strag_cap, (int)ntl, cand_buf, cand_cnt, lb, lb_cum, lb_margin);
                        }
                    }
                }
            }
            if (!lean && strag_on) return W3D_ERR_UNSUPPORTED;          // (the general body has no straggler pass: the plan's lists are incomplete without it)
            if (!lean)
            hipLaunchKernelGGL(kern_emit, dim3((unsigned)std::min<int64_t>(items, std::max(cus - g_knn_reserve, 8))), block, lds_emit, st, X, Y, xsc, ysc, xnorm, ynh, thr_arg, N, M, k, idx, d2,
                               mode, (int64_t)0, p, fl, row_perm, (const int32_t *)tord, tcnt, bord, qcursor, (int)items, g_knn_clk, cand_buf, cand_cnt, lb, lb_cum, lb_margin);
            hipLaunchKernelGGL(knn_select_kernel, dim3((unsigned)nreg), dim3(64), 0, st, (const Entry *)cand_buf, (const int32_t *)cand_cnt, N, k, (const float *)xnorm, row_perm, idx, d2, fl);
        } else if (full > 0 && hot && rem == 0 && !(mode & 0x8000)) {
            static std::atomic<int> n_cu[W3D_MAX_DEVICES];
            int cus = n_cu[dev].load(std::memory_order_acquire);
            if (cus <= 0) { W3D_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)); n_cu[dev].store(cus, std::memory_order_release); }
            W3D_HIP(hipMemsetAsync(qcursor, 0, KNN_NQ * sizeof(int32_t), st));
            if (mode & 0x2000000)                                        // tuning bit: queue-only selection (no direct list fill) in the pruned sweep — A/B
                hipLaunchKernelGGL(kern_persist_q, dim3((unsigned)std::min<int64_t>(full, std::max(cus - g_knn_reserve, 8))), block, lds, st, X, Y, xsc, ysc, xnorm, ynh, thr_arg, N, M, k, idx, d2, mode, (int64_t)0, 1, fl,
                                   row_perm, (const int32_t *)tord, tcnt, bord, qcursor, (int)nblk, g_knn_clk, (Entry *)nullptr, (int32_t *)nullptr, (uint16_t *)nullptr, (const float *)nullptr, 0.f);
            else
                hipLaunchKernelGGL(kern_persist, dim3((unsigned)std::min<int64_t>(full, std::max(cus - g_knn_reserve, 8))), block, lds, st, X, Y, xsc, ysc, xnorm, ynh, thr_arg, N, M, k, idx, d2, mode, (int64_t)0, 1, fl,
                                   row_perm, (const int32_t *)tord, tcnt, bord, qcursor, (int)nblk, g_knn_clk, (Entry *)nullptr, (int32_t *)nullptr, (uint16_t *)nullptr, (const float *)nullptr, 0.f);
        } else if (full > 0 && hot) hipLaunchKernelGGL(kern_hot, dim3((unsigned)full), block, lds, st, X, Y, xsc, ysc, xnorm, ynh, thr_arg, N, M, k, idx, d2, mode, (int64_t)0, 1, fl, row_perm, (const int32_t *)tord, tcnt, bord, g_knn_clk);
        else if (full > 0) hipLaunchKernelGGL(kern, dim3((unsigned)full), block, lds, st, X, Y, xsc, ysc, xnorm, ynh, thr_arg, N, M, k, idx, d2, mode, (int64_t)0, 1, fl, row_perm, (const int32_t *)nullptr);
        if ((hot && emit) || cold_emit || warm_emit) {
            // (launched above)
        } else if (rem > 0 && hot && full == 0 && !(mode & 0x8000)) {
            // a short pruned sweep cut in slices (see above): the (row block, slice) items go through the persistent kernel's work queues as whole blocks do —
            // the hardware deals workgroups to the XCDs round-robin whatever their length (per-CU busy time 1.20 ... 1.69 ms of a 1.76-ms launch at 125 k rows)
            static std::atomic<int> n_cu2[W3D_MAX_DEVICES];
            int cus = n_cu2[dev].load(std::memory_order_acquire);
            if (cus <= 0) { W3D_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)); n_cu2[dev].store(cus, std::memory_order_release); }
            W3D_HIP(hipMemsetAsync(qcursor, 0, KNN_NQ * sizeof(int32_t), st));
            hipLaunchKernelGGL(kern_persist, dim3((unsigned)std::min<int64_t>(rem * parts, std::max(cus - g_knn_reserve, 8))), block, lds, st, X, Y, xsc, ysc, xnorm, ynh, thr_arg, N, M, k, tail_idx, tail_d2, mode, (int64_t)0, parts, fl,
                               row_perm, (const int32_t *)tord, tcnt, bord, qcursor, (int)(rem * parts), g_knn_clk, (Entry *)nullptr, (int32_t *)nullptr, (uint16_t *)nullptr, (const float *)nullptr, 0.f);
            launch_merge_parts(st, tail_idx, tail_d2, rem * ROWS_B, (int64_t)0, N, parts, k, (const float *)xnorm, idx, d2, fl, row_perm);
        } else if (rem > 0 && hot) {
            hipLaunchKernelGGL(kern_hot, dim3((unsigned)(rem * parts)), block, lds, st, X, Y, xsc, ysc, xnorm, ynh, thr_arg, N, M, k, tail_idx, tail_d2, mode, full, parts, fl, row_perm, (const int32_t *)tord, tcnt, full == 0 ? bord : (const int32_t *)nullptr, g_knn_clk);
            const int64_t rows = rem * ROWS_B;
            launch_merge_parts(st, tail_idx, tail_d2, rows, full * ROWS_B, N, parts, k, (const float *)xnorm, idx, d2, fl, row_perm);
        } else if (rem > 0) {
            hipLaunchKernelGGL(kern, dim3((unsigned)(rem * parts)), block, lds, st, X, Y, xsc, ysc, xnorm, ynh, thr_arg, N, M, k, tail_idx, tail_d2, mode, full, parts, fl, row_perm, (const int32_t *)nullptr);
            const int64_t rows = rem * ROWS_B;
            launch_merge_parts(st, tail_idx, tail_d2, rows, full * ROWS_B, N, parts, k, (const float *)xnorm, idx, d2, fl, row_perm);
        }
        if (fl) {   // repair: flagged rows (normally none) once more with a cold bound; the row count stays on the device
            hipLaunchKernelGGL((knn_gather_flagged_kernel<ROWB>), dim3(64), dim3(256), 0, st, X, FMT == FMT_FP8 ? xsc : (const int32_t *)nullptr, xnorm, fl, Xc, xscc, xnormc);
            // A handful of flagged rows is ONE row block: alone it would sweep the whole style set on one CU (4.5 ms at C2, 3 ms at C3 — longer than the sweep it
            // repairs).  The style range is cut in 8 slices per row block (partial lists in the split-tail scratch, merged as a split sweep's are): 1/8 of that.
            const int rparts = ((M + 63) / 64 >= 64 && (size_t)KNN_FLAG_CAP <= tail_rows_cap) ? 8 : 1;
            if (rparts > 1) {
                hipLaunchKernelGGL(kern_repair, dim3((unsigned)(KNN_FLAG_CAP / ROWS_B * rparts)), block, lds, st, (const uint8_t *)Xc, Y, (const int32_t *)xscc, ysc, (const float *)xnormc, ynh,
                                   (const float *)nullptr, (int64_t)KNN_FLAG_CAP, M, k, tail_idx, tail_d2, mode, (int64_t)0, rparts, (int32_t *)nullptr, (const int32_t *)fl);
                launch_merge_parts(st, tail_idx, tail_d2, (int64_t)KNN_FLAG_CAP, (int64_t)0, (int64_t)KNN_FLAG_CAP, rparts, k, (const float *)xnormc, idxc, d2c, (int32_t *)nullptr,
                                   (const int32_t *)nullptr, (const int32_t *)fl);
            } else {
                hipLaunchKernelGGL(kern_repair, dim3((unsigned)(KNN_FLAG_CAP / ROWS_B)), block, lds, st, (const uint8_t *)Xc, Y, (const int32_t *)xscc, ysc, (const float *)xnormc, ynh,
                                   (const float *)nullptr, (int64_t)KNN_FLAG_CAP, M, k, idxc, d2c, mode, (int64_t)0, 1, (int32_t *)nullptr, (const int32_t *)fl);
            }
            hipLaunchKernelGGL(knn_scatter_flagged_kernel, dim3(64), dim3(256), 0, st, fl, idxc, d2c, k, idx, d2, (cold_emit || warm_emit) ? 1 : 0);
            // cold emit sweep: should its regions have overflowed for more rows than the repair takes (storms of candidates the compaction could not keep up with), the
            // whole sweep is redone by the list kernel — a launch that returns at once otherwise
            if (cold_emit || warm_emit) hipLaunchKernelGGL(kern_cold, dim3((unsigned)nblk), block, lds, st, X, Y, xsc, ysc, xnorm, ynh, (const float *)nullptr, N, M, k, idx, d2, mode, (int64_t)0, 1,
                                              (int32_t *)nullptr, row_perm, (const int32_t *)fl);
        }
        return W3D_OK;
    };
    // 64-row tiles (one barrier per 64 MFMAs per wave) when the ring + lists fit the 160 KB LDS, else 32-row tiles.
    // MFMA shape of the 16-bit formats: 16x16x32 by default since round 2 (on the centred feature rows it is 4.5 % faster than 32x32x16: 24.0 vs 25.2 ms per
    // sweep at C3, same device — the chip holds a higher clock on this shape, and the centred rows toggle like random data); mode bit 32 selects 32x32x16
    constexpr int SHALT = FMT == FMT_FP8 ? 0 : 1;
    auto lds_emit_bytes = [&](int bn) { return 2 * (size_t)bn * ROWB + 2 * bn * 4 + (FMT == FMT_FP8 ? 2 * bn * 4 : 0) + 16 + (size_t)NWAVES * ((size_t)QCAP_EMIT * sizeof(Entry) + ROWS_W * 4); };
    if (emit && (!cand_buf || !cand_cnt)) return W3D_ERR_WORKSPACE;
    int rc;
    if (lds_bytes(64, QCAP64) <= 160 * 1024 && !(mode & 8)) {
        rc = !(mode & 32) ? run(&knn_mfma_kernel<D, 64, QCAP64, SHALT, FMT>, &knn_mfma_cold_kernel<D, 64, QCAP64, SHALT, FMT>, &knn_mfma_hot_kernel<D, 64, QCAP64, SHALT, FMT>, &knn_mfma_persistent_kernel<D, 64, QCAP64, SHALT, FMT>, &knn_mfma_persistent_kernel<D, 64, QCAP64, SHALT, FMT, false>, &knn_mfma_repair_kernel<D, 64, QCAP64, SHALT, FMT>, &knn_mfma_persistent_kernel<D, 64, QCAP64, SHALT, FMT, true, 1>, lds_bytes(64, QCAP64), lds_emit_bytes(64))
                         : run(&knn_mfma_kernel<D, 64, QCAP64, 0, FMT>, &knn_mfma_cold_kernel<D, 64, QCAP64, 0, FMT>, &knn_mfma_hot_kernel<D, 64, QCAP64, 0, FMT>, &knn_mfma_persistent_kernel<D, 64, QCAP64, 0, FMT>, &knn_mfma_persistent_kernel<D, 64, QCAP64, 0, FMT, false>, &knn_mfma_repair_kernel<D, 64, QCAP64, 0, FMT>, &knn_mfma_persistent_kernel<D, 64, QCAP64, 0, FMT, true, 1>, lds_bytes(64, QCAP64), lds_emit_bytes(64));
    } else {
        if (hot) return W3D_ERR_UNSUPPORTED;                        // the tile order is built for 64-row tiles
        if (lds_bytes(32, QCAP32) > 160 * 1024) return W3D_ERR_UNSUPPORTED;
        rc = !(mode & 32) ? run(&knn_mfma_kernel<D, 32, QCAP32, SHALT, FMT>, &knn_mfma_cold_kernel<D, 32, QCAP32, SHALT, FMT>, &knn_mfma_hot_kernel<D, 32, QCAP32, SHALT, FMT>, &knn_mfma_persistent_kernel<D, 32, QCAP32, SHALT, FMT>, &knn_mfma_persistent_kernel<D, 32, QCAP32, SHALT, FMT, false>, &knn_mfma_repair_kernel<D, 32, QCAP32, SHALT, FMT>, &knn_mfma_persistent_kernel<D, 32, QCAP32, SHALT, FMT, true, 1>, lds_bytes(32, QCAP32), lds_emit_bytes(32))
                         : run(&knn_mfma_kernel<D, 32, QCAP32, 0, FMT>, &knn_mfma_cold_kernel<D, 32, QCAP32, 0, FMT>, &knn_mfma_hot_kernel<D, 32, QCAP32, 0, FMT>, &knn_mfma_persistent_kernel<D, 32, QCAP32, 0, FMT>, &knn_mfma_persistent_kernel<D, 32, QCAP32, 0, FMT, false>, &knn_mfma_repair_kernel<D, 32, QCAP32, 0, FMT>, &knn_mfma_persistent_kernel<D, 32, QCAP32, 0, FMT, true, 1>, lds_bytes(32, QCAP32), lds_emit_bytes(32));
    }
    if (rc != W3D_OK) return rc;
    return w3d_launch_status();
}

// repair scratch: flags [1 + CAP] | feature rows [CAP x row bytes (<= 2 d)] | scales [CAP] | norms [CAP] | idx [CAP x k] | dist2 [CAP x k], each 256-B aligned
size_t tile_order_bytes(int64_t N, int64_t M) {   // [row blocks][64-row tiles] positions -> tiles, then [row blocks] tiles to sweep (pruned sweeps)
    // + [row blocks] counts, + [row blocks] launch order (longest first), + the queue cursors of the persistent sweep
    return ((size_t)((N + ROWS_B - 1) / ROWS_B) * ((size_t)((M + 63) / 64) + 2) + KNN_Q_INTS) * sizeof(int32_t);
}

// candidate buffer of the emit sweeps: [regions of 64 rows][KNN_REGION_CAP] entries, then the regions' counts
size_t cand_bytes(int64_t N) {
    const size_t nreg = (size_t)knn_regions(N);
    return 256 + nreg * KNN_REGION_CAP * sizeof(Entry) + ((nreg * sizeof(int32_t) + 255) & ~(size_t)255);
}

size_t repair_bytes(int d, int k) {
    auto up256 = [](size_t b) { return (b + 255) & ~(size_t)255; };
    return up256((size_t)(KNN_FLAG_ROW0 + KNN_FLAG_CAP) * 4) + up256((size_t)KNN_FLAG_CAP * d * 2) + 2 * up256((size_t)KNN_FLAG_CAP * 4) + 2 * up256((size_t)KNN_FLAG_CAP * k * 4) + 256;
}

}  // namespace

#if !W3D_KNN_LEAF
extern "C" size_t w3d_knn_mfma_workspace_bytes(int64_t N, int64_t M, int d, int k) {
    const size_t tail = (size_t)255 * ROWS_B * 8 * (size_t)k * 8;      // partial lists of the split tail round (<= 255 row blocks x 8 parts)
    const size_t strag = (size_t)((M + 63) / 64) * ((size_t)knn_strag_cap(N, M) + 1) * sizeof(uint32_t) + 768;      // straggler buckets (+ their counts, + alignment)
    return (size_t)(2 * (((N + 63) / 64) * 64) + (M + YPAD - 1) / YPAD * YPAD) * sizeof(float) + tail + repair_bytes(d, k) + tile_order_bytes(N, M) + cand_bytes(N) + strag + 1024 + 256;
}

extern "C" int w3d_knn_mfma_launches(int64_t N, int64_t M) {
    int64_t full, rem;
    const int mode = g_knn_mode;
    const int parts = tail_split((N + ROWS_B - 1) / ROWS_B, M, (size_t)255 * ROWS_B, mode, &full, &rem);
    (void)parts;
    return (full > 0 ? 1 : 0) + (rem > 0 ? 1 : 0);
}

namespace {
// address of the repair flags inside a workspace laid out by w3d_knn_mfma_fwd_fmt
const int32_t *ws_flags(const void *ws, int64_t N, int64_t M, int k) {
    const float *xnorm = reinterpret_cast<const float *>(ws);
    const float *ynh = xnorm + 2 * (((N + 63) / 64) * 64);
    uintptr_t tp = (reinterpret_cast<uintptr_t>(ynh + (M + YPAD - 1) / YPAD * YPAD) + 255) & ~(uintptr_t)255;
    tp = (tp + (size_t)255 * ROWS_B * 8 * (size_t)k * 8 + 255) & ~(uintptr_t)255;
    return reinterpret_cast<const int32_t *>(tp);
}
}  // namespace

// Number of rows the last warm-started call on this workspace had to redo with a cold bound (normally 0; more than KNN_FLAG_CAP would mean rows
// left unrepaired — the caller must treat that as an error).  Blocks on the stream.
extern "C" int w3d_knn_mfma_repair_count(const void *ws, int64_t N, int64_t M, int k, int32_t *count_host, int32_t *cap_host, w3d_stream_t stream) {
    hipStream_t st = w3d_s(stream);
    if (!ws || !count_host) return W3D_ERR_ARG;
    W3D_HIP(hipMemcpyAsync(count_host, ws_flags(ws, N, M, k), sizeof(int32_t), hipMemcpyDeviceToHost, st));
    if (hipStreamSynchronize(st) != hipSuccess) return W3D_ERR_LAUNCH;
    if (cap_host) *cap_host = KNN_FLAG_CAP;
    return W3D_OK;
}

// Non-blocking form for loops: enqueues a copy of {rows flagged by the last call, largest such count of any call since the workspace was zeroed}
// into host_pair[2] (pinned host memory, or the copy is not asynchronous) and returns; the caller reads it after a later event / synchronisation.
// host_pair[1] > *cap_host means some sweep left rows with fewer than k neighbours (sentinel indices 0x3ffffff in their lists).
extern "C" int w3d_knn_mfma_repair_poll(const void *ws, int64_t N, int64_t M, int k, int32_t *host_pair, int32_t *cap_host, w3d_stream_t stream) {
    if (!ws || !host_pair) return W3D_ERR_ARG;
    W3D_HIP(hipMemcpyAsync(host_pair, ws_flags(ws, N, M, k), 2 * sizeof(int32_t), hipMemcpyDeviceToHost, w3d_s(stream)));
    if (cap_host) *cap_host = KNN_FLAG_CAP;
    return W3D_OK;
}

// Diagnostic knobs of the fused kNN (process-global; tests and A/B scripts only): mode = the W3D_KNN_MODE bit set (16 = no split tail round,
// 32 = 32x32x16 MFMA shape instead of 16x16x32, 8 = 32-row tiles; the in-kernel bits need a -DW3D_KNN_DEBUG build), slack_rel <= 0 = the built-in bound slack.
// Diagnostics (process-global; bench.py): two HIP events (hipEvent_t, created by the caller with timing enabled) the following pruned emit sweeps record on their stream
// right before and right behind the emit kernel — the kernel's own duration inside the launch group (reset, emit, stragglers, select).  NULL, NULL: off.
extern "C" int w3d_knn_mfma_debug_events(void *before, void *after) {
    if ((before == nullptr) != (after == nullptr)) return W3D_ERR_ARG;
    g_knn_ev[0] = before; g_knn_ev[1] = after;
    return W3D_OK;
}
extern "C" int w3d_knn_mfma_debug_clocks(int64_t *clk, int64_t *plan_clk) {
    g_knn_clk = clk;
    g_knn_plan_clk = plan_clk;
    return W3D_OK;
}

// CUs a persistent pruned sweep leaves FREE (process-wide; one process per GPU): its workgroups hold every VGPR and all LDS of their CU for the whole launch, so
// a kernel of another stream runs only on CUs the sweep did not take.  A rank of a sharded scene renders its image band next to its sweep: the band's ~250 tile
// workgroups are latency-bound and fit on 32 CUs, the sweep gives up 1/8 of its rate for them.  0 (default): the sweep takes every CU.
extern "C" int w3d_knn_mfma_set_cu_reserve(int cus) {
    if (cus < 0 || cus > 128) return W3D_ERR_ARG;
    g_knn_reserve = cus;
    return W3D_OK;
}

extern "C" int w3d_knn_mfma_set_tuning(int mode, float slack_rel) {
    g_knn_mode = mode;
    g_knn_slack = slack_rel > 0.f ? slack_rel : 0.f;
    return W3D_OK;
}
#endif   // !W3D_KNN_LEAF

struct WsLayout { float *xnorm, *thr, *ynh; int32_t *tail_idx; float *tail_d2; int32_t *flags; uint8_t *Xc; int32_t *xscc; float *xnormc; int32_t *idxc; float *d2c; int32_t *tord;
                  Entry *cand; int32_t *cand_cnt; uint32_t *strag_cnt; uint32_t *strag; int strag_cap; size_t used; };
namespace {

WsLayout carve(void *ws, int64_t N, int64_t M, int d, int k) {
    WsLayout w;
    w.xnorm = reinterpret_cast<float *>(ws);
    w.thr = w.xnorm + ((N + 63) / 64) * 64;
    w.ynh = w.thr + ((N + 63) / 64) * 64;
    auto up256p = [](char *q) { return reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(q) + 255) & ~(uintptr_t)255); };
    char *tp = up256p(reinterpret_cast<char *>(w.ynh + (M + YPAD - 1) / YPAD * YPAD));
    const size_t tail_elems = (size_t)255 * ROWS_B * 8 * (size_t)k;
    w.tail_idx = reinterpret_cast<int32_t *>(tp);
    w.tail_d2 = reinterpret_cast<float *>(tp + tail_elems * 4);
    char *rp = up256p(tp + tail_elems * 8);
    w.flags = reinterpret_cast<int32_t *>(rp); rp = up256p(rp + (size_t)(KNN_FLAG_ROW0 + KNN_FLAG_CAP) * 4);
    w.Xc = reinterpret_cast<uint8_t *>(rp); rp = up256p(rp + (size_t)KNN_FLAG_CAP * d * 2);
    w.xscc = reinterpret_cast<int32_t *>(rp); rp = up256p(rp + (size_t)KNN_FLAG_CAP * 4);
    w.xnormc = reinterpret_cast<float *>(rp); rp = up256p(rp + (size_t)KNN_FLAG_CAP * 4);
    w.idxc = reinterpret_cast<int32_t *>(rp); rp = up256p(rp + (size_t)KNN_FLAG_CAP * k * 4);
    w.d2c = reinterpret_cast<float *>(rp); rp = up256p(rp + (size_t)KNN_FLAG_CAP * k * 4);
    w.tord = reinterpret_cast<int32_t *>(rp); rp = up256p(rp + tile_order_bytes(N, M));
    w.cand = reinterpret_cast<Entry *>(rp); rp = up256p(rp + (size_t)knn_regions(N) * KNN_REGION_CAP * sizeof(Entry));
    w.cand_cnt = reinterpret_cast<int32_t *>(rp); rp = up256p(rp + (size_t)knn_regions(N) * sizeof(int32_t));
    w.strag_cap = knn_strag_cap(N, M);
    w.strag_cnt = reinterpret_cast<uint32_t *>(rp); rp = up256p(rp + (size_t)((M + 63) / 64) * sizeof(uint32_t));
    w.strag = reinterpret_cast<uint32_t *>(rp); rp += (size_t)((M + 63) / 64) * (size_t)w.strag_cap * sizeof(uint32_t);
    w.used = (size_t)(rp - (char *)ws);
    return w;
}

// Chunked sweep (multi-GPU: the style set arrives shard by shard while the distance-GEMM already runs on the shards that are there).
//   op 0: -0.5|y|^2 of style rows [m0, m_valid) into the workspace (the rows must have arrived); rows [m_valid, m1) are padding of the shard layout
//         (and, for the last chunk, of the array): -inf, never selected
//   op 1: all row blocks sweep style tiles [m0/64, ceil(m1/64)) with the prepared bound; partial lists (raw scores, score order) go to slot `part` of
//         part_idx / part_val [N, parts, k]
//   op 2: merge of the `parts` partial lists into idx / dist2 (+ the cold repair of rows a warm bound left short, over the WHOLE style set)
// X norms and the bound come from w3d_knn_prepare* called with Y = NULL beforehand.
template <int D, int FMT>
int launch_part(int op, const uint8_t *X, const uint8_t *Y, const int32_t *xsc, const int32_t *ysc, int64_t N, int64_t M, int k, int warm, int64_t m0, int64_t m1, int part,
                int parts, int32_t *part_idx, float *part_val, int32_t *idx, float *d2, const WsLayout &w, hipStream_t st, int64_t m_valid, const int32_t *row_perm) {
    constexpr int ROWB = D * Fmt<FMT>::EB;
    constexpr int LPR = ROWB / 16;
    const int mode = g_knn_mode;
    const size_t lds = 2 * (size_t)64 * ROWB + 2 * 64 * 4 + (FMT == FMT_FP8 ? 2 * 64 * 4 : 0) + 16 + (size_t)NWAVES * ((QCAP64 + (size_t)((k + 3) & ~3) * ROWS_W) * sizeof(Entry) + 3 * ROWS_W * 4);
    if (lds > 160 * 1024) return W3D_ERR_UNSUPPORTED;                   // chunked sweeps use the 64-row-tile kernel only (k <= 20 at d = 256)
    if (op == 0) {
        const int64_t Mp = (M + YPAD - 1) / YPAD * YPAD;
        const int64_t e = m1 >= M ? Mp : m1;                             // the last chunk also writes the -inf padding of the array
        if (m_valid < m0 || m_valid > m1 || m_valid > M) return W3D_ERR_ARG;
        hipLaunchKernelGGL((row_sqnorm_kernel<D, FMT>), dim3((unsigned)(((e - m0) * LPR + 255) / 256)), dim3(256), 0, st, Y + m0 * ROWB, ysc ? ysc + m0 : nullptr,
                           m_valid - m0, e - m0, -0.5f, w.ynh + m0);
        return w3d_launch_status();
    }
    constexpr int SHP = FMT == FMT_FP8 ? 0 : 1;                         // the same MFMA shape as the whole-set sweep (results must agree bit for bit)
    auto kern = &knn_mfma_part_kernel<D, 64, QCAP64, SHP, FMT>;
    auto kern_repair = &knn_mfma_repair_kernel<D, 64, QCAP64, SHP, FMT>;
    static std::atomic<size_t> lds_set[W3D_MAX_DEVICES];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= W3D_MAX_DEVICES) return W3D_ERR_LAUNCH;
    if (lds_set[dev].load(std::memory_order_acquire) != lds) {
        W3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        W3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern_repair), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        lds_set[dev].store(lds, std::memory_order_release);
    }
    const float *thr_arg = warm ? w.thr : nullptr;
    const int64_t nblk = (N + ROWS_B - 1) / ROWS_B;
    if (op == 1) {
        const int64_t t0 = m0 / 64, t1 = (m1 + 63) / 64;
        if (m0 % 64 || t0 >= t1 || t1 > (M + 63) / 64) return W3D_ERR_ARG;
        hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(NWAVES * 64), lds, st, X, Y, xsc, ysc, (const float *)w.xnorm, (const float *)w.ynh, thr_arg, N, M, k, part_idx, part_val,
                           mode, (int64_t)0, parts, (int32_t *)nullptr, t0, t1, part, row_perm);
        return w3d_launch_status();
    }
    int32_t *fl = thr_arg ? w.flags : nullptr;
    W3D_HIP(hipMemsetAsync(w.flags, 0, sizeof(int32_t), st));
    launch_merge_parts(st, (const int32_t *)part_idx, (const float *)part_val, N, (int64_t)0, N, parts, k, (const float *)w.xnorm, idx, d2, fl, row_perm);
    if (fl) {
        hipLaunchKernelGGL((knn_gather_flagged_kernel<ROWB>), dim3(64), dim3(256), 0, st, X, FMT == FMT_FP8 ? xsc : (const int32_t *)nullptr, (const float *)w.xnorm, (const int32_t *)fl, w.Xc, w.xscc, w.xnormc);
        hipLaunchKernelGGL(kern_repair, dim3((unsigned)(KNN_FLAG_CAP / ROWS_B)), dim3(NWAVES * 64), lds, st, (const uint8_t *)w.Xc, Y, (const int32_t *)w.xscc, ysc, (const float *)w.xnormc, (const float *)w.ynh,
                           (const float *)nullptr, (int64_t)KNN_FLAG_CAP, M, k, w.idxc, w.d2c, mode, (int64_t)0, 1, (int32_t *)nullptr, (const int32_t *)fl);
        hipLaunchKernelGGL(knn_scatter_flagged_kernel, dim3(64), dim3(256), 0, st, fl, (const int32_t *)w.idxc, (const float *)w.d2c, k, idx, d2);
    }
    return w3d_launch_status();
}
}  // namespace

// ---- main <-> leaf interface (plain structs of arguments; the leaf of (fmt, d) exports w3d_knn_leaf_{fwd,part,prune}_<fmt>_<d>)
struct W3dKnnFwdArgs { const uint8_t *X, *Y; const int32_t *xscale, *yscale; int64_t N, M; int k; int32_t *idx; float *dist2; const int32_t *prev_idx; int phases, warm; hipStream_t st;
                       const float *prev_dist2, *motion2; size_t tail_rows_cap; const float *xnorm_in; const int32_t *row_perm; uint16_t *lb; const float *lb_cum; float lb_margin; float motion_ratio; };
struct W3dKnnPartArgs { int op; const uint8_t *X, *Y; const int32_t *xscale, *yscale; int64_t N, M; int k, warm; int64_t m0, m1; int part, parts; int32_t *part_idx; float *part_val;
                        int32_t *idx; float *dist2; hipStream_t st; int64_t m_valid; const int32_t *row_perm; };
struct W3dKnnPruneArgs { const uint8_t *X; const int32_t *row_order; const uint8_t *C; const float *cnorm, *cradius; int64_t N, ntiles, nblk; int32_t *tcnt; float cnorm_max;
                         const uint16_t *lb; const float *lb_cum; float *summ; float *dref; size_t lds; hipStream_t st; float mrg; };
#define W3D_KNN_CAT3(a, b, c) a##b##_##c
#define W3D_KNN_LEAF_NAME(stem, f, d) W3D_KNN_CAT3(stem, f, d)
#if W3D_KNN_LEAF
extern "C" int W3D_KNN_LEAF_NAME(w3d_knn_leaf_fwd_, W3D_KNN_FMT, W3D_KNN_D)(const W3dKnnFwdArgs *a, const WsLayout *w) {
    return launch<W3D_KNN_D, W3D_KNN_FMT>(a->X, a->Y, a->xscale, a->yscale, a->N, a->M, a->k, a->idx, a->dist2, w->xnorm, w->ynh, a->prev_idx, w->thr, a->phases, a->warm, a->st, a->prev_dist2,
                                          a->motion2, w->tail_idx, w->tail_d2, a->tail_rows_cap, a->xnorm_in, w->flags, w->Xc, w->xscc, w->xnormc, w->idxc, w->d2c, a->row_perm, w->tord,
                                          w->cand, w->cand_cnt, a->lb, a->lb_cum, a->lb_margin, a->motion_ratio, w->strag_cnt, w->strag, w->strag_cap);
}
extern "C" int W3D_KNN_LEAF_NAME(w3d_knn_leaf_part_, W3D_KNN_FMT, W3D_KNN_D)(const W3dKnnPartArgs *a, const WsLayout *w) {
    return launch_part<W3D_KNN_D, W3D_KNN_FMT>(a->op, a->X, a->Y, a->xscale, a->yscale, a->N, a->M, a->k, a->warm, a->m0, a->m1, a->part, a->parts, a->part_idx, a->part_val, a->idx, a->dist2,
                                               *w, a->st, a->m_valid, a->row_perm);
}
#if W3D_KNN_FMT != 2
extern "C" int W3D_KNN_LEAF_NAME(w3d_knn_leaf_prune_, W3D_KNN_FMT, W3D_KNN_D)(const W3dKnnPruneArgs *a, const WsLayout *w) {
    if (a->summ) {                                                      // the plan right after a reset of the cached bounds: ball test + block summaries of the ball bounds
        auto kern = &knn_tile_prune_kernel<W3D_KNN_D, W3D_KNN_FMT, true>;
        W3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)a->lds));
        hipLaunchKernelGGL(kern, dim3((unsigned)a->nblk), dim3(512), a->lds, a->st, a->X, a->row_order, (const float *)w->xnorm, (const float *)w->thr, a->C, a->cnorm, a->cradius, a->N,
                           a->ntiles, w->tord, a->tcnt, a->cnorm_max, g_knn_plan_clk, a->lb, a->lb_cum, a->summ, a->dref, a->mrg);
        return w3d_launch_status();
    }
    auto kern = &knn_tile_prune_kernel<W3D_KNN_D, W3D_KNN_FMT, false>;
    W3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)a->lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)a->nblk), dim3(512), a->lds, a->st, a->X, a->row_order, (const float *)w->xnorm, (const float *)w->thr, a->C, a->cnorm, a->cradius, a->N, a->ntiles,
                       w->tord, a->tcnt, a->cnorm_max, g_knn_plan_clk, a->lb, a->lb_cum, (float *)nullptr, (float *)nullptr, a->mrg);
    return w3d_launch_status();
}
#endif
#else   // ---------------------------------------------------------------- main object
#define W3D_KNN_DECL_LEAF(f, d) extern "C" int W3D_KNN_LEAF_NAME(w3d_knn_leaf_fwd_, f, d)(const W3dKnnFwdArgs *, const WsLayout *); \
                                extern "C" int W3D_KNN_LEAF_NAME(w3d_knn_leaf_part_, f, d)(const W3dKnnPartArgs *, const WsLayout *);
W3D_KNN_DECL_LEAF(0, 64) W3D_KNN_DECL_LEAF(0, 128) W3D_KNN_DECL_LEAF(0, 256) W3D_KNN_DECL_LEAF(1, 64) W3D_KNN_DECL_LEAF(1, 128) W3D_KNN_DECL_LEAF(1, 256)
W3D_KNN_DECL_LEAF(2, 128) W3D_KNN_DECL_LEAF(2, 256)
#undef W3D_KNN_DECL_LEAF
#define W3D_KNN_DECL_PRUNE(f, d) extern "C" int W3D_KNN_LEAF_NAME(w3d_knn_leaf_prune_, f, d)(const W3dKnnPruneArgs *, const WsLayout *);
W3D_KNN_DECL_PRUNE(0, 64) W3D_KNN_DECL_PRUNE(0, 128) W3D_KNN_DECL_PRUNE(0, 256) W3D_KNN_DECL_PRUNE(1, 64) W3D_KNN_DECL_PRUNE(1, 128) W3D_KNN_DECL_PRUNE(1, 256)
#undef W3D_KNN_DECL_PRUNE
#endif


#if !W3D_KNN_LEAF
#define W3D_KNN_DISPATCH(stem, fmt, d, args, w)                                                                                              \
    do {                                                                                                                                     \
        if ((fmt) == FMT_BF16) { switch (d) { case 64: return W3D_KNN_LEAF_NAME(stem, 0, 64)(args, w); case 128: return W3D_KNN_LEAF_NAME(stem, 0, 128)(args, w);  \
                                              case 256: return W3D_KNN_LEAF_NAME(stem, 0, 256)(args, w); default: return W3D_ERR_UNSUPPORTED; } }                     \
        if ((fmt) == FMT_F16) { switch (d) { case 64: return W3D_KNN_LEAF_NAME(stem, 1, 64)(args, w); case 128: return W3D_KNN_LEAF_NAME(stem, 1, 128)(args, w);   \
                                             case 256: return W3D_KNN_LEAF_NAME(stem, 1, 256)(args, w); default: return W3D_ERR_UNSUPPORTED; } }                      \
    } while (0)

extern "C" int w3d_knn_mfma_part_fmt(int op, const void *X, const void *Y, const int32_t *xscale, const int32_t *yscale, int64_t N, int64_t M, int d, int k, int fmt, int warm,
                                     int64_t m0, int64_t m1, int64_t m_valid, int part, int parts, int32_t *part_idx, float *part_val, int32_t *idx, float *dist2, void *ws,
                                     size_t ws_bytes, hipStream_t st, const int32_t *row_perm) {
    if (!ws || ws_bytes < w3d_knn_mfma_workspace_bytes(N, M, d, k)) return W3D_ERR_WORKSPACE;
    if (M >= (1ll << 26) || k > 64 || parts < 1 || parts > 8 || part < 0 || part >= parts || m0 < 0 || m1 <= m0) return W3D_ERR_ARG;
    if (((uintptr_t)X | (uintptr_t)Y) & 15) return W3D_ERR_ARG;
    if (op != 0 && (!part_idx || !part_val)) return W3D_ERR_ARG;
    if (op == 2 && (!idx || !dist2)) return W3D_ERR_ARG;
    const WsLayout w = carve(ws, N, M, d, k);
    if (w.used > ws_bytes) return W3D_ERR_WORKSPACE;
    const W3dKnnPartArgs a{op, reinterpret_cast<const uint8_t *>(X), reinterpret_cast<const uint8_t *>(Y), xscale, yscale, N, M, k, warm, m0, m1, part, parts, part_idx, part_val, idx, dist2, st,
                           m_valid, row_perm};
    W3D_KNN_DISPATCH(w3d_knn_leaf_part_, fmt, d, &a, &w);
    if (fmt == FMT_FP8) { switch (d) { case 128: return w3d_knn_leaf_part_2_128(&a, &w); case 256: return w3d_knn_leaf_part_2_256(&a, &w); default: return W3D_ERR_UNSUPPORTED; } }
    return W3D_ERR_ARG;
}

// fmt: 0 bf16, 1 f16, 2 fp8 e4m3 + per-row E8M0 scales (xscale [N], yscale [M] int32, low byte used)
// warm: 0 cold, 1 warm-start bound in the workspace, 3 pruned sweep (plan in the workspace), 4 pruned sweep with EMIT selection (+ cached lower bounds when lb != NULL:
// lb [row blocks][tiles][512] f16 bits, lb_cum [N] f32 by row, lb_margin = 4e-6 * 2 * cnorm_max of the plan)
extern "C" int w3d_knn_mfma_fwd_lb(const void *X, const void *Y, int64_t N, int64_t M, int d, int k, const int32_t *prev_idx, int32_t *idx, float *dist2, void *ws,
                                   size_t ws_bytes, int phases, int warm, hipStream_t st, const float *prev_dist2, const float *motion2, int fmt,
                                   const int32_t *xscale, const int32_t *yscale, const float *xnorm_in, const int32_t *row_perm, uint16_t *lb, const float *lb_cum,
                                   float lb_margin, float motion_ratio) {
    if (!ws || ws_bytes < w3d_knn_mfma_workspace_bytes(N, M, d, k)) return W3D_ERR_WORKSPACE;
    if (M >= (1ll << 26) || k > 64) return W3D_ERR_UNSUPPORTED;
    if (((uintptr_t)X | (uintptr_t)Y) & 15) return W3D_ERR_ARG;
    if (fmt == FMT_FP8 && (!xscale || !yscale)) return W3D_ERR_ARG;
    if ((lb == nullptr) != (lb_cum == nullptr)) return W3D_ERR_ARG;
    const WsLayout w = carve(ws, N, M, d, k);
    if (w.used > ws_bytes) return W3D_ERR_WORKSPACE;
    const W3dKnnFwdArgs a{reinterpret_cast<const uint8_t *>(X), reinterpret_cast<const uint8_t *>(Y), xscale, yscale, N, M, k, idx, dist2, prev_idx, phases, warm, st, prev_dist2, motion2,
                          (size_t)255 * ROWS_B, xnorm_in, row_perm, lb, lb_cum, lb_margin, motion_ratio};
    W3D_KNN_DISPATCH(w3d_knn_leaf_fwd_, fmt, d, &a, &w);
    if (fmt == FMT_FP8) { switch (d) { case 128: return w3d_knn_leaf_fwd_2_128(&a, &w); case 256: return w3d_knn_leaf_fwd_2_256(&a, &w); default: return W3D_ERR_UNSUPPORTED; } }
    return W3D_ERR_ARG;
}
extern "C" int w3d_knn_mfma_fwd_fmt(const void *X, const void *Y, int64_t N, int64_t M, int d, int k, const int32_t *prev_idx, int32_t *idx, float *dist2, void *ws,
                                    size_t ws_bytes, int phases, int warm, hipStream_t st, const float *prev_dist2, const float *motion2, int fmt,
                                    const int32_t *xscale, const int32_t *yscale, const float *xnorm_in, const int32_t *row_perm) {
    return w3d_knn_mfma_fwd_lb(X, Y, N, M, d, k, prev_idx, idx, dist2, ws, ws_bytes, phases, warm, st, prev_dist2, motion2, fmt, xscale, yscale, xnorm_in, row_perm, nullptr, nullptr, 0.f, 0.f);
}

namespace {
// launch order of a pruned sweep's blocks: by tile count, longest first (counting sort in one block over at most 4096 count classes; ties and classes in any
// order — the order changes no result).  Used by the one-block-per-workgroup launch (sliced sweeps of short row counts).
constexpr int KNN_ORDER_BINS = 4096;
__global__ __launch_bounds__(1024) void knn_block_order_kernel(const int32_t *__restrict__ tile_count, int64_t nblk, int64_t ntiles, int32_t *__restrict__ block_order) {
    if (threadIdx.x == 0) block_order[nblk + KNN_NQ] = 0;               // (the word after the queue cursors) the plans of knn_tile_prune / knn_plan_steady list their tiles in ascending order
    __shared__ int hist[KNN_ORDER_BINS + 1];
    auto bin = [&](int64_t b) -> int { const int64_t c = min((int64_t)tile_count[b], ntiles); return (int)(ntiles <= KNN_ORDER_BINS ? c : c * KNN_ORDER_BINS / ntiles); };
    for (int i = threadIdx.x; i <= KNN_ORDER_BINS; i += 1024) hist[i] = 0;
    __syncthreads();
    for (int64_t b = threadIdx.x; b < nblk; b += 1024) atomicAdd(&hist[bin(b)], 1);
    __syncthreads();
    // start position of every class, longest first: an exclusive scan over the classes in DESCENDING order (position p = BINS - class), four classes per thread
    // (one thread walking the 4097 classes took 44 us — on the critical path between the plan and the sweep)
    static_assert(KNN_ORDER_BINS == 4 * 1024, "four classes per thread of the 1024-thread block (+ the last one)");
    __shared__ int wave_sum_s[16];
    {
        const int t = threadIdx.x, lane = t & 63, w = t >> 6;
        int h[5] = {0, 0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < 4; ++q) h[q] = hist[KNN_ORDER_BINS - (4 * t + q)];
        if (t == 1023) h[4] = hist[0];                                   // position 4096 = class 0
        const int mine = h[0] + h[1] + h[2] + h[3] + h[4];
        int incl = mine;                                                 // inclusive scan inside the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
        if (lane == 63) wave_sum_s[w] = incl;
        __syncthreads();
        int base = 0;
        for (int ww = 0; ww < w; ++ww) base += wave_sum_s[ww];
        int run = base + incl - mine;
#pragma unroll
        for (int q = 0; q < 4; ++q) { hist[KNN_ORDER_BINS - (4 * t + q)] = run; run += h[q]; }
        if (t == 1023) hist[0] = run;
    }
    __syncthreads();
    for (int64_t b = threadIdx.x; b < nblk; b += 1024) block_order[atomicAdd(&hist[bin(b)], 1)] = (int32_t)b;
}

}  // namespace

// byte offset, inside a workspace of w3d_knn_workspace_bytes(N, M, d, k, .), of the per-row-block tile counts a prune plan leaves ([ceil(N/512)] int32): statistics
extern "C" size_t w3d_knn_prune_counts_offset(int64_t N, int64_t M, int d, int k) {
    const WsLayout w = carve(nullptr, N, M, d, k);
    return (size_t)(reinterpret_cast<char *>(w.tord + ((N + ROWS_B - 1) / ROWS_B) * ((M + 63) / 64)) - reinterpret_cast<char *>(0));
}

// byte offset of the straggler buckets' counts ([ceil(M/64)] uint32: (row, tile) pairs the last plan with a straggler threshold took out of the blocks' lists, per tile;
// a count above w3d_knn_straggler_capacity = that tile's bucket overflowed and the pairs past it stayed with their blocks): statistics
extern "C" size_t w3d_knn_straggler_counts_offset(int64_t N, int64_t M, int d, int k) {
    const WsLayout w = carve(nullptr, N, M, d, k);
    return (size_t)(reinterpret_cast<char *>(w.strag_cnt) - reinterpret_cast<char *>(0));
}
extern "C" int w3d_knn_straggler_capacity(int64_t N, int64_t M) { return knn_strag_cap(N, M); }

// A caller-given tile ORDER as the plan of the next sweep: every row block takes ALL tiles, in the order tile_order [row blocks][tiles] names them (each row a
// permutation of 0 ... tiles-1), blocks in natural order; cold != 0 also sets every row's bound to -inf.  What it is for — a loop's FIRST sweep: without a bound the
// running k-th best is all the selection has, and over style rows stored in tree order a row block meets its tiles in an order in which the distance falls for long
// stretches: every tile beats the one before and the candidates come in storms (194 ms for the first sweep at C3, 0.07 of the MFMA peak).  With the tiles nearest to the
// block first the k-th best collapses inside the first few tiles and the rest of the sweep runs like a warm one.  Follow with w3d_knn_match_prepared(warm = 3).
namespace {
__global__ __launch_bounds__(256) void knn_plan_fill_kernel(const int32_t *__restrict__ order_in, int64_t nblk, int64_t ntiles, int32_t *__restrict__ tord,
                                                            int32_t *__restrict__ tcnt, int32_t *__restrict__ bord, float *__restrict__ thr, int64_t N, int cold) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < nblk * ntiles) tord[i] = order_in[i];
    if (i < nblk) { tcnt[i] = (int32_t)ntiles; bord[i] = (int32_t)i; }
    if (i == 0) bord[nblk + KNN_NQ] = 1;                                 // not ascending: the sweep keeps its inclusive threshold after tile-end merges
    if (cold && i < N) thr[i] = -__builtin_inff();
}
}  // namespace
extern "C" int w3d_knn_plan_order(int64_t N, int64_t M, int d, int k, const int32_t *tile_order, int cold, void *workspace, size_t workspace_bytes, w3d_stream_t stream) {
    if (!tile_order || !workspace || N <= 0 || M <= 0 || k <= 0) return W3D_ERR_ARG;
    const WsLayout w = carve(workspace, N, M, d, k);
    if (w.used > workspace_bytes) return W3D_ERR_WORKSPACE;
    const int64_t ntiles = (M + 63) / 64, nblk = (N + ROWS_B - 1) / ROWS_B;
    int32_t *tcnt = w.tord + nblk * ntiles;
    const int64_t n = std::max<int64_t>(nblk * ntiles, N);
    hipLaunchKernelGGL(knn_plan_fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, w3d_s(stream), tile_order, nblk, ntiles, w.tord, tcnt, tcnt + nblk, w.thr, N, cold);
    return w3d_launch_status();
}

// Tile plan of a PRUNED sweep (see knn_tile_prune_kernel): call between w3d_knn_prepare* (which leaves the rows' norms and bounds in the workspace) and
// w3d_knn_match_prepared(warm = 3).  centres [ceil(M/64)][d] in the feature format, cnorm = |c|^2, cradius = max |y - c| over the tile's rows (f32, rounded UP).
namespace {
// STEADY plan (round 4): every iteration between two resets of the cached bounds except the first.  No content rows, no centres, no matrix cores: per block the 512
// rows' q_x = cum_x + D_x and their maxima Q_s per 64 rows; level 1 — a tile whose summaries (per 64 rows the smallest bound any of them holds for it: the ball bound from
// the plan after the reset, or the minimum of what the sweep recorded) exceed Q_s in all eight sub-blocks is out for every row; level 2 — the per-row bounds of the survivors, exactly as in
// the ball-test kernel — and while they are in registers their minimum becomes the tile's new summary.  The summary of a tile the plan KEEPS is set to -inf (unknown: the
// sweep is about to rewrite its rows' bounds), so the next plan sends it through level 2 and recomputes it: the summaries are maintained by the plans alone, the sweep
// does not know they exist.  A tile that passes level 1 without ever having been swept has per-row bounds of -inf and is swept once; from then on its exact bounds speak.
__global__ __launch_bounds__(512) void knn_plan_steady_kernel(const int32_t *__restrict__ row_perm, const float *__restrict__ xnorm, const float *__restrict__ thr, int64_t N,
                                                              int64_t ntiles, int32_t *__restrict__ tile_order, int32_t *__restrict__ tile_count, float cnorm_max,
                                                              const uint16_t *__restrict__ lb, const float *__restrict__ lb_cum, float *__restrict__ summ,
                                                              float *__restrict__ dref, int write_ref, float mrg, uint32_t *__restrict__ strag_cnt, uint32_t *__restrict__ strag,
                                                              int strag_cap, int strag_T) {
    extern __shared__ __attribute__((aligned(16))) uint8_t steady_lds[];
    float *qlds = reinterpret_cast<float *>(steady_lds);                 // [512] q_x, then [512] the rows' references
    float *dlds = qlds + ROWS_B;
    uint32_t *need = reinterpret_cast<uint32_t *>(steady_lds + 2 * ROWS_B * 4);
    __shared__ int wave_tot[8];
    __shared__ float wave_q[8];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int words = (int)((ntiles + 31) / 32);
    for (int i = threadIdx.x; i < words; i += 512) need[i] = 0u;
    // write_ref (the first plan after w3d_knn_bounds_reorder, whose summaries are all unknown): the rows' references move to this plan's q; level 2 — which every tile
    // reaches then — forms the summaries against them
    float q = plan_fill_q(qlds, row_perm, xnorm, thr, lb_cum, N, cnorm_max, mrg, dlds, dref, write_ref != 0);      // q_x - reference_x
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q = fmaxf(q, __shfl_xor(q, o, 64));
    if (lane == 0) wave_q[wid] = q;
    __syncthreads();
    float Qs[8];                                                         // per 64 rows (sub-block s = the rows a sweep's wave s holds): max of cum + D
#pragma unroll
    for (int w = 0; w < 8; ++w) Qs[w] = wave_q[w];
    float *srow = summ + (int64_t)blockIdx.x * ntiles * NWAVES;
    for (int64_t t0 = 0; t0 < ntiles; t0 += 512) {                       // level 1: 64 consecutive tiles per wave and step = two bitmap words
        const int64_t t = t0 + threadIdx.x;
        bool keep = false;
        if (t < ntiles) {
            const float4 s0 = reinterpret_cast<const float4 *>(srow + t * NWAVES)[0], s1 = reinterpret_cast<const float4 *>(srow + t * NWAVES)[1];
            keep = !(s0.x > Qs[0]) || !(s0.y > Qs[1]) || !(s0.z > Qs[2]) || !(s0.w > Qs[3]) || !(s1.x > Qs[4]) || !(s1.y > Qs[5]) || !(s1.z > Qs[6]) || !(s1.w > Qs[7]);
        }
        const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
        if (lane == 0) { const int64_t w0 = (t0 + wid * 64) >> 5; if (w0 < words) need[w0] = (uint32_t)m; if (w0 + 1 < words) need[w0 + 1] = (uint32_t)(m >> 32); }
    }
    __syncthreads();
    // (stragglers: whole blocks only — the positions past the end of a last partial block are copies of the last row, not rows)
    const int sT = ((int64_t)(blockIdx.x + 1) * ROWS_B <= N) ? strag_T : 0;
    uint8_t *mjl = steady_lds + 2 * ROWS_B * 4 + (size_t)((words + 3) & ~3) * 4 + (size_t)wid * 32 * 64;      // behind the tile bitmap: [NWAVES][32][64] bytes
    plan_bounds_phase(need, words, qlds, lb + (int64_t)blockIdx.x * ntiles * ROWS_B, lane, wid, srow, dlds, strag_cnt, strag, strag_cap, sT, (uint32_t)blockIdx.x * ROWS_B, mjl);
    __syncthreads();
    plan_compact(need, ntiles, tile_order + (int64_t)blockIdx.x * ntiles, tile_count + blockIdx.x, wave_tot, lane, wid, srow);
}

__global__ __launch_bounds__(256) void knn_bounds_advance_kernel(const float *__restrict__ motion2, int64_t N, float *__restrict__ cum) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < N) cum[i] += sqrtf(fmaxf(motion2[i], 0.f)) * (1.0f + 1e-5f) + 1e-12f;       // rounded up: a row never moves farther than its cum says
}
constexpr int REORDER_TG = 16;
// f16 bits of the largest half <= v (the cache's bounds are always rounded DOWN)
__device__ __forceinline__ uint16_t f16_floor_bits(float v) {
    _Float16 h = (_Float16)v;
    uint16_t b; __builtin_memcpy(&b, &h, 2);
    if ((float)h > v) b = (b & 0x8000u) ? (uint16_t)(b + 1u) : (b == 0u ? (uint16_t)0x8001u : (uint16_t)(b - 1u));      // one step towards -inf (+inf -> 65504)
    return b;
}
// The cached bounds follow their ROWS into a new sweep order (instead of a reset, which costs a sweep of everything the ball test keeps): position j of the new order
// takes the bounds of the position src_pos[j] the same row had in the old one.  Per (row, tile) the better of the row's own recorded bound and its old sub-block's
// summary + the row's reference (a summary is the minimum over 64 rows of bound - reference: a valid bound of each of them) — the second is what covers tiles the old
// block never swept.  Rebased by the row's cum (the caller zeroes cum afterwards: f16 keeps its resolution over a long run) and rounded down.
__global__ __launch_bounds__(512) void knn_bounds_reorder_kernel(const uint16_t *__restrict__ src, uint16_t *__restrict__ dst, const float *__restrict__ summ,
                                                                 const float *__restrict__ dref, const float *__restrict__ cum, const int32_t *__restrict__ src_pos,
                                                                 const int32_t *__restrict__ new_order, int64_t N, int64_t ntiles) {
    int64_t pos = (int64_t)blockIdx.x * ROWS_B + threadIdx.x;
    if (pos >= N) pos = N - 1;                                           // positions past the end are copies of the last row (plan_fill_q)
    const int64_t row = new_order ? (int64_t)new_order[pos] : pos;
    const int64_t q = src_pos[pos];
    const int64_t ob = q / ROWS_B; const int op = (int)(q % ROWS_B);
    const float c = cum[row], dr = summ ? dref[row] : 0.f;
    const uint16_t *sp = src + ob * ntiles * ROWS_B + op;
    const float *sm = summ ? summ + ob * ntiles * NWAVES + (op >> 6) : nullptr;
    uint16_t *dp = dst + (int64_t)blockIdx.x * ntiles * ROWS_B + threadIdx.x;
    constexpr int U = 8;
    // grid.y = groups of REORDER_TG tiles, the SLOW dimension of the dispatch: the workgroups in flight cover all row blocks for a narrow band of tiles, so the 32-B
    // sectors of the old array that rows from several new blocks share (a refresh of the C3 loop's order scatters 40 % of the rows: 20 sectors per 64 positions
    // instead of 4) are fetched from HBM once and found in the caches by the others
    const int64_t tbeg = (int64_t)blockIdx.y * REORDER_TG, tend = min(ntiles, tbeg + REORDER_TG);
    for (int64_t t0 = tbeg; t0 < tend; t0 += U) {
        uint16_t b[U]; float s[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t t = min(t0 + u, ntiles - 1);
            b[u] = sp[t * ROWS_B];
            s[u] = sm ? sm[t * NWAVES] : -__builtin_inff();
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (t0 + u >= tend) break;
            _Float16 h; __builtin_memcpy(&h, &b[u], 2);
            float v = (float)h;
            const float sv = (s[u] + dr) * (s[u] + dr >= 0.f ? 1.0f - 2e-7f : 1.0f + 2e-7f);      // the sum rounded towards -inf
            v = fmaxf(v, sv);
            v = v - c; v -= 2e-7f * fabsf(v) + 1e-30f;
            dp[(t0 + u) * ROWS_B] = f16_floor_bits(v);
        }
    }
}
}  // namespace

// Cached lower bounds of a loop's pruned sweeps: cum[x] += |x_new - x_old| (motion2 = the squared motion of every row since the previous sweep, as the featurisation
// pass reports it).  Call once per iteration, before the plan.  A reset of the cache (lb = -inf bits 0xFC00, cum = 0) is the caller's fill.
extern "C" int w3d_knn_bounds_advance(const float *motion2, int64_t N, float *cum, w3d_stream_t stream) {
    if (!motion2 || !cum || N <= 0) return W3D_ERR_ARG;
    hipLaunchKernelGGL(knn_bounds_advance_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, w3d_s(stream), motion2, N, cum);
    return w3d_launch_status();
}

// The cached bounds under a NEW sweep order without a reset: dst (a second array of lb's size) = the bounds of src re-keyed by src_pos [ceil(N/512) * 512 ... at least N]
// (old position of the row at each new position), new_order = the new row order (NULL: identity).  summ / dref (both or neither): the old order's summaries and the
// rows' references, folded into the per-row bounds.  Afterwards the caller zeroes cum (the bounds are rebased), fills summ with -inf and makes the next plan a
// steady = 2 one.
extern "C" int w3d_knn_bounds_reorder(const uint16_t *lb_src, uint16_t *lb_dst, const float *summ, const float *dref, const float *cum, const int32_t *src_pos,
                                      const int32_t *new_order, int64_t N, int64_t M, w3d_stream_t stream) {
    if (!lb_src || !lb_dst || lb_src == lb_dst || !cum || !src_pos || N <= 0 || M <= 0 || ((summ == nullptr) != (dref == nullptr))) return W3D_ERR_ARG;
    const int64_t ntiles = (M + 63) / 64, nblk = (N + ROWS_B - 1) / ROWS_B;
    hipLaunchKernelGGL(knn_bounds_reorder_kernel, dim3((unsigned)nblk, (unsigned)((ntiles + REORDER_TG - 1) / REORDER_TG)), dim3(512), 0, w3d_s(stream), lb_src, lb_dst, summ, dref,
                       cum, src_pos, new_order, N, ntiles);
    return w3d_launch_status();
}

// score_err2: the allowance for the sweep's score error, as 2E relative to |x|^2 + 2 max |y|^2 (<= 0: the 16-bit sweeps' 4e-6; plans for fp8 sweeps: W3D_KNN_SCORE_ERR2_FP8)
extern "C" int w3d_knn_prune_plan_err(const void *X, int64_t N, int64_t M, int d, int k, int dtype, const int32_t *row_order, const void *centres, const float *cnorm,
                                      const float *cradius, float cnorm_max, float score_err2, const uint16_t *lb, const float *lb_cum, float *summ, float *dref, int steady,
                                      void *workspace, size_t workspace_bytes, w3d_stream_t stream) {
    const float mrg = score_err2 > 0.f ? score_err2 : KNN_SCORE_ERR2;
    if (!X || !centres || !cnorm || !cradius || !workspace || N <= 0 || M <= 0 || k <= 0 || !(cnorm_max >= 0.f) || !(mrg < 1e-3f)) return W3D_ERR_ARG;
    if ((lb == nullptr) != (lb_cum == nullptr) || (summ && !lb) || (steady && !summ) || ((summ == nullptr) != (dref == nullptr))) return W3D_ERR_ARG;
    if ((dtype != W3D_DTYPE_BF16 && dtype != W3D_DTYPE_F16) || (d != 64 && d != 128 && d != 256)) return W3D_ERR_UNSUPPORTED;
    const WsLayout w = carve(workspace, N, M, d, k);
    if (w.used > workspace_bytes) return W3D_ERR_WORKSPACE;
    const int64_t ntiles = (M + 63) / 64, nblk = (N + ROWS_B - 1) / ROWS_B;
    int32_t *tcnt = w.tord + nblk * ntiles;
    const size_t lds = (size_t)2 * 64 * (d * 2) + (size_t)std::min<int64_t>((ntiles + 63) / 64, PRUNE_CG) * 128 * 4 + (size_t)((((ntiles + 31) / 32) + 3) & ~(int64_t)3) * 4;
    if (lds > 160 * 1024) return W3D_ERR_UNSUPPORTED;
    hipStream_t st = w3d_s(stream);
    // bits 8-15 of `steady`: the straggler threshold (rows; 0 = none).  Every plan that is given one starts from empty buckets — also a ball-test plan, which fills none
    const int strag_T = (steady >> 8) & 0xff;
    if ((steady & ~0xffff) || (strag_T && !lb)) return W3D_ERR_ARG;
    steady &= 0xff;
    if (strag_T) W3D_HIP(hipMemsetAsync(w.strag_cnt, 0, (size_t)ntiles * sizeof(uint32_t), st));
    if (steady) {
        // between two resets of the cached bounds every plan but the first: block summaries + per-row bounds only (no x.c pass over the centres)
        const size_t lds_s = (size_t)2 * ROWS_B * 4 + (size_t)((((ntiles + 31) / 32) + 3) & ~(int64_t)3) * 4 + (strag_T ? (size_t)NWAVES * 32 * 64 : 0);
        if (lds_s > 64 * 1024) return W3D_ERR_UNSUPPORTED;
        hipLaunchKernelGGL(knn_plan_steady_kernel, dim3((unsigned)nblk), dim3(512), lds_s, st, row_order, (const float *)w.xnorm, (const float *)w.thr, N, ntiles, w.tord, tcnt, cnorm_max,
                           lb, lb_cum, summ, dref, steady == 2 ? 1 : 0, mrg, w.strag_cnt, w.strag, w.strag_cap, strag_T);
        hipLaunchKernelGGL(knn_block_order_kernel, dim3(1), dim3(1024), 0, st, (const int32_t *)tcnt, nblk, ntiles, tcnt + nblk);
        return w3d_launch_status();
    }
    const W3dKnnPruneArgs a{reinterpret_cast<const uint8_t *>(X), row_order, reinterpret_cast<const uint8_t *>(centres), cnorm, cradius, N, ntiles, nblk, tcnt, cnorm_max, lb, lb_cum, summ, dref, lds, st, mrg};
    auto leaf = [&]() -> int { W3D_KNN_DISPATCH(w3d_knn_leaf_prune_, dtype == W3D_DTYPE_BF16 ? FMT_BF16 : FMT_F16, d, &a, &w); return W3D_ERR_UNSUPPORTED; };
    const int rc = leaf();
    if (rc != W3D_OK) return rc;
    hipLaunchKernelGGL(knn_block_order_kernel, dim3(1), dim3(1024), 0, st, (const int32_t *)tcnt, nblk, ntiles, tcnt + nblk);
    return w3d_launch_status();
}
extern "C" int w3d_knn_prune_plan_lb(const void *X, int64_t N, int64_t M, int d, int k, int dtype, const int32_t *row_order, const void *centres, const float *cnorm,
                                     const float *cradius, float cnorm_max, const uint16_t *lb, const float *lb_cum, float *summ, float *dref, int steady, void *workspace,
                                     size_t workspace_bytes, w3d_stream_t stream) {
    return w3d_knn_prune_plan_err(X, N, M, d, k, dtype, row_order, centres, cnorm, cradius, cnorm_max, 0.f, lb, lb_cum, summ, dref, steady, workspace, workspace_bytes, stream);
}
extern "C" int w3d_knn_prune_plan(const void *X, int64_t N, int64_t M, int d, int k, int dtype, const int32_t *row_order, const void *centres, const float *cnorm,
                                  const float *cradius, float cnorm_max, void *workspace, size_t workspace_bytes, w3d_stream_t stream) {
    return w3d_knn_prune_plan_lb(X, N, M, d, k, dtype, row_order, centres, cnorm, cradius, cnorm_max, nullptr, nullptr, nullptr, nullptr, 0, workspace, workspace_bytes, stream);
}

extern "C" int w3d_knn_mfma_fwd(const void *X, const void *Y, int64_t N, int64_t M, int d, int k, const int32_t *prev_idx, int32_t *idx, float *dist2, void *ws,
                                size_t ws_bytes, int phases, int warm, hipStream_t st, const float *prev_dist2, const float *motion2) {
    return w3d_knn_mfma_fwd_fmt(X, Y, N, M, d, k, prev_idx, idx, dist2, ws, ws_bytes, phases, warm, st, prev_dist2, motion2, FMT_BF16, nullptr, nullptr, nullptr, nullptr);
}
#endif   // !W3D_KNN_LEAF
