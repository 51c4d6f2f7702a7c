// a3 backward (index_put accumulate=True of NB11:994-1009) as ONE row-order pass with an LDS accumulation window, optionally fused with
// the NN-loss backward (dL/dF of a4) so that the [N,d] feature gradient never goes through HBM.
//
// Why: in a scene stored along a space-filling curve, the neighbours of rows r0..r0+B sit in a narrow index range around them.
// A block takes B consecutive rows and keeps dL/dX for a WINDOW of `win` points (12 B each, up to 156 KB of the 160 KB LDS):
//   * every (row, neighbour) contribution whose destination falls inside the window is an LDS atomic (no HBM traffic, no L2 atomics),
//   * the rest (and any list that is not local at all) falls back to a global float atomic — correct for ANY list,
//   * at the end the window is added to gX once (3*win atomics per B rows instead of 3*K per row).
// The LDS accumulators are 32-bit FIXED POINT: measured on gfx950 (scripts/micro/lds_atomic_rate.hip, random addresses, 16 waves/CU)
// ds_add_u32 retires 11.4 lanes/clk/CU, ds_add_f32 0.38 — a float window is 30x slower and was the whole kernel's bound.
// Per block: bound G >= max |contribution| of its rows (exact max |gF| in the plain form; k * 2*scale * (max|F_X row| + max|F_Y|)
// in the fused form), fan-in bound H (power of two >= the largest number of list entries naming one point), S = 2^e the largest
// power of two with G*H*S <= 2^30; a contribution c is added as rint(c*S) (clamped to +-2^30/H against rounding in the bound);
// the window is flushed as float(acc)/S.  H comes from the caller (w3d_list_max_fanin, once per frozen list).  Integer adds are associative: the in-window part of the sum does not
// depend on the order the waves run in.  Resolution: G / 2^(30 - log2 H) per contribution (H = 1024: 1e-6 G).
// HBM traffic: idx (4 B) + gF (4 B, read twice in the plain form, never in the fused form) per contribution, streamed, coalesced;
// points are 16-B L2 gathers.  The CSR form (geometry.hip) stays as the bit-reproducible option.
#include "common.h"
#include <cstdlib>

namespace {

constexpr int WB_THREADS = 1024;
constexpr int WB_WAVES = WB_THREADS / 64;
constexpr int WB_MAX_WIN = 13312;                 // 159 744 B of LDS
constexpr int WB_DEFAULT_WIN = 12288;             // 147 456 B: leaves 16 KB of the CU's LDS to co-resident blocks of other kernels (the render leg's
                                                  // blend kernels need 10.5 KB), so a VALU-bound kernel can share the CU with this latency-bound one

__device__ __forceinline__ void lds_add_i(int *p, int v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

__device__ __forceinline__ float block_max(float v, float *red /*[16]*/) {
    v = wave_max(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = red[0];
#pragma unroll
    for (int w = 1; w < WB_WAVES; ++w) r = fmaxf(r, red[w]);
    __syncthreads();
    return r;
}

// FUSED: g[c] = scale2 * (k * FX[r,c] - sum_j FY[nn[r,j], c])   (w3d_knn_match_bwd), formed in registers per row
// F16: the feature rows are f16 instead of bf16 (fused form)
template <bool FUSED, bool F16, bool STAGE = false>
__global__ __launch_bounds__(WB_THREADS) void descriptor_bwd_window_kernel(const float4 *__restrict__ X4, const int32_t *__restrict__ idx, int64_t R, int K,
                                                                           const float *__restrict__ gF, const uint16_t *__restrict__ FX,
                                                                           const uint16_t *__restrict__ FY, uint32_t M, const int32_t *__restrict__ nn_idx, int knn, float scale2,
                                                                           float fy_absmax, float *__restrict__ gF_out, int rows_per_block, int win, int fanin_pow2,
                                                                           int64_t n_points, float *__restrict__ gX, const float *__restrict__ nn_dist2, float loss_scale,
                                                                           float *__restrict__ loss_accum, const uint16_t *__restrict__ nbr_slot, const int32_t *__restrict__ dest,
                                                                           const int32_t *__restrict__ dest_off, const int32_t *__restrict__ root_idx,
                                                                           float *__restrict__ sum_cache, int32_t *__restrict__ sum_idx) {
    // sum_cache [R][d] f32 + sum_idx [R][knn] (fused form, optional, d <= 256; round 4): the NEIGHBOUR-SUM CACHE of a loop.  sum_j FY[nn[r, j]] depends on the SET of a
    // row's neighbours only (the style rows are constant), and in a running loop 75-96 % of the rows keep their set from one iteration to the next (almost all others
    // exchange ONE neighbour; profiles/r04s_nn_sum_cache.log): a row whose list names the same k style rows as the list its cached sum was formed from (sum_idx; exact set
    // comparison, k ballots) takes the sum from the cache — one coalesced 1-KB read instead of k gathered 512-B rows out of the Infinity Cache — and every other row
    // gathers as before and refreshes both.  Cached sum and list are prefetched with the row's other operands, one row ahead.  A cached sum was added in ITS list's
    // order: against a fresh sum it may differ in the last bit (f32 addition order).  Fill sum_idx with -1 to invalidate (new style rows, new N).
    // root_idx (slot plan, optional, [R]): the rows' roots as a dense array — given when EVERY entry of the lists has a slot (no 0xffff): the index rows are then
    // not read at all (1 GB per pass at C3 and a quarter of the remaining L1 accesses: the slots say where an entry's point sits)
    constexpr bool stage_pos = STAGE;                                // compile-time: the test per entry cost the unstaged kernel 5 %
    // nbr_slot / dest / dest_off (optional, FROZEN lists: w3d "slot plan"): the block's window is not an index range but the SET of points its rows name —
    // dest[dest_off[b] .. dest_off[b+1]) — and nbr_slot[r][c] is the place of entry (r, c)'s point in that set (0xffff: not in it -> global atomic).  Lists that
    // are local in space but not along the storage curve (the exact 256 nearest neighbours of a Z-ordered scene: 16 % of the entries fall outside a 12 288-point
    // index window) then stay in LDS entirely; no window arithmetic per entry.
    extern __shared__ __attribute__((aligned(16))) int acc[];        // [win][3] fixed point
    __shared__ float red[WB_WAVES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave-uniform: row pointers are scalar, the neighbour indices of a row become scalar loads
    const int64_t doff = nbr_slot ? (int64_t)dest_off[blockIdx.x] : 0;
    // stage_pos (slot plan, sets of at most 6 826 points): the positions of the block's set are staged behind the accumulators (12 + 12 B per point) and the
    // neighbour positions of every entry come from LDS — the 256 16-B gathers per row were a third of the kernel's L1 accesses (770 per row, one per clock per CU)
    float *pts = reinterpret_cast<float *>(acc + 3 * win);           // `win` here = the launch's max_set: the accumulators of every block end before it
    if (nbr_slot) win = dest_off[blockIdx.x + 1] - (int)doff;         // host: every block's set fits the LDS the launch asked for
    for (int i = tid; i < win * 3; i += WB_THREADS) acc[i] = 0;
    if (stage_pos) for (int i = tid; i < win; i += WB_THREADS) { const float4 q = X4[dest[doff + i]]; pts[3 * i] = q.x; pts[3 * i + 1] = q.y; pts[3 * i + 2] = q.z; }
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = min(R, r0 + (int64_t)rows_per_block);
    const int d = K - 1;
    // ---- bound on |contribution| over the block's rows (|unit vector| <= 1)
    float gmax = 0.f;
    if (FUSED) {
        const int64_t n8 = (r1 - r0) * d / 8;                        // d % 8 == 0 on this path (host check): 16-B loads
        const uint4 *p = reinterpret_cast<const uint4 *>(FX + r0 * d);
        uint32_t m = 0u;                                             // max of |bf16| compared as integers (sign cleared)
        // eight loads in flight per thread (one per turn waited for a full memory latency 32 times per block: 512 KB of rows at one 16-B load per thread and round trip)
        auto fold = [&](const uint4 &v) {
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) { m = max(m, (w[q] & 0x7fffu) << 16); m = max(m, w[q] & 0x7fff0000u); }   // both 16-bit formats order by magnitude bits
        };
        int64_t i = tid;
        for (; i + 7 * WB_THREADS < n8; i += 8 * WB_THREADS) {
            uint4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[i + u * WB_THREADS];
#pragma unroll
            for (int u = 0; u < 8; ++u) fold(v[u]);
        }
        for (; i < n8; i += WB_THREADS) fold(p[i]);
        float fxmax;
        if (F16) { const uint16_t hb = (uint16_t)(m >> 16); _Float16 hh; __builtin_memcpy(&hh, &hb, 2); fxmax = (float)hh; } else fxmax = __uint_as_float(m);
        gmax = fabsf(scale2) * (float)knn * (fxmax + fy_absmax);
    } else {
        const float *p = gF + r0 * d;
        const int64_t n = (r1 - r0) * d;
        for (int64_t i = tid; i < n; i += WB_THREADS) gmax = fmaxf(gmax, fabsf(p[i]));
    }
    gmax = block_max(gmax, red);
    // S = 2^e with gmax * fanin * S <= 2^30
    int ex = 0;
    (void)frexpf(gmax * (float)fanin_pow2, &ex);                     // gmax*fanin < 2^ex
    ex = max(ex, -90);                                               // keep S finite for vanishing gradients
    const bool finite = gmax < 3.0e38f;                              // inf / NaN in the gradient: no window, plain float atomics propagate it
    const float S = gmax > 0.f ? ldexpf(1.0f, 30 - ex) : 0.f;
    const float invS = gmax > 0.f ? ldexpf(1.0f, ex - 30) : 0.f;
    if (!finite) win = 0;
    // window centred on the block's roots
    const int64_t centre = ((int64_t)idx[r0 * K] + (int64_t)idx[(r1 - 1) * K]) >> 1;
    int64_t base = centre - win / 2;
    if (base > n_points - win) base = n_points - win;
    if (base < 0) base = 0;
    __syncthreads();
    // a destination outside the window (index-window form; never under a covering slot plan) is a global float atomic
    auto add = [&](int64_t n, float gx, float gy, float gz, uint32_t lslot) {
        const int64_t off = nbr_slot ? (lslot == 0xffffu ? -1 : (int64_t)lslot) : n - base;
        if ((uint64_t)off < (uint64_t)win) {
            int *p = acc + 3 * off;
            // |contribution| <= G by construction (|unit vector| <= 1, G bounds |dL/dF| of the block), so c * S fits 2^30 / fan-in: no clamp needed
            lds_add_i(p, __float2int_rn(gx * S));
            lds_add_i(p + 1, __float2int_rn(gy * S));
            lds_add_i(p + 2, __float2int_rn(gz * S));
        } else {
            // slot plan without index rows (root_idx): the destination is the slot's point (reached when a non-finite gradient has switched the window off)
            if (nbr_slot && lslot != 0xffffu) n = dest[doff + lslot];
            float *p = gX + 3 * n;
            unsafeAtomicAdd(p, gx); unsafeAtomicAdd(p + 1, gy); unsafeAtomicAdd(p + 2, gz);
        }
    };
    const bool fast = stage_pos && root_idx != nullptr && finite && gmax > 0.f && (d & 3) == 0;      // block-uniform
    float lsum = 0.f;                                                // NN-loss head (w3d_knn_loss) of the block's rows, if asked for
    // buffer resource of the style set (fused form): raw, no stride, bounds = the whole set (gfx9 raw-buffer word 3: 32-bit data format)
    const __amdgpu_buffer_rsrc_t fy_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(FY), (short)0, FUSED ? (int)(M * (uint32_t)(K - 1) * 2u) : 0, 0x00020000);
    // a wave walks its rows one after the other: the index quad (and, fused, the own feature quad) of the NEXT row are loaded while the current
    // row is processed, so that a row's gathers can all issue at its start instead of waiting for its index loads
    const bool one_group = d <= 256;
    const bool use_cache = FUSED && sum_cache != nullptr && one_group && knn <= 64;
    // Round 4 — the row walk is a two-deep software pipeline (pipe): EVERYTHING a row needs from global memory is in registers when its turn comes — index / feature /
    // slot quads, its neighbour list as a VECTOR (lane j = nn[r][j]; the gathers' scalar offsets and the cache's set comparison come out of it by v_readlane), its
    // selected squared distances, the cached sum + list, and the ROOT's position, whose index is fetched two rows ahead.  Before, root index -> root position and the
    // k scalar index loads sat on every row's critical path (two dependent memory latencies, and an s_load in flight also blocks the row's LDS reads: lgkmcnt counts
    // both): the kernel ran at ~6 us per row and wave whether it gathered or not (1.52 ms with every row served by the cache against 1.59 without, 1.47 with the LDS
    // atomics removed: neither the gathers nor the atomics were its time).  All of these are VECTOR loads (addresses laundered through a VGPR where they are uniform), so
    // the waits are vmcnt waits, in issue order.
    const bool pipe = one_group && (!FUSED || knn <= 64);
    int vz = 0;
    asm volatile("" : "+v"(vz));                                     // a zero the compiler cannot see through: uniform addresses + vz = vector loads
    auto load_group = [&](int64_t r_, int c0_, int32_t (&nb_)[4], uint2 &xv_, uint2 &sq_, float4 &cs_) {
        if (use_cache && c0_ < d) cs_ = *reinterpret_cast<const float4 *>(sum_cache + r_ * d + c0_);
        const int32_t *row_ = idx + r_ * K;
        const int nc_ = min(4, d - c0_);
        if (!root_idx) {
#pragma unroll
            for (int c = 0; c < 4; ++c) nb_[c] = c < nc_ ? row_[c0_ + c + 1] : row_[0];
        }
        if (FUSED && c0_ < d) xv_ = *reinterpret_cast<const uint2 *>(FX + r_ * d + c0_);
        if (nbr_slot && c0_ < d) sq_ = *reinterpret_cast<const uint2 *>(nbr_slot + r_ * d + c0_);       // d % 4 == 0 in slot mode (host check)
    };
    auto load_root = [&](int64_t r_) -> int32_t { return root_idx ? root_idx[r_ + vz] : idx[(r_ + vz) * K]; };
    auto load_extras = [&](int64_t r_, int32_t root_, int32_t &old_, int32_t &nnv_, float &d2_, float4 &rp_) {
        if (FUSED) {
            nnv_ = lane < knn ? nn_idx[r_ * knn + lane] : -1;
            if (nn_dist2) d2_ = lane < knn ? nn_dist2[r_ * knn + lane] : 0.f;
        }
        if (use_cache) old_ = lane < knn ? sum_idx[r_ * knn + lane] : -1;
        rp_ = X4[root_ + vz];
    };
    int32_t pre_nb[4] = {0, 0, 0, 0};
    uint2 pre_xv = make_uint2(0u, 0u), pre_sq = make_uint2(0xffffffffu, 0xffffffffu);
    int32_t pre_old = -1, pre_nnv = -1, pre_root = 0, pre2_root = 0;
    float pre_d2 = 0.f;
    float4 pre_cs = make_float4(0.f, 0.f, 0.f, 0.f), pre_rp = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pipe && r0 + wid < r1) {
        pre_root = __builtin_amdgcn_readfirstlane(load_root(r0 + wid));
        if (lane * 4 < d) load_group(r0 + wid, lane * 4, pre_nb, pre_xv, pre_sq, pre_cs);
        load_extras(r0 + wid, pre_root, pre_old, pre_nnv, pre_d2, pre_rp);
        if (r0 + wid + WB_WAVES < r1) pre2_root = load_root(r0 + wid + WB_WAVES);
    }
    // ---- THE LOOP'S CASE, specialised (round 4): fused form, covering slot plan with staged positions, finite gradients, d <= 256, k <= 64 — its own row loop, with
    // nothing of the general one in it (index rows, window tests, fallback fetches, scalar list loads): every operand of a row is prefetched one row ahead as a vector load,
    // the positions of a lane's four entries are read from LDS before any of its accumulator atomics (the compiler may not move an LDS read across an LDS atomic), and a
    // row the cache serves issues no gather.  General loop below: 416 vector + 250 scalar instructions per row at C3 (profiles/r04s_bwd_pmc.log); this one: 303 + 211 in the loop's mix of cache hits and
    // misses (profiles/r04t_inst_mix.log).
    if (FUSED && fast && pipe && gF_out == nullptr) {
        const uint32_t row_bytes = (uint32_t)d * 2u, lane_off = (uint32_t)lane * 8u;
        const bool act = lane * 4 < d;                                       // feature lanes (all 64 at d = 256)
        auto lo16 = [](uint32_t w) -> float { if (F16) { const uint16_t b = (uint16_t)(w & 0xffffu); _Float16 hh; __builtin_memcpy(&hh, &b, 2); return (float)hh; } return __uint_as_float(w << 16); };
        auto hi16 = [](uint32_t w) -> float { if (F16) { const uint16_t b = (uint16_t)(w >> 16); _Float16 hh; __builtin_memcpy(&hh, &b, 2); return (float)hh; } return __uint_as_float(w & 0xffff0000u); };
        auto ldrow = [&](uint32_t soff) -> uint2 {
            typedef int i32x2_t __attribute__((ext_vector_type(2)));
            const i32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(fy_rsrc, (int)lane_off, (int)soff, 0);
            return make_uint2((uint32_t)v[0], (uint32_t)v[1]);
        };
        uint2 p_xv = make_uint2(0u, 0u), p_sq = make_uint2(0u, 0u);
        float4 p_cs = make_float4(0.f, 0.f, 0.f, 0.f), p_rp = make_float4(0.f, 0.f, 0.f, 0.f);
        int32_t p_old = -1, p_nnv = -1, p_root = 0, p2_root = 0;
        float p_d2 = 0.f;
        auto prefetch = [&](int64_t r_, int32_t root_) {
            if (act) {
                p_xv = *reinterpret_cast<const uint2 *>(FX + r_ * d + lane * 4);
                p_sq = *reinterpret_cast<const uint2 *>(nbr_slot + r_ * d + lane * 4);
                if (use_cache) p_cs = *reinterpret_cast<const float4 *>(sum_cache + r_ * d + lane * 4);
            }
            p_nnv = lane < knn ? nn_idx[r_ * knn + lane] : -1;
            if (nn_dist2) p_d2 = lane < knn ? nn_dist2[r_ * knn + lane] : 0.f;
            if (use_cache) p_old = lane < knn ? sum_idx[r_ * knn + lane] : -1;
            p_rp = X4[root_ + vz];
        };
        if (r0 + wid < r1) {
            p_root = __builtin_amdgcn_readfirstlane(root_idx[r0 + wid + vz]);
            prefetch(r0 + wid, p_root);
            if (r0 + wid + WB_WAVES < r1) p2_root = root_idx[r0 + wid + WB_WAVES + vz];
        }
        const float kf = (float)knn;
        for (int64_t r = r0 + wid; r < r1; r += WB_WAVES) {
            const int32_t root = p_root;
            const float4 rp = p_rp;
            const uint2 xv = p_xv, sq = p_sq;
            const float4 cs = p_cs;
            const int32_t old = p_old, nnv = p_nnv;
            lsum += p_d2;
            if (r + WB_WAVES < r1) {
                const int32_t nroot = __builtin_amdgcn_readfirstlane(p2_root);
                prefetch(r + WB_WAVES, nroot);
                p_root = nroot;
                if (r + 2 * WB_WAVES < r1) p2_root = root_idx[r + 2 * WB_WAVES + vz];
            }
            // the row's list (lanes 0 ... k - 1): range check + clamp once, then the cache's set comparison
            const bool oob = lane < knn && (uint32_t)nnv >= M;
            const bool bad = __builtin_amdgcn_ballot_w64(oob) != 0ull;
            const int32_t nnc = oob ? 0 : nnv;
            bool cached = false;
            if (use_cache) {
                cached = __builtin_amdgcn_ballot_w64(old != nnv) == 0ull;      // same list, same order (lanes >= k hold -1 in both): the common case
                if (!cached) {                                                 // same SET (k distinct rows each): every new entry is among the old ones
                    cached = true;
                    for (int jj = 0; jj < knn; ++jj) cached = cached && (__builtin_amdgcn_ballot_w64(old == __builtin_amdgcn_readlane(nnv, jj)) != 0ull);
                }
                // a list with an out-of-range entry (-1, the kNN kernel's sentinel) never hits — such an entry would "find itself" among the padding lanes or in
                // an invalidated cache entry; the row gathers, clamped, and its gradient is poisoned as on the uncached path.  An in-range entry never equals the padding's
                // -1, so the valid lists need no lane mask (restricting every ballot to the list's lanes cost the cached pass 0.15 ms: profiles/r05v_bwd_ab.log).
                cached = cached && !bad;
            }
            float a0 = cs.x, a1 = cs.y, a2 = cs.z, a3 = cs.w;
            if (!cached) {
                a0 = a1 = a2 = a3 = 0.f;
                auto acc4 = [&](const uint2 v) { a0 += lo16(v.x); a1 += hi16(v.x); a2 += lo16(v.y); a3 += hi16(v.y); };
                int j = 0;
                for (; j + 10 <= knn; j += 10) {
                    uint2 pv[10];
#pragma unroll
                    for (int u = 0; u < 10; ++u) pv[u] = ldrow((uint32_t)__builtin_amdgcn_readlane(nnc, j + u) * row_bytes);
#pragma unroll
                    for (int u = 0; u < 10; ++u) acc4(pv[u]);
                }
                for (; j < knn; ++j) acc4(ldrow((uint32_t)__builtin_amdgcn_readlane(nnc, j) * row_bytes));
                if (bad) a0 = a1 = a2 = a3 = __builtin_nanf("");
                if (use_cache) {
                    if (act) *reinterpret_cast<float4 *>(sum_cache + r * d + lane * 4) = make_float4(a0, a1, a2, a3);
                    if (lane < knn) sum_idx[r * knn + lane] = nnv;
                }
            }
            float sx = 0.f, sy = 0.f, sz = 0.f;
            if (act) {
                const float g[4] = {scale2 * (kf * lo16(xv.x) - a0), scale2 * (kf * hi16(xv.x) - a1), scale2 * (kf * lo16(xv.y) - a2), scale2 * (kf * hi16(xv.y) - a3)};
                const uint32_t sl3[4] = {3u * (sq.x & 0xffffu), 3u * (sq.x >> 16), 3u * (sq.y & 0xffffu), 3u * (sq.y >> 16)};
                float px[4], py[4], pz[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) { const float *pp = pts + sl3[c]; px[c] = pp[0]; py[c] = pp[1]; pz[c] = pp[2]; }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float vx = px[c] - rp.x, vy = py[c] - rp.y, vz_ = pz[c] - rp.z;
                    const float d2 = vx * vx + vy * vy + vz_ * vz_;
                    const float w = d2 > 0.f ? g[c] * __builtin_amdgcn_rsqf(d2) : 0.f;   // g / |v| (torch.norm backward: 0 at 0); v_rsq_f32: 1 ulp
                    const float gx = w * vx, gy = w * vy, gz = w * vz_;
                    int *pa = acc + sl3[c];
                    lds_add_i(pa, __float2int_rn(gx * S));
                    lds_add_i(pa + 1, __float2int_rn(gy * S));
                    lds_add_i(pa + 2, __float2int_rn(gz * S));
                    sx += gx; sy += gy; sz += gz;
                }
            }
            // the root's share is a sum of up to d contributions (outside the fan-in bound): one global float atomic per row and axis
            sx = wave_sum(sx); sy = wave_sum(sy); sz = wave_sum(sz);
            // (round 6: ONE atomic instruction with three lanes — x, y, z are 12 consecutive bytes — instead of three one-lane instructions; worth 0.01 ms: the pass is
            // not bound by its atomics, profiles/r06n)
            if (lane < 3) unsafeAtomicAdd(gX + 3 * (int64_t)root + lane, lane == 0 ? -sx : (lane == 1 ? -sy : -sz));
        }
    } else
    for (int64_t r = r0 + wid; r < r1; r += WB_WAVES) {
        const int32_t *row = idx + r * K;
        int64_t root;
        float4 rp;
        int32_t cur_nb[4] = {pre_nb[0], pre_nb[1], pre_nb[2], pre_nb[3]};
        const uint2 cur_xv = pre_xv, cur_sq = pre_sq;
        const int32_t cur_old = pre_old; const float4 cur_cs = pre_cs;
        int32_t cur_nnv = pre_nnv;
        if (pipe) {
            root = pre_root; rp = pre_rp;
            if (FUSED && nn_dist2) lsum += pre_d2;
            if (r + WB_WAVES < r1) {
                const int32_t nroot = __builtin_amdgcn_readfirstlane(pre2_root);
                if (lane * 4 < d) load_group(r + WB_WAVES, lane * 4, pre_nb, pre_xv, pre_sq, pre_cs);
                load_extras(r + WB_WAVES, nroot, pre_old, pre_nnv, pre_d2, pre_rp);
                pre_root = nroot;
                if (r + 2 * WB_WAVES < r1) pre2_root = load_root(r + 2 * WB_WAVES);
            }
        } else {
            if (FUSED && nn_dist2) { for (int jq = lane; jq < knn; jq += 64) lsum += nn_dist2[r * knn + jq]; }
            root = root_idx ? root_idx[r] : row[0];
            rp = X4[root];
        }
        // row level, ALL lanes (the list sits in lanes 0 ... k - 1, whatever d is): range check + clamp of the list, and the cache's set comparison
        int32_t nnc = cur_nnv;
        uint32_t row_bad = 0u;
        bool row_cached = false;
        if (FUSED && pipe) {
            const bool oob = lane < knn && (uint32_t)cur_nnv >= M;
            row_bad = __builtin_amdgcn_ballot_w64(oob) != 0ull ? 1u : 0u;
            nnc = oob ? 0 : cur_nnv;
            if (use_cache) {                                  // same SET as the list the cached sum was formed from?  (lists hold k distinct rows: every new entry found = equal sets)
                row_cached = __builtin_amdgcn_ballot_w64(cur_old != cur_nnv) == 0ull;      // the same list in the same order (lanes >= k hold -1 in both): the common case
                if (!row_cached) {
                    row_cached = true;
                    for (int jj = 0; jj < knn; ++jj) row_cached = row_cached && (__builtin_amdgcn_ballot_w64(cur_old == __builtin_amdgcn_readlane(cur_nnv, jj)) != 0ull);
                }
                row_cached = row_cached && !row_bad;         //  a list with an out-of-range entry never hits: see the specialised row loop above
            }
        }
        float sx = 0.f, sy = 0.f, sz = 0.f;
        for (int c0 = lane * 4; c0 < d; c0 += 256) {
            const int nc = min(4, d - c0);
            float g[4] = {0.f, 0.f, 0.f, 0.f};
            int32_t nb[4];
            uint2 xv = cur_xv, sq = cur_sq;
            if (one_group) {
#pragma unroll
                for (int c = 0; c < 4; ++c) nb[c] = cur_nb[c];
            } else {
                float4 c_ = make_float4(0.f, 0.f, 0.f, 0.f);
                load_group(r, c0, nb, xv, sq, c_);
            }
            const uint32_t ls[4] = {sq.x & 0xffffu, sq.x >> 16, sq.y & 0xffffu, sq.y >> 16};
            if (FUSED) {
                auto lo16 = [](uint32_t w) -> float { if (F16) { const uint16_t b = (uint16_t)(w & 0xffffu); _Float16 hh; __builtin_memcpy(&hh, &b, 2); return (float)hh; } return __uint_as_float(w << 16); };
                auto hi16 = [](uint32_t w) -> float { if (F16) { const uint16_t b = (uint16_t)(w >> 16); _Float16 hh; __builtin_memcpy(&hh, &b, 2); return (float)hh; } return __uint_as_float(w & 0xffff0000u); };
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
                const int32_t *nn = nn_idx + r * knn;
                const uint32_t row_bytes = (uint32_t)d * 2u, lane_off = (uint32_t)c0 * 2u;
                // style rows in batches of 10 gathers IN FLIGHT (the compiler otherwise waits for each load before issuing the next: the kernel's
                // time is the latency of these gathers).  Buffer loads: the style set is ONE resource (base, M * d * 2 bytes: < 4 GB is checked on the host), a row's byte
                // offset rides in the instruction's SCALAR offset (the neighbour indices of a row are wave-uniform scalar loads), the lane's 8 bytes in its VGPR offset
                // — no vector address arithmetic and one SGPR per gather in flight.
                // An unfilled list slot (the kNN kernel's sentinel index: a NaN / Inf feature row collects no candidates, and rows beyond the repair
                // capacity stay short) names no style row: row 0 is read instead and the row's gradient is poisoned with NaN — what the reference's
                // cdist -> mask -> mean propagates.
                uint32_t bad = 0u;
                // pipe: the row's list sits in the lanes (cur_nnv): out-of-range entries are found and clamped ONCE per row on the vector, a gather's scalar offset is one
                // v_readlane + one s_mul (the list as k scalar loads cost an s_load + full lgkmcnt wait at the head of every row, and 6 scalar + 2 vector instructions
                // per gather for the range check)
                if (pipe) bad = row_bad;
                auto yrow = [&](int jj) -> uint32_t {
                    if (pipe) return (uint32_t)__builtin_amdgcn_readlane(nnc, jj) * row_bytes;
                    const uint32_t y = (uint32_t)nn[jj]; const bool ok = y < M; bad |= ok ? 0u : 1u; return (ok ? y : 0u) * row_bytes;
                };
                auto ldrow = [&](uint32_t soff) -> uint2 {
                    typedef int i32x2_t __attribute__((ext_vector_type(2)));
                    const i32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(fy_rsrc, (int)lane_off, (int)soff, 0);
                    return make_uint2((uint32_t)v[0], (uint32_t)v[1]);
                };
                // (v_dot2c_f32_bf16 with a (1, 0) / (0, 1) mask adds a half per instruction instead of shift + add; measured: no faster — the kernel waits for its gathers, not
                // for the VALU — and its sums round differently from plain adds, so the unpack-and-add form, bit-equal to w3d_knn_match_bwd, stays)
                auto acc4 = [&](const uint2 v) { a0 += lo16(v.x); a1 += hi16(v.x); a2 += lo16(v.y); a3 += hi16(v.y); };
                const bool cached = row_cached;
                int j = cached ? knn : 0;
                if (cached) { a0 = cur_cs.x; a1 = cur_cs.y; a2 = cur_cs.z; a3 = cur_cs.w; }
                for (; j + 5 <= knn; j += 5) {                   // 5 gathers in flight (with the row pipeline's registers the 128 VGPRs a 1024-thread block may use are full; the
                    uint2 pv[5];                                 // compiler overlaps consecutive batches as far as the registers reach)
#pragma unroll
                    for (int u = 0; u < 5; ++u) pv[u] = ldrow(yrow(j + u));
#pragma unroll
                    for (int u = 0; u < 5; ++u) acc4(pv[u]);
                }
                for (; j < knn; ++j) acc4(ldrow(yrow(j)));
                if (bad) a0 = a1 = a2 = a3 = __builtin_nanf("");
                if (use_cache && !cached) {                       // refresh the row's cache entry (a NaN sum with its sentinel list is consistent: the same list hits again)
                    *reinterpret_cast<float4 *>(sum_cache + r * d + c0) = make_float4(a0, a1, a2, a3);
                }
                const float kf = (float)knn;
                g[0] = scale2 * (kf * lo16(xv.x) - a0); g[1] = scale2 * (kf * hi16(xv.x) - a1);
                g[2] = scale2 * (kf * lo16(xv.y) - a2); g[3] = scale2 * (kf * hi16(xv.y) - a3);
                if (gF_out) *reinterpret_cast<float4 *>(gF_out + r * d + c0) = make_float4(g[0], g[1], g[2], g[3]);
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) if (c < nc) g[c] = gF[r * d + c0 + c];
            }
            float4 q[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (stage_pos && ls[c] != 0xffffu) { const float *pp = pts + 3 * ls[c]; q[c] = make_float4(pp[0], pp[1], pp[2], 0.f); }
                else q[c] = X4[nb[c]];
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (c < nc) {
                    const float vx = q[c].x - rp.x, vy = q[c].y - rp.y, vz = q[c].z - rp.z;
                    const float d2 = vx * vx + vy * vy + vz * vz;
                    const float w = d2 > 0.f ? g[c] * __builtin_amdgcn_rsqf(d2) : 0.f;   // g / |v| (torch.norm backward: 0 at 0); v_rsq_f32: 1 ulp
                    const float gx = w * vx, gy = w * vy, gz = w * vz;
                    add(nb[c], gx, gy, gz, ls[c]);
                    sx += gx; sy += gy; sz += gz;
                }
            }
        }
        if (use_cache && !row_cached && lane < knn) sum_idx[r * knn + lane] = cur_nnv;      // (all lanes: the list's lanes need not be feature lanes)
        // the root's share is a sum of up to d contributions (outside the fan-in bound): one global float atomic per row and axis
        sx = wave_sum(sx); sy = wave_sum(sy); sz = wave_sum(sz);
        if (lane == 0) { float *p = gX + 3 * root; unsafeAtomicAdd(p, -sx); unsafeAtomicAdd(p + 1, -sy); unsafeAtomicAdd(p + 2, -sz); }
    }
    __syncthreads();
    if (FUSED && nn_dist2) {                                         // one float atomic per block
        lsum = wave_sum(lsum);
        if (lane == 0) red[wid] = lsum;
        __syncthreads();
        if (tid == 0) { float t = 0.f; for (int w = 0; w < WB_WAVES; ++w) t += red[w]; atomicAdd(loss_accum, loss_scale * t); }
    }
    if (nbr_slot) {                                                  // the window is a set of points: each goes to its own place
        for (int i = tid; i < win * 3; i += WB_THREADS) {
            const int v = acc[i];
            if (v != 0) unsafeAtomicAdd(gX + 3 * (int64_t)dest[doff + i / 3] + i % 3, (float)v * invS);
        }
        return;
    }
    const int64_t lim3 = min((int64_t)win, n_points - base) * 3;
    float *out = gX + 3 * base;
    for (int i = tid; i < lim3; i += WB_THREADS) {
        const int v = acc[i];
        if (v != 0) unsafeAtomicAdd(out + i, (float)v * invS);
    }
}

// a3 forward with the same window idea: the block's 12288-point window of positions (12 B each, 144 KB) is staged in LDS once, and the K-1
// neighbour fetches of every row become LDS reads instead of 16-B L2 gathers (257 M of them per iteration at C3); entries outside the window
// fall back to the global gather, so any list is handled.  HBM traffic is then the streams only: idx (4 B per entry), the 16-bit feature rows
// (written; read once more for the row motion).  Also produces |F_new - F_old|^2 and |F_new|^2 of the rounded rows (both optional).
template <bool F16>
__global__ __launch_bounds__(WB_THREADS) void descriptor_fwd_window_kernel(const float4 *__restrict__ X4, int64_t n_points, const int32_t *__restrict__ idx, int64_t R,
                                                                           int K, uint16_t *__restrict__ F, float *__restrict__ motion2, float *__restrict__ xnorm,
                                                                           int rows_per_block, int win, const float *__restrict__ center,
                                                                           const uint16_t *__restrict__ nbr_slot, const int32_t *__restrict__ dest,
                                                                           const int32_t *__restrict__ dest_off, const int32_t *__restrict__ root_idx) {
    // root_idx (optional, with the slot plan when every entry has a slot): the rows' roots as a dense array — the index rows are not read (half the kernel's bytes)
    // nbr_slot / dest / dest_off (optional): the slot plan of the backward (see descriptor_bwd_window_kernel) — the block stages the positions of the SET of points
    // its rows name and every neighbour fetch is an LDS read at the entry's slot (0xffff: a global gather); the root's position is one global load per row.
    extern __shared__ __attribute__((aligned(16))) float pts[];     // [win][3]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = min(R, r0 + (int64_t)rows_per_block);
    int64_t base = 0;
    if (nbr_slot) {
        const int64_t doff = dest_off[blockIdx.x];
        win = dest_off[blockIdx.x + 1] - (int)doff;
        for (int i = tid; i < win; i += WB_THREADS) { const float4 q = X4[dest[doff + i]]; pts[3 * i] = q.x; pts[3 * i + 1] = q.y; pts[3 * i + 2] = q.z; }
    } else {
        const int64_t centre = ((int64_t)idx[r0 * K] + (int64_t)idx[(r1 - 1) * K]) >> 1;
        base = centre - win / 2;
        if (base > n_points - win) base = n_points - win;
        if (base < 0) base = 0;
        for (int i = tid; i < win; i += WB_THREADS) {
            const int64_t p = base + i;
            if (p < n_points) { const float4 q = X4[p]; pts[3 * i] = q.x; pts[3 * i + 1] = q.y; pts[3 * i + 2] = q.z; }
        }
    }
    __syncthreads();
    auto fetch = [&](int64_t n, float &x, float &y, float &z, uint32_t lslot = 0xffffu) {
        const int64_t off = nbr_slot ? (lslot == 0xffffu ? -1 : (int64_t)lslot) : n - base;
        if ((uint64_t)off < (uint64_t)win) { const float *p = pts + 3 * off; x = p[0]; y = p[1]; z = p[2]; }
        else { const float4 q = X4[n]; x = q.x; y = q.y; z = q.z; }
    };
    auto enc = [](float v) -> uint16_t { if (F16) { const _Float16 h = (_Float16)v; uint16_t b; __builtin_memcpy(&b, &h, 2); return b; } return f32_to_bf16(v); };
    auto dec = [](uint16_t b) -> float { if (F16) { _Float16 h; __builtin_memcpy(&h, &b, 2); return (float)h; } return bf16_to_f32(b); };
    const int d = K - 1;
    const bool vec = (d & 3) == 0;                                   // 8-B aligned groups of four features
    if (vec && d <= 256) {
        // one group of four features per lane: the index quad, the root and the old feature quad of the NEXT row are loaded while the current
        // row is computed (a wave walks its rows one after the other: without this every row pays the full latency of its index loads)
        const int c0 = lane * 4;
        const bool on = c0 < d;
        // centre of the feature rows (optional, [d]): subtracted before the 16-bit rounding (distances between rows are unchanged, their
        // rounding error is not); the lane's four values are loaded once
        float mu[4] = {0.f, 0.f, 0.f, 0.f};
        if (center && on) { const float4 m = *reinterpret_cast<const float4 *>(center + c0); mu[0] = m.x; mu[1] = m.y; mu[2] = m.z; mu[3] = m.w; }
        if (nbr_slot && root_idx) {
            // The loop's configuration (slot plan with a slot for every entry, roots as a dense array; round 4: with or without the row motion — the cached lower bounds
            // of the pruned sweeps need it every iteration, and the old feature quad is one more unconditional load a row ahead) as a straight-line software pipeline: the root INDEX
            // is loaded two rows ahead, the root's POSITION and the row's slot quad one row ahead, unconditionally (indices clamped, no exec branches), so that the
            // compiler's waits are counted ones.  The general form below issues the next row's loads inside divergent branches; every branch with a global load in it
            // (the root's position is one per row) then drains vmcnt(0) — the prefetch just issued included — and a row paid a full memory latency
            // (SQ_WAIT_ANY 68 % of the wave cycles at C3).  Entries without a slot (0xffff: none in the loop's plans) take the index row + a global gather, out of line.
            // Same arithmetic, same order: bit-identical rows and norms.
            const int c0c = on ? c0 : 0;
            auto ld_root = [&](int64_t rr) -> int32_t { return root_idx[rr < r1 ? rr : r1 - 1]; };
            auto ld_slots = [&](int64_t rr) -> uint2 { return *reinterpret_cast<const uint2 *>(nbr_slot + (rr < r1 ? rr : r1 - 1) * d + c0c); };
            auto ld_old = [&](int64_t rr) -> uint2 { return *reinterpret_cast<const uint2 *>(F + (rr < r1 ? rr : r1 - 1) * d + c0c); };   // (a later row: not yet rewritten)
            int64_t r = r0 + wid;
            if (r >= r1) return;
            int32_t root_b = ld_root(r), root_a = ld_root(r + WB_WAVES);
            float4 pos_b = X4[root_b];
            uint2 sq_b = ld_slots(r);
            uint2 old_b = make_uint2(0u, 0u);
            if (motion2) old_b = ld_old(r);
            for (; r < r1; r += WB_WAVES) {
                const int32_t root_n2 = ld_root(r + 2 * WB_WAVES);
                const float4 pos_n = X4[root_a];
                const uint2 sq_n = ld_slots(r + WB_WAVES);
                uint2 old_n = make_uint2(0u, 0u);
                if (motion2) old_n = ld_old(r + WB_WAVES);
                const uint32_t ls[4] = {sq_b.x & 0xffffu, sq_b.x >> 16, sq_b.y & 0xffffu, sq_b.y >> 16};
                const float rx = pos_b.x, ry = pos_b.y, rz = pos_b.z;
                float q[4][3];
                bool miss = false;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const uint32_t sl = ls[c] < (uint32_t)win ? ls[c] : 0u;       // (an entry without a slot reads slot 0 here and is redone below)
                    miss |= ls[c] >= (uint32_t)win;
                    const float *pp = pts + 3 * sl;
                    q[c][0] = pp[0]; q[c][1] = pp[1]; q[c][2] = pp[2];
                }
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(miss && on) != 0ull, 0)) {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (on && ls[c] >= (uint32_t)win) {
                            // the gathered values pass through a register move INSIDE this branch: the wait for the gather then sits here, not after the merge (where
                            // the waitcnt pass would otherwise drain vmcnt(0) on every row — the prefetches just issued included — for registers this path may have loaded)
                            const float4 g = X4[idx[r * K + c0 + c + 1]];
                            asm volatile("v_mov_b32 %0, %1" : "=v"(q[c][0]) : "v"(g.x));
                            asm volatile("v_mov_b32 %0, %1" : "=v"(q[c][1]) : "v"(g.y));
                            asm volatile("v_mov_b32 %0, %1" : "=v"(q[c][2]) : "v"(g.z));
                        }
                }
                float n2 = 0.f, m2 = 0.f;
                if (on) {
                    uint16_t nb[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float vx = q[c][0] - rx, vy = q[c][1] - ry, vz = q[c][2] - rz;
                        nb[c] = enc(sqrtf(vx * vx + vy * vy + vz * vz) - mu[c]);
                    }
                    const uint16_t ob[4] = {(uint16_t)(old_b.x & 0xffffu), (uint16_t)(old_b.x >> 16), (uint16_t)(old_b.y & 0xffffu), (uint16_t)(old_b.y >> 16)};
#pragma unroll
                    for (int c = 0; c < 4; ++c) { const float fn = dec(nb[c]); n2 = fmaf(fn, fn, n2); const float dd = fn - dec(ob[c]); m2 = fmaf(dd, dd, m2); }
                    *reinterpret_cast<uint2 *>(F + r * d + c0) = make_uint2((uint32_t)nb[0] | ((uint32_t)nb[1] << 16), (uint32_t)nb[2] | ((uint32_t)nb[3] << 16));
                }
                if (xnorm) {
                    n2 = wave_sum(n2);
                    if (lane == 0) xnorm[r] = n2;
                }
                if (motion2) {
                    m2 = wave_sum(m2);
                    if (lane == 0) motion2[r] = m2;
                }
                root_b = root_a; root_a = root_n2; pos_b = pos_n; sq_b = sq_n; old_b = old_n;
            }
            return;
        }
        auto load_row = [&](int64_t r, int32_t (&nbi)[4], int32_t &root, uint2 &oldq, uint2 &sq) {
            const int32_t *row = idx + r * K;
            root = root_idx ? root_idx[r] : row[0];
            if (on) {
                if (!root_idx) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) nbi[c] = row[c0 + c + 1];
                }
                if (motion2) oldq = *reinterpret_cast<const uint2 *>(F + r * d + c0);
                if (nbr_slot) sq = *reinterpret_cast<const uint2 *>(nbr_slot + r * d + c0);
            }
        };
        int32_t cur_i[4] = {0, 0, 0, 0}, nxt_i[4] = {0, 0, 0, 0}, cur_root = 0, nxt_root = 0;
        uint2 cur_old = make_uint2(0u, 0u), nxt_old = make_uint2(0u, 0u), cur_sq = make_uint2(0xffffffffu, 0xffffffffu), nxt_sq = cur_sq;
        int64_t r = r0 + wid;
        if (r < r1) load_row(r, cur_i, cur_root, cur_old, cur_sq);
        for (; r < r1; r += WB_WAVES) {
            const int64_t rn = r + WB_WAVES;
            if (rn < r1) load_row(rn, nxt_i, nxt_root, nxt_old, nxt_sq);
            const uint32_t ls[4] = {cur_sq.x & 0xffffu, cur_sq.x >> 16, cur_sq.y & 0xffffu, cur_sq.y >> 16};
            float rx, ry, rz;
            fetch(cur_root, rx, ry, rz);
            float m2 = 0.f, n2 = 0.f;
            if (on) {
                uint16_t nb[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float qx, qy, qz;
                    fetch(cur_i[c], qx, qy, qz, ls[c]);
                    const float vx = qx - rx, vy = qy - ry, vz = qz - rz;
                    nb[c] = enc(sqrtf(vx * vx + vy * vy + vz * vz) - mu[c]);
                }
                const uint16_t old[4] = {(uint16_t)(cur_old.x & 0xffffu), (uint16_t)(cur_old.x >> 16), (uint16_t)(cur_old.y & 0xffffu), (uint16_t)(cur_old.y >> 16)};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float fn = dec(nb[c]);
                    const float dd = fn - dec(old[c]);
                    m2 = fmaf(dd, dd, m2);
                    n2 = fmaf(fn, fn, n2);
                }
                *reinterpret_cast<uint2 *>(F + r * d + c0) = make_uint2((uint32_t)nb[0] | ((uint32_t)nb[1] << 16), (uint32_t)nb[2] | ((uint32_t)nb[3] << 16));
            }
            if (motion2 || xnorm) {
                m2 = wave_sum(m2); n2 = wave_sum(n2);
                if (lane == 0) { if (motion2) motion2[r] = m2; if (xnorm) xnorm[r] = n2; }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) cur_i[c] = nxt_i[c];
            cur_root = nxt_root; cur_old = nxt_old; cur_sq = nxt_sq;
        }
        return;
    }
    for (int64_t r = r0 + wid; r < r1; r += WB_WAVES) {
        const int32_t *row = idx + r * K;
        float rx, ry, rz;
        fetch(row[0], rx, ry, rz);
        float m2 = 0.f, n2 = 0.f;
        for (int c0 = lane * 4; c0 < d; c0 += 256) {
            const int nc = min(4, d - c0);
            uint16_t nb[4] = {0, 0, 0, 0};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (c < nc) {
                    float qx, qy, qz;
                    fetch(row[c0 + c + 1], qx, qy, qz);
                    const float vx = qx - rx, vy = qy - ry, vz = qz - rz;
                    nb[c] = enc(sqrtf(vx * vx + vy * vy + vz * vz) - (center ? center[c0 + c] : 0.f));
                }
            }
            uint16_t *f = F + r * d + c0;
            uint16_t old[4] = {0, 0, 0, 0};
            if (motion2) {
                if (vec) { const uint2 o = *reinterpret_cast<const uint2 *>(f); old[0] = (uint16_t)(o.x & 0xffffu); old[1] = (uint16_t)(o.x >> 16); old[2] = (uint16_t)(o.y & 0xffffu); old[3] = (uint16_t)(o.y >> 16); }
                else { for (int c = 0; c < nc; ++c) old[c] = f[c]; }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (c < nc) {
                    const float fn = dec(nb[c]);
                    const float dd = fn - dec(old[c]);
                    m2 = fmaf(dd, dd, m2);
                    n2 = fmaf(fn, fn, n2);
                }
            }
            if (vec) *reinterpret_cast<uint2 *>(f) = make_uint2((uint32_t)nb[0] | ((uint32_t)nb[1] << 16), (uint32_t)nb[2] | ((uint32_t)nb[3] << 16));
            else { for (int c = 0; c < nc; ++c) f[c] = nb[c]; }
        }
        if (motion2 || xnorm) {
            m2 = wave_sum(m2); n2 = wave_sum(n2);
            if (lane == 0) { if (motion2) motion2[r] = m2; if (xnorm) xnorm[r] = n2; }
        }
    }
}

// fan-in of a neighbour list: counts[p] = number of entries in columns >= 1 naming p; *out_max = max_p counts[p]
__global__ __launch_bounds__(256) void fanin_count_kernel(const int32_t *__restrict__ idx, int64_t R, int K, int64_t n_points, int32_t *__restrict__ counts) {
    const int64_t nnz = R * K;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < nnz; e += (int64_t)gridDim.x * 256) {
        if (e % K == 0) continue;
        const int64_t p = idx[e];
        if (p >= 0 && p < n_points) atomicAdd(&counts[p], 1);
    }
}
__global__ __launch_bounds__(256) void int_max_kernel(const int32_t *__restrict__ v, int64_t n, int32_t *__restrict__ out) {
    int m = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) m = max(m, v[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}
template <bool F16>
__global__ __launch_bounds__(256) void absmax_bf16_kernel(const uint16_t *__restrict__ v, int64_t n, uint32_t *__restrict__ out_bits) {
    uint32_t m = 0u;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const uint16_t a = v[i] & 0x7fffu;
        if (F16) { _Float16 hh; __builtin_memcpy(&hh, &a, 2); m = max(m, __float_as_uint((float)hh)); } else m = max(m, ((uint32_t)a) << 16);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(out_bits, m);       // |bf16| as f32 bits: non-negative floats order like unsigned integers
}

struct WinPlan { int win, rows; unsigned blocks; };

WinPlan plan(int64_t R, int64_t n_points, int window_points) {
    WinPlan p;
    int64_t win = window_points > 0 ? window_points : WB_DEFAULT_WIN;
    if (win > WB_MAX_WIN) win = WB_MAX_WIN;
    if (win > n_points) win = n_points;
    if (win < 1) win = 1;
    // rows per block: a third of the window at most (the rest is halo), at least 256, and ~one block per CU on large inputs
    int64_t rows = (R + 255) / 256;
    if (rows > win / 3) rows = win / 3;
    if (rows < 256) rows = 256;
    p.win = (int)win; p.rows = (int)rows; p.blocks = (unsigned)((R + rows - 1) / rows);
    return p;
}

int pow2_at_least(int v) { int p = 1; while (p < v && p < (1 << 24)) p <<= 1; return p; }

template <bool FUSED, bool F16>
int launch(const float *X4, const int32_t *idx, int64_t R, int K, const float *gF, const uint16_t *FX, const uint16_t *FY, const int32_t *nn, int knn, float scale2,
           float fy_absmax, float *gF_out, int64_t n_points, int window_points, int max_fanin, float *gX, hipStream_t st, uint32_t M = 0, const float *nn_dist2 = nullptr,
           float loss_scale = 0.f, float *loss_accum = nullptr, const uint16_t *nbr_slot = nullptr, const int32_t *dest = nullptr, const int32_t *dest_off = nullptr, int slot_rows = 0, int slot_max_set = 0,
           const int32_t *root_idx = nullptr, float *sum_cache = nullptr, int32_t *sum_idx = nullptr) {
    WinPlan p = plan(R, n_points, window_points);
    if (nbr_slot) {                                                  // slot plan: the caller's row blocks and the largest destination set among them
        p.rows = slot_rows; p.win = slot_max_set; p.blocks = (unsigned)((R + slot_rows - 1) / slot_rows);
    }
    const bool stage_pos = FUSED && nbr_slot && (size_t)p.win * 24 + 256 <= 160 * 1024;
    const size_t lds = (size_t)p.win * (stage_pos ? 6 : 3) * sizeof(int);
    auto kern = stage_pos ? &descriptor_bwd_window_kernel<FUSED, F16, FUSED> : &descriptor_bwd_window_kernel<FUSED, F16, false>;
    W3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(p.blocks), dim3(WB_THREADS), lds, st, reinterpret_cast<const float4 *>(X4), idx, R, K, gF, FX, FY, M, nn, knn, scale2, fy_absmax, gF_out,
                       p.rows, p.win, pow2_at_least(max_fanin), n_points, gX, nn_dist2, loss_scale, loss_accum, nbr_slot, dest, dest_off, stage_pos ? root_idx : nullptr,
                       sum_cache, sum_idx);
    W3D_CHECK_LAUNCH();
    return W3D_OK;
}

}  // namespace

extern "C" int w3d_list_max_fanin(const int32_t *idx, int64_t R, int K, int64_t n_points, int32_t *counts_scratch, int32_t *out_max, w3d_stream_t stream) {
    if (!idx || !counts_scratch || !out_max || R < 0 || K < 1 || n_points < 1) return W3D_ERR_ARG;
    W3D_HIP(hipMemsetAsync(counts_scratch, 0, (size_t)n_points * 4, w3d_s(stream)));
    W3D_HIP(hipMemsetAsync(out_max, 0, 4, w3d_s(stream)));
    if (R > 0 && K > 1) hipLaunchKernelGGL(fanin_count_kernel, dim3(w3d_grid(R * K, 256, 8192)), dim3(256), 0, w3d_s(stream), idx, R, K, n_points, counts_scratch);
    hipLaunchKernelGGL(int_max_kernel, dim3(w3d_grid(n_points, 256, 1024)), dim3(256), 0, w3d_s(stream), counts_scratch, n_points, out_max);
    W3D_CHECK_LAUNCH();
    return W3D_OK;
}

extern "C" int w3d_absmax_16(const void *v, int dtype, int64_t n, float *out, w3d_stream_t stream) {
    if (!v || !out || n < 0 || (dtype != W3D_DTYPE_BF16 && dtype != W3D_DTYPE_F16)) return W3D_ERR_ARG;
    W3D_HIP(hipMemsetAsync(out, 0, 4, w3d_s(stream)));
    if (n > 0 && dtype == W3D_DTYPE_BF16)
        hipLaunchKernelGGL(absmax_bf16_kernel<false>, dim3(w3d_grid(n, 256, 2048)), dim3(256), 0, w3d_s(stream), reinterpret_cast<const uint16_t *>(v), n, reinterpret_cast<uint32_t *>(out));
    if (n > 0 && dtype == W3D_DTYPE_F16)
        hipLaunchKernelGGL(absmax_bf16_kernel<true>, dim3(w3d_grid(n, 256, 2048)), dim3(256), 0, w3d_s(stream), reinterpret_cast<const uint16_t *>(v), n, reinterpret_cast<uint32_t *>(out));
    W3D_CHECK_LAUNCH();
    return W3D_OK;
}

extern "C" int w3d_descriptor_bwd_window_x4(const float *X4, const int32_t *idx, int64_t R, int K, const float *gF, int64_t n_points, int window_points, int max_fanin,
                                            float *gX, w3d_stream_t stream) {
    if (!X4 || !idx || !gF || !gX || R < 0 || K < 2 || n_points < 1 || max_fanin < 1) return W3D_ERR_ARG;
    if (((uintptr_t)X4) & 15) return W3D_ERR_ARG;
    if (R == 0) return W3D_OK;
    return launch<false, false>(X4, idx, R, K, gF, nullptr, nullptr, nullptr, 0, 0.f, 0.f, nullptr, n_points, window_points, max_fanin, gX, w3d_s(stream));
}

extern "C" int w3d_knn_descriptor_bwd_fused_x4(const float *X4, const int32_t *desc_idx, int64_t R, int K, int dtype, const void *FX_bf16, const void *FY_bf16, int64_t M,
                                               float fy_absmax, const int32_t *nn_idx, int k, float scale, int64_t n_points, int window_points, int max_fanin,
                                               float *gF_out, float *gX, const float *nn_dist2, float *loss_accum, w3d_stream_t stream) {
    if (!X4 || !desc_idx || !FX_bf16 || !FY_bf16 || !nn_idx || !gX || R < 0 || K < 2 || k < 1 || n_points < 1 || M < 1 || max_fanin < 1 || !(fy_absmax >= 0.f))
        return W3D_ERR_ARG;
    if ((((uintptr_t)X4) & 15) || (((uintptr_t)FX_bf16 | (uintptr_t)FY_bf16) & 15) || (gF_out && (((uintptr_t)gF_out) & 15))) return W3D_ERR_ARG;
    if ((K - 1) % 8 || (uint64_t)M * (uint64_t)(K - 1) * 2ull >= (1ull << 32)) return W3D_ERR_UNSUPPORTED;   // 16-B feature loads; 32-bit byte offsets into the style rows
    if ((nn_dist2 == nullptr) != (loss_accum == nullptr)) return W3D_ERR_ARG;
    if (R == 0) return W3D_OK;
    if (dtype == W3D_DTYPE_F16)
        return launch<true, true>(X4, desc_idx, R, K, nullptr, reinterpret_cast<const uint16_t *>(FX_bf16), reinterpret_cast<const uint16_t *>(FY_bf16), nn_idx, k,
                                  2.0f * scale, fy_absmax, gF_out, n_points, window_points, max_fanin, gX, w3d_s(stream), (uint32_t)M, nn_dist2, scale, loss_accum);
    if (dtype != W3D_DTYPE_BF16) return W3D_ERR_ARG;
    return launch<true, false>(X4, desc_idx, R, K, nullptr, reinterpret_cast<const uint16_t *>(FX_bf16), reinterpret_cast<const uint16_t *>(FY_bf16), nn_idx, k,
                               2.0f * scale, fy_absmax, gF_out, n_points, window_points, max_fanin, gX, w3d_s(stream), (uint32_t)M, nn_dist2, scale, loss_accum);
}

// The fused backward over a SLOT PLAN of the frozen lists (see the kernel): row blocks of `rows_per_block` rows, block b's destination set dest[dest_off[b]..dest_off[b+1])
// (at most max_set <= 13312 points), nbr_slot [R][K-1] = each entry's place in its block's set (0xffff: outside -> global atomic).
extern "C" int w3d_knn_descriptor_bwd_fused_slots(const float *X4, const int32_t *desc_idx, int64_t R, int K, int dtype, const void *FX_bf16, const void *FY_bf16, int64_t M,
                                                  float fy_absmax, const int32_t *nn_idx, int k, float scale, int64_t n_points, int max_fanin, float *gX,
                                                  const float *nn_dist2, float *loss_accum, int rows_per_block, int max_set, const uint16_t *nbr_slot,
                                                  const int32_t *dest, const int32_t *dest_off, const int32_t *root_idx, w3d_stream_t stream) {
    if (!X4 || !desc_idx || !FX_bf16 || !FY_bf16 || !nn_idx || !gX || !nbr_slot || !dest || !dest_off || R < 0 || K < 2 || k < 1 || n_points < 1 || M < 1 || max_fanin < 1 ||
        !(fy_absmax >= 0.f) || rows_per_block < 1 || max_set < 1)
        return W3D_ERR_ARG;
    if ((((uintptr_t)X4) & 15) || (((uintptr_t)FX_bf16 | (uintptr_t)FY_bf16) & 15) || (((uintptr_t)nbr_slot) & 7)) return W3D_ERR_ARG;
    if ((K - 1) % 8 || K - 1 > 256 || (uint64_t)M * (uint64_t)(K - 1) * 2ull >= (1ull << 32) || max_set > WB_MAX_WIN) return W3D_ERR_UNSUPPORTED;
    if ((nn_dist2 == nullptr) != (loss_accum == nullptr)) return W3D_ERR_ARG;
    if (R == 0) return W3D_OK;
    if (dtype == W3D_DTYPE_F16)
        return launch<true, true>(X4, desc_idx, R, K, nullptr, reinterpret_cast<const uint16_t *>(FX_bf16), reinterpret_cast<const uint16_t *>(FY_bf16), nn_idx, k,
                                  2.0f * scale, fy_absmax, nullptr, n_points, 0, max_fanin, gX, w3d_s(stream), (uint32_t)M, nn_dist2, scale, loss_accum,
                                  nbr_slot, dest, dest_off, rows_per_block, max_set, root_idx);
    if (dtype != W3D_DTYPE_BF16) return W3D_ERR_ARG;
    return launch<true, false>(X4, desc_idx, R, K, nullptr, reinterpret_cast<const uint16_t *>(FX_bf16), reinterpret_cast<const uint16_t *>(FY_bf16), nn_idx, k,
                               2.0f * scale, fy_absmax, nullptr, n_points, 0, max_fanin, gX, w3d_s(stream), (uint32_t)M, nn_dist2, scale, loss_accum,
                               nbr_slot, dest, dest_off, rows_per_block, max_set, root_idx);
}

// ... with the NEIGHBOUR-SUM CACHE of a loop (see the kernel): sum_cache [R][K-1] f32 and sum_idx [R][k] int32 are the caller's, kept from call to call; sum_idx filled
// with -1 before the first call and whenever the style rows change.  Rows whose k neighbours are the same set as at the call that filled their cache entry read the
// sum of their style rows from it instead of gathering them.
extern "C" int w3d_knn_descriptor_bwd_fused_slots_cached(const float *X4, const int32_t *desc_idx, int64_t R, int K, int dtype, const void *FX_bf16, const void *FY_bf16,
                                                         int64_t M, float fy_absmax, const int32_t *nn_idx, int k, float scale, int64_t n_points, int max_fanin, float *gX,
                                                         const float *nn_dist2, float *loss_accum, int rows_per_block, int max_set, const uint16_t *nbr_slot,
                                                         const int32_t *dest, const int32_t *dest_off, const int32_t *root_idx, float *sum_cache, int32_t *sum_idx,
                                                         w3d_stream_t stream) {
    if (!X4 || !desc_idx || !FX_bf16 || !FY_bf16 || !nn_idx || !gX || !nbr_slot || !dest || !dest_off || R < 0 || K < 2 || k < 1 || n_points < 1 || M < 1 || max_fanin < 1 ||
        !(fy_absmax >= 0.f) || rows_per_block < 1 || max_set < 1 || !sum_cache || !sum_idx)
        return W3D_ERR_ARG;
    if ((((uintptr_t)X4) & 15) || (((uintptr_t)FX_bf16 | (uintptr_t)FY_bf16 | (uintptr_t)sum_cache) & 15) || (((uintptr_t)nbr_slot) & 7)) return W3D_ERR_ARG;
    if ((K - 1) % 8 || K - 1 > 256 || k > 64 || (uint64_t)M * (uint64_t)(K - 1) * 2ull >= (1ull << 32) || max_set > WB_MAX_WIN) return W3D_ERR_UNSUPPORTED;
    if ((nn_dist2 == nullptr) != (loss_accum == nullptr)) return W3D_ERR_ARG;
    if (R == 0) return W3D_OK;
    if (dtype == W3D_DTYPE_F16)
        return launch<true, true>(X4, desc_idx, R, K, nullptr, reinterpret_cast<const uint16_t *>(FX_bf16), reinterpret_cast<const uint16_t *>(FY_bf16), nn_idx, k,
                                  2.0f * scale, fy_absmax, nullptr, n_points, 0, max_fanin, gX, w3d_s(stream), (uint32_t)M, nn_dist2, scale, loss_accum,
                                  nbr_slot, dest, dest_off, rows_per_block, max_set, root_idx, sum_cache, sum_idx);
    if (dtype != W3D_DTYPE_BF16) return W3D_ERR_ARG;
    return launch<true, false>(X4, desc_idx, R, K, nullptr, reinterpret_cast<const uint16_t *>(FX_bf16), reinterpret_cast<const uint16_t *>(FY_bf16), nn_idx, k,
                               2.0f * scale, fy_absmax, nullptr, n_points, 0, max_fanin, gX, w3d_s(stream), (uint32_t)M, nn_dist2, scale, loss_accum,
                               nbr_slot, dest, dest_off, rows_per_block, max_set, root_idx, sum_cache, sum_idx);
}

static int fwd_window_launch(const float *X4, int64_t n_points, const int32_t *idx, int64_t R, int K, int dtype, void *F, float *motion2, float *xnorm, int window_points,
                             const float *center, int slot_rows, int slot_max_set, const uint16_t *nbr_slot, const int32_t *dest, const int32_t *dest_off, const int32_t *root_idx,
                             w3d_stream_t stream);

extern "C" int w3d_descriptor_fwd_window_x4(const float *X4, int64_t n_points, const int32_t *idx, int64_t R, int K, int dtype, void *F, float *motion2, float *xnorm,
                                            int window_points, const float *center, w3d_stream_t stream) {
    return fwd_window_launch(X4, n_points, idx, R, K, dtype, F, motion2, xnorm, window_points, center, 0, 0, nullptr, nullptr, nullptr, nullptr, stream);
}

// the forward over the backward's slot plan (w3d_knn_descriptor_bwd_fused_slots): every neighbour position comes from LDS
extern "C" int w3d_descriptor_fwd_slots_x4(const float *X4, int64_t n_points, const int32_t *idx, int64_t R, int K, int dtype, void *F, float *motion2, float *xnorm,
                                           const float *center, int rows_per_block, int max_set, const uint16_t *nbr_slot, const int32_t *dest, const int32_t *dest_off,
                                           const int32_t *root_idx, w3d_stream_t stream) {
    if (!nbr_slot || !dest || !dest_off || rows_per_block < 1 || max_set < 1 || max_set > WB_MAX_WIN || (K - 1) % 4 || K - 1 > 256 || (((uintptr_t)nbr_slot) & 7)) return W3D_ERR_ARG;
    return fwd_window_launch(X4, n_points, idx, R, K, dtype, F, motion2, xnorm, 0, center, rows_per_block, max_set, nbr_slot, dest, dest_off, root_idx, stream);
}

static int fwd_window_launch(const float *X4, int64_t n_points, const int32_t *idx, int64_t R, int K, int dtype, void *F, float *motion2, float *xnorm, int window_points,
                             const float *center, int slot_rows, int slot_max_set, const uint16_t *nbr_slot, const int32_t *dest, const int32_t *dest_off, const int32_t *root_idx,
                             w3d_stream_t stream) {
    if (!X4 || !idx || !F || R < 0 || K < 2 || n_points < 1 || (((uintptr_t)X4) & 15) || (((uintptr_t)F) & 7) || (((uintptr_t)center) & 15)) return W3D_ERR_ARG;
    if (dtype != W3D_DTYPE_BF16 && dtype != W3D_DTYPE_F16) return W3D_ERR_ARG;
    if (R == 0) return W3D_OK;
    WinPlan p = plan(R, n_points, window_points);
    if (nbr_slot) { p.rows = slot_rows; p.win = slot_max_set; p.blocks = (unsigned)((R + slot_rows - 1) / slot_rows); }
    const size_t lds = (size_t)p.win * 3 * sizeof(float);
    const float4 *X = reinterpret_cast<const float4 *>(X4);
    uint16_t *Fo = reinterpret_cast<uint16_t *>(F);
    if (dtype == W3D_DTYPE_F16) {
        auto kern = &descriptor_fwd_window_kernel<true>;
        W3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(p.blocks), dim3(WB_THREADS), lds, w3d_s(stream), X, n_points, idx, R, K, Fo, motion2, xnorm, p.rows, p.win, center, nbr_slot, dest, dest_off, nbr_slot ? root_idx : nullptr);
    } else {
        auto kern = &descriptor_fwd_window_kernel<false>;
        W3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(p.blocks), dim3(WB_THREADS), lds, w3d_s(stream), X, n_points, idx, R, K, Fo, motion2, xnorm, p.rows, p.win, center, nbr_slot, dest, dest_off, nbr_slot ? root_idx : nullptr);
    }
    W3D_CHECK_LAUNCH();
    return W3D_OK;
}
