// a6/a7  Differentiable tile-based Gaussian rasteriser for gfx950 (depth output + per-pixel sampling offsets).
//
// Pipeline (same stages as CudaRasterizer::Rasterizer::forward/backward, rasterizer_impl.cu:198-446):
//   preprocess (per Gaussian, streaming)  -> inclusive scan of tile counts -> host reads R (one blocking D2H)
//   -> emit (tile|depth, id) pairs -> radix sort (rocPRIM) -> tile ranges -> blend (one 256-thread block per tile,
//   Gaussians staged through LDS in batches of 256).
// Backward: re-traverse back-to-front; every (pixel, Gaussian) contribution is reduced across the 64 lanes of its
// wave with DPP/shuffle adds and leaves the wave as ONE 10-lane atomic instruction into a packed 12-float
// accumulator per Gaussian (the reference issues 10 scalar atomics per pixel per Gaussian, backward.cu:549-583);
// a single per-Gaussian kernel then does the conic/cov2D/cov3D/SH/projection chain (backward.cu:144-410).
#include "raster_math.h"
#include <rocprim/device/device_radix_sort.hpp>
#include <vector>
#include <rocprim/device/device_scan.hpp>

namespace {

constexpr size_t ALIGN = 256;
inline size_t up(size_t n) { return (n + ALIGN - 1) / ALIGN * ALIGN; }

// Splat record (12 floats, 48 B) — the unit that crosses GPUs when Gaussians are sharded:
//   [0] px  [1] py  [2] view depth  [3] radius (int bits; 0 = culled)  [4..6] conic xx,xy,yy  [7] opacity  [8..10] rgb  [11] 0
constexpr int SPLAT = W3D_SPLAT_FLOATS;
struct Geom {
    float *splats; float *cov3D; uint8_t *clamped;
    uint32_t *tiles_touched; uint32_t *point_offsets; void *scan_tmp; size_t scan_tmp_bytes;
};
struct Img { float *final_T; uint32_t *n_contrib; uint2 *ranges; uint32_t *cursor; uint32_t *total; };   // cursor [tiles], total [4]: {R, max tile count, overflow, -}
struct Bin { uint64_t *keys_unsorted, *keys; uint32_t *vals_unsorted, *point_list; void *sort_tmp; size_t sort_tmp_bytes; };

template <typename T> T *take(char *&p, size_t count) { T *r = reinterpret_cast<T *>(p); p += up(count * sizeof(T)); return r; }

size_t scan_tmp_bytes(int P) {
    size_t b = 0;
    rocprim::inclusive_scan(nullptr, b, (uint32_t *)nullptr, (uint32_t *)nullptr, (size_t)P, rocprim::plus<uint32_t>(), (hipStream_t)0);
    return b;
}
size_t sort_tmp_bytes(int64_t R) {
    size_t b = 0;
    rocprim::radix_sort_pairs(nullptr, b, (const uint64_t *)nullptr, (uint64_t *)nullptr, (const uint32_t *)nullptr, (uint32_t *)nullptr, (size_t)R, 0u, 64u,
                              (hipStream_t)0);
    return b;
}
Geom geom_from(void *buf, int P) {
    char *p = reinterpret_cast<char *>(buf);
    Geom g;
    g.splats = take<float>(p, SPLAT * (size_t)P); g.cov3D = take<float>(p, 6 * (size_t)P);
    g.clamped = take<uint8_t>(p, P); g.tiles_touched = take<uint32_t>(p, P); g.point_offsets = take<uint32_t>(p, P);
    g.scan_tmp_bytes = scan_tmp_bytes(P); g.scan_tmp = p;
    return g;
}
Img img_from(void *buf, int H, int W) {
    char *p = reinterpret_cast<char *>(buf);
    Img i;
    const size_t n = (size_t)H * W, tiles = (size_t)((W + TILE_X - 1) / TILE_X) * ((H + TILE_Y - 1) / TILE_Y);
    i.final_T = take<float>(p, n); i.n_contrib = take<uint32_t>(p, n); i.ranges = take<uint2>(p, tiles); i.cursor = take<uint32_t>(p, tiles); i.total = take<uint32_t>(p, 4);
    return i;
}
Bin bin_from(void *buf, int64_t R) {
    char *p = reinterpret_cast<char *>(buf);
    Bin b;
    const size_t n = (size_t)(R > 0 ? R : 1);
    // point_list FIRST, as in the per-tile layout below: the backward reads it without knowing which binning path the forward took
    b.point_list = take<uint32_t>(p, n); b.keys_unsorted = take<uint64_t>(p, n); b.keys = take<uint64_t>(p, n); b.vals_unsorted = take<uint32_t>(p, n);
    b.sort_tmp_bytes = sort_tmp_bytes(n); b.sort_tmp = p;
    return b;
}

// Binning by tile (default): instances are counted per tile, placed by an atomic cursor and sorted PER TILE in LDS on (depth bits, Gaussian index) —
// the order a stable radix sort on (tile, depth) of instances emitted in Gaussian order gives (rasterizer_impl.cu:70-138, 310-332), in 4 launches
// instead of ~22 (rocPRIM scan + six onesweep passes with their fills).  Layout: point_list u32 [n] | keys u64 [n].
struct BinT { uint32_t *point_list; uint64_t *keys; };
BinT bint_from(void *buf, int64_t n_) {
    char *p = reinterpret_cast<char *>(buf);
    const size_t n = (size_t)(n_ > 0 ? n_ : 1);
    BinT b; b.point_list = take<uint32_t>(p, n); b.keys = take<uint64_t>(p, n);
    return b;
}
constexpr int TB_LDS_TILES = 8192;          // tiles of a band the binning kernels can count in LDS (32 / 64 KB tables); larger bands use global atomics
constexpr int TS_LDS_KEYS = 8192;           // keys one tile-sort block holds in LDS (64 KB); larger tiles are sorted in place in global memory

struct CamConst { float view[16]; float proj[16]; float cam[3]; float bg[3]; };

__global__ void load_cam_kernel(const float *view, const float *proj, const float *cam, const float *bg, CamConst *out) {
    const int t = threadIdx.x;
    if (t < 16) { out->view[t] = view[t]; out->proj[t] = proj[t]; }
    if (t < 3) { out->cam[t] = cam ? cam[t] : 0.f; out->bg[t] = bg ? bg[t] : 0.f; }
}

// ------------------------------------------------------------------------------------------------ forward kernels
// Round 6 — STAGE_SH (the loop's layout: DC and higher-order SH coefficients apart, degree-3 storage = SH_REST floats per Gaussian): the block's 256 rows of higher-order
// coefficients (46 KB, contiguous) come in by coalesced 16-B loads into LDS and every thread reads ITS row from there (row stride 45 words: odd, conflict-free).  Read
// straight from global memory a thread walks its 180-B row in 4-B steps: every load instruction of a wave touches 64 different rows (the kernel moved 0.32 GB in
// 204 us).  The arithmetic — and so every output bit — is unchanged.
constexpr int SH_REST = 45;          // (16 - 1) coefficients x 3 channels
template <int NROW>
__device__ __forceinline__ void stage_rows_to_lds(const float *__restrict__ src, int64_t P, float *lds) {
    const int64_t base = (int64_t)blockIdx.x * 256 * NROW, total = P * NROW;
    const int cnt = (int)min((int64_t)256 * NROW, total - base);
    const float4 *s4 = reinterpret_cast<const float4 *>(src + base);             // 256 * NROW * 4 B per block: a multiple of 16
    for (int i = threadIdx.x; i < (cnt >> 2); i += 256) reinterpret_cast<float4 *>(lds)[i] = s4[i];
    for (int i = (cnt & ~3) + threadIdx.x; i < cnt; i += 256) lds[i] = src[base + i];
    __syncthreads();
}

template <bool STAGE_SH>
__global__ __launch_bounds__(256) void preprocess_kernel(int P, int deg, int M, const float *__restrict__ means3D, const float *__restrict__ scales, float mod,
                                                         const float *__restrict__ rots, const float *__restrict__ opac, const float *__restrict__ shs,
                                                         const float *__restrict__ shs_rest, const float *__restrict__ cov3D_pre, const float *__restrict__ colors_pre,
                                                         const float *__restrict__ view, const float *__restrict__ proj, const float *__restrict__ campos, int W, int H,
                                                         float tanx, float tany, float fx, float fy, int gx, int gy, int32_t *__restrict__ radii, Geom g) {
    extern __shared__ float sh_lds[];                                              // STAGE_SH: [256][SH_REST]
    if (STAGE_SH) stage_rows_to_lds<SH_REST>(shs_rest, P, sh_lds);
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= P) return;
    radii[idx] = 0;
    g.tiles_touched[idx] = 0;
    float *rec = g.splats + (size_t)idx * SPLAT;
    rec[3] = 0.f;                                                          // radius 0 = culled (all-zero bits)
    const rm::V3 p = {means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
    const rm::V3 pv = rm::xform43(view, p);
    if (pv.z <= 0.2f) return;                                            // near cull, auxiliary.h:154
    const float4 ph = rm::xform44(proj, p);
    const float pw = 1.0f / (ph.w + 0.0000001f);
    const float ndc_x = ph.x * pw, ndc_y = ph.y * pw;
    float cov[6];
    if (cov3D_pre) {
#pragma unroll
        for (int i = 0; i < 6; ++i) cov[i] = cov3D_pre[6 * (size_t)idx + i];
    } else {
        rm::cov3d_from_scale_rot(scales + 3 * (size_t)idx, mod, rots + 4 * (size_t)idx, cov);
#pragma unroll
        for (int i = 0; i < 6; ++i) g.cov3D[6 * (size_t)idx + i] = cov[i];
    }
    const rm::Cov2D c2 = rm::project_cov(p, fx, fy, tanx, tany, cov, view);
    const float det = c2.a * c2.c - c2.b * c2.b;
    if (det == 0.0f) return;
    const float di = 1.f / det;
    const float mid = 0.5f * (c2.a + c2.c);
    const float root = sqrtf(fmaxf(0.1f, mid * mid - det));
    const float rad = ceilf(3.f * sqrtf(fmaxf(mid + root, mid - root)));
    const float px = rm::ndc_to_pix(ndc_x, W), py = rm::ndc_to_pix(ndc_y, H);
    int x0, y0, x1, y1;
    rm::tile_rect(px, py, (int)rad, gx, gy, x0, y0, x1, y1);
    if ((x1 - x0) * (y1 - y0) == 0) return;
    rm::V3 c;
    if (!colors_pre) {
        uint32_t cm;
        // one [P,M,3] array, or DC [P,1,3] + higher orders [P,M-1,3] apart (the reference's two parameter tensors, no concatenation)
        const float *sh0 = shs_rest ? shs + (size_t)idx * 3 : shs + (size_t)idx * M * 3;
        const float *shhi = STAGE_SH ? sh_lds + threadIdx.x * SH_REST - 3 : (shs_rest ? shs_rest + (size_t)idx * (M - 1) * 3 - 3 : sh0);
        c = rm::sh_to_rgb(deg, sh0, shhi, p, {campos[0], campos[1], campos[2]}, cm);
        g.clamped[idx] = (uint8_t)cm;
    } else {
        c = {colors_pre[3 * (size_t)idx], colors_pre[3 * (size_t)idx + 1], colors_pre[3 * (size_t)idx + 2]};
    }
    radii[idx] = (int)rad;
    reinterpret_cast<float4 *>(rec)[0] = make_float4(px, py, pv.z, __int_as_float((int)rad));
    reinterpret_cast<float4 *>(rec)[1] = make_float4(c2.c * di, -c2.b * di, c2.a * di, opac[idx]);
    reinterpret_cast<float4 *>(rec)[2] = make_float4(c.x, c.y, c.z, 0.f);
    g.tiles_touched[idx] = (uint32_t)((y1 - y0) * (x1 - x0));
}

// tiles of rows [ty0, ty1) overlapped by each splat (the band a rank blends when Gaussians are sharded; whole image otherwise)
__global__ __launch_bounds__(256) void band_count_kernel(int P, const float *__restrict__ splats, int gx, int gy, int ty0, int ty1, uint32_t *__restrict__ touched) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= P) return;
    const float4 a = reinterpret_cast<const float4 *>(splats + (size_t)idx * SPLAT)[0];
    const int rad = __float_as_int(a.w);
    uint32_t n = 0;
    if (rad > 0) {
        int x0, y0, x1, y1;
        rm::tile_rect(a.x, a.y, rad, gx, gy, x0, y0, x1, y1);
        y0 = max(y0, ty0); y1 = min(y1, ty1);
        n = y1 > y0 ? (uint32_t)((y1 - y0) * (x1 - x0)) : 0u;
    }
    touched[idx] = n;
}

__global__ __launch_bounds__(256) void emit_keys_kernel(int P, const float *__restrict__ splats, const uint32_t *__restrict__ offsets, int gx, int gy, int ty0,
                                                        int ty1, uint64_t *__restrict__ keys, uint32_t *__restrict__ vals) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= P) return;
    const float4 a = reinterpret_cast<const float4 *>(splats + (size_t)idx * SPLAT)[0];
    const int rad = __float_as_int(a.w);
    if (rad <= 0) return;
    uint32_t off = idx == 0 ? 0u : offsets[idx - 1];
    int x0, y0, x1, y1;
    rm::tile_rect(a.x, a.y, rad, gx, gy, x0, y0, x1, y1);
    y0 = max(y0, ty0); y1 = min(y1, ty1);
    const uint64_t dbits = __float_as_uint(a.z);
    for (int y = y0; y < y1; ++y)
        for (int x = x0; x < x1; ++x) {
            keys[off] = ((uint64_t)(uint32_t)((y - ty0) * gx + x) << 32) | dbits;   // | band-local tile | depth |, rasterizer_impl.cu:102-106
            vals[off] = (uint32_t)idx;
            ++off;
        }
}

// instances per tile of rows [ty0, ty1).  Consecutive Gaussians (stored along a space-filling curve, or simply clustered) hit the same few tiles, and
// same-address global atomics serialise at L2 (2.4 M of them on 2500 counters: 0.45 ms): a block counts in an LDS table first and adds each touched
// tile to the global counter once (lds_tiles = 0: band too large for the table, global atomics directly).
__global__ __launch_bounds__(256) void tile_hist_kernel(int P, const float *__restrict__ splats, int gx, int gy, int ty0, int ty1, uint32_t *__restrict__ cnt, int lds_tiles) {
    extern __shared__ uint32_t lcnt[];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    for (int i = threadIdx.x; i < lds_tiles; i += 256) lcnt[i] = 0u;
    if (lds_tiles) __syncthreads();
    int rad = 0, x0 = 0, y0 = 0, x1 = 0, y1 = 0;
    if (idx < P) {
        const float4 a = reinterpret_cast<const float4 *>(splats + (size_t)idx * SPLAT)[0];
        rad = __float_as_int(a.w);
        if (rad > 0) { rm::tile_rect(a.x, a.y, rad, gx, gy, x0, y0, x1, y1); y0 = max(y0, ty0); y1 = min(y1, ty1); }
    }
    if (rad > 0)
        for (int y = y0; y < y1; ++y)
            for (int x = x0; x < x1; ++x) { const int t = (y - ty0) * gx + x; if (lds_tiles) atomicAdd(&lcnt[t], 1u); else atomicAdd(&cnt[t], 1u); }
    if (lds_tiles) {
        __syncthreads();
        for (int i = threadIdx.x; i < lds_tiles; i += 256) { const uint32_t c = lcnt[i]; if (c) atomicAdd(&cnt[i], c); }
    }
}
// one block: exclusive scan of the tile counts -> ranges[t] = (start, end), cursor[t] = start, total = {R, largest tile, 0, 0}
__global__ __launch_bounds__(1024) void tile_scan_kernel(int tiles, uint32_t *__restrict__ cursor, uint2 *__restrict__ ranges, uint32_t *__restrict__ total) {
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry_s, max_s;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) { carry_s = 0u; max_s = 0u; }
    __syncthreads();
    for (int base = 0; base < tiles; base += 1024) {
        const int t = base + tid;
        const uint32_t c = t < tiles ? cursor[t] : 0u;
        uint32_t incl = c;
        for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
        uint32_t mx = c;
        for (int o = 32; o > 0; o >>= 1) mx = max(mx, (uint32_t)__shfl_xor(mx, o, 64));
        if (lane == 63) wsum[wid] = incl;
        if (lane == 0) atomicMax(&max_s, mx);
        __syncthreads();
        uint32_t woff = 0u;
        for (int w = 0; w < wid; ++w) woff += wsum[w];
        const uint32_t start = carry_s + woff + incl - c;
        if (t < tiles) { ranges[t] = make_uint2(start, start + c); cursor[t] = start; }
        __syncthreads();
        if (tid == 1023) carry_s = start + c;
        __syncthreads();
    }
    if (tid == 0) { total[0] = carry_s; total[1] = max_s; total[2] = 0u; total[3] = 0u; }
}
// place every instance at its tile's cursor: key = depth bits << 32 | Gaussian index.  Same two levels as the histogram: the block counts its
// instances per tile in LDS, reserves one contiguous range per touched tile with a single global atomic, then hands out slots from LDS.
__global__ __launch_bounds__(256) void emit_binned_kernel(int P, const float *__restrict__ splats, int gx, int gy, int ty0, int ty1, uint32_t *__restrict__ cursor,
                                                          uint64_t *__restrict__ keys, uint32_t capacity, uint32_t *__restrict__ total, int lds_tiles) {
    extern __shared__ uint32_t ltab[];                   // [lds_tiles] counts, then [lds_tiles] bases
    uint32_t *lcnt = ltab, *lbase = ltab + lds_tiles;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    for (int i = threadIdx.x; i < lds_tiles; i += 256) lcnt[i] = 0u;
    if (lds_tiles) __syncthreads();
    int rad = 0, x0 = 0, y0 = 0, x1 = 0, y1 = 0;
    uint64_t key = 0ull;
    if (idx < P) {
        const float4 a = reinterpret_cast<const float4 *>(splats + (size_t)idx * SPLAT)[0];
        rad = __float_as_int(a.w);
        if (rad > 0) {
            rm::tile_rect(a.x, a.y, rad, gx, gy, x0, y0, x1, y1); y0 = max(y0, ty0); y1 = min(y1, ty1);
            key = ((uint64_t)__float_as_uint(a.z) << 32) | (uint32_t)idx;
        }
    }
    if (lds_tiles) {
        if (rad > 0)
            for (int y = y0; y < y1; ++y)
                for (int x = x0; x < x1; ++x) atomicAdd(&lcnt[(y - ty0) * gx + x], 1u);
        __syncthreads();
        for (int i = threadIdx.x; i < lds_tiles; i += 256) { const uint32_t c = lcnt[i]; if (c) { lbase[i] = atomicAdd(&cursor[i], c); lcnt[i] = 0u; } }
        __syncthreads();
    }
    if (rad > 0)
        for (int y = y0; y < y1; ++y)
            for (int x = x0; x < x1; ++x) {
                const int t = (y - ty0) * gx + x;
                const uint32_t slot = lds_tiles ? lbase[t] + atomicAdd(&lcnt[t], 1u) : atomicAdd(&cursor[t], 1u);
                if (slot < capacity) keys[slot] = key; else total[2] = 1u;           // caller-sized buffer too small: flagged, never written past
            }
}
// one block per tile: bitonic sort of the tile's keys (LDS up to TS_LDS_KEYS, else in place in global memory), Gaussian indices out.
// All comparators ascending ("flip" form of the merge: first stage pairs i with its mirror in the block), so entries past n can stay VIRTUAL
// +inf: a comparator whose upper index is >= n never swaps.
template <typename KeyArray>
__device__ __forceinline__ void bitonic_ascending(KeyArray &ka, int n, int m, int tid) {
    auto cx = [&](int a, int b) {
        if (b < n) { const uint64_t va = ka[a], vb = ka[b]; if (va > vb) { ka[a] = vb; ka[b] = va; } }
    };
    for (int k = 2; k <= m; k <<= 1) {
        const int hk = k >> 1;
        for (int i = tid; i < (m >> 1); i += 256) { const int blk = (i / hk) * k, o = i % hk; cx(blk + o, blk + k - 1 - o); }
        __syncthreads();
        for (int j = k >> 2; j > 0; j >>= 1) {
            for (int i = tid; i < (m >> 1); i += 256) { const int a = ((i & ~(j - 1)) << 1) | (i & (j - 1)); cx(a, a | j); }
            __syncthreads();
        }
    }
}
__global__ __launch_bounds__(256) void tile_sort_kernel(const uint2 *__restrict__ ranges, uint64_t *__restrict__ keys, uint32_t *__restrict__ point_list, uint32_t capacity) {
    extern __shared__ __attribute__((aligned(16))) uint64_t sk[];
    const uint2 rg = ranges[blockIdx.x];
    const uint32_t lo = min(rg.x, capacity), hi = min(rg.y, capacity);
    const int n = (int)(hi - lo), tid = threadIdx.x;
    if (n <= 0) return;
    int m = 1;
    while (m < n) m <<= 1;
    uint64_t *g = keys + lo;
    if (n <= TS_LDS_KEYS) {
        for (int i = tid; i < n; i += 256) sk[i] = g[i];
        __syncthreads();
        uint64_t *arr = sk;
        bitonic_ascending(arr, n, m, tid);
        for (int i = tid; i < n; i += 256) point_list[lo + i] = (uint32_t)sk[i];
    } else {                                                     // huge tile: the same network on global memory (one block: barriers order the stages)
        volatile uint64_t *arr = g;
        bitonic_ascending(arr, n, m, tid);
        for (int i = tid; i < n; i += 256) point_list[lo + i] = (uint32_t)g[i];
    }
}

// Round 6 — the per-tile sort as a MERGE sort (the bitonic network above is n log^2 n: 55 barrier stages at 1 024 keys, 91 at 8 192 — 0.14 ms of the C3 forward, and the reason
// tiles beyond 4 096 keys went to a global radix sort).  One workgroup per tile: runs of 8 keys sorted in registers (a 19-comparator network), then log2(n / 8) merge passes between two
// LDS buffers — every thread produces 8 consecutive outputs of the pass: a merge-path binary search for where its diagonal cuts the two input runs, then 8 sequential picks; the last
// pass writes the Gaussian indices straight to point_list.  Keys are distinct (depth bits << 32 | index), so the order is the bitonic kernel's and the radix path's.  Tiles of
// 8 193 ... 16 384 keys: two halves sorted that way back into the key array, merged from there.  A launch takes the tiles with nmin < n <= nmax (the caller issues one launch per size
// class, each with the LDS its class needs: a few large tiles do not cost the small ones their occupancy).
constexpr int MS_R = 8, MS_LDS_KEYS = 10240;      // keys per LDS buffer of the large class: 2 x 10 240 x 8 B = the CU's 160 KB (tiles up to 20 480 keys in two halves)
__device__ __forceinline__ void ms_cx(uint64_t &a, uint64_t &b) { const uint64_t lo = a < b ? a : b, hi = a < b ? b : a; a = lo; b = hi; }
// `n` keys at src (global) -> sorted; out_pts != NULL: indices to out_pts, else sorted keys back to dst_keys (global).  buf: 2 x cap keys of LDS, cap >= n rounded up to 8.
template <int MS_T>
__device__ __forceinline__ void ms_sort_chunk(const uint64_t *__restrict__ src, int n, uint64_t *buf, int cap, uint32_t *__restrict__ out_pts, uint64_t *__restrict__ dst_keys, int tid) {
    uint64_t *a = buf, *b = buf + cap;
    const int nruns = (n + MS_R - 1) / MS_R, np = nruns * MS_R;
    for (int r = tid; r < nruns; r += MS_T) {
        uint64_t v[MS_R];
#pragma unroll
        for (int i = 0; i < MS_R; ++i) { const int e = r * MS_R + i; v[i] = e < n ? src[e] : ~0ull; }
        // 19-comparator sorting network for 8 inputs
        ms_cx(v[0], v[1]); ms_cx(v[2], v[3]); ms_cx(v[4], v[5]); ms_cx(v[6], v[7]);
        ms_cx(v[0], v[2]); ms_cx(v[1], v[3]); ms_cx(v[4], v[6]); ms_cx(v[5], v[7]);
        ms_cx(v[1], v[2]); ms_cx(v[5], v[6]); ms_cx(v[0], v[4]); ms_cx(v[3], v[7]);
        ms_cx(v[1], v[5]); ms_cx(v[2], v[6]);
        ms_cx(v[1], v[4]); ms_cx(v[3], v[6]);
        ms_cx(v[2], v[4]); ms_cx(v[3], v[5]);
        ms_cx(v[3], v[4]);
#pragma unroll
        for (int i = 0; i < MS_R; ++i) a[r * MS_R + i] = v[i];
    }
    __syncthreads();
    bool wrote = false;
    for (int L = MS_R; L < np || !wrote; L <<= 1) {
        const bool last = 2 * L >= np;                                   // (np <= 8: one run — this pass only copies it out)
        for (int seg = tid; seg < nruns; seg += MS_T) {
            const int o = seg * MS_R;
            const int base = (o / (2 * L)) * (2 * L);
            const int la = min(L, np - base), lb = max(0, min(L, np - base - L));
            const uint64_t *A = a + base, *B = a + base + la;
            const int d = o - base;
            int lo = max(0, d - lb), hi = min(d, la);
            while (lo < hi) {                                           // merge path: the smallest i with A[i] > B[d - 1 - i]
                const int mid = (lo + hi) >> 1;
                if (A[mid] <= B[d - 1 - mid]) lo = mid + 1; else hi = mid;
            }
            int i = lo, j = d - lo;
            uint64_t av = i < la ? A[i] : ~0ull, bv = j < lb ? B[j] : ~0ull;
#pragma unroll
            for (int t = 0; t < MS_R; ++t) {
                const bool ta = (j >= lb) || (i < la && av <= bv);
                const uint64_t out = ta ? av : bv;
                if (ta) { ++i; av = i < la ? A[i] : ~0ull; } else { ++j; bv = j < lb ? B[j] : ~0ull; }
                if (last) {
                    if (o + t < n) { if (out_pts) out_pts[o + t] = (uint32_t)out; else dst_keys[o + t] = out; }
                } else b[o + t] = out;
            }
        }
        __syncthreads();
        uint64_t *tmp = a; a = b; b = tmp;
        wrote = last;
        if (last) break;
    }
}
template <int MS_T>
__global__ __launch_bounds__(MS_T) void tile_merge_sort_kernel(const uint2 *__restrict__ ranges, uint64_t *__restrict__ keys, uint32_t *__restrict__ point_list, uint32_t capacity,
                                                               int cap, uint32_t nmin, uint32_t nmax) {
    extern __shared__ __attribute__((aligned(16))) uint64_t ms_buf[];     // [2][cap]
    const uint2 rg = ranges[blockIdx.x];
    const uint32_t lo = min(rg.x, capacity), hi = min(rg.y, capacity);
    const uint32_t n = hi - lo;
    if (n <= nmin || n > nmax) return;
    const int tid = threadIdx.x;
    uint64_t *g = keys + lo;
    if (n <= (uint32_t)cap) { ms_sort_chunk<MS_T>(g, (int)n, ms_buf, cap, point_list + lo, nullptr, tid); return; }
    // two halves (n <= 2 cap), each sorted in LDS and written back in place, then merged from the key array
    const int n0 = (int)((n / 2 + MS_R - 1) / MS_R * MS_R), n1 = (int)n - n0;
    ms_sort_chunk<MS_T>(g, n0, ms_buf, cap, nullptr, g, tid);
    __syncthreads();
    ms_sort_chunk<MS_T>(g + n0, n1, ms_buf, cap, nullptr, g + n0, tid);
    __threadfence_block();
    __syncthreads();
    const volatile uint64_t *A = g, *B = g + n0;
    for (int o = tid * MS_R; o < (int)n; o += MS_T * MS_R) {
        int l2 = max(0, o - n1), h2 = min(o, n0);
        while (l2 < h2) { const int mid = (l2 + h2) >> 1; if (A[mid] <= B[o - 1 - mid]) l2 = mid + 1; else h2 = mid; }
        int i = l2, j = o - l2;
        uint64_t av = i < n0 ? A[i] : ~0ull, bv = j < n1 ? B[j] : ~0ull;
        for (int t = 0; t < MS_R && o + t < (int)n; ++t) {
            const bool ta = (j >= n1) || (i < n0 && av <= bv);
            point_list[lo + o + t] = (uint32_t)(ta ? av : bv);
            if (ta) { ++i; av = i < n0 ? A[i] : ~0ull; } else { ++j; bv = j < n1 ? B[j] : ~0ull; }
        }
    }
}

__global__ __launch_bounds__(256) void tile_ranges_kernel(int64_t R, const uint64_t *__restrict__ keys, uint2 *__restrict__ ranges) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= R) return;
    const uint32_t cur = (uint32_t)(keys[i] >> 32);
    if (i == 0) ranges[cur].x = 0;
    else {
        const uint32_t prev = (uint32_t)(keys[i - 1] >> 32);
        if (cur != prev) { ranges[prev].y = (uint32_t)i; ranges[cur].x = (uint32_t)i; }
    }
    if (i == R - 1) ranges[cur].y = (uint32_t)R;
}

// Pixel of thread `tid` inside its 16x16 tile.  A wave owns an 8x8 QUADRANT (not a 16x4 strip): a splat of ~10 px touches ~2 quadrants but ~3.5
// strips, so fewer (wave, Gaussian) pairs have a live lane — the unit the backward pays a 10-value wave reduction + atomics for.
static_assert(TILE_X == 16 && TILE_Y == 16, "quadrant mapping assumes 16x16 tiles");
__device__ __forceinline__ int wave_px(int tid) { return ((tid >> 6) & 1) * 8 + (tid & 7); }
__device__ __forceinline__ int wave_py(int tid) { return (tid >> 7) * 8 + ((tid >> 3) & 7); }

// Jittered sample positions of a wave's 64 pixels span a rectangle; rect[q] = (xmin, xmax, ymin, ymax) of quadrant q, written once per block.
__device__ __forceinline__ void publish_wave_rect(float pfx, float pfy, int tid, float (*rect)[4]) {
    float x0 = pfx, x1 = pfx, y0 = pfy, y1 = pfy;
    for (int o = 32; o > 0; o >>= 1) {
        x0 = fminf(x0, __shfl_xor(x0, o, 64)); x1 = fmaxf(x1, __shfl_xor(x1, o, 64));
        y0 = fminf(y0, __shfl_xor(y0, o, 64)); y1 = fmaxf(y1, __shfl_xor(y1, o, 64));
    }
    if ((tid & 63) == 0) { rect[tid >> 6][0] = x0; rect[tid >> 6][1] = x1; rect[tid >> 6][2] = y0; rect[tid >> 6][3] = y1; }
}
// Which quadrants can a splat reach at all?  A pixel takes a contribution only if power <= 0 and opacity * exp(power) >= 1/255, i.e. inside the
// ellipse  0.5 (A dx^2 + C dy^2) + B dx dy <= ln(255 opacity); its axis-aligned bounding box (half extents sqrt(2 tau C / det), sqrt(2 tau A / det),
// widened by 0.1 % + 0.01 px against rounding) is tested against the four sample rectangles ONCE per (tile, splat) by the thread that stages the
// splat; a wave then skips a splat on one bit test instead of evaluating the quadratic form for 64 pixels.  Conservative: results are unchanged.
__device__ __forceinline__ uint32_t quadrant_mask(float cx, float cy, float4 co, const float (*rect)[4]) {
    const float s = 255.0f * co.w;
    if (!(s >= 1.0f)) return s < 1.0f ? 0u : 0xfu;                      // opacity < 1/255 never passes the alpha test; NaN: no culling
    const float det = co.x * co.z - co.y * co.y;
    if (!(det > 0.f) || !(co.x > 0.f) || !(co.z > 0.f)) return 0xfu;
    const float tau2 = 2.0f * (logf(s) * 1.001f + 1e-3f);
    const float ex = sqrtf(tau2 * co.z / det) * 1.001f + 0.01f, ey = sqrtf(tau2 * co.x / det) * 1.001f + 0.01f;
    if (!(ex < 3.0e38f) || !(ey < 3.0e38f)) return 0xfu;
    uint32_t m = 0u;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (cx + ex >= rect[q][0] && cx - ex <= rect[q][1] && cy + ey >= rect[q][2] && cy - ey <= rect[q][3]) m |= 1u << q;
    return m;
}

// exp of the blend kernels' Gaussian exponent (power <= 0, results below 1/255 / opacity are cut anyway): v_exp_f32 of power * log2(e) — two instructions, ~2 ulp — instead
// of expf's 12 (range reduction, ldexp, overflow / underflow selects).  The blend kernels issue vector instructions 0.6-0.9 of the time (profiles/r04t_inst_mix.log);
// forward and backward use the SAME function, so the backward's recomputed alpha is the forward's bit for bit.
__device__ __forceinline__ float blend_exp(float power) { return __builtin_amdgcn_exp2f(power * 1.44269504088896340736f); }

__global__ __launch_bounds__(256) void render_fwd_kernel(const uint2 *__restrict__ ranges, const uint32_t *__restrict__ point_list, int W, int H, int ty0,
                                                         const float *__restrict__ splats, const float *__restrict__ bg, const float *__restrict__ offs,
                                                         float *__restrict__ final_T, uint32_t *__restrict__ n_contrib, float *__restrict__ out_color,
                                                         float *__restrict__ out_depth) {
    __shared__ float4 s_xy[TILE_PIX];   // x, y, quadrant mask (int bits), -
    __shared__ float4 s_co[TILE_PIX];
    __shared__ float4 s_fd[TILE_PIX];   // rgb + depth
    __shared__ float s_rect[4][4];
    __shared__ unsigned long long s_ball[4][4];   // [quadrant][staging wave]: which of the wave's 64 staged splats reach the quadrant (round 6)
    const int tid = threadIdx.y * TILE_X + threadIdx.x;
    const int lx = wave_px(tid), ly = wave_py(tid);                                // a wave = an 8x8 pixel quadrant of the tile
    const int px = blockIdx.x * TILE_X + lx, py = (blockIdx.y + ty0) * TILE_Y + ly;
    const bool inside = px < W && py < H;
    const int pix = W * py + px;
    float ox = 0.f, oy = 0.f;
    if (inside) { ox = offs[2 * (size_t)pix]; oy = offs[2 * (size_t)pix + 1]; }   // (the reference reads these unguarded: OOB on ragged tiles)
    const float pfx = (float)px + ox, pfy = (float)py + oy;
    publish_wave_rect(pfx, pfy, tid, s_rect);                                     // visible after the first barrier of the loop below
    const uint2 range = ranges[blockIdx.y * gridDim.x + blockIdx.x];
    const int rounds = (int)((range.y - range.x + TILE_PIX - 1) / TILE_PIX);
    bool done = !inside;
    float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, Dp = 0.f;
    uint32_t last = 0;
    for (int i = 0; i < rounds; ++i) {
        if (__syncthreads_count(done) == TILE_PIX) break;
        const int prog = i * TILE_PIX + tid;
        uint32_t qm = 0u;
        if (range.x + prog < range.y) {
            const uint32_t id = point_list[range.x + prog];
            const float4 *rec = reinterpret_cast<const float4 *>(splats + (size_t)id * SPLAT);
            const float4 a = rec[0], c = rec[2], co = rec[1];
            qm = quadrant_mask(a.x, a.y, co, s_rect);
            s_xy[tid] = make_float4(a.x, a.y, __uint_as_float(qm), 0.f);
            s_co[tid] = co;
            s_fd[tid] = make_float4(c.x, c.y, c.z, a.z);
        }
        // Round 6: a wave walks only the staged splats that REACH its quadrant — their positions as bits of four ballots per staging wave, taken in list order (ctz) —
        // instead of testing every staged splat's mask: at C3 a splat reaches 1.4 of the 4 quadrants, and the skipped iterations (an LDS read, a test and a scalar
        // branch each) were two thirds of the walk; the kernel's scalar instructions outnumbered its vector ones (profiles/r04t_inst_mix.log)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned long long b = __builtin_amdgcn_ballot_w64((qm >> q) & 1u);
            if ((tid & 63) == 0) s_ball[q][tid >> 6] = b;
        }
        __syncthreads();
        for (int sw = 0; sw < 4; ++sw) {
            const unsigned long long bv = s_ball[tid >> 6][sw];
            unsigned long long bits = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(bv >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)bv);
            while (bits) {
                if (__builtin_amdgcn_ballot_w64(!done) == 0ull) break;             // every pixel of the quadrant is through
                const int j = sw * 64 + __builtin_ctzll(bits);
                bits &= bits - 1;
                if (done) continue;
                const float4 c = s_xy[j];
                const float dx = c.x - pfx, dy = c.y - pfy;
                const float4 co = s_co[j];
                const float power = -0.5f * (co.x * dx * dx + co.z * dy * dy) - co.y * dx * dy;
                if (power > 0.0f) continue;
                const float alpha = fminf(0.99f, co.w * blend_exp(power));
                if (alpha < 1.0f / 255.0f) continue;
                const float test_T = T * (1.f - alpha);
                if (test_T < 0.0001f) { done = true; continue; }
                const float4 f = s_fd[j];
                const float w = alpha * T;
                C0 += f.x * w; C1 += f.y * w; C2 += f.z * w; Dp += f.w * w;
                T = test_T;
                last = (uint32_t)(i * TILE_PIX + j + 1);                           // position in the tile's list, counted from 1 (forward.cu:333 `contributor`)
            }
        }
    }
    if (inside) {
        final_T[pix] = T;
        n_contrib[pix] = last;
        const size_t hw = (size_t)H * W;
        out_color[pix] = C0 + T * bg[0];
        out_color[hw + pix] = C1 + T * bg[1];
        out_color[2 * hw + pix] = C2 + T * bg[2];
        out_depth[pix] = Dp;
    }
}

// ------------------------------------------------------------------------------------------------ backward kernels
// accumulator slots per Gaussian (12 floats, 48 B): 0-2 dL/dcolor, 3 dL/dviewdepth, 4-5 dL/dmean2D, 6-8 dL/dconic (xx,xy,yy), 9 dL/dopacity
constexpr int ACC = 12;

__global__ __launch_bounds__(256) void render_bwd_kernel(const uint2 *__restrict__ ranges, const uint32_t *__restrict__ point_list, int W, int H, int ty0,
                                                         const float *__restrict__ bg, const float *__restrict__ splats,
                                                         const float *__restrict__ final_T, const uint32_t *__restrict__ n_contrib,
                                                         const float *__restrict__ dL_dpix, const float *__restrict__ dL_ddepth, const float *__restrict__ offs,
                                                         float *__restrict__ accum) {
    __shared__ uint32_t s_id[TILE_PIX];
    __shared__ float4 s_xy[TILE_PIX];   // x, y, quadrant mask (int bits), -
    __shared__ float4 s_co[TILE_PIX];
    __shared__ float4 s_col[TILE_PIX];
    __shared__ float s_rect[4][4];
    __shared__ unsigned long long s_ball[4][4];   // [quadrant][staging wave] (see render_fwd_kernel)
    const int tid = threadIdx.y * TILE_X + threadIdx.x;
    const int lane = tid & 63;
    const int px = blockIdx.x * TILE_X + wave_px(tid), py = (blockIdx.y + ty0) * TILE_Y + wave_py(tid);
    const bool inside = px < W && py < H;
    const int pix = W * py + px;
    float ox = 0.f, oy = 0.f;
    if (inside) { ox = offs[2 * (size_t)pix]; oy = offs[2 * (size_t)pix + 1]; }
    const float pfx = (float)px + ox, pfy = (float)py + oy;
    publish_wave_rect(pfx, pfy, tid, s_rect);                                     // visible after the first barrier of the loop below
    uint2 range = ranges[blockIdx.y * gridDim.x + blockIdx.x];
    const float T_final = inside ? final_T[pix] : 0.f;
    float T = T_final;
    const uint32_t last_contributor = inside ? n_contrib[pix] : 0u;
    // Round 4: the tail of the tile's list beyond the LAST contributor of any of its pixels reaches nobody (the forward stopped every pixel before it: T < 1e-4) —
    // the reference walks it anyway, skipping per pixel (backward.cu:497-499).  In a trained scene a tile lists 4 400 instances and its pixels stop after 660 on
    // average (profiles/r04raster_*): the walk starts at the tile's largest n_contrib instead of at the end of the list.  Same sums, same order.
    __shared__ uint32_t s_maxlast;
    if (tid == 0) s_maxlast = 0u;
    __syncthreads();
    {
        uint32_t m = last_contributor;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
        if (lane == 0) atomicMax(&s_maxlast, m);
    }
    __syncthreads();
    range.y = min(range.y, range.x + s_maxlast);
    const int rounds = (int)((range.y - range.x + TILE_PIX - 1) / TILE_PIX);
    int todo = (int)(range.y - range.x);
    const size_t hw = (size_t)H * W;
    float g0 = 0.f, g1 = 0.f, g2 = 0.f, gd = 0.f;
    if (inside) { g0 = dL_dpix[pix]; g1 = dL_dpix[hw + pix]; g2 = dL_dpix[2 * hw + pix]; gd = dL_ddepth[pix]; }   // (reference reads dL_ddepth unguarded)
    const float bg_dot = bg[0] * g0 + bg[1] * g1 + bg[2] * g2;
    float ar0 = 0.f, ar1 = 0.f, ar2 = 0.f, lc0 = 0.f, lc1 = 0.f, lc2 = 0.f, last_alpha = 0.f;
    const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;
    for (int i = 0; i < rounds; ++i, todo -= TILE_PIX) {
        __syncthreads();
        const int prog = i * TILE_PIX + tid;
        uint32_t qm = 0u;
        if (range.x + prog < range.y) {
            const uint32_t id = point_list[range.y - prog - 1];        // back to front
            s_id[tid] = id;
            const float4 *rec = reinterpret_cast<const float4 *>(splats + (size_t)id * SPLAT);
            const float4 a = rec[0], co = rec[1];
            qm = quadrant_mask(a.x, a.y, co, s_rect);
            s_xy[tid] = make_float4(a.x, a.y, __uint_as_float(qm), 0.f);
            s_co[tid] = co;
            s_col[tid] = rec[2];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {                                  // (round 6: the wave walks only the staged splats that reach its quadrant — render_fwd_kernel)
            const unsigned long long b = __builtin_amdgcn_ballot_w64((qm >> q) & 1u);
            if (lane == 0) s_ball[q][tid >> 6] = b;
        }
        __syncthreads();
        for (int sw = 0; sw < 4; ++sw) {
        const unsigned long long bv = s_ball[tid >> 6][sw];
        unsigned long long bits = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(bv >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)bv);
        while (bits) {
            const int j = sw * 64 + __builtin_ctzll(bits);
            bits &= bits - 1;
            const uint32_t contributor = (uint32_t)(todo - j - 1);     // position of this entry in the tile's list, counted from 0
            const float4 c = s_xy[j];
            bool live = inside && contributor < last_contributor;
            float G = 0.f, alpha = 0.f, dx = 0.f, dy = 0.f;
            const float4 co = s_co[j];
            if (live) {
                dx = c.x - pfx; dy = c.y - pfy;
                const float power = -0.5f * (co.x * dx * dx + co.z * dy * dy) - co.y * dx * dy;
                live = power <= 0.0f;
                if (live) {
                    G = blend_exp(power);
                    alpha = fminf(0.99f, co.w * G);
                    live = alpha >= 1.0f / 255.0f;
                }
            }
            if (__builtin_amdgcn_ballot_w64(live) == 0ull) continue;     // nothing from this wave's 64 pixels
            float v[10];
#pragma unroll
            for (int q = 0; q < 10; ++q) v[q] = 0.f;
            if (live) {
                // 1 / (1 - alpha), alpha <= 0.99, by v_rcp_f32 (1 ulp) instead of an IEEE division (10 vector instructions of the ~110 per (wave, splat) pair): T picks up
                // <= 1.5 ulp per contributor against the reference's T / (1 - alpha) — three orders of magnitude inside the parity tolerance of the gradients
                const float rcp1a = __builtin_amdgcn_rcpf(1.f - alpha);
                T = T * rcp1a;
                const float w = alpha * T;
                const float4 col = s_col[j];
                ar0 = last_alpha * lc0 + (1.f - last_alpha) * ar0; lc0 = col.x;
                ar1 = last_alpha * lc1 + (1.f - last_alpha) * ar1; lc1 = col.y;
                ar2 = last_alpha * lc2 + (1.f - last_alpha) * ar2; lc2 = col.z;
                float dL_dalpha = (col.x - ar0) * g0 + (col.y - ar1) * g1 + (col.z - ar2) * g2;
                v[0] = w * g0; v[1] = w * g1; v[2] = w * g2;
                v[3] = w * gd;                                           // depth reaches only the view depth (backward.cu:552)
                dL_dalpha *= T;
                last_alpha = alpha;
                dL_dalpha += (-T_final * rcp1a) * bg_dot;
                const float dL_dG = co.w * dL_dalpha;
                const float gdx = G * dx, gdy = G * dy;
                v[4] = dL_dG * (-gdx * co.x - gdy * co.y) * ddelx_dx;
                v[5] = dL_dG * (-gdy * co.z - gdx * co.y) * ddely_dy;
                v[6] = -0.5f * gdx * dx * dL_dG;
                v[7] = -0.5f * gdx * dy * dL_dG;
                v[8] = -0.5f * gdy * dy * dL_dG;
                v[9] = G * dL_dalpha;
            }
            const float mine = wave_sum10_scatter<true>(v);              // lane q < 10: total of v[q] over the wave (DPP reduce-scatter + lane swaps, common.h)
            if (lane < 10) atomicAdd(&accum[(size_t)s_id[j] * ACC + lane], mine);
        }
        }
    }
}

__device__ __forceinline__ void unit_dir_grad(rm::V3 v, rm::V3 dv, float out[3]) {   // auxiliary.h:107-117
    const float s2 = v.x * v.x + v.y * v.y + v.z * v.z;
    const float inv = 1.0f / sqrtf(s2 * s2 * s2);
    out[0] = ((s2 - v.x * v.x) * dv.x - v.y * v.x * dv.y - v.z * v.x * dv.z) * inv;
    out[1] = (-v.x * v.y * dv.x + (s2 - v.y * v.y) * dv.y - v.z * v.y * dv.z) * inv;
    out[2] = (-v.x * v.z * dv.x - v.y * v.z * dv.y + (s2 - v.z * v.z) * dv.z) * inv;
}

// One thread per Gaussian: conic -> cov2D -> (cov3D, mean) ; mean2D -> mean ; view depth -> mean ; colour -> SH (+ mean) ;
// cov3D -> scale, quaternion.  backward.cu:144-410 fused into one streaming pass.
// ADD (round 4): the outputs are added to instead of overwritten — the per-Gaussian gradients of SEVERAL views (BASELINE C5: 4 per iteration) accumulate in one set of
// buffers without per-view copies and sums (w3d_raster_backward_accumulate)
// STAGE_SH (round 6; see preprocess_kernel): the higher-order SH rows come in through LDS, and dL/dSH of those coefficients — 45 floats per Gaussian, until now 45 store
// instructions per wave that each touched 64 different rows — is written into the thread's own LDS row (every coefficient it still needs was read before) and leaves by
// coalesced stores; rows of culled Gaussians are not written (their gradients stay whatever the caller's buffer holds: zero), as before.
template <bool ADD, bool STAGE_SH>
__global__ __launch_bounds__(256) void preprocess_bwd_kernel(int P, int deg, int M, const float *__restrict__ means3D, const int32_t *__restrict__ radii,
                                                             const float *__restrict__ shs, const float *__restrict__ shs_rest, const uint8_t *__restrict__ clamped,
                                                             const float *__restrict__ scales,
                                                             const float *__restrict__ rots, float mod, const float *__restrict__ cov3Ds,
                                                             const float *__restrict__ view, const float *__restrict__ proj, const float *__restrict__ campos,
                                                             float fx, float fy, float tanx, float tany, float *__restrict__ accum, W3dRasterGradOut o, int clean) {
    extern __shared__ float sh_lds[];                                              // STAGE_SH: [256][SH_REST] (the SH rows in, their gradients out) + [256] visibility
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const bool vis = idx < P && radii[idx] > 0;
    // clean (round 6, w3d_raster_backward_selfclean): a loop's persistent buffers without a fill per iteration — the accumulator row a Gaussian consumed is zeroed on the spot
    // (rows of culled Gaussians are never written by the blend backward: the array is zero again when this kernel is through), and the three outputs the caller reads for EVERY
    // Gaussian (opacity, scale, rotation: the activation backward runs over all of them) are written as zeros for the culled ones instead of relying on a pre-zeroed buffer
    if (clean && !ADD && idx < P && !vis) {
        o.dL_dopacity[idx] = 0.f;
        if (o.dL_dscales) { o.dL_dscales[3 * (size_t)idx] = 0.f; o.dL_dscales[3 * (size_t)idx + 1] = 0.f; o.dL_dscales[3 * (size_t)idx + 2] = 0.f; }
        if (o.dL_drotations) *reinterpret_cast<float4 *>(o.dL_drotations + 4 * (size_t)idx) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (STAGE_SH) {
        if (__syncthreads_or(vis) == 0) return;                                   // nothing of this block is on screen
        reinterpret_cast<int *>(sh_lds + 256 * SH_REST)[threadIdx.x] = vis;
        stage_rows_to_lds<SH_REST>(shs_rest, P, sh_lds);
    }
    if (vis) {
    auto st = [](float *p, float v) { if (ADD) *p += v; else *p = v; };
    float *a = accum + (size_t)idx * ACC;
    const float dcol[3] = {a[0], a[1], a[2]};
    const float dvz = a[3], dm2x = a[4], dm2y = a[5], dcxx = a[6], dcxy = a[7], dcyy = a[8];
    const float dopac = a[9];
    if (clean) {                                                                   // (48-byte rows, 16-byte aligned: ACC = 12 floats)
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        reinterpret_cast<float4 *>(a)[0] = z4; reinterpret_cast<float4 *>(a)[1] = z4; reinterpret_cast<float4 *>(a)[2] = z4;
    }
    // (dL_dmeans2D, dL_dcolors, dL_dcov3D may be NULL — round 6: the loop consumes none of the three, and each is a handful of 4-B stores at a 12- / 24-B stride)
    if (o.dL_dmeans2D) { st(o.dL_dmeans2D + 3 * (size_t)idx, dm2x); st(o.dL_dmeans2D + 3 * (size_t)idx + 1, dm2y); }
    if (o.dL_dcolors) { st(o.dL_dcolors + 3 * (size_t)idx, dcol[0]); st(o.dL_dcolors + 3 * (size_t)idx + 1, dcol[1]); st(o.dL_dcolors + 3 * (size_t)idx + 2, dcol[2]); }
    st(o.dL_dopacity + idx, dopac);
    const rm::V3 m = {means3D[3 * (size_t)idx], means3D[3 * (size_t)idx + 1], means3D[3 * (size_t)idx + 2]};
    float cov[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) cov[i] = cov3Ds[6 * (size_t)idx + i];
    const rm::Cov2D c2 = rm::project_cov(m, fx, fy, tanx, tany, cov, view);
    const float limx = 1.3f * tanx, limy = 1.3f * tany;
    const float xmul = (c2.txtz < -limx || c2.txtz > limx) ? 0.f : 1.f, ymul = (c2.tytz < -limy || c2.tytz > limy) ? 0.f : 1.f;
    const float ca = c2.a, cb = c2.b, cc = c2.c;
    const float denom = ca * cc - cb * cb;
    const float d2i = 1.0f / (denom * denom + 0.0000001f);
    float da = 0.f, db = 0.f, dc = 0.f;
    const float *T0 = c2.T0, *T1 = c2.T1;     // T0[r] = T[0][r], T1[r] = T[1][r] in the reference's column-major notation
    float dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (d2i != 0.f) {
        da = d2i * (-cc * cc * dcxx + 2 * cb * cc * dcxy + (denom - ca * cc) * dcyy);
        dc = d2i * (-ca * ca * dcyy + 2 * ca * cb * dcxy + (denom - ca * cc) * dcxx);
        db = d2i * 2 * (cb * cc * dcxx - (denom + 2 * cb * cb) * dcxy + ca * cb * dcyy);
        dcov[0] = T0[0] * T0[0] * da + T0[0] * T1[0] * db + T1[0] * T1[0] * dc;
        dcov[3] = T0[1] * T0[1] * da + T0[1] * T1[1] * db + T1[1] * T1[1] * dc;
        dcov[5] = T0[2] * T0[2] * da + T0[2] * T1[2] * db + T1[2] * T1[2] * dc;
        dcov[1] = 2 * T0[0] * T0[1] * da + (T0[0] * T1[1] + T0[1] * T1[0]) * db + 2 * T1[0] * T1[1] * dc;
        dcov[2] = 2 * T0[0] * T0[2] * da + (T0[0] * T1[2] + T0[2] * T1[0]) * db + 2 * T1[0] * T1[2] * dc;
        dcov[4] = 2 * T0[2] * T0[1] * da + (T0[1] * T1[2] + T0[2] * T1[1]) * db + 2 * T1[1] * T1[2] * dc;
    }
    if (o.dL_dcov3D) {
#pragma unroll
        for (int i = 0; i < 6; ++i) st(o.dL_dcov3D + 6 * (size_t)idx + i, dcov[i]);
    }
    // dL/dT (upper 2x3), then dL/dJ, then dL/dt (view-space mean)
    const float V[9] = {cov[0], cov[1], cov[2], cov[1], cov[3], cov[4], cov[2], cov[4], cov[5]};
    float tv0[3], tv1[3];   // T0 . V[:,c], T1 . V[:,c]
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        tv0[c] = T0[0] * V[c] + T0[1] * V[3 + c] + T0[2] * V[6 + c];
        tv1[c] = T1[0] * V[c] + T1[1] * V[3 + c] + T1[2] * V[6 + c];
    }
    float dT0[3], dT1[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { dT0[c] = 2 * tv0[c] * da + tv1[c] * db; dT1[c] = 2 * tv1[c] * dc + tv0[c] * db; }
    // W[c][r] (reference notation) = view[4r + c]
    const float dJ00 = view[0] * dT0[0] + view[4] * dT0[1] + view[8] * dT0[2];
    const float dJ02 = view[2] * dT0[0] + view[6] * dT0[1] + view[10] * dT0[2];
    const float dJ11 = view[1] * dT1[0] + view[5] * dT1[1] + view[9] * dT1[2];
    const float dJ12 = view[2] * dT1[0] + view[6] * dT1[1] + view[10] * dT1[2];
    const float tz = 1.f / c2.tz, tz2 = tz * tz, tz3 = tz2 * tz;
    const float dtx = xmul * -fx * tz2 * dJ02;
    const float dty = ymul * -fy * tz2 * dJ12;
    const float dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + (2 * fx * c2.tx) * tz3 * dJ02 + (2 * fy * c2.ty) * tz3 * dJ12;
    float dmean[3] = {view[0] * dtx + view[1] * dty + view[2] * dtz, view[4] * dtx + view[5] * dty + view[6] * dtz, view[8] * dtx + view[9] * dty + view[10] * dtz};
    // screen-space mean -> world mean through the projection (backward.cu:375-392)
    const float4 mh = rm::xform44(proj, m);
    const float mw = 1.0f / (mh.w + 0.0000001f);
    const float mul1 = (proj[0] * m.x + proj[4] * m.y + proj[8] * m.z + proj[12]) * mw * mw;
    const float mul2 = (proj[1] * m.x + proj[5] * m.y + proj[9] * m.z + proj[13]) * mw * mw;
    dmean[0] += (proj[0] * mw - proj[3] * mul1) * dm2x + (proj[1] * mw - proj[3] * mul2) * dm2y + view[2] * dvz;
    dmean[1] += (proj[4] * mw - proj[7] * mul1) * dm2x + (proj[5] * mw - proj[7] * mul2) * dm2y + view[6] * dvz;
    dmean[2] += (proj[8] * mw - proj[11] * mul1) * dm2x + (proj[9] * mw - proj[11] * mul2) * dm2y + view[10] * dvz;
    // colour -> SH coefficients and view direction (backward.cu:20-139)
    if (shs) {
        const float *sh = STAGE_SH ? sh_lds + threadIdx.x * SH_REST - 3 : (shs_rest ? shs_rest + (size_t)idx * (M - 1) * 3 - 3 : shs + (size_t)idx * M * 3);     // sh[3 i + c], i >= 1
        float *dsh0 = shs_rest ? o.dL_dsh + (size_t)idx * 3 : o.dL_dsh + (size_t)idx * M * 3;                  // DC slot
        float *dsh = shs_rest ? o.dL_dsh_rest + (size_t)idx * (M - 1) * 3 - 3 : dsh0;                           // dsh[3 i + c], i >= 1
        float *lrow = sh_lds + threadIdx.x * SH_REST - 3;                                                       // STAGE_SH: the gradients of coefficients i >= 1 leave through the thread's LDS row
        const rm::V3 dor = {m.x - campos[0], m.y - campos[1], m.z - campos[2]};
        const float len = sqrtf(dor.x * dor.x + dor.y * dor.y + dor.z * dor.z);
        const float x = dor.x / len, y = dor.y / len, z = dor.z / len;
        const uint32_t cm = clamped[idx];
        const float dR[3] = {(cm & 1u) ? 0.f : dcol[0], (cm & 2u) ? 0.f : dcol[1], (cm & 4u) ? 0.f : dcol[2]};
        float ddx = 0.f, ddy = 0.f, ddz = 0.f;    // dL/d(dir)
        auto put = [&](int i, float w) {
            if (STAGE_SH && i > 0) { float *t = lrow + 3 * i; t[0] = w * dR[0]; t[1] = w * dR[1]; t[2] = w * dR[2]; return; }
            float *t = i == 0 ? dsh0 : dsh + 3 * i; st(t, w * dR[0]); st(t + 1, w * dR[1]); st(t + 2, w * dR[2]);
        };
        auto dotsh = [&](int i) { return sh[3 * i] * dR[0] + sh[3 * i + 1] * dR[1] + sh[3 * i + 2] * dR[2]; };
        // every coefficient is READ (dotsh) before any gradient is written (put): with STAGE_SH both live in the same LDS row
        float sd[16];
#pragma unroll
        for (int i = 1; i < 16; ++i) sd[i] = (i < (deg + 1) * (deg + 1)) ? dotsh(i) : 0.f;
        put(0, rm::kSH0);
        if (deg > 0) {
            put(1, -rm::kSH1 * y); put(2, rm::kSH1 * z); put(3, -rm::kSH1 * x);
            ddx = -rm::kSH1 * sd[3]; ddy = -rm::kSH1 * sd[1]; ddz = rm::kSH1 * sd[2];
            if (deg > 1) {
                const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                put(4, rm::kSH2[0] * xy); put(5, rm::kSH2[1] * yz); put(6, rm::kSH2[2] * (2.f * zz - xx - yy)); put(7, rm::kSH2[3] * xz);
                put(8, rm::kSH2[4] * (xx - yy));
                const float s4 = sd[4], s5 = sd[5], s6 = sd[6], s7 = sd[7], s8 = sd[8];
                ddx += rm::kSH2[0] * y * s4 + rm::kSH2[2] * 2.f * -x * s6 + rm::kSH2[3] * z * s7 + rm::kSH2[4] * 2.f * x * s8;
                ddy += rm::kSH2[0] * x * s4 + rm::kSH2[1] * z * s5 + rm::kSH2[2] * 2.f * -y * s6 + rm::kSH2[4] * 2.f * -y * s8;
                ddz += rm::kSH2[1] * y * s5 + rm::kSH2[2] * 2.f * 2.f * z * s6 + rm::kSH2[3] * x * s7;
                if (deg > 2) {
                    put(9, rm::kSH3[0] * y * (3.f * xx - yy)); put(10, rm::kSH3[1] * xy * z); put(11, rm::kSH3[2] * y * (4.f * zz - xx - yy));
                    put(12, rm::kSH3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy)); put(13, rm::kSH3[4] * x * (4.f * zz - xx - yy));
                    put(14, rm::kSH3[5] * z * (xx - yy)); put(15, rm::kSH3[6] * x * (xx - 3.f * yy));
                    const float s9 = sd[9], s10 = sd[10], s11 = sd[11], s12 = sd[12], s13 = sd[13], s14 = sd[14], s15 = sd[15];
                    ddx += rm::kSH3[0] * s9 * 3.f * 2.f * xy + rm::kSH3[1] * s10 * yz + rm::kSH3[2] * s11 * -2.f * xy + rm::kSH3[3] * s12 * -3.f * 2.f * xz +
                           rm::kSH3[4] * s13 * (-3.f * xx + 4.f * zz - yy) + rm::kSH3[5] * s14 * 2.f * xz + rm::kSH3[6] * s15 * 3.f * (xx - yy);
                    ddy += rm::kSH3[0] * s9 * 3.f * (xx - yy) + rm::kSH3[1] * s10 * xz + rm::kSH3[2] * s11 * (-3.f * yy + 4.f * zz - xx) +
                           rm::kSH3[3] * s12 * -3.f * 2.f * yz + rm::kSH3[4] * s13 * -2.f * xy + rm::kSH3[5] * s14 * -2.f * yz + rm::kSH3[6] * s15 * -3.f * 2.f * xy;
                    ddz += rm::kSH3[1] * s10 * xy + rm::kSH3[2] * s11 * 4.f * 2.f * yz + rm::kSH3[3] * s12 * 3.f * (2.f * zz - xx - yy) +
                           rm::kSH3[4] * s13 * 4.f * 2.f * xz + rm::kSH3[5] * s14 * (xx - yy);
                }
            }
        }
        float dm[3];
        unit_dir_grad(dor, {ddx, ddy, ddz}, dm);
        dmean[0] += dm[0]; dmean[1] += dm[1]; dmean[2] += dm[2];
    }
    st(o.dL_dmeans3D + 3 * (size_t)idx, dmean[0]); st(o.dL_dmeans3D + 3 * (size_t)idx + 1, dmean[1]); st(o.dL_dmeans3D + 3 * (size_t)idx + 2, dmean[2]);
    // cov3D -> scale and (un-normalised) quaternion (backward.cu:278-341)
    if (scales) {
        const float *q = rots + 4 * (size_t)idx;
        const float r = q[0], x = q[1], y = q[2], z = q[3];
        float R[9];
        rm::quat_rows(q, R);
        const float s[3] = {mod * scales[3 * (size_t)idx], mod * scales[3 * (size_t)idx + 1], mod * scales[3 * (size_t)idx + 2]};
        const float dS[9] = {dcov[0], 0.5f * dcov[1], 0.5f * dcov[2], 0.5f * dcov[1], dcov[3], 0.5f * dcov[4], 0.5f * dcov[2], 0.5f * dcov[4], dcov[5]};
        // M[k][i] = s_k R[i][k];  dM = 2 M dSigma;  dscale_k = sum_j R[j][k] dM[k][j];  G[k][j] = s_k dM[k][j]
        float Gm[9];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float ds = 0.f;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float dMkj = 2.0f * (s[k] * R[k] * dS[j] + s[k] * R[3 + k] * dS[3 + j] + s[k] * R[6 + k] * dS[6 + j]);
                ds += R[3 * j + k] * dMkj;
                Gm[3 * k + j] = s[k] * dMkj;
            }
            st(o.dL_dscales + 3 * (size_t)idx + k, ds);
        }
#define GM(i, j) Gm[3 * (i) + (j)]
        float *dq = o.dL_drotations + 4 * (size_t)idx;
        st(dq, 2 * z * (GM(0, 1) - GM(1, 0)) + 2 * y * (GM(2, 0) - GM(0, 2)) + 2 * x * (GM(1, 2) - GM(2, 1)));
        st(dq + 1, 2 * y * (GM(1, 0) + GM(0, 1)) + 2 * z * (GM(2, 0) + GM(0, 2)) + 2 * r * (GM(1, 2) - GM(2, 1)) - 4 * x * (GM(2, 2) + GM(1, 1)));
        st(dq + 2, 2 * x * (GM(1, 0) + GM(0, 1)) + 2 * r * (GM(2, 0) - GM(0, 2)) + 2 * z * (GM(1, 2) + GM(2, 1)) - 4 * y * (GM(2, 2) + GM(0, 0)));
        st(dq + 3, 2 * r * (GM(0, 1) - GM(1, 0)) + 2 * x * (GM(2, 0) + GM(0, 2)) + 2 * y * (GM(1, 2) + GM(2, 1)) - 4 * z * (GM(1, 1) + GM(0, 0)));
#undef GM
    }
    }   // vis
    if (STAGE_SH) {
        // the block's dL/dSH rows (higher orders) out of LDS: element e of the block = (row e / 45, coefficient slot e % 45), consecutive lanes = consecutive floats
        __syncthreads();
        const int *s_vis = reinterpret_cast<const int *>(sh_lds + 256 * SH_REST);
        const int64_t base = (int64_t)blockIdx.x * 256 * SH_REST;
        const int cnt = (int)min((int64_t)256 * SH_REST, (int64_t)P * SH_REST - base);
        float *dst = o.dL_dsh_rest + base;
        const int live = 3 * ((deg + 1) * (deg + 1) - 1);                                         // slots of the active degree's coefficients (the direct path writes no others)
        for (int e = threadIdx.x; e < cnt; e += 256) {
            const int row = e / SH_REST;
            if (s_vis[row] && e - row * SH_REST < live) { if (ADD) dst[e] += sh_lds[e]; else dst[e] = sh_lds[e]; }
        }
    }
}

__global__ __launch_bounds__(256) void mark_visible_kernel(const float *__restrict__ means3D, const float *__restrict__ view, int P, uint8_t *__restrict__ present) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= P) return;
    const rm::V3 pv = rm::xform43(view, {means3D[3 * (size_t)idx], means3D[3 * (size_t)idx + 1], means3D[3 * (size_t)idx + 2]});
    present[idx] = pv.z > 0.2f ? 1 : 0;
}

// a10 activations (scene/gaussian_model.py:33-41): exp / normalize (eps 1e-12) / sigmoid
__global__ __launch_bounds__(256) void activations_fwd_kernel(const float *__restrict__ s, const float *__restrict__ q, const float *__restrict__ o, int P,
                                                              float *__restrict__ so, float *__restrict__ qo, float *__restrict__ oo) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
#pragma unroll
    for (int c = 0; c < 3; ++c) so[3 * (size_t)i + c] = expf(s[3 * (size_t)i + c]);
    const float4 v = *reinterpret_cast<const float4 *>(q + 4 * (size_t)i);
    const float n = fmaxf(sqrtf(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w), 1e-12f);
    *reinterpret_cast<float4 *>(qo + 4 * (size_t)i) = make_float4(v.x / n, v.y / n, v.z / n, v.w / n);
    oo[i] = 1.0f / (1.0f + expf(-o[i]));
}
__global__ __launch_bounds__(256) void activations_bwd_kernel(const float *__restrict__ s, const float *__restrict__ q, const float *__restrict__ o, int P,
                                                              const float *__restrict__ gs, const float *__restrict__ gq, const float *__restrict__ go,
                                                              float *__restrict__ ds, float *__restrict__ dq, float *__restrict__ dop) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
#pragma unroll
    for (int c = 0; c < 3; ++c) ds[3 * (size_t)i + c] = gs[3 * (size_t)i + c] * expf(s[3 * (size_t)i + c]);
    const float4 v = *reinterpret_cast<const float4 *>(q + 4 * (size_t)i);
    const float4 g = *reinterpret_cast<const float4 *>(gq + 4 * (size_t)i);
    const float n = fmaxf(sqrtf(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w), 1e-12f);
    const float4 u = make_float4(v.x / n, v.y / n, v.z / n, v.w / n);
    const float d = u.x * g.x + u.y * g.y + u.z * g.z + u.w * g.w;
    *reinterpret_cast<float4 *>(dq + 4 * (size_t)i) = make_float4((g.x - u.x * d) / n, (g.y - u.y * d) / n, (g.z - u.z * d) / n, (g.w - u.w * d) / n);
    const float sg = 1.0f / (1.0f + expf(-o[i]));
    dop[i] = go[i] * sg * (1.0f - sg);
}

}  // namespace

// ------------------------------------------------------------------------------------------------ C ABI
extern "C" size_t w3d_raster_geom_bytes(int P) {
    if (P <= 0) return ALIGN;
    const size_t p = (size_t)P;
    return up(p * 4 * SPLAT) + up(p * 24) + up(p) + up(p * 4) + up(p * 4) + up(scan_tmp_bytes(P)) + ALIGN;
}
extern "C" size_t w3d_raster_image_bytes(int H, int W) {
    const size_t n = (size_t)H * W, tiles = (size_t)((W + TILE_X - 1) / TILE_X) * ((H + TILE_Y - 1) / TILE_Y);
    return up(n * 4) + up(n * 4) + up(tiles * 8) + up(tiles * 4) + up(16) + ALIGN;
}
extern "C" size_t w3d_raster_binning_bytes(int64_t R) {
    const size_t n = (size_t)(R > 0 ? R : 1);
    return up(n * 8) * 2 + up(n * 4) * 2 + up(sort_tmp_bytes((int64_t)n)) + ALIGN;
}

// Binning path.  Default (auto): count per tile, and then — the largest tile is known before anything is placed — per-tile placement + the per-tile merge sort in LDS
// (round 6; tiles up to 20 480 instances: C3 0.10 ms, the trained scene's 11 M instances 0.5 ms with the placement), the reference-shaped global radix sort (rocPRIM, keys
// tile|depth: rasterizer_impl.cu:305-310) when some tile is larger still.  (Rounds 4-6: a bitonic network per tile up to 4 096 keys, the radix sort beyond — a trained scene had
// spent 6.1 of its 8.7 ms forward in bitonic sorts of tiles beyond 8 192 keys in global memory: profiles/r04a_raster_kernel_stats.csv.)
// W3D_RASTER_BIN=radix / binned / auto-bitonic forces a path (diagnostics; read once per process).  Every path leaves point_list at the head of the binning buffer and the
// tile ranges in the image buffer: the backward does not care which one ran.
static int bin_mode() {                    // 0 auto, 1 binned, 2 radix, 3 rounds 4-6's auto (bitonic per-tile sort up to 4 096 keys, global radix sort beyond: A/B)
    static const int v = [] { const char *e = getenv("W3D_RASTER_BIN"); if (!e) return 0; const std::string m(e); return m == "radix" ? 2 : m == "binned" ? 1 : m == "auto-bitonic" ? 3 : 0; }();
    return v;
}
static bool use_binned() { return bin_mode() != 2; }
constexpr uint32_t BIN_RADIX_ABOVE = 4096;   // (rounds 4-6, mode 3) largest tile (instances) beyond which the auto mode took the global radix sort
constexpr uint32_t BIN_MERGE_MAX = 2 * MS_LDS_KEYS;      // largest tile the per-tile merge sort takes (two LDS halves); beyond: the global radix sort
static int higher_msb(uint32_t n) {   // number of bits needed for tile ids (rasterizer_impl.cu:35-50 computes the same bound)
    int b = 0;
    while (b < 32 && (n >> b)) ++b;
    return b;
}

// ---- stage A: per-Gaussian projection -> splat records (the first SPLAT*P floats of geom_buffer)
extern "C" int w3d_raster_preprocess(const W3dRasterIn *in, void *geom_buffer, int32_t *radii, w3d_stream_t stream) {
    if (!in || !geom_buffer || !radii) return W3D_ERR_ARG;
    if (in->P <= 0 || in->H <= 0 || in->W <= 0 || !in->means3D || !in->opacities || !in->viewmatrix || !in->projmatrix) return W3D_ERR_ARG;
    if ((!in->shs) == (!in->colors_precomp)) return W3D_ERR_ARG;                       // exactly one of SHs / precomputed colours
    if (!in->cov3D_precomp && (!in->scales || !in->rotations)) return W3D_ERR_ARG;      // scale/rotation pair or precomputed covariance
    if (in->shs && (!in->campos || in->sh_coeffs < (in->sh_degree + 1) * (in->sh_degree + 1) || in->sh_degree > 3)) return W3D_ERR_ARG;
    const int P = in->P, H = in->H, W = in->W;
    const int gx = (W + TILE_X - 1) / TILE_X, gy = (H + TILE_Y - 1) / TILE_Y;
    const float fy = H / (2.0f * in->tan_fovy), fx = W / (2.0f * in->tan_fovx);
    Geom g = geom_from(geom_buffer, P);
    // (the higher-order SH rows through LDS when they are stored apart at degree-3 width and are 16-B aligned: the loop's and the reference's own parameter layout)
    const bool stage_sh = in->shs && in->shs_rest && !in->colors_precomp && in->sh_coeffs == 16 && !((uintptr_t)in->shs_rest & 15);
    if (stage_sh)
        hipLaunchKernelGGL(preprocess_kernel<true>, dim3((P + 255) / 256), dim3(256), 256 * SH_REST * sizeof(float), w3d_s(stream), P, in->sh_degree, in->sh_coeffs, in->means3D,
                           in->scales, in->scale_modifier, in->rotations, in->opacities, in->shs, in->shs_rest, in->cov3D_precomp, in->colors_precomp, in->viewmatrix,
                           in->projmatrix, in->campos, W, H, in->tan_fovx, in->tan_fovy, fx, fy, gx, gy, radii, g);
    else
        hipLaunchKernelGGL(preprocess_kernel<false>, dim3((P + 255) / 256), dim3(256), 0, w3d_s(stream), P, in->sh_degree, in->sh_coeffs, in->means3D, in->scales,
                           in->scale_modifier, in->rotations, in->opacities, in->shs, in->shs_rest, in->cov3D_precomp, in->colors_precomp, in->viewmatrix, in->projmatrix,
                           in->campos, W, H, in->tan_fovx, in->tan_fovy, fx, fy, gx, gy, radii, g);
    W3D_CHECK_LAUNCH();
    return W3D_OK;
}

extern "C" size_t w3d_raster_band_scratch_bytes(int P) {
    if (P <= 0) return ALIGN;
    return up((size_t)P * 4) * 2 + up(scan_tmp_bytes(P)) + ALIGN;
}

// ---- stage B: bin + sort + blend the tile rows [tile_row0, tile_row1) from splat records (own or all-gathered)
static int blend_forward_impl(const float *splats, int P, int H, int W, int ty0, int ty1, uint32_t *touched, bool have_counts, uint32_t *offsets, void *scan_tmp,
                              size_t scan_bytes, const float *bg, const float *offs, void *image_buffer, w3d_alloc_fn binning_alloc, void *alloc_user,
                              float *out_color, float *out_depth, int *num_rendered_host, hipStream_t st) {
    const int gx = (W + TILE_X - 1) / TILE_X, gy = (H + TILE_Y - 1) / TILE_Y;
    if (ty0 < 0 || ty1 > gy || ty0 >= ty1) return W3D_ERR_ARG;
    Img im = img_from(image_buffer, H, W);
    if (use_binned()) {
        // count per tile -> scan over tiles -> (host reads R: the allocation callback needs it, rasterizer_impl.cu:283) -> place -> sort per tile -> blend
        const int band_tiles = gx * (ty1 - ty0);
        W3D_HIP(hipMemsetAsync(im.cursor, 0, (size_t)band_tiles * sizeof(uint32_t), st));
        // LDS tables of the two binning kernels: one (histogram) / two (placement) u32 per tile of the band, when they fit
        const int lds_tiles = band_tiles <= TB_LDS_TILES ? band_tiles : 0;
        hipLaunchKernelGGL(tile_hist_kernel, dim3((P + 255) / 256), dim3(256), (size_t)lds_tiles * 4, st, P, splats, gx, gy, ty0, ty1, im.cursor, lds_tiles);
        hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, st, band_tiles, im.cursor, im.ranges, im.total);
        uint32_t tot[2] = {0u, 0u};                                                         // {R, largest tile}
        W3D_HIP(hipMemcpyAsync(tot, im.total, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        W3D_HIP(hipStreamSynchronize(st));
        const uint32_t R = tot[0];
        *num_rendered_host = (int)R;
        static const bool debug_tiles = getenv("W3D_RASTER_DEBUG_TILES") != nullptr;      // (read once; dev aid: the tile-size histogram of every forward on stderr — one blocking copy)
        if (debug_tiles) {
            std::vector<uint2> hr((size_t)band_tiles);
            (void)hipMemcpy(hr.data(), im.ranges, (size_t)band_tiles * sizeof(uint2), hipMemcpyDeviceToHost);
            int hb[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (const uint2 &r : hr) { const uint32_t n = r.y - r.x; hb[n <= 1024 ? 0 : n <= 2048 ? 1 : n <= 4096 ? 2 : n <= 5120 ? 3 : n <= 8192 ? 4 : n <= 10240 ? 5 : n <= 16384 ? 6 : 7]++; }
            fprintf(stderr, "[w3d raster] R %u largest tile %u; tiles <=1024 %d <=2048 %d <=4096 %d <=5120 %d <=8192 %d <=10240 %d <=16384 %d more %d\n", tot[0], tot[1], hb[0], hb[1], hb[2], hb[3],
                    hb[4], hb[5], hb[6], hb[7]);
        }
        void *bbuf = binning_alloc(alloc_user, w3d_raster_binning_bytes(R));
        if (!bbuf) return W3D_ERR_WORKSPACE;
        if (R > 0 && ((bin_mode() == 0 && tot[1] > BIN_MERGE_MAX) || (bin_mode() == 3 && tot[1] > BIN_RADIX_ABOVE))) {
            // some tile is large: instances in Gaussian order with keys tile|depth, one stable global radix sort (the order the per-tile sort on (depth, index) gives);
            // the tile ranges are the histogram's scan, already in place
            Bin br = bin_from(bbuf, R);
            if (!have_counts) hipLaunchKernelGGL(band_count_kernel, dim3((P + 255) / 256), dim3(256), 0, st, P, splats, gx, gy, ty0, ty1, touched);
            if (rocprim::inclusive_scan(scan_tmp, scan_bytes, touched, offsets, (size_t)P, rocprim::plus<uint32_t>(), st) != hipSuccess) return W3D_ERR_LAUNCH;
            hipLaunchKernelGGL(emit_keys_kernel, dim3((P + 255) / 256), dim3(256), 0, st, P, splats, offsets, gx, gy, ty0, ty1, br.keys_unsorted, br.vals_unsorted);
            const unsigned end_bit = 32u + (unsigned)higher_msb((uint32_t)band_tiles);
            if (rocprim::radix_sort_pairs(br.sort_tmp, br.sort_tmp_bytes, (const uint64_t *)br.keys_unsorted, br.keys, (const uint32_t *)br.vals_unsorted, br.point_list,
                                          (size_t)R, 0u, end_bit, st) != hipSuccess)
                return W3D_ERR_LAUNCH;
            hipLaunchKernelGGL(render_fwd_kernel, dim3(gx, ty1 - ty0), dim3(TILE_X, TILE_Y), 0, st, im.ranges, br.point_list, W, H, ty0, splats, bg, offs, im.final_T,
                               im.n_contrib, out_color, out_depth);
            return w3d_launch_status();
        }
        BinT b = bint_from(bbuf, R);
        if (R > 0) {
            static const int lds_ok2 = hipFuncSetAttribute(reinterpret_cast<const void *>(&emit_binned_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, TB_LDS_TILES * 8) == hipSuccess;
            if (!lds_ok2) return W3D_ERR_LAUNCH;
            hipLaunchKernelGGL(emit_binned_kernel, dim3((P + 255) / 256), dim3(256), (size_t)lds_tiles * 8, st, P, splats, gx, gy, ty0, ty1, im.cursor, b.keys, R, im.total, lds_tiles);
            if (bin_mode() == 0 && tot[1] <= BIN_MERGE_MAX) {
                // per-tile merge sort, one launch per size class (the LDS of a launch = its class's largest tile, two buffers): tiles up to 2 048 keys, and the rest
                // (small class: 256 threads, up to 64 KB of LDS — several tiles per CU; large class: 1 024 threads on the CU's whole LDS)
                static const int lds_ok4 = hipFuncSetAttribute(reinterpret_cast<const void *>(&tile_merge_sort_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 4096 * 8) == hipSuccess &&
                                           hipFuncSetAttribute(reinterpret_cast<const void *>(&tile_merge_sort_kernel<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * MS_LDS_KEYS * 8) == hipSuccess;
                if (!lds_ok4) return W3D_ERR_LAUNCH;
                constexpr uint32_t SMALL = 4096;
                int cap_s = (int)((std::min<uint32_t>(tot[1], SMALL) + 255u) / 256u * 256u);
                hipLaunchKernelGGL(tile_merge_sort_kernel<256>, dim3(band_tiles), dim3(256), (size_t)2 * cap_s * 8, st, im.ranges, b.keys, b.point_list, R, cap_s, 0u, SMALL);
                if (tot[1] > SMALL)
                    hipLaunchKernelGGL(tile_merge_sort_kernel<1024>, dim3(band_tiles), dim3(1024), (size_t)2 * MS_LDS_KEYS * 8, st, im.ranges, b.keys, b.point_list, R, MS_LDS_KEYS, SMALL,
                                       BIN_MERGE_MAX);
            } else {
            static const int lds_ok = hipFuncSetAttribute(reinterpret_cast<const void *>(&tile_sort_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, TS_LDS_KEYS * 8) == hipSuccess;
            if (!lds_ok) return W3D_ERR_LAUNCH;
            size_t sort_keys = 256;                                                         // LDS of the sort = the largest tile, rounded up (occupancy)
            while (sort_keys < tot[1] && sort_keys < (size_t)TS_LDS_KEYS) sort_keys <<= 1;
            hipLaunchKernelGGL(tile_sort_kernel, dim3(band_tiles), dim3(256), sort_keys * 8, st, im.ranges, b.keys, b.point_list, R);
            }
        }
        hipLaunchKernelGGL(render_fwd_kernel, dim3(gx, ty1 - ty0), dim3(TILE_X, TILE_Y), 0, st, im.ranges, b.point_list, W, H, ty0, splats, bg, offs, im.final_T,
                           im.n_contrib, out_color, out_depth);
        return w3d_launch_status();
    }
    if (!have_counts) hipLaunchKernelGGL(band_count_kernel, dim3((P + 255) / 256), dim3(256), 0, st, P, splats, gx, gy, ty0, ty1, touched);
    if (rocprim::inclusive_scan(scan_tmp, scan_bytes, touched, offsets, (size_t)P, rocprim::plus<uint32_t>(), st) != hipSuccess) return W3D_ERR_LAUNCH;
    uint32_t R = 0;
    W3D_HIP(hipMemcpyAsync(&R, offsets + (P - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    W3D_HIP(hipStreamSynchronize(st));                                                  // the reference's blocking cudaMemcpy (rasterizer_impl.cu:283)
    *num_rendered_host = (int)R;
    void *bbuf = binning_alloc(alloc_user, w3d_raster_binning_bytes(R));
    if (!bbuf) return W3D_ERR_WORKSPACE;
    Bin b = bin_from(bbuf, R);
    const int band_tiles = gx * (ty1 - ty0);
    W3D_HIP(hipMemsetAsync(im.ranges, 0, (size_t)band_tiles * sizeof(uint2), st));
    if (R > 0) {
        hipLaunchKernelGGL(emit_keys_kernel, dim3((P + 255) / 256), dim3(256), 0, st, P, splats, offsets, gx, gy, ty0, ty1, b.keys_unsorted, b.vals_unsorted);
        const unsigned end_bit = 32u + (unsigned)higher_msb((uint32_t)band_tiles);
        if (rocprim::radix_sort_pairs(b.sort_tmp, b.sort_tmp_bytes, (const uint64_t *)b.keys_unsorted, b.keys, (const uint32_t *)b.vals_unsorted, b.point_list,
                                      (size_t)R, 0u, end_bit, st) != hipSuccess)
            return W3D_ERR_LAUNCH;
        hipLaunchKernelGGL(tile_ranges_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, st, (int64_t)R, b.keys, im.ranges);
    }
    hipLaunchKernelGGL(render_fwd_kernel, dim3(gx, ty1 - ty0), dim3(TILE_X, TILE_Y), 0, st, im.ranges, b.point_list, W, H, ty0, splats, bg, offs, im.final_T,
                       im.n_contrib, out_color, out_depth);
    return w3d_launch_status();
}

extern "C" int w3d_raster_blend_forward(const float *splats, int P, int H, int W, int tile_row0, int tile_row1, const float *background,
                                        const float *sampling_offsets, void *band_scratch, void *image_buffer, w3d_alloc_fn binning_alloc, void *alloc_user,
                                        float *out_color, float *out_depth, int *num_rendered_host, w3d_stream_t stream) {
    if (!splats || P <= 0 || !background || !sampling_offsets || !band_scratch || !image_buffer || !binning_alloc || !out_color || !out_depth || !num_rendered_host)
        return W3D_ERR_ARG;
    char *p = reinterpret_cast<char *>(band_scratch);
    uint32_t *touched = take<uint32_t>(p, P);
    uint32_t *offsets = take<uint32_t>(p, P);
    return blend_forward_impl(splats, P, H, W, tile_row0, tile_row1, touched, false, offsets, p, scan_tmp_bytes(P), background, sampling_offsets, image_buffer,
                              binning_alloc, alloc_user, out_color, out_depth, num_rendered_host, w3d_s(stream));
}

extern "C" int w3d_raster_forward(const W3dRasterIn *in, void *geom_buffer, void *image_buffer, w3d_alloc_fn binning_alloc, void *alloc_user, float *out_color,
                                  float *out_depth, int32_t *radii, int *num_rendered_host, w3d_stream_t stream) {
    if (!in || !geom_buffer || !image_buffer || !binning_alloc || !out_color || !out_depth || !radii || !num_rendered_host) return W3D_ERR_ARG;
    if (!in->background || !in->sampling_offsets) return W3D_ERR_ARG;
    const int rc = w3d_raster_preprocess(in, geom_buffer, radii, stream);
    if (rc != W3D_OK) return rc;
    Geom g = geom_from(geom_buffer, in->P);
    const int gy = (in->H + TILE_Y - 1) / TILE_Y;
    return blend_forward_impl(g.splats, in->P, in->H, in->W, 0, gy, g.tiles_touched, true, g.point_offsets, g.scan_tmp, g.scan_tmp_bytes, in->background,
                              in->sampling_offsets, image_buffer, binning_alloc, alloc_user, out_color, out_depth, num_rendered_host, w3d_s(stream));
}

// ---- stage C: blend backward over a band -> packed accumulators accum[P,12] (+=)
extern "C" int w3d_raster_blend_backward(const float *splats, int P, int H, int W, int tile_row0, int tile_row1, const float *background,
                                         const float *sampling_offsets, const void *image_buffer, const void *binning_buffer, int num_rendered,
                                         const float *dL_dout_color, const float *dL_dout_depth, float *accum, w3d_stream_t stream) {
    if (!splats || P <= 0 || !background || !sampling_offsets || !image_buffer || !dL_dout_color || !dL_dout_depth || !accum) return W3D_ERR_ARG;
    const int gx = (W + TILE_X - 1) / TILE_X, gy = (H + TILE_Y - 1) / TILE_Y;
    if (tile_row0 < 0 || tile_row1 > gy || tile_row0 >= tile_row1) return W3D_ERR_ARG;
    if (num_rendered <= 0) return W3D_OK;
    if (!binning_buffer) return W3D_ERR_ARG;
    Img im = img_from(const_cast<void *>(image_buffer), H, W);
    const uint32_t *point_list = bint_from(const_cast<void *>(binning_buffer), num_rendered).point_list;     // head of the buffer in either layout
    hipLaunchKernelGGL(render_bwd_kernel, dim3(gx, tile_row1 - tile_row0), dim3(TILE_X, TILE_Y), 0, w3d_s(stream), im.ranges, point_list, W, H, tile_row0, background,
                       splats, im.final_T, im.n_contrib, dL_dout_color, dL_dout_depth, sampling_offsets, accum);
    W3D_CHECK_LAUNCH();
    return W3D_OK;
}

// ---- stage D: per-Gaussian chain from the accumulators to the parameter gradients
static int preprocess_backward_impl(const W3dRasterIn *in, const int32_t *radii, const void *geom_buffer, W3dRasterGradOut *gout, w3d_stream_t stream, bool add, int clean = 0) {
    if (!in || !radii || !gout || !geom_buffer) return W3D_ERR_ARG;
    if (!gout->dL_dopacity || !gout->dL_dmeans3D || !gout->accum) return W3D_ERR_ARG;          // (dL_dmeans2D, dL_dcolors, dL_dcov3D: NULL = not wanted)
    if (!in->scales && !gout->dL_dcov3D) return W3D_ERR_ARG;                                     // with a precomputed covariance its gradient is the output
    if (in->shs && !gout->dL_dsh) return W3D_ERR_ARG;
    if (in->shs_rest && (!in->shs || !gout->dL_dsh_rest || in->sh_coeffs < 2)) return W3D_ERR_ARG;
    if (in->scales && (!gout->dL_dscales || !gout->dL_drotations)) return W3D_ERR_ARG;
    const int P = in->P, H = in->H, W = in->W;
    const float fy = H / (2.0f * in->tan_fovy), fx = W / (2.0f * in->tan_fovx);
    Geom g = geom_from(const_cast<void *>(geom_buffer), P);
    const float *cov3D = in->cov3D_precomp ? in->cov3D_precomp : g.cov3D;
    const bool stage_sh = in->shs && in->shs_rest && in->sh_coeffs == 16 && !((uintptr_t)in->shs_rest & 15);
    const size_t lds = stage_sh ? (256 * SH_REST + 256) * sizeof(float) : 0;
#define W3D_PRE_BWD(ADD_, ST_)                                                                                                                                              \
    hipLaunchKernelGGL((preprocess_bwd_kernel<ADD_, ST_>), dim3((P + 255) / 256), dim3(256), lds, w3d_s(stream), P, in->sh_degree, in->sh_coeffs, in->means3D, radii, in->shs, \
                       in->shs_rest, g.clamped, in->scales, in->rotations, in->scale_modifier, cov3D, in->viewmatrix, in->projmatrix, in->campos, fx, fy, in->tan_fovx,      \
                       in->tan_fovy, gout->accum, *gout, clean)
    if (add) { if (stage_sh) W3D_PRE_BWD(true, true); else W3D_PRE_BWD(true, false); }
    else { if (stage_sh) W3D_PRE_BWD(false, true); else W3D_PRE_BWD(false, false); }
#undef W3D_PRE_BWD
    W3D_CHECK_LAUNCH();
    return W3D_OK;
}
extern "C" int w3d_raster_preprocess_backward(const W3dRasterIn *in, const int32_t *radii, const void *geom_buffer, W3dRasterGradOut *gout, w3d_stream_t stream) {
    return preprocess_backward_impl(in, radii, geom_buffer, gout, stream, false);
}

static int raster_backward_impl(const W3dRasterIn *in, const int32_t *radii, const W3dRasterGradIn *gin, W3dRasterGradOut *gout, const void *geom_buffer,
                                const void *binning_buffer, const void *image_buffer, int num_rendered, w3d_stream_t stream, bool add, int clean = 0) {
    if (!in || !radii || !gin || !gout || !geom_buffer || !image_buffer || !gin->dL_dout_color || !gin->dL_dout_depth || !gout->accum) return W3D_ERR_ARG;
    if (!in->background || !in->sampling_offsets) return W3D_ERR_ARG;
    Geom g = geom_from(const_cast<void *>(geom_buffer), in->P);
    const int gy = (in->H + TILE_Y - 1) / TILE_Y;
    int rc = w3d_raster_blend_backward(g.splats, in->P, in->H, in->W, 0, gy, in->background, in->sampling_offsets, image_buffer, binning_buffer, num_rendered,
                                       gin->dL_dout_color, gin->dL_dout_depth, gout->accum, stream);
    if (rc != W3D_OK) return rc;
    return preprocess_backward_impl(in, radii, geom_buffer, gout, stream, add, clean);
}
extern "C" int w3d_raster_backward(const W3dRasterIn *in, const int32_t *radii, const W3dRasterGradIn *gin, W3dRasterGradOut *gout, const void *geom_buffer,
                                   const void *binning_buffer, const void *image_buffer, int num_rendered, w3d_stream_t stream) {
    return raster_backward_impl(in, radii, gin, gout, geom_buffer, binning_buffer, image_buffer, num_rendered, stream, false);
}
// The same backward ADDING to every per-Gaussian output instead of overwriting it (gout->accum, the blend backward's scratch, must still be zeroed by the caller before
// each call): the gradients of several views of one iteration accumulate in place.  Gaussians culled in this view are not touched.
extern "C" int w3d_raster_backward_accumulate(const W3dRasterIn *in, const int32_t *radii, const W3dRasterGradIn *gin, W3dRasterGradOut *gout, const void *geom_buffer,
                                              const void *binning_buffer, const void *image_buffer, int num_rendered, w3d_stream_t stream) {
    return raster_backward_impl(in, radii, gin, gout, geom_buffer, binning_buffer, image_buffer, num_rendered, stream, true);
}

// w3d_raster_backward (accumulate = 0) / w3d_raster_backward_accumulate (accumulate != 0) for PERSISTENT buffers: gout->accum must be all zero on entry (zero-fill it once, when it is
// allocated) and is all zero again on return; with accumulate = 0 the outputs dL_dopacity, dL_dscales, dL_drotations are written for EVERY Gaussian (zeros for the culled ones), so
// they need no fill either.  The other outputs (dL_dmeans3D, dL_dsh, dL_dsh_rest, ...) are written for visible Gaussians only, as before.
extern "C" int w3d_raster_backward_selfclean(const W3dRasterIn *in, const int32_t *radii, const W3dRasterGradIn *gin, W3dRasterGradOut *gout, const void *geom_buffer,
                                             const void *binning_buffer, const void *image_buffer, int num_rendered, int accumulate, w3d_stream_t stream) {
    return raster_backward_impl(in, radii, gin, gout, geom_buffer, binning_buffer, image_buffer, num_rendered, stream, accumulate != 0, 1);
}

extern "C" int w3d_mark_visible(const float *means3D, const float *viewmatrix, const float *projmatrix, int P, uint8_t *present, w3d_stream_t stream) {
    if (!means3D || !viewmatrix || !present || P < 0) return W3D_ERR_ARG;
    (void)projmatrix;   // the frustum test of the fork only uses the view depth (auxiliary.h:154; the side planes are commented out)
    if (P == 0) return W3D_OK;
    hipLaunchKernelGGL(mark_visible_kernel, dim3((P + 255) / 256), dim3(256), 0, w3d_s(stream), means3D, viewmatrix, P, present);
    W3D_CHECK_LAUNCH();
    return W3D_OK;
}

extern "C" int w3d_activations_fwd(const float *scaling_raw, const float *rotation_raw, const float *opacity_raw, int P, float *scales, float *rotations,
                                   float *opacities, w3d_stream_t stream) {
    if (!scaling_raw || !rotation_raw || !opacity_raw || !scales || !rotations || !opacities || P < 0) return W3D_ERR_ARG;
    if (P == 0) return W3D_OK;
    hipLaunchKernelGGL(activations_fwd_kernel, dim3((P + 255) / 256), dim3(256), 0, w3d_s(stream), scaling_raw, rotation_raw, opacity_raw, P, scales, rotations, opacities);
    W3D_CHECK_LAUNCH();
    return W3D_OK;
}
extern "C" int w3d_activations_bwd(const float *scaling_raw, const float *rotation_raw, const float *opacity_raw, int P, const float *g_scales,
                                   const float *g_rotations, const float *g_opacities, float *g_scaling_raw, float *g_rotation_raw, float *g_opacity_raw,
                                   w3d_stream_t stream) {
    if (!scaling_raw || !rotation_raw || !opacity_raw || !g_scales || !g_rotations || !g_opacities || !g_scaling_raw || !g_rotation_raw || !g_opacity_raw || P < 0)
        return W3D_ERR_ARG;
    if (P == 0) return W3D_OK;
    hipLaunchKernelGGL(activations_bwd_kernel, dim3((P + 255) / 256), dim3(256), 0, w3d_s(stream), scaling_raw, rotation_raw, opacity_raw, P, g_scales, g_rotations,
                       g_opacities, g_scaling_raw, g_rotation_raw, g_opacity_raw);
    W3D_CHECK_LAUNCH();
    return W3D_OK;
}
