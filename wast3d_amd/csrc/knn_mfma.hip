// a1+a2  fused distance-GEMM + running top-k on the gfx950 matrix cores (bf16 in, fp32 accumulate).
//
// Layout of the work ("flash-attention without the PV product"):
//   block  = 512 threads = 8 waves, owns 512 content rows of X; streams ALL style rows of Y once.
//   wave   = 64 X rows (RB = 2 blocks of 32), kept as MFMA B-operand fragments in VGPRs for the whole kernel.
//   Y tile = 32 style rows x D, double-buffered in LDS (XOR-swizzled 16-B chunks: conflict-free ds_read_b128),
//            shared by the 8 waves.
//   MFMA   = v_mfma_f32_32x32x16_bf16, operands swapped (A = Y rows, B = X^T) so that the X row sits on the
//            lane (col = lane & 31) and the 16 accumulator registers of a lane are 16 different candidates of
//            ONE row -> the running-threshold test is a per-lane v_max3 tree + one compare.
//   acc    = -0.5|y|^2 (C-in) + x.y   ("val", bigger = closer);  d2 = |x|^2 - 2 val.
// Selection: candidates that beat the row's current k-th best are appended to a per-wave LDS queue
// (scalar slot counter, no atomics); when the queue fills the wave counting-sorts it by row and every lane
// merges its own row's entries into its k-entry list (LDS, lane-strided) and refreshes the threshold.
// Algorithmic work: 2*N*M*D flop; HBM bytes (N+M)*D*2 + N*k*8 (negligible) -> MFMA-bound.
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BN = 32;        // style rows per tile
constexpr int QCAP = 128;     // queue entries per wave
constexpr int NWAVES = 8;
constexpr int RB = 2;         // 32-row blocks per wave
constexpr int ROWS_W = RB * 32;
constexpr int ROWS_B = NWAVES * ROWS_W;

struct __attribute__((aligned(8))) Entry {
    float val;
    uint32_t packed;   // (local row << 26) | style index     (queue)   /   style index (lists)
};

// "a is a better neighbour than b": larger val, ties -> lower style index (torch.topk / stable sort order)
__device__ __forceinline__ bool better(float va, uint32_t ia, float vb, uint32_t ib) { return va > vb || (va == vb && ia < ib); }

template <int D>
__device__ __forceinline__ int swz(int r) { return D >= 128 ? (r & 15) : ((r >> 1) & 7); }

// out[r] = scale * sum_c A[r,c]^2, (D/8) lanes per row, 16-B loads
template <int D>
__global__ __launch_bounds__(256) void row_sqnorm_kernel(const uint16_t *__restrict__ A, int64_t rows, float scale, float *__restrict__ out) {
    constexpr int LPR = D / 8;                    // lanes per row (8, 16, 32)
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t r = gid / LPR;
    const int c = (int)(gid % LPR);
    float s = 0.f;
    if (r < rows) {
        const uint4 v = *reinterpret_cast<const uint4 *>(A + r * D + c * 8);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float lo = __uint_as_float(w[i] << 16), hi = __uint_as_float(w[i] & 0xffff0000u);
            s = fmaf(lo, lo, s); s = fmaf(hi, hi, s);
        }
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (r < rows && c == 0) out[r] = scale * s;
}

template <int D>
__global__ __launch_bounds__(512, 2) void knn_mfma_kernel(const uint16_t *__restrict__ X, const uint16_t *__restrict__ Y, const float *__restrict__ xnorm,
                                                          const float *__restrict__ ynh, int64_t N, int64_t M, int k, int32_t *__restrict__ out_idx,
                                                          float *__restrict__ out_d2) {
    constexpr int KS = D / 16;            // MFMA k-steps per tile
    constexpr int CPR = D / 8;            // 16-B chunks per row
    constexpr int TILE_BYTES = BN * D * 2;
    constexpr int CHUNKS_PER_THREAD = (BN * CPR) / 512;   // 2 (D=256), 1 (D=128); D=64 handled by predicate
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *ytile = smem;                                               // [2][TILE_BYTES]
    float *ynh_t = reinterpret_cast<float *>(smem + 2 * TILE_BYTES);     // [2][BN]
    uint8_t *wave_base = smem + 2 * TILE_BYTES + 2 * BN * 4;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int per_wave = (2 * QCAP + k * ROWS_W) * (int)sizeof(Entry);
    Entry *queue = reinterpret_cast<Entry *>(wave_base + wid * per_wave);
    Entry *sorted = queue + QCAP;
    Entry *lists = sorted + QCAP;                                        // [k][ROWS_W], lane-strided

    const int xr = lane & 31, h = lane >> 5;
    const int64_t wave_row0 = (int64_t)blockIdx.x * ROWS_B + wid * ROWS_W;

    // ---- X fragments (B operand): lane holds X[row = rb*32 + xr][16 s + 8 h .. +8] for every k-step s
    bf16x8 xf[RB][KS];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        int64_t row = wave_row0 + rb * 32 + xr;
        if (row >= N) row = N - 1;                                       // clamp; never written out
#pragma unroll
        for (int s = 0; s < KS; ++s) xf[rb][s] = *reinterpret_cast<const bf16x8 *>(X + row * D + 16 * s + 8 * h);
    }

    // ---- per-lane list state: lane L owns local row L of this wave
    int lcount = 0, lminpos = 0;
    float lmin_v = __builtin_inff();
    uint32_t lmin_i = 0;
    float thr[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) thr[rb] = -__builtin_inff();
    int qcnt = 0;

    auto flush = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // 1. how many queued entries belong to my row
        int c = 0;
        for (int e = 0; e < qcnt; ++e) c += (int)((queue[e].packed >> 26) == (uint32_t)lane);
        // 2. exclusive prefix over lanes
        int incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
        const int off = incl - c;
        // 3. scatter into row order
        int pos = off;
        for (int e = 0; e < qcnt; ++e) {
            const Entry en = queue[e];
            if ((en.packed >> 26) == (uint32_t)lane) { sorted[pos] = en; ++pos; }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // 4. merge my entries into my list
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
                for (int t = 0; t < k; ++t) {
                    const Entry le = lists[t * ROWS_W + lane];
                    if (t == 0 || better(wv, wi, le.val, le.packed)) { wv = le.val; wi = le.packed; wp = t; }
                }
                lmin_v = wv; lmin_i = wi; lminpos = wp;
            }
        }
        // 5. thresholds back into the MFMA layout (row rb*32 + xr lives in lane rb*32 + xr)
        const float mythr = (lcount == k) ? lmin_v : -__builtin_inff();
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) thr[rb] = __shfl(mythr, rb * 32 + xr, 64);
        qcnt = 0;
        __builtin_amdgcn_wave_barrier();
    };

    // ---- Y tile staging: global -> registers -> swizzled LDS
    const int64_t ntiles = (M + BN - 1) / BN;
    uint4 stage[CHUNKS_PER_THREAD > 0 ? CHUNKS_PER_THREAD : 1];
    float stage_n = 0.f;
    auto load_tile = [&](int64_t t) {
#pragma unroll
        for (int i = 0; i < (CHUNKS_PER_THREAD > 0 ? CHUNKS_PER_THREAD : 1); ++i) {
            const int q = threadIdx.x + 512 * i;
            if (q < BN * CPR) {
                const int r = q / CPR, c = q % CPR;
                int64_t yrow = t * BN + r;
                if (yrow >= M) yrow = M - 1;
                stage[i] = *reinterpret_cast<const uint4 *>(Y + yrow * D + c * 8);
            }
        }
        if (threadIdx.x < BN) {
            const int64_t yrow = t * BN + threadIdx.x;
            stage_n = yrow < M ? ynh[yrow] : -__builtin_inff();      // padded rows can never be selected
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < (CHUNKS_PER_THREAD > 0 ? CHUNKS_PER_THREAD : 1); ++i) {
            const int q = threadIdx.x + 512 * i;
            if (q < BN * CPR) {
                const int r = q / CPR, c = q % CPR;
                *reinterpret_cast<uint4 *>(ytile + buf * TILE_BYTES + r * (D * 2) + ((c ^ swz<D>(r)) << 4)) = stage[i];
            }
        }
        if (threadIdx.x < BN) ynh_t[buf * BN + threadIdx.x] = stage_n;
    };

    load_tile(0);
    store_tile(0);

    for (int64_t t = 0; t < ntiles; ++t) {
        const int cur = (int)(t & 1);
        if (t + 1 < ntiles) load_tile(t + 1);
        __syncthreads();

        // ---- distance block: acc[rb][reg] = -0.5|y|^2 + x.y  for y = tile row (reg&3)+8(reg>>2)+4h
        f32x16 cinit;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 v = *reinterpret_cast<const float4 *>(&ynh_t[cur * BN + 8 * g + 4 * h]);
            cinit[4 * g + 0] = v.x; cinit[4 * g + 1] = v.y; cinit[4 * g + 2] = v.z; cinit[4 * g + 3] = v.w;
        }
        f32x16 acc[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) acc[rb] = cinit;
        const uint8_t *trow = ytile + cur * TILE_BYTES + xr * (D * 2);
        const int sw = swz<D>(xr);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const bf16x8 a = *reinterpret_cast<const bf16x8 *>(trow + (((2 * s + h) ^ sw) << 4));
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, xf[rb][s], acc[rb], 0, 0, 0);
        }

        // ---- selection epilogue
        const uint32_t ybase = (uint32_t)(t * BN) + 4 * h;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            float m01 = fmaxf(fmaxf(acc[rb][0], acc[rb][1]), fmaxf(acc[rb][2], acc[rb][3]));
            float m23 = fmaxf(fmaxf(acc[rb][4], acc[rb][5]), fmaxf(acc[rb][6], acc[rb][7]));
            float m45 = fmaxf(fmaxf(acc[rb][8], acc[rb][9]), fmaxf(acc[rb][10], acc[rb][11]));
            float m67 = fmaxf(fmaxf(acc[rb][12], acc[rb][13]), fmaxf(acc[rb][14], acc[rb][15]));
            const float mx = fmaxf(fmaxf(m01, m23), fmaxf(m45, m67));
            if (__builtin_amdgcn_ballot_w64(mx >= thr[rb]) != 0ull) {
                int r_start = 0;
                while (true) {
                    bool overflow = false;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        if (r >= r_start && !overflow) {
                            const bool hit = acc[rb][r] >= thr[rb];
                            const unsigned long long mask = __builtin_amdgcn_ballot_w64(hit);
                            if (mask != 0ull) {
                                const int n = __builtin_popcountll(mask);
                                if (qcnt + n > QCAP) {
                                    overflow = true; r_start = r;
                                } else {
                                    const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
                                    if (hit) {
                                        const uint32_t yi = ybase + (uint32_t)((r & 3) + 8 * (r >> 2));
                                        queue[qcnt + rank] = Entry{acc[rb][r], ((uint32_t)(rb * 32 + xr) << 26) | yi};
                                    }
                                    qcnt += n;
                                }
                            }
                        }
                    }
                    if (!overflow) break;
                    flush();
                }
            }
        }

        if (t + 1 < ntiles) store_tile(cur ^ 1);
    }
    flush();

    // ---- write out: each lane selection-sorts its k entries (best first)
    const int64_t row = wave_row0 + lane;
    if (row < N) {
        const float xn = xnorm[row];
        for (int t = 0; t < k; ++t) {
            float bv = -__builtin_inff(); uint32_t bi = 0xffffffffu; int bp = t;
            for (int u = t; u < k; ++u) {
                const Entry le = lists[u * ROWS_W + lane];
                if (better(le.val, le.packed, bv, bi)) { bv = le.val; bi = le.packed; bp = u; }
            }
            const Entry tmp = lists[t * ROWS_W + lane];
            lists[bp * ROWS_W + lane] = tmp;
            lists[t * ROWS_W + lane] = Entry{bv, bi};
            out_idx[row * k + t] = (int32_t)bi;
            out_d2[row * k + t] = fmaxf(xn - 2.0f * bv, 0.f);
        }
    }
}

template <int D>
int launch(const uint16_t *X, const uint16_t *Y, int64_t N, int64_t M, int k, int32_t *idx, float *d2, float *xnorm, float *ynh, hipStream_t st) {
    constexpr int LPR = D / 8;
    hipLaunchKernelGGL(row_sqnorm_kernel<D>, dim3((unsigned)((N * LPR + 255) / 256)), dim3(256), 0, st, X, N, 1.0f, xnorm);
    hipLaunchKernelGGL(row_sqnorm_kernel<D>, dim3((unsigned)((M * LPR + 255) / 256)), dim3(256), 0, st, Y, M, -0.5f, ynh);
    const size_t lds = 2 * (size_t)BN * D * 2 + 2 * BN * 4 + (size_t)NWAVES * (2 * QCAP + (size_t)k * ROWS_W) * sizeof(Entry);
    if (lds > 160 * 1024) return W3D_ERR_UNSUPPORTED;
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&knn_mfma_kernel<D>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return W3D_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(knn_mfma_kernel<D>, dim3((unsigned)((N + ROWS_B - 1) / ROWS_B)), dim3(512), lds, st, X, Y, xnorm, ynh, N, M, k, idx, d2);
    return hipGetLastError() == hipSuccess ? W3D_OK : W3D_ERR_LAUNCH;
}

}  // namespace

extern "C" size_t w3d_knn_mfma_workspace_bytes(int64_t N, int64_t M, int d, int k) {
    (void)d; (void)k;
    return (size_t)(((N + 63) / 64) * 64 + M) * sizeof(float) + 512;
}

extern "C" int w3d_knn_mfma_fwd(const void *X, const void *Y, int64_t N, int64_t M, int d, int k, int32_t *idx, float *dist2, void *ws, size_t ws_bytes,
                                hipStream_t st) {
    if (!ws || ws_bytes < w3d_knn_mfma_workspace_bytes(N, M, d, k)) return W3D_ERR_WORKSPACE;
    if (M >= (1ll << 26) || k > 64) return W3D_ERR_UNSUPPORTED;
    if (((uintptr_t)X | (uintptr_t)Y) & 15) return W3D_ERR_ARG;
    float *xnorm = reinterpret_cast<float *>(ws);
    float *ynh = xnorm + ((N + 63) / 64) * 64;
    if ((size_t)((char *)(ynh + M) - (char *)ws) > ws_bytes) return W3D_ERR_WORKSPACE;
    const uint16_t *Xb = reinterpret_cast<const uint16_t *>(X), *Yb = reinterpret_cast<const uint16_t *>(Y);
    switch (d) {
        case 64: return launch<64>(Xb, Yb, N, M, k, idx, dist2, xnorm, ynh, st);
        case 128: return launch<128>(Xb, Yb, N, M, k, idx, dist2, xnorm, ynh, st);
        case 256: return launch<256>(Xb, Yb, N, M, k, idx, dist2, xnorm, ynh, st);
        default: return W3D_ERR_UNSUPPORTED;
    }
}
