// Dev micro-benchmark: LDS atomic throughput by type on gfx950 (random addresses, 16 waves per CU, 156 KB window).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE>
__global__ __launch_bounds__(1024) void k(int iters, int win, uint32_t *out) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    for (int i = threadIdx.x; i < win * 3; i += 1024) lds[i] = 0;
    __syncthreads();
    uint32_t s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u;
        const uint32_t p = (s >> 8) % (uint32_t)win;
        if (MODE == 0) {
            float *a = reinterpret_cast<float *>(lds) + 3 * p;
            __hip_atomic_fetch_add(a, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(a + 1, 2.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(a + 2, 3.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (MODE == 1) {
            uint32_t *a = lds + 3 * p;
            __hip_atomic_fetch_add(a, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(a + 1, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(a + 2, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (MODE == 2) {   // one u64 (two packed fields) + one u32, layout [win] u64 | [win] u32
            unsigned long long *a = reinterpret_cast<unsigned long long *>(lds) + p;
            uint32_t *b = lds + 2 * win + p;
            __hip_atomic_fetch_add(a, 0x100000002ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(b, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (MODE == 3) {   // plain (non-atomic) read-modify-write: upper bound
            uint32_t *a = lds + 3 * p;
            a[0] += 1u; a[1] += 2u; a[2] += 3u;
        } else if (MODE == 4) {   // f32, 4-float stride (bank pattern check)
            float *a = reinterpret_cast<float *>(lds) + 3 * p;
            __hip_atomic_fetch_add(a, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (MODE == 5) {   // pk f16 x2 + f16... : ds_pk_add_f16 rate
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            h2 *a = reinterpret_cast<h2 *>(lds) + 3 * p;
            h2 v = {(_Float16)1.0f, (_Float16)2.0f};
            __builtin_amdgcn_ds_atomic_fadd_v2f16((__attribute__((address_space(3))) h2 *)a, v);
        }
    }
    __syncthreads();
    uint32_t acc = 0;
    for (int i = threadIdx.x; i < win * 3; i += 1024) acc += lds[i];
    if (acc == 0x12345678u) out[0] = acc;
}
template <int MODE>
void run(const char *name, int perlane) {
    const int win = 13312, iters = 4096; uint32_t *out; hipMalloc(&out, 4);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, win * 12);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<256, 1024, win * 12>>>(iters, win, out); hipDeviceSynchronize();
    hipEventRecord(e0); k<MODE><<<256, 1024, win * 12>>>(iters, win, out); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ops = 256.0 * 1024 * iters * perlane;
    printf("%-28s %7.3f ms  %8.1f G lane-atomics/s  (%0.2f per clk per CU @2.1GHz)\n", name, ms, ops / ms / 1e6, ops / ms / 1e6 / 256 / 2.1);
}
int main() {
    run<0>("ds_add_f32 x3", 3); run<1>("ds_add_u32 x3", 3); run<2>("ds_add_u64 + ds_add_u32", 2); run<3>("plain RMW x3", 3); run<4>("ds_add_f32 x1", 1);
    run<5>("ds_pk_add_f16 x1", 1);
    return 0;
}
