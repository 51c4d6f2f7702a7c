// Which SIMD does wave w of a 512-thread workgroup land on?  (The out-of-phase schedule of csrc/knn_mfma.hip pairs wave w with wave w + 4.)
// hipcc --offload-arch=gfx950 -O2 scripts/micro/wave_simd_map.hip -o /tmp/wave_simd_map && /tmp/wave_simd_map
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void k(unsigned *out) {
    extern __shared__ char lds[];
    lds[threadIdx.x] = 1;                                            // 150 KB of dynamic LDS: one workgroup per CU, as the kNN kernel
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = hw;
}
int main() {
    unsigned *d, h[8 * 8];
    hipMalloc(&d, sizeof(h));
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipLaunchKernelGGL(k, dim3(8), dim3(512), 150 * 1024, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int b = 0; b < 8; ++b) {
        printf("block %d:", b);
        for (int w = 0; w < 8; ++w) printf("  w%d simd %u slot %u cu %u", w, (h[b * 8 + w] >> 4) & 3, h[b * 8 + w] & 15, (h[b * 8 + w] >> 8) & 15);
        printf("\n");
    }
    return 0;
}
