// micro-benchmark (scripts/under_sweep_probe.py): a pure HBM stream in <= 32 VGPRs and no LDS — does it run UNDER the persistent emit sweep (two 240-VGPR waves per SIMD)?
#include <hip/hip_runtime.h>
#include <stdint.h>
__global__ __launch_bounds__(256) void stream_kernel(float2 *p, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) { float2 v = p[i]; v.x = v.x * 1.0001f + 1e-6f; v.y = v.y * 0.9999f; p[i] = v; }
}
extern "C" int stream_launch(void *p, int64_t n, int blocks, void *stream) {
    hipLaunchKernelGGL(stream_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (float2 *)p, n);
    return (int)hipGetLastError();
}
