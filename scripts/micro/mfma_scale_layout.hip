// Dev check: operand / scale lane maps of v_mfma_scale_f32_32x32x64_f8f6f4 with e4m3 operands (exact integer data, asymmetric B).
// Hypothesis: lane l holds A[row l&31][k = 32*(l>>5) + j] and B[k = 32*(l>>5) + j][col l&31] in byte j = 0..31 of its 8 VGPRs;
// scale operand: byte 0 (opsel 0) of the lane's VGPR is the E8M0 scale of that lane's row (A) / column (B) for its 32-wide K block.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
static uint8_t enc_e4m3(int v) {   // exact small non-negative integers 0..15
    if (v == 0) return 0;
    int e = 0; while ((1 << (e + 1)) <= v) ++e;        // v in [2^e, 2^(e+1))
    int mant = ((v << 3) >> e) & 7;                    // 3 mantissa bits (exact for v < 16)
    return (uint8_t)(((e + 7) << 3) | mant);
}
__global__ void k(const uint8_t *A /*[32][64]*/, const uint8_t *B /*[64][32]*/, const int *sa /*[32]*/, const int *sb /*[32]*/, float *D /*[32][32]*/) {
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    union { v8i v; uint8_t b[32]; } a, b;
    for (int j = 0; j < 32; ++j) { a.b[j] = A[r * 64 + 32 * h + j]; b.b[j] = B[(32 * h + j) * 32 + r]; }
    v16f c = {0};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a.v, b.v, c, 0, 0, 0, sa[r], 0, sb[r]);
    for (int reg = 0; reg < 16; ++reg) D[((reg & 3) + 8 * (reg >> 2) + 4 * h) * 32 + r] = c[reg];
}
int main() {
    uint8_t hA[32 * 64], hB[64 * 32]; int ia[32 * 64], ib[64 * 32], hsa[32], hsb[32];
    for (int i = 0; i < 32; ++i) for (int kk = 0; kk < 64; ++kk) { ia[i * 64 + kk] = (i + 2 * kk) % 7; hA[i * 64 + kk] = enc_e4m3(ia[i * 64 + kk]); }
    for (int kk = 0; kk < 64; ++kk) for (int j = 0; j < 32; ++j) { ib[kk * 32 + j] = (3 * kk + j) % 5; hB[kk * 32 + j] = enc_e4m3(ib[kk * 32 + j]); }
    for (int i = 0; i < 32; ++i) { hsa[i] = 127 + (i % 3); hsb[i] = 127 - (i % 2); }
    uint8_t *dA, *dB; int *dsa, *dsb; float *dD; float hD[32 * 32];
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dsa, sizeof hsa); hipMalloc(&dsb, sizeof hsb); hipMalloc(&dD, sizeof hD);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipMemcpy(dsa, hsa, sizeof hsa, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, sizeof hsb, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dA, dB, dsa, dsb, dD); hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        double s = 0; for (int kk = 0; kk < 64; ++kk) s += (double)ia[i * 64 + kk] * ib[kk * 32 + j];
        s *= std::ldexp(1.0, (i % 3)) * std::ldexp(1.0, -(j % 2));
        if (std::fabs(hD[i * 32 + j] - s) > 1e-3) { if (bad < 5) printf("mismatch D[%d][%d] = %f expected %f\n", i, j, hD[i * 32 + j], s); ++bad; }
    }
    printf("mfma_scale 32x32x64 e4m3 layout check: %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);
    return bad != 0;
}
