// Per-Gaussian maths of the splatting pipeline, written out in scalar form (no matrix library).
// Semantics follow RAST/cuda_rasterizer/forward.cu:20-152 and backward.cu:20-341; matrices arrive transposed
// (row-vector convention, scene/cameras.py:54-57) and are indexed m[4*col + row].
#pragma once
#include "common.h"

#define TILE_X 16
#define TILE_Y 16
#define TILE_PIX 256

namespace rm {

__device__ __constant__ const float kSH0 = 0.28209479177387814f;
__device__ __constant__ const float kSH1 = 0.4886025119029199f;
__device__ __constant__ const float kSH2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f, 0.5462742152960396f};
__device__ __constant__ const float kSH3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                                               -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};

struct V3 { float x, y, z; };

__device__ __forceinline__ V3 xform43(const float *m, V3 p) {   // auxiliary.h:58-66
    return {m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13], m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]};
}
__device__ __forceinline__ float4 xform44(const float *m, V3 p) {   // auxiliary.h:68-77
    return make_float4(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
                       m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14], m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15]);
}
__device__ __forceinline__ float ndc_to_pix(float v, int S) { return (float)((((double)v + 1.0) * S - 1.0) * 0.5); }   // auxiliary.h:41-44

__device__ __forceinline__ void tile_rect(float px, float py, int radius, int gx, int gy, int &x0, int &y0, int &x1, int &y1) {   // auxiliary.h:46-56
    x0 = min(gx, max(0, (int)((px - radius) / TILE_X)));
    y0 = min(gy, max(0, (int)((py - radius) / TILE_Y)));
    x1 = min(gx, max(0, (int)((px + radius + TILE_X - 1) / TILE_X)));
    y1 = min(gy, max(0, (int)((py + radius + TILE_Y - 1) / TILE_Y)));
}

// Rotation (textbook, row-major) of an UN-normalised quaternion (r,x,y,z): forward.cu:127 leaves normalisation to the caller
__device__ __forceinline__ void quat_rows(const float *q, float R[9]) {
    const float r = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = 1.f - 2.f * (y * y + z * z); R[1] = 2.f * (x * y - r * z); R[2] = 2.f * (x * z + r * y);
    R[3] = 2.f * (x * y + r * z); R[4] = 1.f - 2.f * (x * x + z * z); R[5] = 2.f * (y * z - r * x);
    R[6] = 2.f * (x * z - r * y); R[7] = 2.f * (y * z + r * x); R[8] = 1.f - 2.f * (x * x + y * y);
}

// Sigma = R diag(s)^2 R^T, evaluated as M^T M with M[k][i] = s_k R[i][k] (forward.cu:140-143)
__device__ __forceinline__ void cov3d_from_scale_rot(const float *scale, float mod, const float *q, float cov[6]) {
    float R[9];
    quat_rows(q, R);
    const float s[3] = {mod * scale[0], mod * scale[1], mod * scale[2]};
    float M[9];   // M[k*3+i]
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int i = 0; i < 3; ++i) M[3 * k + i] = s[k] * R[3 * i + k];
    auto dotc = [&](int i, int j) { return M[i] * M[j] + M[3 + i] * M[3 + j] + M[6 + i] * M[6 + j]; };
    cov[0] = dotc(0, 0); cov[1] = dotc(0, 1); cov[2] = dotc(0, 2); cov[3] = dotc(1, 1); cov[4] = dotc(1, 2); cov[5] = dotc(2, 2);
}

// EWA projection.  T (3x2 used part): T[r][0], T[r][1] for r = 0..2; returns cov2D entries with the 0.3 px low-pass.
struct Cov2D { float a, b, c; float T0[3], T1[3]; float tx, ty, tz; float txtz, tytz; };
__device__ __forceinline__ Cov2D project_cov(V3 mean, float fx, float fy, float tanx, float tany, const float *cov3D, const float *view) {
    Cov2D o;
    V3 t = xform43(view, mean);
    const float limx = 1.3f * tanx, limy = 1.3f * tany;
    o.txtz = t.x / t.z; o.tytz = t.y / t.z;
    t.x = fminf(limx, fmaxf(-limx, o.txtz)) * t.z;
    t.y = fminf(limy, fmaxf(-limy, o.tytz)) * t.z;
    o.tx = t.x; o.ty = t.y; o.tz = t.z;
    const float j00 = fx / t.z, j11 = fy / t.z, j20 = -(fx * t.x) / (t.z * t.z), j21 = -(fy * t.y) / (t.z * t.z);
    // W(r,k) = view[4r + k]  (world->view rotation, transposed storage)
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        o.T0[r] = view[4 * r + 0] * j00 + view[4 * r + 2] * j20;
        o.T1[r] = view[4 * r + 1] * j11 + view[4 * r + 2] * j21;
    }
    const float V[9] = {cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]};
    float u0[3], u1[3];   // (T^T V) rows 0 and 1
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        u0[c] = o.T0[0] * V[c] + o.T0[1] * V[3 + c] + o.T0[2] * V[6 + c];
        u1[c] = o.T1[0] * V[c] + o.T1[1] * V[3 + c] + o.T1[2] * V[6 + c];
    }
    o.a = u0[0] * o.T0[0] + u0[1] * o.T0[1] + u0[2] * o.T0[2] + 0.3f;
    o.b = u1[0] * o.T0[0] + u1[1] * o.T0[1] + u1[2] * o.T0[2];
    o.c = u1[0] * o.T1[0] + u1[1] * o.T1[1] + u1[2] * o.T1[2] + 0.3f;
    return o;
}

// SH -> RGB (forward.cu:20-71); sh0 = the DC coefficient [3] of this Gaussian, sh[3*i + c] = coefficient i >= 1 (one [M][3] array: sh == sh0;
// DC and higher orders stored apart, as the reference's _features_dc / _features_rest: sh = rest - 3); returns clamp bits in `clampmask`
__device__ __forceinline__ V3 sh_to_rgb(int deg, const float *sh0, const float *sh, V3 pos, V3 cam, uint32_t &clampmask) {
    float dx = pos.x - cam.x, dy = pos.y - cam.y, dz = pos.z - cam.z;
    const float len = sqrtf(dx * dx + dy * dy + dz * dz);
    const float x = dx / len, y = dy / len, z = dz / len;
    float res[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float r = kSH0 * sh0[c];
        if (deg > 0) {
            r = r - kSH1 * y * sh[3 + c] + kSH1 * z * sh[6 + c] - kSH1 * x * sh[9 + c];
            if (deg > 1) {
                const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                r = r + kSH2[0] * xy * sh[12 + c] + kSH2[1] * yz * sh[15 + c] + kSH2[2] * (2.0f * zz - xx - yy) * sh[18 + c] + kSH2[3] * xz * sh[21 + c] +
                    kSH2[4] * (xx - yy) * sh[24 + c];
                if (deg > 2) {
                    r = r + kSH3[0] * y * (3.0f * xx - yy) * sh[27 + c] + kSH3[1] * xy * z * sh[30 + c] + kSH3[2] * y * (4.0f * zz - xx - yy) * sh[33 + c] +
                        kSH3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh[36 + c] + kSH3[4] * x * (4.0f * zz - xx - yy) * sh[39 + c] +
                        kSH3[5] * z * (xx - yy) * sh[42 + c] + kSH3[6] * x * (xx - 3.0f * yy) * sh[45 + c];
                }
            }
        }
        res[c] = r + 0.5f;
    }
    clampmask = (res[0] < 0.f ? 1u : 0u) | (res[1] < 0.f ? 2u : 0u) | (res[2] < 0.f ? 4u : 0u);
    return {fmaxf(res[0], 0.f), fmaxf(res[1], 0.f), fmaxf(res[2], 0.f)};
}

}  // namespace rm
