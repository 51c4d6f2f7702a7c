/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Plain-C, single-threaded CPU restatement of
 * the reference's differentiable Gaussian rasteriser fork (depth output + per-pixel sampling offsets).
 * Never linked into, imported by, or shipped with the product library.
 *
 * The reference implementation is CUDA-only (RAST = submodules/diff-gaussian-rasterization) and cannot be
 * built in this image (no nvcc): "parity unpinned" at whole-image level.  What pins this file:
 *   - sub-steps against the reference's own Python fallbacks (tests/golden/sh_cov_cam.npz: eval_sh,
 *     build_covariance_from_scaling_rotation, getProjectionMatrix/getWorld2View2);
 *   - fp64 finite differences of the forward against the analytic backward (tests/test_oracle_raster.py);
 *   - closed-form single-Gaussian cases.
 *
 * Lines restated (RAST/cuda_rasterizer/...):
 *   forward.cu:20-71   computeColorFromSH        forward.cu:74-113  computeCov2D
 *   forward.cu:118-152 computeCov3D              forward.cu:155-256 preprocessCUDA
 *   forward.cu:261-390 renderCUDA (fwd)          auxiliary.h:41-164 ndc2Pix/getRect/transform/in_frustum
 *   rasterizer_impl.cu:70-138 duplicateWithKeys/identifyTileRanges, :198-341 forward orchestration
 *   backward.cu:20-139 SH bwd   :144-274 computeCov2DCUDA   :278-341 computeCov3D bwd
 *   backward.cu:346-410 preprocessCUDA (bwd)     backward.cu:413-586 renderCUDA (bwd)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define BX 16
#define BY 16
#define BS 256

static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                               -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};

typedef struct {
    int P, D, M, H, W;
    const float *bg, *means3D, *shs, *colors_precomp, *opacities, *scales, *rotations, *cov3D_precomp;
    const float *view, *proj, *campos, *offsets;
    float scale_modifier, tan_fovx, tan_fovy;
} RIn;

/* per-Gaussian state kept between forward and backward (the reference's geomBuffer) */
typedef struct {
    float *depths, *means2D /*[P,2]*/, *cov3D /*[P,6]*/, *conic_opacity /*[P,4]*/, *rgb /*[P,3]*/;
    uint8_t *clamped /*[P,3]*/;
    int *radii;
    uint32_t *tiles_touched;
    /* binning */
    int64_t R;
    uint32_t *point_list;
    uint32_t *ranges; /* [tiles,2] */
    /* image */
    float *final_T;
    uint32_t *n_contrib;
} RState;

static float ndc2pix(float v, int S) { return (float)((((double)v + 1.0) * S - 1.0) * 0.5); }

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

static void get_rect(float px, float py, int max_radius, int gx, int gy, int *rmin, int *rmax) {
    rmin[0] = imin(gx, imax(0, (int)((px - max_radius) / BX)));
    rmin[1] = imin(gy, imax(0, (int)((py - max_radius) / BY)));
    rmax[0] = imin(gx, imax(0, (int)((px + max_radius + BX - 1) / BX)));
    rmax[1] = imin(gy, imax(0, (int)((py + max_radius + BY - 1) / BY)));
}

static void tp4x3(const float *p, const float *m, float *o) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}
static void tp4x4(const float *p, const float *m, float *o) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
    o[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}

/* 3x3 helpers, row-major a[r][c] in MATH convention (GLM's column-major constructors are unrolled by hand:
 * glm::mat3(a,b,c, d,e,f, g,h,i) has COLUMNS (a,b,c),(d,e,f),(g,h,i)). */
static void mm3(const float A[9], const float B[9], float C[9]) {
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) C[3 * r + c] = A[3 * r] * B[c] + A[3 * r + 1] * B[3 + c] + A[3 * r + 2] * B[6 + c];
}
static void tr3(const float A[9], float T[9]) {
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) T[3 * r + c] = A[3 * c + r];
}

/* forward.cu:118-152.  GLM: S diag; R = mat3(...) column-major => R_glm columns are the listed triples, i.e. the
 * math matrix Rm is the TRANSPOSE of the textbook rotation matrix laid out row-wise in the source.  M = S * R
 * (GLM, column-major product = math product), Sigma = transpose(M) * M. */
static void quat_to_Rglm(const float *q, float Rm[9]) {
    float r = q[0], x = q[1], y = q[2], z = q[3];
    /* textbook rows */
    float Rt[9] = {1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
                   2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                   2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)};
    /* glm::mat3(Rt...) fills columns with these triples: math matrix = transpose(Rt) */
    tr3(Rt, Rm);
}
static void compute_cov3D(const float *scale, float mod, const float *rot, float *cov) {
    float Rm[9];
    quat_to_Rglm(rot, Rm);
    float S[9] = {mod * scale[0], 0, 0, 0, mod * scale[1], 0, 0, 0, mod * scale[2]};
    float Mx[9], Mt[9], Sg[9];
    mm3(S, Rm, Mx);
    tr3(Mx, Mt);
    mm3(Mt, Mx, Sg);
    /* GLM Sigma[c][r]; symmetric, so index order is immaterial */
    cov[0] = Sg[0]; cov[1] = Sg[1]; cov[2] = Sg[2]; cov[3] = Sg[4]; cov[4] = Sg[5]; cov[5] = Sg[8];
}

/* forward.cu:74-113.  Returns cov2D (a,b,c) with the 0.3 low-pass; optionally the math matrix T (3x3). */
static void compute_cov2D(const float *mean, float fx, float fy, float tfx, float tfy, const float *cov3D, const float *view, float *out3, float *Tout) {
    float t[3];
    tp4x3(mean, view, t);
    const float limx = 1.3f * tfx, limy = 1.3f * tfy;
    const float txtz = t[0] / t[2], tytz = t[1] / t[2];
    t[0] = fminf(limx, fmaxf(-limx, txtz)) * t[2];
    t[1] = fminf(limy, fmaxf(-limy, tytz)) * t[2];
    /* glm J columns: (fx/tz,0,-(fx tx)/tz^2), (0,fy/tz,-(fy ty)/tz^2), (0,0,0)  => math J = transpose of listed rows */
    float Jl[9] = {fx / t[2], 0.f, -(fx * t[0]) / (t[2] * t[2]), 0.f, fy / t[2], -(fy * t[1]) / (t[2] * t[2]), 0.f, 0.f, 0.f};
    float Wl[9] = {view[0], view[4], view[8], view[1], view[5], view[9], view[2], view[6], view[10]};
    float J[9], Wm[9], T[9], Tt[9], V[9], tmp[9], cov[9];
    tr3(Jl, J);
    tr3(Wl, Wm);
    mm3(Wm, J, T); /* T = W * J (GLM) */
    float Vl[9] = {cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]};
    memcpy(V, Vl, sizeof(V)); /* symmetric */
    tr3(T, Tt);
    mm3(Tt, V, tmp);
    mm3(tmp, T, cov); /* transpose(T) * transpose(Vrk) * T */
    out3[0] = cov[0] + 0.3f;
    out3[1] = cov[3]; /* GLM cov[0][1] = column 0, row 1 = math (1,0) */
    out3[2] = cov[4] + 0.3f;
    if (Tout) memcpy(Tout, T, sizeof(T));
}

static void color_from_sh(int idx, int deg, int M, const float *means, const float *campos, const float *shs, uint8_t *clamped, float *out) {
    const float *pos = means + 3 * idx;
    float dir[3] = {pos[0] - campos[0], pos[1] - campos[1], pos[2] - campos[2]};
    float len = sqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
    dir[0] /= len; dir[1] /= len; dir[2] /= len;
    const float *sh = shs + (size_t)idx * M * 3;
    float x = dir[0], y = dir[1], z = dir[2];
    for (int c = 0; c < 3; c++) {
        float res = SH_C0 * sh[c];
        if (deg > 0) {
            res = res - SH_C1 * y * sh[3 + c] + SH_C1 * z * sh[6 + c] - SH_C1 * x * sh[9 + c];
            if (deg > 1) {
                float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                res = res + SH_C2[0] * xy * sh[12 + c] + SH_C2[1] * yz * sh[15 + c] + SH_C2[2] * (2.0f * zz - xx - yy) * sh[18 + c] +
                      SH_C2[3] * xz * sh[21 + c] + SH_C2[4] * (xx - yy) * sh[24 + c];
                if (deg > 2) {
                    res = res + SH_C3[0] * y * (3.0f * xx - yy) * sh[27 + c] + SH_C3[1] * xy * z * sh[30 + c] +
                          SH_C3[2] * y * (4.0f * zz - xx - yy) * sh[33 + c] + SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh[36 + c] +
                          SH_C3[4] * x * (4.0f * zz - xx - yy) * sh[39 + c] + SH_C3[5] * z * (xx - yy) * sh[42 + c] +
                          SH_C3[6] * x * (xx - 3.0f * yy) * sh[45 + c];
                }
            }
        }
        res += 0.5f;
        clamped[3 * idx + c] = (res < 0);
        out[c] = fmaxf(res, 0.0f);
    }
}

static RState *state_new(int P, int H, int W) {
    RState *s = (RState *)calloc(1, sizeof(RState));
    int tiles = ((W + BX - 1) / BX) * ((H + BY - 1) / BY);
    s->depths = calloc(P, 4); s->means2D = calloc(2 * (size_t)P, 4); s->cov3D = calloc(6 * (size_t)P, 4);
    s->conic_opacity = calloc(4 * (size_t)P, 4); s->rgb = calloc(3 * (size_t)P, 4); s->clamped = calloc(3 * (size_t)P, 1);
    s->radii = calloc(P, 4); s->tiles_touched = calloc(P, 4);
    s->ranges = calloc(2 * (size_t)tiles, 4);
    s->final_T = calloc((size_t)H * W, 4); s->n_contrib = calloc((size_t)H * W, 4);
    return s;
}
void raster_ref_free(RState *s) {
    if (!s) return;
    free(s->depths); free(s->means2D); free(s->cov3D); free(s->conic_opacity); free(s->rgb); free(s->clamped);
    free(s->radii); free(s->tiles_touched); free(s->point_list); free(s->ranges); free(s->final_T); free(s->n_contrib);
    free(s);
}

typedef struct { uint64_t key; uint32_t val; } KV;
static void merge_sort(KV *a, KV *tmp, int64_t n) { /* stable, like the radix sort the reference uses */
    if (n < 2) return;
    int64_t h = n / 2;
    merge_sort(a, tmp, h);
    merge_sort(a + h, tmp, n - h);
    int64_t i = 0, j = h, k = 0;
    while (i < h && j < n) tmp[k++] = (a[j].key < a[i].key) ? a[j++] : a[i++];
    while (i < h) tmp[k++] = a[i++];
    while (j < n) tmp[k++] = a[j++];
    memcpy(a, tmp, n * sizeof(KV));
}

/* accessors for tests */
int64_t raster_ref_num_rendered(const RState *s) { return s->R; }
const int *raster_ref_radii(const RState *s) { return s->radii; }
const float *raster_ref_rgb(const RState *s) { return s->rgb; }
const float *raster_ref_cov3D(const RState *s) { return s->cov3D; }
const float *raster_ref_means2D(const RState *s) { return s->means2D; }
const float *raster_ref_conic_opacity(const RState *s) { return s->conic_opacity; }
const float *raster_ref_depths(const RState *s) { return s->depths; }
const uint32_t *raster_ref_point_list(const RState *s) { return s->point_list; }
const uint32_t *raster_ref_ranges(const RState *s) { return s->ranges; }
const uint32_t *raster_ref_n_contrib(const RState *s) { return s->n_contrib; }
const float *raster_ref_final_T(const RState *s) { return s->final_T; }

RState *raster_ref_forward(const RIn *in, float *out_color, float *out_depth, int *radii_out) {
    const int P = in->P, H = in->H, W = in->W;
    const int gx = (W + BX - 1) / BX, gy = (H + BY - 1) / BY;
    const float focal_y = H / (2.0f * in->tan_fovy), focal_x = W / (2.0f * in->tan_fovx);
    RState *s = state_new(P, H, W);
    /* ---- preprocess (forward.cu:155-256) ---- */
    for (int idx = 0; idx < P; idx++) {
        s->radii[idx] = 0; s->tiles_touched[idx] = 0;
        const float *po = in->means3D + 3 * idx;
        float pv[3];
        tp4x3(po, in->view, pv);
        if (pv[2] <= 0.2f) continue; /* auxiliary.h:154 */
        float ph[4];
        tp4x4(po, in->proj, ph);
        float pw = 1.0f / (ph[3] + 0.0000001f);
        float pp[3] = {ph[0] * pw, ph[1] * pw, ph[2] * pw};
        const float *cov3D;
        if (in->cov3D_precomp) cov3D = in->cov3D_precomp + 6 * idx;
        else { compute_cov3D(in->scales + 3 * idx, in->scale_modifier, in->rotations + 4 * idx, s->cov3D + 6 * idx); cov3D = s->cov3D + 6 * idx; }
        float cov[3];
        compute_cov2D(po, focal_x, focal_y, in->tan_fovx, in->tan_fovy, cov3D, in->view, cov, NULL);
        float det = cov[0] * cov[2] - cov[1] * cov[1];
        if (det == 0.0f) continue;
        float det_inv = 1.f / det;
        float conic[3] = {cov[2] * det_inv, -cov[1] * det_inv, cov[0] * det_inv};
        float mid = 0.5f * (cov[0] + cov[2]);
        float l1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det)), l2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
        float my_radius = ceilf(3.f * sqrtf(fmaxf(l1, l2)));
        float pix[2] = {ndc2pix(pp[0], W), ndc2pix(pp[1], H)};
        int rmin[2], rmax[2];
        get_rect(pix[0], pix[1], (int)my_radius, gx, gy, rmin, rmax);
        if ((rmax[0] - rmin[0]) * (rmax[1] - rmin[1]) == 0) continue;
        if (!in->colors_precomp) color_from_sh(idx, in->D, in->M, in->means3D, in->campos, in->shs, s->clamped, s->rgb + 3 * idx);
        s->depths[idx] = pv[2];
        s->radii[idx] = (int)my_radius;
        s->means2D[2 * idx] = pix[0]; s->means2D[2 * idx + 1] = pix[1];
        s->conic_opacity[4 * idx] = conic[0]; s->conic_opacity[4 * idx + 1] = conic[1]; s->conic_opacity[4 * idx + 2] = conic[2];
        s->conic_opacity[4 * idx + 3] = in->opacities[idx];
        s->tiles_touched[idx] = (uint32_t)((rmax[1] - rmin[1]) * (rmax[0] - rmin[0]));
    }
    /* ---- binning (rasterizer_impl.cu:70-138,277-320) ---- */
    int64_t R = 0;
    for (int i = 0; i < P; i++) R += s->tiles_touched[i];
    s->R = R;
    KV *kv = (KV *)malloc((size_t)(R > 0 ? R : 1) * sizeof(KV)), *tmp = (KV *)malloc((size_t)(R > 0 ? R : 1) * sizeof(KV));
    int64_t off = 0;
    for (int idx = 0; idx < P; idx++) {
        if (s->radii[idx] <= 0) continue;
        int rmin[2], rmax[2];
        get_rect(s->means2D[2 * idx], s->means2D[2 * idx + 1], s->radii[idx], gx, gy, rmin, rmax);
        uint32_t dbits;
        memcpy(&dbits, &s->depths[idx], 4);
        for (int y = rmin[1]; y < rmax[1]; y++)
            for (int x = rmin[0]; x < rmax[0]; x++) {
                kv[off].key = ((uint64_t)(y * gx + x) << 32) | dbits;
                kv[off].val = (uint32_t)idx;
                off++;
            }
    }
    merge_sort(kv, tmp, R);
    s->point_list = (uint32_t *)malloc((size_t)(R > 0 ? R : 1) * 4);
    for (int64_t i = 0; i < R; i++) {
        s->point_list[i] = kv[i].val;
        uint32_t cur = (uint32_t)(kv[i].key >> 32);
        if (i == 0) s->ranges[2 * cur] = 0;
        else {
            uint32_t prev = (uint32_t)(kv[i - 1].key >> 32);
            if (cur != prev) { s->ranges[2 * prev + 1] = (uint32_t)i; s->ranges[2 * cur] = (uint32_t)i; }
        }
        if (i == R - 1) s->ranges[2 * cur + 1] = (uint32_t)R;
    }
    free(kv); free(tmp);
    /* ---- render (forward.cu:261-390); per pixel, same batch/termination semantics ---- */
    const float *feat = in->colors_precomp ? in->colors_precomp : s->rgb;
    for (int py = 0; py < H; py++)
        for (int px = 0; px < W; px++) {
            const int pix_id = W * py + px;
            const int tile = (py / BY) * gx + (px / BX);
            const uint32_t r0 = s->ranges[2 * tile], r1 = s->ranges[2 * tile + 1];
            const float pfx = (float)px + in->offsets[2 * pix_id], pfy = (float)py + in->offsets[2 * pix_id + 1];
            float T = 1.0f, C[3] = {0, 0, 0}, Dd = 0;
            uint32_t contributor = 0, last = 0;
            for (uint32_t i = r0; i < r1; i++) {
                contributor++;
                const uint32_t id = s->point_list[i];
                const float dx = s->means2D[2 * id] - pfx, dy = s->means2D[2 * id + 1] - pfy;
                const float *co = s->conic_opacity + 4 * id;
                const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                if (power > 0.0f) continue;
                const float alpha = fminf(0.99f, co[3] * expf(power));
                if (alpha < 1.0f / 255.0f) continue;
                const float test_T = T * (1 - alpha);
                if (test_T < 0.0001f) break; /* done = true */
                for (int ch = 0; ch < 3; ch++) C[ch] += feat[3 * id + ch] * alpha * T;
                Dd += s->depths[id] * alpha * T;
                T = test_T;
                last = contributor;
            }
            s->final_T[pix_id] = T;
            s->n_contrib[pix_id] = last;
            for (int ch = 0; ch < 3; ch++) out_color[ch * H * W + pix_id] = C[ch] + T * in->bg[ch];
            out_depth[pix_id] = Dd;
        }
    if (radii_out) memcpy(radii_out, s->radii, (size_t)P * 4);
    return s;
}

/* ------------------------------------------------------------------------------------------------ backward */
typedef struct {
    float *dL_dmeans2D /*[P,3]*/, *dL_dcolors /*[P,3]*/, *dL_dopacity /*[P]*/, *dL_dmeans3D /*[P,3]*/, *dL_dcov3D /*[P,6]*/,
        *dL_dsh /*[P,M,3]*/, *dL_dscales /*[P,3]*/, *dL_drots /*[P,4]*/;
} RGrad;

static void dnormvdv3(const float v[3], const float dv[3], float o[3]) {
    float sum2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    float inv = 1.0f / sqrtf(sum2 * sum2 * sum2);
    o[0] = ((+sum2 - v[0] * v[0]) * dv[0] - v[1] * v[0] * dv[1] - v[2] * v[0] * dv[2]) * inv;
    o[1] = (-v[0] * v[1] * dv[0] + (sum2 - v[1] * v[1]) * dv[1] - v[2] * v[1] * dv[2]) * inv;
    o[2] = (-v[0] * v[2] * dv[0] - v[1] * v[2] * dv[1] + (sum2 - v[2] * v[2]) * dv[2]) * inv;
}

void raster_ref_backward(const RIn *in, const RState *s, const float *dL_dpix, const float *dL_ddepth, RGrad *g) {
    const int P = in->P, H = in->H, W = in->W, M = in->M;
    const int gx = (W + BX - 1) / BX;
    const float focal_y = H / (2.0f * in->tan_fovy), focal_x = W / (2.0f * in->tan_fovx);
    double *a_col = calloc(3 * (size_t)P, 8), *a_m2d = calloc(2 * (size_t)P, 8), *a_con = calloc(3 * (size_t)P, 8), *a_op = calloc(P, 8),
           *a_dep = calloc(P, 8);
    const float *colors = in->colors_precomp ? in->colors_precomp : s->rgb;
    /* ---- render backward (backward.cu:413-586), accumulators in double (the reference uses float atomics) ---- */
    for (int py = 0; py < H; py++)
        for (int px = 0; px < W; px++) {
            const int pix_id = W * py + px;
            const int tile = (py / BY) * gx + (px / BX);
            const uint32_t r0 = s->ranges[2 * tile], r1 = s->ranges[2 * tile + 1];
            const float pfx = (float)px + in->offsets[2 * pix_id], pfy = (float)py + in->offsets[2 * pix_id + 1];
            const float T_final = s->final_T[pix_id];
            float T = T_final;
            const uint32_t last_contributor = s->n_contrib[pix_id];
            float accum_rec[3] = {0, 0, 0}, last_color[3] = {0, 0, 0}, last_alpha = 0;
            float dpx[3] = {dL_dpix[0 * H * W + pix_id], dL_dpix[1 * H * W + pix_id], dL_dpix[2 * H * W + pix_id]};
            const float ddep = dL_ddepth[pix_id];
            const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;
            uint32_t contributor = r1 - r0;
            for (uint32_t ii = 0; ii < r1 - r0; ii++) {
                const uint32_t id = s->point_list[r1 - ii - 1];
                contributor--;
                if (contributor >= last_contributor) continue;
                const float dx = s->means2D[2 * id] - pfx, dy = s->means2D[2 * id + 1] - pfy;
                const float *co = s->conic_opacity + 4 * id;
                const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                if (power > 0.0f) continue;
                const float G = expf(power);
                const float alpha = fminf(0.99f, co[3] * G);
                if (alpha < 1.0f / 255.0f) continue;
                T = T / (1.f - alpha);
                const float dchannel_dcolor = alpha * T;
                float dL_dalpha = 0.0f;
                for (int ch = 0; ch < 3; ch++) {
                    const float c = colors[3 * id + ch];
                    accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
                    last_color[ch] = c;
                    dL_dalpha += (c - accum_rec[ch]) * dpx[ch];
                    a_col[3 * id + ch] += dchannel_dcolor * dpx[ch];
                }
                a_dep[id] += dchannel_dcolor * ddep; /* depth gradient does NOT enter dL_dalpha (backward.cu:552) */
                dL_dalpha *= T;
                last_alpha = alpha;
                float bg_dot = 0;
                for (int i = 0; i < 3; i++) bg_dot += in->bg[i] * dpx[i];
                dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot;
                const float dL_dG = co[3] * dL_dalpha;
                const float gdx = G * dx, gdy = G * dy;
                const float dG_ddelx = -gdx * co[0] - gdy * co[1];
                const float dG_ddely = -gdy * co[2] - gdx * co[1];
                a_m2d[2 * id] += dL_dG * dG_ddelx * ddelx_dx;
                a_m2d[2 * id + 1] += dL_dG * dG_ddely * ddely_dy;
                a_con[3 * id] += -0.5f * gdx * dx * dL_dG;
                a_con[3 * id + 1] += -0.5f * gdx * dy * dL_dG;
                a_con[3 * id + 2] += -0.5f * gdy * dy * dL_dG;
                a_op[id] += G * dL_dalpha;
            }
        }
    for (int i = 0; i < P; i++) {
        g->dL_dmeans2D[3 * i] = (float)a_m2d[2 * i]; g->dL_dmeans2D[3 * i + 1] = (float)a_m2d[2 * i + 1]; g->dL_dmeans2D[3 * i + 2] = 0.f;
        for (int c = 0; c < 3; c++) g->dL_dcolors[3 * i + c] = (float)a_col[3 * i + c];
        g->dL_dopacity[i] = (float)a_op[i];
    }
    /* ---- computeCov2DCUDA (backward.cu:144-274) + preprocessCUDA bwd (:346-410) ---- */
    for (int idx = 0; idx < P; idx++) {
        for (int c = 0; c < 3; c++) g->dL_dmeans3D[3 * idx + c] = 0.f;
        for (int c = 0; c < 6; c++) g->dL_dcov3D[6 * idx + c] = 0.f;
        if (!(s->radii[idx] > 0)) continue;
        const float *cov3D = in->cov3D_precomp ? in->cov3D_precomp + 6 * idx : s->cov3D + 6 * idx;
        const float *mean = in->means3D + 3 * idx;
        const float dcon[3] = {(float)a_con[3 * idx], (float)a_con[3 * idx + 1], (float)a_con[3 * idx + 2]}; /* .x .y .w */
        float t[3];
        tp4x3(mean, in->view, t);
        const float limx = 1.3f * in->tan_fovx, limy = 1.3f * in->tan_fovy;
        const float txtz = t[0] / t[2], tytz = t[1] / t[2];
        t[0] = fminf(limx, fmaxf(-limx, txtz)) * t[2];
        t[1] = fminf(limy, fmaxf(-limy, tytz)) * t[2];
        const float x_grad_mul = (txtz < -limx || txtz > limx) ? 0 : 1, y_grad_mul = (tytz < -limy || tytz > limy) ? 0 : 1;
        float cov2[3], Tm[9];
        compute_cov2D(mean, focal_x, focal_y, in->tan_fovx, in->tan_fovy, cov3D, in->view, cov2, Tm);
        /* GLM T[c][r] = math Tm[r][c]: write Tg(c,r) for the reference's T[c][r] */
#define TG(c, r) Tm[3 * (r) + (c)]
        const float a = cov2[0], b = cov2[1], c_ = cov2[2];
        const float denom = a * c_ - b * b;
        float dL_da = 0, dL_db = 0, dL_dc = 0;
        const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
        float V[9] = {cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]};
#define VR(c, r) V[3 * (r) + (c)]
        if (denom2inv != 0) {
            dL_da = denom2inv * (-c_ * c_ * dcon[0] + 2 * b * c_ * dcon[1] + (denom - a * c_) * dcon[2]);
            dL_dc = denom2inv * (-a * a * dcon[2] + 2 * a * b * dcon[1] + (denom - a * c_) * dcon[0]);
            dL_db = denom2inv * 2 * (b * c_ * dcon[0] - (denom + 2 * b * b) * dcon[1] + a * b * dcon[2]);
            float *dc = g->dL_dcov3D + 6 * idx;
            dc[0] = (TG(0, 0) * TG(0, 0) * dL_da + TG(0, 0) * TG(1, 0) * dL_db + TG(1, 0) * TG(1, 0) * dL_dc);
            dc[3] = (TG(0, 1) * TG(0, 1) * dL_da + TG(0, 1) * TG(1, 1) * dL_db + TG(1, 1) * TG(1, 1) * dL_dc);
            dc[5] = (TG(0, 2) * TG(0, 2) * dL_da + TG(0, 2) * TG(1, 2) * dL_db + TG(1, 2) * TG(1, 2) * dL_dc);
            dc[1] = 2 * TG(0, 0) * TG(0, 1) * dL_da + (TG(0, 0) * TG(1, 1) + TG(0, 1) * TG(1, 0)) * dL_db + 2 * TG(1, 0) * TG(1, 1) * dL_dc;
            dc[2] = 2 * TG(0, 0) * TG(0, 2) * dL_da + (TG(0, 0) * TG(1, 2) + TG(0, 2) * TG(1, 0)) * dL_db + 2 * TG(1, 0) * TG(1, 2) * dL_dc;
            dc[4] = 2 * TG(0, 2) * TG(0, 1) * dL_da + (TG(0, 1) * TG(1, 2) + TG(0, 2) * TG(1, 1)) * dL_db + 2 * TG(1, 1) * TG(1, 2) * dL_dc;
        }
        float dL_dT00 = 2 * (TG(0, 0) * VR(0, 0) + TG(0, 1) * VR(0, 1) + TG(0, 2) * VR(0, 2)) * dL_da + (TG(1, 0) * VR(0, 0) + TG(1, 1) * VR(0, 1) + TG(1, 2) * VR(0, 2)) * dL_db;
        float dL_dT01 = 2 * (TG(0, 0) * VR(1, 0) + TG(0, 1) * VR(1, 1) + TG(0, 2) * VR(1, 2)) * dL_da + (TG(1, 0) * VR(1, 0) + TG(1, 1) * VR(1, 1) + TG(1, 2) * VR(1, 2)) * dL_db;
        float dL_dT02 = 2 * (TG(0, 0) * VR(2, 0) + TG(0, 1) * VR(2, 1) + TG(0, 2) * VR(2, 2)) * dL_da + (TG(1, 0) * VR(2, 0) + TG(1, 1) * VR(2, 1) + TG(1, 2) * VR(2, 2)) * dL_db;
        float dL_dT10 = 2 * (TG(1, 0) * VR(0, 0) + TG(1, 1) * VR(0, 1) + TG(1, 2) * VR(0, 2)) * dL_dc + (TG(0, 0) * VR(0, 0) + TG(0, 1) * VR(0, 1) + TG(0, 2) * VR(0, 2)) * dL_db;
        float dL_dT11 = 2 * (TG(1, 0) * VR(1, 0) + TG(1, 1) * VR(1, 1) + TG(1, 2) * VR(1, 2)) * dL_dc + (TG(0, 0) * VR(1, 0) + TG(0, 1) * VR(1, 1) + TG(0, 2) * VR(1, 2)) * dL_db;
        float dL_dT12 = 2 * (TG(1, 0) * VR(2, 0) + TG(1, 1) * VR(2, 1) + TG(1, 2) * VR(2, 2)) * dL_dc + (TG(0, 0) * VR(2, 0) + TG(0, 1) * VR(2, 1) + TG(0, 2) * VR(2, 2)) * dL_db;
        /* W GLM: W[c][r] with columns (view[0],view[4],view[8]),...  => W[c][r] = view[4*r + c] */
#define WG(c, r) in->view[4 * (r) + (c)]
        float dL_dJ00 = WG(0, 0) * dL_dT00 + WG(0, 1) * dL_dT01 + WG(0, 2) * dL_dT02;
        float dL_dJ02 = WG(2, 0) * dL_dT00 + WG(2, 1) * dL_dT01 + WG(2, 2) * dL_dT02;
        float dL_dJ11 = WG(1, 0) * dL_dT10 + WG(1, 1) * dL_dT11 + WG(1, 2) * dL_dT12;
        float dL_dJ12 = WG(2, 0) * dL_dT10 + WG(2, 1) * dL_dT11 + WG(2, 2) * dL_dT12;
        float tz = 1.f / t[2], tz2 = tz * tz, tz3 = tz2 * tz;
        float dL_dtx = x_grad_mul * -focal_x * tz2 * dL_dJ02;
        float dL_dty = y_grad_mul * -focal_y * tz2 * dL_dJ12;
        float dL_dtz = -focal_x * tz2 * dL_dJ00 - focal_y * tz2 * dL_dJ11 + (2 * focal_x * t[0]) * tz3 * dL_dJ02 + (2 * focal_y * t[1]) * tz3 * dL_dJ12;
        const float *vm = in->view;
        float dmean[3] = {vm[0] * dL_dtx + vm[1] * dL_dty + vm[2] * dL_dtz, vm[4] * dL_dtx + vm[5] * dL_dty + vm[6] * dL_dtz,
                          vm[8] * dL_dtx + vm[9] * dL_dty + vm[10] * dL_dtz};
        /* preprocess bwd */
        const float *pr = in->proj;
        float mh[4];
        tp4x4(mean, pr, mh);
        float m_w = 1.0f / (mh[3] + 0.0000001f);
        float mul1 = (pr[0] * mean[0] + pr[4] * mean[1] + pr[8] * mean[2] + pr[12]) * m_w * m_w;
        float mul2 = (pr[1] * mean[0] + pr[5] * mean[1] + pr[9] * mean[2] + pr[13]) * m_w * m_w;
        float d2x = g->dL_dmeans2D[3 * idx], d2y = g->dL_dmeans2D[3 * idx + 1];
        float dm2[3];
        dm2[0] = (pr[0] * m_w - pr[3] * mul1) * d2x + (pr[1] * m_w - pr[3] * mul2) * d2y;
        dm2[1] = (pr[4] * m_w - pr[7] * mul1) * d2x + (pr[5] * m_w - pr[7] * mul2) * d2y;
        dm2[2] = (pr[8] * m_w - pr[11] * mul1) * d2x + (pr[9] * m_w - pr[11] * mul2) * d2y;
        const float dcv = (float)a_dep[idx];
        dm2[0] += vm[2] * dcv; dm2[1] += vm[6] * dcv; dm2[2] += vm[10] * dcv; /* backward.cu:395-397 */
        for (int c = 0; c < 3; c++) dmean[c] += dm2[c];
        /* SH backward (backward.cu:20-139) */
        if (in->shs) {
            const float *sh = in->shs + (size_t)idx * M * 3;
            float *dsh = g->dL_dsh + (size_t)idx * M * 3;
            float dir_orig[3] = {mean[0] - in->campos[0], mean[1] - in->campos[1], mean[2] - in->campos[2]};
            float len = sqrtf(dir_orig[0] * dir_orig[0] + dir_orig[1] * dir_orig[1] + dir_orig[2] * dir_orig[2]);
            float x = dir_orig[0] / len, y = dir_orig[1] / len, z = dir_orig[2] / len;
            float dRGB[3];
            for (int c = 0; c < 3; c++) dRGB[c] = g->dL_dcolors[3 * idx + c] * (s->clamped[3 * idx + c] ? 0.f : 1.f);
            float dx_[3] = {0, 0, 0}, dy_[3] = {0, 0, 0}, dz_[3] = {0, 0, 0};
#define SH(i) (sh + 3 * (i))
            for (int c = 0; c < 3; c++) dsh[c] = SH_C0 * dRGB[c];
            if (in->D > 0) {
                for (int c = 0; c < 3; c++) {
                    dsh[3 + c] = -SH_C1 * y * dRGB[c]; dsh[6 + c] = SH_C1 * z * dRGB[c]; dsh[9 + c] = -SH_C1 * x * dRGB[c];
                    dx_[c] = -SH_C1 * SH(3)[c]; dy_[c] = -SH_C1 * SH(1)[c]; dz_[c] = SH_C1 * SH(2)[c];
                }
                if (in->D > 1) {
                    float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                    for (int c = 0; c < 3; c++) {
                        dsh[12 + c] = SH_C2[0] * xy * dRGB[c]; dsh[15 + c] = SH_C2[1] * yz * dRGB[c];
                        dsh[18 + c] = SH_C2[2] * (2.f * zz - xx - yy) * dRGB[c]; dsh[21 + c] = SH_C2[3] * xz * dRGB[c];
                        dsh[24 + c] = SH_C2[4] * (xx - yy) * dRGB[c];
                        dx_[c] += SH_C2[0] * y * SH(4)[c] + SH_C2[2] * 2.f * -x * SH(6)[c] + SH_C2[3] * z * SH(7)[c] + SH_C2[4] * 2.f * x * SH(8)[c];
                        dy_[c] += SH_C2[0] * x * SH(4)[c] + SH_C2[1] * z * SH(5)[c] + SH_C2[2] * 2.f * -y * SH(6)[c] + SH_C2[4] * 2.f * -y * SH(8)[c];
                        dz_[c] += SH_C2[1] * y * SH(5)[c] + SH_C2[2] * 2.f * 2.f * z * SH(6)[c] + SH_C2[3] * x * SH(7)[c];
                    }
                    if (in->D > 2) {
                        for (int c = 0; c < 3; c++) {
                            dsh[27 + c] = SH_C3[0] * y * (3.f * xx - yy) * dRGB[c]; dsh[30 + c] = SH_C3[1] * xy * z * dRGB[c];
                            dsh[33 + c] = SH_C3[2] * y * (4.f * zz - xx - yy) * dRGB[c];
                            dsh[36 + c] = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy) * dRGB[c];
                            dsh[39 + c] = SH_C3[4] * x * (4.f * zz - xx - yy) * dRGB[c]; dsh[42 + c] = SH_C3[5] * z * (xx - yy) * dRGB[c];
                            dsh[45 + c] = SH_C3[6] * x * (xx - 3.f * yy) * dRGB[c];
                            dx_[c] += (SH_C3[0] * SH(9)[c] * 3.f * 2.f * xy + SH_C3[1] * SH(10)[c] * yz + SH_C3[2] * SH(11)[c] * -2.f * xy +
                                       SH_C3[3] * SH(12)[c] * -3.f * 2.f * xz + SH_C3[4] * SH(13)[c] * (-3.f * xx + 4.f * zz - yy) +
                                       SH_C3[5] * SH(14)[c] * 2.f * xz + SH_C3[6] * SH(15)[c] * 3.f * (xx - yy));
                            dy_[c] += (SH_C3[0] * SH(9)[c] * 3.f * (xx - yy) + SH_C3[1] * SH(10)[c] * xz + SH_C3[2] * SH(11)[c] * (-3.f * yy + 4.f * zz - xx) +
                                       SH_C3[3] * SH(12)[c] * -3.f * 2.f * yz + SH_C3[4] * SH(13)[c] * -2.f * xy + SH_C3[5] * SH(14)[c] * -2.f * yz +
                                       SH_C3[6] * SH(15)[c] * -3.f * 2.f * xy);
                            dz_[c] += (SH_C3[1] * SH(10)[c] * xy + SH_C3[2] * SH(11)[c] * 4.f * 2.f * yz + SH_C3[3] * SH(12)[c] * 3.f * (2.f * zz - xx - yy) +
                                       SH_C3[4] * SH(13)[c] * 4.f * 2.f * xz + SH_C3[5] * SH(14)[c] * (xx - yy));
                        }
                    }
                }
            }
            float ddir[3] = {dx_[0] * dRGB[0] + dx_[1] * dRGB[1] + dx_[2] * dRGB[2], dy_[0] * dRGB[0] + dy_[1] * dRGB[1] + dy_[2] * dRGB[2],
                             dz_[0] * dRGB[0] + dz_[1] * dRGB[1] + dz_[2] * dRGB[2]};
            float dmm[3];
            dnormvdv3(dir_orig, ddir, dmm);
            for (int c = 0; c < 3; c++) dmean[c] += dmm[c];
        }
        for (int c = 0; c < 3; c++) g->dL_dmeans3D[3 * idx + c] = dmean[c];
        /* cov3D backward (backward.cu:278-341) */
        if (in->scales) {
            const float *q = in->rotations + 4 * idx;
            float r = q[0], x = q[1], y = q[2], z = q[3];
            float Rm[9];
            quat_to_Rglm(q, Rm);
            float sc[3] = {in->scale_modifier * in->scales[3 * idx], in->scale_modifier * in->scales[3 * idx + 1], in->scale_modifier * in->scales[3 * idx + 2]};
            float S[9] = {sc[0], 0, 0, 0, sc[1], 0, 0, 0, sc[2]}, Mx[9];
            mm3(S, Rm, Mx);
            const float *dc = g->dL_dcov3D + 6 * idx;
            float dSig[9] = {dc[0], 0.5f * dc[1], 0.5f * dc[2], 0.5f * dc[1], dc[3], 0.5f * dc[4], 0.5f * dc[2], 0.5f * dc[4], dc[5]};
            float dM[9], tmp9[9];
            mm3(Mx, dSig, tmp9);
            for (int i = 0; i < 9; i++) dM[i] = 2.0f * tmp9[i];
            /* GLM: Rt = transpose(R), dL_dMt = transpose(dL_dM); Rt[i] is COLUMN i of Rt = ROW i of R(math)... in math
             * terms: Rt[i][j] (col i,row j) = Rt_math[j][i] = Rm[i][j]; dL_dMt[i][j] = dM[i][j]. */
            float *ds = g->dL_dscales + 3 * idx;
            for (int i = 0; i < 3; i++) ds[i] = Rm[3 * i] * dM[3 * i] + Rm[3 * i + 1] * dM[3 * i + 1] + Rm[3 * i + 2] * dM[3 * i + 2];
            float dMt[9];
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) dMt[3 * i + j] = dM[3 * i + j] * sc[i]; /* dL_dMt[i] *= s_i ; element [i][j] */
#define DMT(i, j) dMt[3 * (i) + (j)]
            float *dq = g->dL_drots + 4 * idx;
            dq[0] = 2 * z * (DMT(0, 1) - DMT(1, 0)) + 2 * y * (DMT(2, 0) - DMT(0, 2)) + 2 * x * (DMT(1, 2) - DMT(2, 1));
            dq[1] = 2 * y * (DMT(1, 0) + DMT(0, 1)) + 2 * z * (DMT(2, 0) + DMT(0, 2)) + 2 * r * (DMT(1, 2) - DMT(2, 1)) - 4 * x * (DMT(2, 2) + DMT(1, 1));
            dq[2] = 2 * x * (DMT(1, 0) + DMT(0, 1)) + 2 * r * (DMT(2, 0) - DMT(0, 2)) + 2 * z * (DMT(1, 2) + DMT(2, 1)) - 4 * y * (DMT(2, 2) + DMT(0, 0));
            dq[3] = 2 * r * (DMT(0, 1) - DMT(1, 0)) + 2 * x * (DMT(2, 0) + DMT(0, 2)) + 2 * y * (DMT(1, 2) + DMT(2, 1)) - 4 * z * (DMT(1, 1) + DMT(0, 0));
            /* the cov3D-bwd uses s = mod*scale for dL_dMt scaling but dL_dscale is w.r.t. the unscaled parameter? backward.cu:323-325
             * takes dot(Rt[i], dL_dMt[i]) BEFORE the *= s — already done above (no modifier factor, as in the reference). */
        }
    }
    free(a_col); free(a_m2d); free(a_con); free(a_op); free(a_dep);
}

void raster_ref_mark_visible(const float *means3D, const float *view, const float *proj, int P, uint8_t *present) {
    (void)proj;
    for (int i = 0; i < P; i++) {
        float pv[3];
        tp4x3(means3D + 3 * i, view, pv);
        present[i] = pv[2] > 0.2f;
    }
}
