// Per-(view, Gaussian) math shared by the modular kernels (lbs_sh.hip, raster_*.hip) and the
// fused per-instance kernels (k_inst_fwd in raster_fwd.hip, k_inst_gather / k_inst_bwd in raster_bwd.hip): LBS of mean/covariance, SH colour, EWA projection,
// and their analytic backward passes.  All functions are register-only device inlines.
//
// Reference semantics (brown-ivl/manus):
//   LBS            src/modules/hand_dynamic.py:106-127
//   covariance     src/models/gaussian.py:49-53,84-93; src/utils/gaussian_utils.py:279-314
//   SH colour      src/utils/gaussian_utils.py:431-449; src/utils/sh_utils.py:57-104
//   projection     external rasterizer, SURVEY.md Appendix A
#pragma once
#include <hip/hip_fp16.h>

#include "mgr_common.h"

#ifdef __HIPCC__

// ---------------------------------------------------------------------------
// canonical Gaussian: position, normalised quaternion -> R, exp(log scale)
// ---------------------------------------------------------------------------
struct GaussCano {
    float x, y, z;
    float q[4], nrm, R[9], s[3];
};

__device__ __forceinline__ void quat_rot(const float q[4], float R[9]) {
#pragma clang fp contract(off)
    const float r = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = 1.f - 2.f * (y * y + z * z); R[1] = 2.f * (x * y - r * z); R[2] = 2.f * (x * z + r * y);
    R[3] = 2.f * (x * y + r * z); R[4] = 1.f - 2.f * (x * x + z * z); R[5] = 2.f * (y * z - r * x);
    R[6] = 2.f * (x * z - r * y); R[7] = 2.f * (y * z + r * x); R[8] = 1.f - 2.f * (x * x + y * y);
}

// Row loads of the (N,3) / (N,B) / (N,45) parameter arrays: rows are only 4-byte aligned, but gfx950
// global loads take unaligned dwordx3/x4, and one wide load costs the texture path one pass
// instead of three or four.
typedef float mgr_f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float mgr_f3u __attribute__((ext_vector_type(3), aligned(4)));

__device__ __forceinline__ void cano_load(const float* __restrict__ xyz, const float* __restrict__ log_scale,
                                          const float* __restrict__ rot, int i, GaussCano& g) {
#pragma clang fp contract(off)
    const mgr_f3u p = *(const mgr_f3u*)(xyz + 3 * (size_t)i);
    const mgr_f3u ls = *(const mgr_f3u*)(log_scale + 3 * (size_t)i);
    const mgr_f4u q4 = *(const mgr_f4u*)(rot + 4 * (size_t)i);
    g.x = p.x; g.y = p.y; g.z = p.z;
    const float qr[4] = {q4.x, q4.y, q4.z, q4.w};
    g.nrm = sqrtf(qr[0] * qr[0] + qr[1] * qr[1] + qr[2] * qr[2] + qr[3] * qr[3]);
#pragma unroll
    for (int k = 0; k < 4; ++k) g.q[k] = qr[k] / g.nrm;
    quat_rot(g.q, g.R);
    g.s[0] = expf(ls.x); g.s[1] = expf(ls.y); g.s[2] = expf(ls.z);
}

// tf rows 0..2 (3x4, row-major) = sum_b w_b * T_b ; identity when w == nullptr
__device__ __forceinline__ void blend_tf(const float* __restrict__ w_row, const float* __restrict__ Tp,
                                         int B, float tf[12]) {
#pragma clang fp contract(off)
    if (w_row == nullptr) {
#pragma unroll
        for (int k = 0; k < 12; ++k) tf[k] = (k == 0 || k == 5 || k == 10) ? 1.f : 0.f;
        return;
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) tf[k] = 0.f;
    int b = 0;
    for (; b + 4 <= B; b += 4) {  // four weights per load
        const mgr_f4u w4 = *(const mgr_f4u*)(w_row + b);
        const float wj[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float* T = Tp + (size_t)(b + j) * 16;  // wave-uniform address
#pragma unroll
            for (int k = 0; k < 12; ++k) tf[k] = __builtin_fmaf(wj[j], T[k], tf[k]);  // explicit fma: the same rounding in
        }                                                                            // every kernel, half the instructions
    }
    for (; b < B; ++b) {
        const float w = w_row[b];
        const float* T = Tp + (size_t)b * 16;
#pragma unroll
        for (int k = 0; k < 12; ++k) tf[k] = __builtin_fmaf(w, T[k], tf[k]);
    }
}

// posed = A x + t ; Sigma' = (A R S)(A R S)^T packed [xx,xy,xz,yy,yz,zz]
__device__ __forceinline__ void lbs_apply(const float tf[12], const GaussCano& g, float posed[3], float cov6[6]) {
#pragma clang fp contract(off)
    posed[0] = tf[0] * g.x + tf[1] * g.y + tf[2] * g.z + tf[3];
    posed[1] = tf[4] * g.x + tf[5] * g.y + tf[6] * g.z + tf[7];
    posed[2] = tf[8] * g.x + tf[9] * g.y + tf[10] * g.z + tf[11];
    float M[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
            M[3 * r + c] = (tf[4 * r] * g.R[c] + tf[4 * r + 1] * g.R[3 + c] + tf[4 * r + 2] * g.R[6 + c]) * g.s[c];
    cov6[0] = M[0] * M[0] + M[1] * M[1] + M[2] * M[2];
    cov6[1] = M[0] * M[3] + M[1] * M[4] + M[2] * M[5];
    cov6[2] = M[0] * M[6] + M[1] * M[7] + M[2] * M[8];
    cov6[3] = M[3] * M[3] + M[4] * M[4] + M[5] * M[5];
    cov6[4] = M[3] * M[6] + M[4] * M[7] + M[5] * M[8];
    cov6[5] = M[6] * M[6] + M[7] * M[7] + M[8] * M[8];
}

// One pose/view of the LBS backward: accumulates into dxyz, ds (d/ds, not d/dlog s), dR and
// returns dtf (3x4) = gradient w.r.t. the blended transform (incl. the optional extra g_tf).
template <bool HAS_GTF>
__device__ __forceinline__ void lbs_backward_view(const float tf[12], const GaussCano& g, const float gp[3],
                                                  const float g6[6], const float g_tf[12], float dxyz[3],
                                                  float ds[3], float dR[9], float dtf[12]) {
    const float Gs[9] = {2.f * g6[0], g6[1], g6[2], g6[1], 2.f * g6[3], g6[4], g6[2], g6[4], 2.f * g6[5]};
    float L[9], AL[9], dM[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) L[3 * r + c] = g.R[3 * r + c] * g.s[c];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
            AL[3 * r + c] = tf[4 * r] * L[c] + tf[4 * r + 1] * L[3 + c] + tf[4 * r + 2] * L[6 + c];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
            dM[3 * r + c] = Gs[3 * r] * AL[c] + Gs[3 * r + 1] * AL[3 + c] + Gs[3 * r + 2] * AL[6 + c];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float dLrc = tf[r] * dM[c] + tf[4 + r] * dM[3 + c] + tf[8 + r] * dM[6 + c];
            ds[c] += dLrc * g.R[3 * r + c];
            dR[3 * r + c] += dLrc * g.s[c];
            dtf[4 * r + c] = dM[3 * r] * L[3 * c] + dM[3 * r + 1] * L[3 * c + 1] + dM[3 * r + 2] * L[3 * c + 2];
        }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        dtf[4 * r + 0] += gp[r] * g.x;
        dtf[4 * r + 1] += gp[r] * g.y;
        dtf[4 * r + 2] += gp[r] * g.z;
        dtf[4 * r + 3] = gp[r];
    }
    dxyz[0] += tf[0] * gp[0] + tf[4] * gp[1] + tf[8] * gp[2];
    dxyz[1] += tf[1] * gp[0] + tf[5] * gp[1] + tf[9] * gp[2];
    dxyz[2] += tf[2] * gp[0] + tf[6] * gp[1] + tf[10] * gp[2];
    if (HAS_GTF) {
#pragma unroll
        for (int k = 0; k < 12; ++k) dtf[k] += g_tf[k];
    }
}

// dR -> gradient w.r.t. the RAW (un-normalised) quaternion
__device__ __forceinline__ void quat_backward(const GaussCano& g, const float dR[9], float drot[4]) {
    const float r = g.q[0], qx = g.q[1], qy = g.q[2], qz = g.q[3];
    float dq[4];
    dq[0] = 2.f * (-qz * dR[1] + qy * dR[2] + qz * dR[3] - qx * dR[5] - qy * dR[6] + qx * dR[7]);
    dq[1] = 2.f * (qy * dR[1] + qz * dR[2] + qy * dR[3] - 2.f * qx * dR[4] - r * dR[5] + qz * dR[6] + r * dR[7] - 2.f * qx * dR[8]);
    dq[2] = 2.f * (-2.f * qy * dR[0] + qx * dR[1] + r * dR[2] + qx * dR[3] + qz * dR[5] - r * dR[6] + qz * dR[7] - 2.f * qy * dR[8]);
    dq[3] = 2.f * (-2.f * qz * dR[0] - r * dR[1] + qx * dR[2] + r * dR[3] - 2.f * qz * dR[4] + qy * dR[5] + qx * dR[6] + qy * dR[7]);
    const float qd = g.q[0] * dq[0] + g.q[1] * dq[1] + g.q[2] * dq[2] + g.q[3] * dq[3];
#pragma unroll
    for (int k = 0; k < 4; ++k) drot[k] = (dq[k] - g.q[k] * qd) / g.nrm;
}

// ---------------------------------------------------------------------------
// SH colour (degree 3)
// ---------------------------------------------------------------------------
#define SHC0 0.28209479177387814f
#define SHC1 0.4886025119029199f
__device__ static const float SHC2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                         -1.0925484305920792f, 0.5462742152960396f};
__device__ static const float SHC3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                         0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                         -0.5900435899266435f};

__device__ __forceinline__ void sh_basis(float x, float y, float z, float Y[16]) {
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    Y[0] = SHC0;
    Y[1] = -SHC1 * y; Y[2] = SHC1 * z; Y[3] = -SHC1 * x;
    Y[4] = SHC2[0] * xy; Y[5] = SHC2[1] * yz; Y[6] = SHC2[2] * (2.f * zz - xx - yy);
    Y[7] = SHC2[3] * xz; Y[8] = SHC2[4] * (xx - yy);
    Y[9] = SHC3[0] * y * (3.f * xx - yy); Y[10] = SHC3[1] * xy * z;
    Y[11] = SHC3[2] * y * (4.f * zz - xx - yy); Y[12] = SHC3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy);
    Y[13] = SHC3[4] * x * (4.f * zz - xx - yy); Y[14] = SHC3[5] * z * (xx - yy);
    Y[15] = SHC3[6] * x * (xx - 3.f * yy);
}

// direction (un-normalised d, and pulled-back camera) for one (view, Gaussian)
struct ShDir {
    float d[3], n, ci[3];  // d = xyz - cam', n = |d|, ci = inv(tf)*cam (when tf)
    float Ainv[9];         // inverse of tf[:3,:3] (when tf)
};

// HAS_TF: t = the blended transform (rows 0..2); otherwise dir = xyz - cam and t is ignored
template <bool HAS_TF>
__device__ __forceinline__ void sh_dir_xyz(float x, float y, float z, const float t[12], const float cam[3], ShDir& o) {
    if (HAS_TF) {
        const float a = t[0], b = t[1], c = t[2], d = t[4], e = t[5], f = t[6], g = t[8], h = t[9], k = t[10];
        const float c00 = e * k - f * h, c01 = f * g - d * k, c02 = d * h - e * g;
        const float det = a * c00 + b * c01 + c * c02;
        const float id = 1.0f / det;
        o.Ainv[0] = c00 * id; o.Ainv[1] = (c * h - b * k) * id; o.Ainv[2] = (b * f - c * e) * id;
        o.Ainv[3] = c01 * id; o.Ainv[4] = (a * k - c * g) * id; o.Ainv[5] = (c * d - a * f) * id;
        o.Ainv[6] = c02 * id; o.Ainv[7] = (b * g - a * h) * id; o.Ainv[8] = (a * e - b * d) * id;
        const float bx = cam[0] - t[3], by = cam[1] - t[7], bz = cam[2] - t[11];
        o.ci[0] = o.Ainv[0] * bx + o.Ainv[1] * by + o.Ainv[2] * bz;
        o.ci[1] = o.Ainv[3] * bx + o.Ainv[4] * by + o.Ainv[5] * bz;
        o.ci[2] = o.Ainv[6] * bx + o.Ainv[7] * by + o.Ainv[8] * bz;
        o.d[0] = x - o.ci[0]; o.d[1] = y - o.ci[1]; o.d[2] = z - o.ci[2];
    } else {
        o.d[0] = x - cam[0]; o.d[1] = y - cam[1]; o.d[2] = z - cam[2];
    }
    o.n = sqrtf(o.d[0] * o.d[0] + o.d[1] * o.d[1] + o.d[2] * o.d[2]);
}

// un-clamped rgb = sum_k Y_k c[k][ch]; c is the (16,3) coefficient block in registers
__device__ __forceinline__ void sh_rgb(const float c[48], const float Y[16], float rgb[3]) {
    rgb[0] = rgb[1] = rgb[2] = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        rgb[0] += Y[k] * c[3 * k];
        rgb[1] += Y[k] * c[3 * k + 1];
        rgb[2] += Y[k] * c[3 * k + 2];
    }
}

// Backward of one view: gc = dL/dcolour (after clamp).  Accumulates dsh (48); returns the
// gradient w.r.t. the xyz that formed the direction (gd) and, when has_tf, w.r.t. the 3x4
// transform (dtf, written).  The coefficient source C is indexable (register array or a
// small loader that reads from memory, which keeps 48 registers free in the fused kernel).
struct ShCoefMem {  // (16,3) coefficients split as f_dc (3) + f_rest (45), read on demand
    const float* dc;
    const float* rest;
    __device__ __forceinline__ float operator[](int k) const { return k < 3 ? dc[k] : rest[k - 3]; }
};

// the same with f_rest stored as fp16, 48 halves per Gaussian (45 used; rows 16-byte aligned): BASELINE config 5's
// "fp16 SH coeffs" storage option -- the arithmetic stays fp32
struct ShCoefMemH {
    const float* dc;
    const __half* rest;
    __device__ __forceinline__ float operator[](int k) const { return k < 3 ? dc[k] : __half2float(rest[k - 3]); }
};
#define MGR_SH_HALF_ROW 48

template <typename C>
__device__ __forceinline__ void sh_backward_view(const C& c, const ShDir& D, bool has_tf, const float gc[3],
                                                 float dsh[48], float gd[3], float dtf[12]) {
    const float inv_n = 1.0f / D.n;
    const float x = D.d[0] * inv_n, y = D.d[1] * inv_n, z = D.d[2] * inv_n;
    float Y[16], rgb[3] = {0.f, 0.f, 0.f};
    sh_basis(x, y, z, Y);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        rgb[0] += Y[k] * c[3 * k];
        rgb[1] += Y[k] * c[3 * k + 1];
        rgb[2] += Y[k] * c[3 * k + 2];
    }
    float dr[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) dr[ch] = (rgb[ch] + 0.5f >= 0.f) ? gc[ch] : 0.f;
    float t[16];  // dL/dY_k
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        t[k] = c[3 * k] * dr[0] + c[3 * k + 1] * dr[1] + c[3 * k + 2] * dr[2];
        dsh[3 * k] += Y[k] * dr[0];
        dsh[3 * k + 1] += Y[k] * dr[1];
        dsh[3 * k + 2] += Y[k] * dr[2];
    }
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    const float gdx = -SHC1 * t[3] + SHC2[0] * y * t[4] + SHC2[2] * (-2.f * x) * t[6] + SHC2[3] * z * t[7] +
                      SHC2[4] * 2.f * x * t[8] + SHC3[0] * 6.f * xy * t[9] + SHC3[1] * yz * t[10] +
                      SHC3[2] * (-2.f * xy) * t[11] + SHC3[3] * (-6.f * xz) * t[12] +
                      SHC3[4] * (4.f * zz - 3.f * xx - yy) * t[13] + SHC3[5] * 2.f * xz * t[14] +
                      SHC3[6] * (3.f * xx - 3.f * yy) * t[15];
    const float gdy = -SHC1 * t[1] + SHC2[0] * x * t[4] + SHC2[1] * z * t[5] + SHC2[2] * (-2.f * y) * t[6] +
                      SHC2[4] * (-2.f * y) * t[8] + SHC3[0] * (3.f * xx - 3.f * yy) * t[9] +
                      SHC3[1] * xz * t[10] + SHC3[2] * (4.f * zz - xx - 3.f * yy) * t[11] +
                      SHC3[3] * (-6.f * yz) * t[12] + SHC3[4] * (-2.f * xy) * t[13] +
                      SHC3[5] * (-2.f * yz) * t[14] + SHC3[6] * (-6.f * xy) * t[15];
    const float gdz = SHC1 * t[2] + SHC2[1] * y * t[5] + SHC2[2] * 4.f * z * t[6] + SHC2[3] * x * t[7] +
                      SHC3[1] * xy * t[10] + SHC3[2] * 8.f * yz * t[11] +
                      SHC3[3] * (6.f * zz - 3.f * xx - 3.f * yy) * t[12] + SHC3[4] * 8.f * xz * t[13] +
                      SHC3[5] * (xx - yy) * t[14];
    const float dp = x * gdx + y * gdy + z * gdz;  // through dir = d/|d|
    gd[0] = (gdx - x * dp) * inv_n;
    gd[1] = (gdy - y * dp) * inv_n;
    gd[2] = (gdz - z * dp) * inv_n;
    if (has_tf) {
        // cam' = Ainv (cam - t);  g_cam' = -gd;  h = Ainv^T g_cam'
        const float h0 = -(D.Ainv[0] * gd[0] + D.Ainv[3] * gd[1] + D.Ainv[6] * gd[2]);
        const float h1 = -(D.Ainv[1] * gd[0] + D.Ainv[4] * gd[1] + D.Ainv[7] * gd[2]);
        const float h2 = -(D.Ainv[2] * gd[0] + D.Ainv[5] * gd[1] + D.Ainv[8] * gd[2]);
        dtf[0] = -h0 * D.ci[0]; dtf[1] = -h0 * D.ci[1]; dtf[2] = -h0 * D.ci[2]; dtf[3] = -h0;
        dtf[4] = -h1 * D.ci[0]; dtf[5] = -h1 * D.ci[1]; dtf[6] = -h1 * D.ci[2]; dtf[7] = -h1;
        dtf[8] = -h2 * D.ci[0]; dtf[9] = -h2 * D.ci[1]; dtf[10] = -h2 * D.ci[2]; dtf[11] = -h2;
    }
}

// ---------------------------------------------------------------------------
// EWA projection of one Gaussian into one view (rasterizer K1) and its backward (K8+K9)
// ---------------------------------------------------------------------------
struct ProjOut {
    int radius, x0, y0, x1, y1;  // radius 0 = culled; tile rectangle [x0,x1) x [y0,y1)
    float px, py, ca, cb, cc, zv;
};

__device__ __forceinline__ void project_gaussian(const MgrCam& cam, int W, int H, int gx, int gy, const float p[3],
                                                 const float c6[6], ProjOut& o) {
    // integer decisions (radius, tile rectangle, culling) hang off this arithmetic: no fma contraction, so that every
    // kernel that inlines it -- and the scalar oracle, built with -ffp-contract=off -- rounds identically
#pragma clang fp contract(off)
    o.radius = 0; o.x0 = o.y0 = o.x1 = o.y1 = 0;
    o.px = o.py = o.ca = o.cb = o.cc = 0.f;
    const float* vm = cam.view;
    const float* pm = cam.proj;
    o.zv = vm[2] * p[0] + vm[6] * p[1] + vm[10] * p[2] + vm[14];
    if (!(o.zv > 0.2f)) return;
    const float hx = pm[0] * p[0] + pm[4] * p[1] + pm[8] * p[2] + pm[12];
    const float hy = pm[1] * p[0] + pm[5] * p[1] + pm[9] * p[2] + pm[13];
    const float hw = pm[3] * p[0] + pm[7] * p[1] + pm[11] * p[2] + pm[15];
    const float pw = 1.0f / (hw + 0.0000001f);
    const float ndx = hx * pw, ndy = hy * pw;
    float M0[3], M1[3], t[3], xm, ym, fx, fy, S0[3], S1[3];
    mgr_ewa_rows(cam, (float)W, (float)H, p, M0, M1, t, xm, ym, fx, fy);
    mgr_sym_mul(c6, M0, S0);
    mgr_sym_mul(c6, M1, S1);
    const float a = M0[0] * S0[0] + M0[1] * S0[1] + M0[2] * S0[2] + 0.3f;
    const float b = M0[0] * S1[0] + M0[1] * S1[1] + M0[2] * S1[2];
    const float c = M1[0] * S1[0] + M1[1] * S1[1] + M1[2] * S1[2] + 0.3f;
    const float det = a * c - b * b;
    if (det == 0.0f) return;
    const float dinv = 1.0f / det;
    const float mid = 0.5f * (a + c);
    const float sq = sqrtf(fmaxf(0.1f, mid * mid - det));
    const int rad = (int)ceilf(3.0f * sqrtf(fmaxf(mid + sq, mid - sq)));
    o.px = ((ndx + 1.0f) * (float)W - 1.0f) * 0.5f;
    o.py = ((ndy + 1.0f) * (float)H - 1.0f) * 0.5f;
    const float fr = (float)rad;
    const int x0 = min(gx, max(0, (int)((o.px - fr) / 16.0f)));
    const int y0 = min(gy, max(0, (int)((o.py - fr) / 16.0f)));
    const int x1 = min(gx, max(0, (int)((o.px + fr + 15.0f) / 16.0f)));
    const int y1 = min(gy, max(0, (int)((o.py + fr + 15.0f) / 16.0f)));
    if ((x1 - x0) * (y1 - y0) > 0) {
        o.radius = rad;
        o.x0 = x0; o.y0 = y0; o.x1 = x1; o.y1 = y1;
        o.ca = c * dinv;
        o.cb = -b * dinv;
        o.cc = a * dinv;
    }
}

// acc = summed pair records [dmean2D.x, dmean2D.y, dconic A, B, C, dopacity, dr, dg, db]
__device__ __forceinline__ void project_backward(const MgrCam& cam, int W, int H, const float p[3],
                                                 const float c6[6], const float acc[9], float dm[3],
                                                 float dc6[6]) {
    const float* vm = cam.view;
    float M0[3], M1[3], t[3], xm, ym, fx, fy, S0[3], S1[3];
    mgr_ewa_rows(cam, (float)W, (float)H, p, M0, M1, t, xm, ym, fx, fy);
    mgr_sym_mul(c6, M0, S0);
    mgr_sym_mul(c6, M1, S1);
    const float a = M0[0] * S0[0] + M0[1] * S0[1] + M0[2] * S0[2] + 0.3f;
    const float b = M0[0] * S1[0] + M0[1] * S1[1] + M0[2] * S1[2];
    const float c = M1[0] * S1[0] + M1[1] * S1[1] + M1[2] * S1[2] + 0.3f;
    const float dA = acc[2], dB = acc[3], dC = acc[4];
    const float den = a * c - b * b;
    const float k2 = 1.0f / (den * den + 0.0000001f);
    float da = 0.f, db = 0.f, dc = 0.f;
#pragma unroll
    for (int j = 0; j < 6; ++j) dc6[j] = 0.f;
    if (k2 != 0.0f) {
        da = k2 * (-c * c * dA + 2.0f * b * c * dB + (den - a * c) * dC);
        dc = k2 * (-a * a * dC + 2.0f * a * b * dB + (den - a * c) * dA);
        db = k2 * 2.0f * (b * c * dA - (den + 2.0f * b * b) * dB + a * b * dC);
        dc6[0] = M0[0] * M0[0] * da + M0[0] * M1[0] * db + M1[0] * M1[0] * dc;
        dc6[3] = M0[1] * M0[1] * da + M0[1] * M1[1] * db + M1[1] * M1[1] * dc;
        dc6[5] = M0[2] * M0[2] * da + M0[2] * M1[2] * db + M1[2] * M1[2] * dc;
        dc6[1] = 2.0f * M0[0] * M0[1] * da + (M0[0] * M1[1] + M0[1] * M1[0]) * db + 2.0f * M1[0] * M1[1] * dc;
        dc6[2] = 2.0f * M0[0] * M0[2] * da + (M0[0] * M1[2] + M0[2] * M1[0]) * db + 2.0f * M1[0] * M1[2] * dc;
        dc6[4] = 2.0f * M0[2] * M0[1] * da + (M0[1] * M1[2] + M0[2] * M1[1]) * db + 2.0f * M1[1] * M1[2] * dc;
    }
    float dM0[3], dM1[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        dM0[j] = 2.0f * S0[j] * da + S1[j] * db;
        dM1[j] = 2.0f * S1[j] * dc + S0[j] * db;
    }
    const float dJ00 = vm[0] * dM0[0] + vm[4] * dM0[1] + vm[8] * dM0[2];
    const float dJ02 = vm[2] * dM0[0] + vm[6] * dM0[1] + vm[10] * dM0[2];
    const float dJ11 = vm[1] * dM1[0] + vm[5] * dM1[1] + vm[9] * dM1[2];
    const float dJ12 = vm[2] * dM1[0] + vm[6] * dM1[1] + vm[10] * dM1[2];
    const float tz = 1.0f / t[2], tz2 = tz * tz, tz3 = tz2 * tz;
    const float dtx = xm * -fx * tz2 * dJ02;
    const float dty = ym * -fy * tz2 * dJ12;
    const float dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + (2.0f * fx * t[0]) * tz3 * dJ02 +
                      (2.0f * fy * t[1]) * tz3 * dJ12;
    dm[0] = vm[0] * dtx + vm[1] * dty + vm[2] * dtz;
    dm[1] = vm[4] * dtx + vm[5] * dty + vm[6] * dtz;
    dm[2] = vm[8] * dtx + vm[9] * dty + vm[10] * dtz;
    const float* pm = cam.proj;
    const float hw = pm[3] * p[0] + pm[7] * p[1] + pm[11] * p[2] + pm[15];
    const float mw = 1.0f / (hw + 0.0000001f);
    const float mul1 = (pm[0] * p[0] + pm[4] * p[1] + pm[8] * p[2] + pm[12]) * mw * mw;
    const float mul2 = (pm[1] * p[0] + pm[5] * p[1] + pm[9] * p[2] + pm[13]) * mw * mw;
    const float gx2 = acc[0], gy2 = acc[1];
    dm[0] += (pm[0] * mw - pm[3] * mul1) * gx2 + (pm[1] * mw - pm[3] * mul2) * gy2;
    dm[1] += (pm[4] * mw - pm[7] * mul1) * gx2 + (pm[5] * mw - pm[7] * mul2) * gy2;
    dm[2] += (pm[8] * mw - pm[11] * mul1) * gx2 + (pm[9] * mw - pm[11] * mul2) * gy2;
}

// Sum of this Gaussian's pair records that the backward blend wrote in this call.
typedef uint32_t mgr_u4u __attribute__((ext_vector_type(4), aligned(4)));

// Sum the pair-gradient records of one (view, Gaussian): slots [off, off + cnt), valid when their
// tag equals the call's epoch.  Two passes per 64 slots so that neither is a chain of dependent
// loads: (1) the tags, sixteen per step as four unaligned dwordx4 (the tag array is padded by
// 64 bytes for the over-read), folded into a bit mask; (2) the records of the set bits, in
// ascending slot order (fixed summation order).
__device__ __forceinline__ void gather_pair_grads(uint32_t off, uint32_t cnt, const uint32_t* __restrict__ pair_tag,
                                                  const float4* __restrict__ pair_grad, uint32_t cap, uint32_t epoch,
                                                  float acc[9]) {
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = 0.f;
    if (off >= cap) return;
    cnt = min(cnt, cap - off);
    for (uint32_t base = 0; base < cnt; base += 64) {
        const uint32_t n = min(64u, cnt - base);
        unsigned long long hit = 0ull;
        for (uint32_t k = 0; k < n; k += 16) {
            const mgr_u4u* tp = (const mgr_u4u*)(pair_tag + off + base + k);
            const mgr_u4u t0 = tp[0], t1 = tp[1], t2 = tp[2], t3 = tp[3];
            const uint32_t tg[16] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w,
                                     t2.x, t2.y, t2.z, t2.w, t3.x, t3.y, t3.z, t3.w};
            uint32_t m16 = 0;
#pragma unroll
            for (int j = 0; j < 16; ++j) m16 |= (tg[j] == epoch ? 1u : 0u) << j;
            hit |= (unsigned long long)m16 << k;
        }
        if (n < 64) hit &= (1ull << n) - 1ull;
        while (hit) {
            const int j = __builtin_ctzll(hit);
            hit &= hit - 1;
            const float4* r = pair_grad + (size_t)(off + base + (uint32_t)j) * 3;
            const float4 a = r[0], b = r[1], c = r[2];
            acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w;
            acc[4] += b.x; acc[5] += b.y; acc[6] += b.z; acc[7] += b.w;
            acc[8] += c.x;
        }
    }
}

#endif  // __HIPCC__
