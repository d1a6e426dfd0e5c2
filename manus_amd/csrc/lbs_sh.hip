// Articulation kernels for gfx950: skin-weight sampling from the voxel grid,
// linear-blend skinning of means and covariances, SH colour — forward and
// analytic backward.  One thread per Gaussian, bone transforms and cameras are
// wave-uniform (scalar loads), per-pose / per-view sums are taken inside the
// thread in a fixed order (no atomics, deterministic).
//
// Reference semantics (paths relative to /root/reference):
//   skin weights   src/utils/gaussian_utils.py:167-196
//   LBS            src/modules/hand_dynamic.py:106-127
//   covariance     src/models/gaussian.py:49-53,84-93; src/utils/gaussian_utils.py:279-314
//   SH colour      src/utils/gaussian_utils.py:431-449; src/utils/sh_utils.py:57-104
//   projection     src/utils/transforms.py:304-311
//   L1 loss        src/utils/loss_utils.py:22-27
#include "instance_math.h"

// ---------------------------------------------------------------------------
// skin weights: trilinear, align_corners=True, zero padding, then w / sum(w)
// ---------------------------------------------------------------------------
struct TriSetup {
    int x0, y0, z0;
    float fx, fy, fz;  // fractional parts
};

__device__ __forceinline__ TriSetup tri_setup(const float* __restrict__ xyz, int i,
                                              const float* __restrict__ center,
                                              const float* __restrict__ scale, int D, int H, int W) {
    TriSetup s;
    const float ux = (xyz[3 * i + 0] - center[0]) / scale[0];
    const float uy = (xyz[3 * i + 1] - center[1]) / scale[1];
    const float uz = (xyz[3 * i + 2] - center[2]) / scale[2];
    const float ix = ((ux + 1.0f) * 0.5f) * (float)(W - 1);
    const float iy = ((uy + 1.0f) * 0.5f) * (float)(H - 1);
    const float iz = ((uz + 1.0f) * 0.5f) * (float)(D - 1);
    const float flx = floorf(ix), fly = floorf(iy), flz = floorf(iz);
    // clamp far-away points so the int conversion is defined; they are out of
    // bounds either way and contribute zero
    s.x0 = (int)fminf(fmaxf(flx, -2.0f), (float)W + 1.0f);
    s.y0 = (int)fminf(fmaxf(fly, -2.0f), (float)H + 1.0f);
    s.z0 = (int)fminf(fmaxf(flz, -2.0f), (float)D + 1.0f);
    s.fx = ix - flx;
    s.fy = iy - fly;
    s.fz = iz - flz;
    return s;
}

// value of channel b at the 8 corners (0 when out of bounds)
__device__ __forceinline__ void tri_corners(const float* __restrict__ grid, const TriSetup& s, int D,
                                            int H, int W, int B, int b, float c[8]) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int x = s.x0 + (k & 1), y = s.y0 + ((k >> 1) & 1), z = s.z0 + (k >> 2);
        const bool ok = x >= 0 && x < W && y >= 0 && y < H && z >= 0 && z < D;
        c[k] = ok ? grid[(((size_t)z * H + y) * W + x) * B + b] : 0.0f;
    }
}

__device__ __forceinline__ float tri_value(const TriSetup& s, const float c[8]) {
    const float wx0 = 1.0f - s.fx, wx1 = s.fx, wy0 = 1.0f - s.fy, wy1 = s.fy, wz0 = 1.0f - s.fz,
                wz1 = s.fz;
    return c[0] * (wx0 * wy0 * wz0) + c[1] * (wx1 * wy0 * wz0) + c[2] * (wx0 * wy1 * wz0) +
           c[3] * (wx1 * wy1 * wz0) + c[4] * (wx0 * wy0 * wz1) + c[5] * (wx1 * wy0 * wz1) +
           c[6] * (wx0 * wy1 * wz1) + c[7] * (wx1 * wy1 * wz1);
}

__global__ __launch_bounds__(256) void k_skin_fwd(int N, const float* __restrict__ xyz,
                                                  const float* __restrict__ grid, int D, int H, int W,
                                                  int B, const float* __restrict__ center,
                                                  const float* __restrict__ scale,
                                                  float* __restrict__ out_w) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const TriSetup s = tri_setup(xyz, i, center, scale, D, H, W);
    float sum = 0.f;
    for (int b = 0; b < B; ++b) {
        float c[8];
        tri_corners(grid, s, D, H, W, B, b, c);
        const float v = tri_value(s, c);
        out_w[(size_t)i * B + b] = v;
        sum += v;
    }
    for (int b = 0; b < B; ++b) out_w[(size_t)i * B + b] = out_w[(size_t)i * B + b] / sum;
}

__global__ __launch_bounds__(256) void k_skin_bwd(int N, const float* __restrict__ xyz,
                                                  const float* __restrict__ grid, int D, int H, int W,
                                                  int B, const float* __restrict__ center,
                                                  const float* __restrict__ scale,
                                                  const float* __restrict__ dL_dw,
                                                  float* __restrict__ dL_dxyz, int accumulate,
                                                  const uint32_t* __restrict__ index,
                                                  const uint32_t* __restrict__ index_count) {
    const int t = blockIdx.x * 256 + threadIdx.x;   // (the indexed launch covers the list's maximum length)
    if (index ? (uint32_t)t >= *index_count : t >= N) return;
    const int i = index ? (int)index[t] : t;
    if (i >= N) return;
    const TriSetup s = tri_setup(xyz, i, center, scale, D, H, W);
    // pass 1: S = sum raw, dot = sum_c dLdw_c * raw_c
    float S = 0.f, dot = 0.f;
    for (int b = 0; b < B; ++b) {
        float c[8];
        tri_corners(grid, s, D, H, W, B, b, c);
        const float v = tri_value(s, c);
        S += v;
        dot += dL_dw[(size_t)i * B + b] * v;
    }
    const float invS = 1.0f / S;
    dot *= invS;  // = sum_c dLdw_c * w_c
    // pass 2: dL/draw_b = (dLdw_b - dot)/S ; chain through the trilinear weights
    float gx = 0.f, gy = 0.f, gz = 0.f;
    const float wx0 = 1.0f - s.fx, wx1 = s.fx, wy0 = 1.0f - s.fy, wy1 = s.fy, wz0 = 1.0f - s.fz,
                wz1 = s.fz;
    for (int b = 0; b < B; ++b) {
        float c[8];
        tri_corners(grid, s, D, H, W, B, b, c);
        const float gr = (dL_dw[(size_t)i * B + b] - dot) * invS;
        const float ddx = (c[1] - c[0]) * (wy0 * wz0) + (c[3] - c[2]) * (wy1 * wz0) +
                          (c[5] - c[4]) * (wy0 * wz1) + (c[7] - c[6]) * (wy1 * wz1);
        const float ddy = (c[2] - c[0]) * (wx0 * wz0) + (c[3] - c[1]) * (wx1 * wz0) +
                          (c[6] - c[4]) * (wx0 * wz1) + (c[7] - c[5]) * (wx1 * wz1);
        const float ddz = (c[4] - c[0]) * (wx0 * wy0) + (c[5] - c[1]) * (wx1 * wy0) +
                          (c[6] - c[2]) * (wx0 * wy1) + (c[7] - c[3]) * (wx1 * wy1);
        gx += gr * ddx;
        gy += gr * ddy;
        gz += gr * ddz;
    }
    dL_dxyz[3 * i + 0] = (accumulate ? dL_dxyz[3 * i + 0] : 0.f) + gx * (0.5f * (float)(W - 1)) / scale[0];
    dL_dxyz[3 * i + 1] = (accumulate ? dL_dxyz[3 * i + 1] : 0.f) + gy * (0.5f * (float)(H - 1)) / scale[1];
    dL_dxyz[3 * i + 2] = (accumulate ? dL_dxyz[3 * i + 2] : 0.f) + gz * (0.5f * (float)(D - 1)) / scale[2];
}

// ---------------------------------------------------------------------------
// fast path: channel stride padded to 24 floats (96 B per voxel, 16-byte aligned), so every
// corner is six float4 loads; one pass over the 8 corners in both directions.
// Backward algebra: with per-corner scalars P_k = sum_b a_b c_k[b] and Q_k = sum_b c_k[b]
// (a = dL/dw) and trilinear weights W_k, S = sum_k W_k Q_k, dot*S = sum_k W_k P_k and
//   dL/d(ix) = (1/S) [ sum_k dW_k/d(ix) P_k - (dot*S / S) sum_k dW_k/d(ix) Q_k ].
// ---------------------------------------------------------------------------
#define SKIN_BP 24

__device__ __forceinline__ void tri_weights(const TriSetup& s, float Wk[8]) {
    const float wx[2] = {1.0f - s.fx, s.fx}, wy[2] = {1.0f - s.fy, s.fy}, wz[2] = {1.0f - s.fz, s.fz};
#pragma unroll
    for (int k = 0; k < 8; ++k) Wk[k] = wx[k & 1] * wy[(k >> 1) & 1] * wz[k >> 2];
}

__global__ __launch_bounds__(256) void k_skin_fwd24(int N, const float* __restrict__ xyz,
                                                    const float4* __restrict__ grid, int D, int H, int W,
                                                    int B, const float* __restrict__ center,
                                                    const float* __restrict__ scale,
                                                    float* __restrict__ out_w) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const TriSetup s = tri_setup(xyz, i, center, scale, D, H, W);
    float Wk[8];
    tri_weights(s, Wk);
    float acc[SKIN_BP];
#pragma unroll
    for (int b = 0; b < SKIN_BP; ++b) acc[b] = 0.f;
    // All 48 loads are unconditional (coordinates clamped, the weight of an outside corner set to
    // zero = grid_sample's zero padding): predicated loads make hipcc drain the memory queue corner
    // by corner.
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int x = s.x0 + (k & 1), y = s.y0 + ((k >> 1) & 1), z = s.z0 + (k >> 2);
        const bool inb = x >= 0 && x < W && y >= 0 && y < H && z >= 0 && z < D;
        const int xc = min(max(x, 0), W - 1), yc = min(max(y, 0), H - 1), zc = min(max(z, 0), D - 1);
        const float wk = inb ? Wk[k] : 0.f;
        const float4* p = grid + (((size_t)zc * H + yc) * W + xc) * (SKIN_BP / 4);
#pragma unroll
        for (int q = 0; q < SKIN_BP / 4; ++q) {
            const float4 c = p[q];
            acc[4 * q + 0] += wk * c.x;
            acc[4 * q + 1] += wk * c.y;
            acc[4 * q + 2] += wk * c.z;
            acc[4 * q + 3] += wk * c.w;
        }
    }
    float sum = 0.f;
#pragma unroll
    for (int b = 0; b < SKIN_BP; ++b) sum += acc[b];  // pad channels are zero
#pragma unroll
    for (int b = 0; b < SKIN_BP; ++b)
        if (b < B) out_w[(size_t)i * B + b] = acc[b] / sum;
}

// Eight lanes per Gaussian, one per corner: a lane fetches its corner's 96-byte row (six float4 of one or two cache lines),
// weights it, and the eight rows are summed with a transposing DPP reduction that leaves channel (8 m + k) in lane k.
// k_skin_fwd24 above keeps all 48 loads of a Gaussian in one thread: 190 VGPRs, two waves per SIMD, 1.5 resident on
// average over the launch -- the gather is latency-bound (VALU 0.08, HBM 0.43 of peak by the counters) and wants waves
// in flight, not loads per wave.  Here a wave holds 8 Gaussians and ~50 VGPRs.  The sum over the corners is a tree instead
// of a chain (last-bit differences against the one-thread kernel).
__global__ __launch_bounds__(256) void k_skin_fwd24x8(int N, const float* __restrict__ xyz,
                                                      const float4* __restrict__ grid, int D, int H, int W,
                                                      int B, const float* __restrict__ center,
                                                      const float* __restrict__ scale,
                                                      float* __restrict__ out_w) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int i = min(t >> 3, N - 1), k = t & 7;   // (the tail group recomputes the last Gaussian: lanes stay convergent for the DPP sums)
    const TriSetup s = tri_setup(xyz, i, center, scale, D, H, W);
    const float wx = (k & 1) ? s.fx : 1.0f - s.fx, wy = (k & 2) ? s.fy : 1.0f - s.fy, wz = (k & 4) ? s.fz : 1.0f - s.fz;
    const int x = s.x0 + (k & 1), y = s.y0 + ((k >> 1) & 1), z = s.z0 + (k >> 2);
    const bool inb = x >= 0 && x < W && y >= 0 && y < H && z >= 0 && z < D;
    const int xc = min(max(x, 0), W - 1), yc = min(max(y, 0), H - 1), zc = min(max(z, 0), D - 1);
    const float wk = inb ? (wx * wy) * wz : 0.f;
    const float4* p = grid + (((size_t)zc * H + yc) * W + xc) * (SKIN_BP / 4);
    float v[SKIN_BP];
#pragma unroll
    for (int q = 0; q < SKIN_BP / 4; ++q) {
        const float4 c = p[q];
        v[4 * q + 0] = wk * c.x; v[4 * q + 1] = wk * c.y; v[4 * q + 2] = wk * c.z; v[4 * q + 3] = wk * c.w;
    }
    float tot[3];
#pragma unroll
    for (int m = 0; m < 3; ++m) tot[m] = grp8_reduce_scatter(v + 8 * m, k);   // lane k: channel 8 m + k summed over the corners
    const float sum = grp_sum<8>((tot[0] + tot[1]) + tot[2]);                  // pad channels are zero
    if ((t >> 3) < N) {
#pragma unroll
        for (int m = 0; m < 3; ++m)
            if (8 * m + k < B) out_w[(size_t)i * B + 8 * m + k] = tot[m] / sum;
    }
}

__global__ __launch_bounds__(256) void k_skin_bwd24(int N, const float* __restrict__ xyz,
                                                    const float4* __restrict__ grid, int D, int H, int W,
                                                    int B, const float* __restrict__ center,
                                                    const float* __restrict__ scale,
                                                    const float* __restrict__ dL_dw,
                                                    float* __restrict__ dL_dxyz, int accumulate,
                                                    const uint32_t* __restrict__ index,
                                                    const uint32_t* __restrict__ index_count) {
    // index != nullptr: only the Gaussians index[0 .. *index_count) (the backward's active list) are processed
    const int t = blockIdx.x * 256 + threadIdx.x;   // (the indexed launch covers the list's maximum length)
    if (index ? (uint32_t)t >= *index_count : t >= N) return;
    const int i = index ? (int)index[t] : t;
    if (i >= N) return;
    const TriSetup s = tri_setup(xyz, i, center, scale, D, H, W);
    float a[SKIN_BP];
#pragma unroll
    for (int b = 0; b < SKIN_BP; ++b) a[b] = 0.f;
    {   // the (N,B) row in unaligned dwordx4 pieces
        const float* row = dL_dw + (size_t)i * B;
#pragma unroll
        for (int q = 0; q < SKIN_BP / 4; ++q) {
            if (4 * q + 4 <= B) {
                const mgr_f4u t = *(const mgr_f4u*)(row + 4 * q);
                a[4 * q] = t.x; a[4 * q + 1] = t.y; a[4 * q + 2] = t.z; a[4 * q + 3] = t.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (4 * q + e < B) a[4 * q + e] = row[4 * q + e];
            }
        }
    }
    const float wx[2] = {1.0f - s.fx, s.fx}, wy[2] = {1.0f - s.fy, s.fy}, wz[2] = {1.0f - s.fz, s.fz};
    float S = 0.f, dS = 0.f, gxP = 0.f, gyP = 0.f, gzP = 0.f, gxQ = 0.f, gyQ = 0.f, gzQ = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int bx = k & 1, by = (k >> 1) & 1, bz = k >> 2;
        const int x = s.x0 + bx, y = s.y0 + by, z = s.z0 + bz;
        const float inb = (x >= 0 && x < W && y >= 0 && y < H && z >= 0 && z < D) ? 1.f : 0.f;  // zero padding
        const int xc = min(max(x, 0), W - 1), yc = min(max(y, 0), H - 1), zc = min(max(z, 0), D - 1);
        const float4* p = grid + (((size_t)zc * H + yc) * W + xc) * (SKIN_BP / 4);
        float Pk = 0.f, Qk = 0.f;
#pragma unroll
        for (int q = 0; q < SKIN_BP / 4; ++q) {
            const float4 c = p[q];
            Pk += a[4 * q] * c.x + a[4 * q + 1] * c.y + a[4 * q + 2] * c.z + a[4 * q + 3] * c.w;
            Qk += (c.x + c.y) + (c.z + c.w);
        }
        Pk *= inb;
        Qk *= inb;
        const float Wk = wx[bx] * wy[by] * wz[bz];
        const float Dx = (bx ? 1.f : -1.f) * wy[by] * wz[bz];
        const float Dy = (by ? 1.f : -1.f) * wx[bx] * wz[bz];
        const float Dz = (bz ? 1.f : -1.f) * wx[bx] * wy[by];
        S += Wk * Qk; dS += Wk * Pk;
        gxP += Dx * Pk; gyP += Dy * Pk; gzP += Dz * Pk;
        gxQ += Dx * Qk; gyQ += Dy * Qk; gzQ += Dz * Qk;
    }
    const float invS = 1.0f / S, dot = dS * invS;
    dL_dxyz[3 * i + 0] = (accumulate ? dL_dxyz[3 * i + 0] : 0.f) + (gxP - dot * gxQ) * invS * (0.5f * (float)(W - 1)) / scale[0];
    dL_dxyz[3 * i + 1] = (accumulate ? dL_dxyz[3 * i + 1] : 0.f) + (gyP - dot * gyQ) * invS * (0.5f * (float)(H - 1)) / scale[1];
    dL_dxyz[3 * i + 2] = (accumulate ? dL_dxyz[3 * i + 2] : 0.f) + (gzP - dot * gzQ) * invS * (0.5f * (float)(D - 1)) / scale[2];
}

// (An eight-lanes-per-Gaussian version of this kernel, like k_skin_fwd24x8, was measured and is slower: 0.046 against
// 0.032 ms.  The backward runs on the active list only -- 130 k threads = two waves per SIMD, all resident at once with 48
// loads each in flight -- and already moves 0.58 of the HBM peak in 96-byte rows; there is no occupancy to win.)
// ---------------------------------------------------------------------------
// LBS of means and covariances
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lbs_fwd(int N, int B, const float* __restrict__ xyz,
                                                 const float* __restrict__ log_scale,
                                                 const float* __restrict__ rot,
                                                 const float* __restrict__ skin_w,
                                                 const float* __restrict__ transforms,
                                                 float* __restrict__ posed_xyz,
                                                 float* __restrict__ posed_cov,
                                                 float* __restrict__ tf_out, int tfr) {
    // tfr: floats per row of tf_out -- 12 (rows 0..2 of the 4x4) or 16 (the reference's (N,4,4) layout: the constant last row is written too)
    const int i = blockIdx.x * 256 + threadIdx.x, p = blockIdx.y;
    if (i >= N) return;
    float tf[12];
    blend_tf(skin_w ? skin_w + (size_t)i * B : nullptr, transforms ? transforms + (size_t)p * B * 16 : nullptr, B, tf);
    GaussCano g;
    cano_load(xyz, log_scale, rot, i, g);
    float posed[3], cov6[6];
    lbs_apply(tf, g, posed, cov6);
    const size_t pi = (size_t)p * N + i;
#pragma unroll
    for (int k = 0; k < 3; ++k) posed_xyz[pi * 3 + k] = posed[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) posed_cov[pi * 6 + k] = cov6[k];
    if (tf_out) {
        float* to = tf_out + pi * (size_t)tfr;
#pragma unroll
        for (int k = 0; k < 12; ++k) to[k] = tf[k];
        if (tfr == 16) { to[12] = 0.f; to[13] = 0.f; to[14] = 0.f; to[15] = 1.f; }
    }
}

__global__ __launch_bounds__(256) void k_lbs_bwd(int P, int N, int B, const float* __restrict__ xyz,
                                                 const float* __restrict__ log_scale,
                                                 const float* __restrict__ rot,
                                                 const float* __restrict__ skin_w,
                                                 const float* __restrict__ transforms,
                                                 const float* __restrict__ g_xyz,
                                                 const float* __restrict__ g_cov,
                                                 const float* __restrict__ g_tf,
                                                 float* __restrict__ dL_dxyz,
                                                 float* __restrict__ dL_dls,
                                                 float* __restrict__ dL_drot,
                                                 float* __restrict__ dL_dw, int tfr) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    GaussCano g;
    cano_load(xyz, log_scale, rot, i, g);
    float dxyz[3] = {0.f, 0.f, 0.f}, ds[3] = {0.f, 0.f, 0.f};
    float dR[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float dw[MGR_MAX_BONES];
#pragma unroll
    for (int b = 0; b < MGR_MAX_BONES; ++b) dw[b] = 0.f;

    for (int p = 0; p < P; ++p) {
        const size_t pi = (size_t)p * N + i;
        const float* Tp = transforms ? transforms + (size_t)p * B * 16 : nullptr;
        float tf[12], dtf[12];
        blend_tf(skin_w ? skin_w + (size_t)i * B : nullptr, Tp, B, tf);
        const float gp[3] = {g_xyz[pi * 3], g_xyz[pi * 3 + 1], g_xyz[pi * 3 + 2]};
        const float g6[6] = {g_cov[pi * 6], g_cov[pi * 6 + 1], g_cov[pi * 6 + 2],
                             g_cov[pi * 6 + 3], g_cov[pi * 6 + 4], g_cov[pi * 6 + 5]};
        float gt[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) gt[k] = g_tf ? g_tf[pi * (size_t)tfr + k] : 0.f;
        lbs_backward_view<true>(tf, g, gp, g6, gt, dxyz, ds, dR, dtf);
        if (skin_w) {
#pragma unroll
            for (int b = 0; b < MGR_MAX_BONES; ++b) {
                if (b < B) {
                    const float* T = Tp + (size_t)b * 16;
                    float a = 0.f;
#pragma unroll
                    for (int k = 0; k < 12; ++k) a += dtf[k] * T[k];
                    dw[b] += a;
                }
            }
        }
    }
    dL_dxyz[3 * i] = dxyz[0]; dL_dxyz[3 * i + 1] = dxyz[1]; dL_dxyz[3 * i + 2] = dxyz[2];
    dL_dls[3 * i] = ds[0] * g.s[0]; dL_dls[3 * i + 1] = ds[1] * g.s[1]; dL_dls[3 * i + 2] = ds[2] * g.s[2];
    float drot[4];
    quat_backward(g, dR, drot);
#pragma unroll
    for (int k = 0; k < 4; ++k) dL_drot[4 * i + k] = drot[k];
    if (dL_dw && skin_w) {
#pragma unroll
        for (int b = 0; b < MGR_MAX_BONES; ++b)
            if (b < B) dL_dw[(size_t)i * B + b] = dw[b];
    }
}

// ---------------------------------------------------------------------------
// SH colour (degree 3)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sh_fwd(int N, const float* __restrict__ sh,
                                                const float* __restrict__ xyz, int64_t s_xyz,
                                                const float* __restrict__ tf, int64_t s_tf,
                                                const float* __restrict__ cams,
                                                float* __restrict__ colors, int tfr) {
    const int i = blockIdx.x * 256 + threadIdx.x, v = blockIdx.y;
    if (i >= N) return;
    const float* cp = cams + (size_t)v * MGR_CAM_FLOATS + 34;
    const float cam[3] = {cp[0], cp[1], cp[2]};
    const float* xv = xyz + (size_t)v * s_xyz;
    ShDir D;
    float t12[12];
    if (tf) {
#pragma unroll
        for (int k = 0; k < 12; ++k) t12[k] = tf[(size_t)v * s_tf + (size_t)i * tfr + k];
        sh_dir_xyz<true>(xv[3 * i], xv[3 * i + 1], xv[3 * i + 2], t12, cam, D);
    } else {
        sh_dir_xyz<false>(xv[3 * i], xv[3 * i + 1], xv[3 * i + 2], t12, cam, D);
    }
    float Y[16], c[48], rgb[3];
    sh_basis(D.d[0] / D.n, D.d[1] / D.n, D.d[2] / D.n, Y);
#pragma unroll
    for (int k = 0; k < 48; ++k) c[k] = sh[(size_t)i * 48 + k];
    sh_rgb(c, Y, rgb);
    float* o = colors + ((size_t)v * N + i) * 3;
    o[0] = fmaxf(rgb[0] + 0.5f, 0.f);
    o[1] = fmaxf(rgb[1] + 0.5f, 0.f);
    o[2] = fmaxf(rgb[2] + 0.5f, 0.f);
}

__global__ __launch_bounds__(256, 2) void k_sh_bwd(int V, int N, const float* __restrict__ sh,
                                                const float* __restrict__ xyz, int64_t s_xyz,
                                                const float* __restrict__ tf, int64_t s_tf,
                                                const float* __restrict__ cams,
                                                const float* __restrict__ g_col,
                                                float* __restrict__ dL_dsh,
                                                float* __restrict__ dL_dxyz,
                                                float* __restrict__ dL_dtf, int tfr) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    float c[48], dsh[48];
#pragma unroll
    for (int k = 0; k < 48; ++k) {
        c[k] = sh[(size_t)i * 48 + k];
        dsh[k] = 0.f;
    }
    for (int v = 0; v < V; ++v) {
        const float* cp = cams + (size_t)v * MGR_CAM_FLOATS + 34;
        const float cam[3] = {cp[0], cp[1], cp[2]};
        const float* xv = xyz + (size_t)v * s_xyz;
        const bool has_tf = tf != nullptr;
        ShDir D;
        float t12[12];
        if (has_tf) {
#pragma unroll
            for (int k = 0; k < 12; ++k) t12[k] = tf[(size_t)v * s_tf + (size_t)i * tfr + k];
            sh_dir_xyz<true>(xv[3 * i], xv[3 * i + 1], xv[3 * i + 2], t12, cam, D);
        } else {
            sh_dir_xyz<false>(xv[3 * i], xv[3 * i + 1], xv[3 * i + 2], t12, cam, D);
        }
        const float* gcp = g_col + ((size_t)v * N + i) * 3;
        const float gc[3] = {gcp[0], gcp[1], gcp[2]};
        float gd[3], dtf[12];
        sh_backward_view(c, D, has_tf, gc, dsh, gd, dtf);
        float* ox = dL_dxyz + ((size_t)v * N + i) * 3;
        ox[0] = gd[0]; ox[1] = gd[1]; ox[2] = gd[2];
        if (has_tf && dL_dtf) {
            float* ot = dL_dtf + ((size_t)v * N + i) * (size_t)tfr;
#pragma unroll
            for (int k = 0; k < 12; ++k) ot[k] = dtf[k];
            if (tfr == 16) { ot[12] = 0.f; ot[13] = 0.f; ot[14] = 0.f; ot[15] = 0.f; }      // (the constant row has no gradient)
        }
    }
#pragma unroll
    for (int k = 0; k < 48; ++k) dL_dsh[(size_t)i * 48 + k] = dsh[k];
}

// ---------------------------------------------------------------------------
// misc
// ---------------------------------------------------------------------------
__global__ void k_project(int N, const float* __restrict__ xyz, const float* __restrict__ K,
                          const float* __restrict__ E, float* __restrict__ uv) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    float Pm[12];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) Pm[4 * r + c] = K[3 * r] * E[c] + K[3 * r + 1] * E[4 + c] + K[3 * r + 2] * E[8 + c];
    const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    const float a = Pm[0] * x + Pm[1] * y + Pm[2] * z + Pm[3];
    const float b = Pm[4] * x + Pm[5] * y + Pm[6] * z + Pm[7];
    const float c = Pm[8] * x + Pm[9] * y + Pm[10] * z + Pm[11];
    uv[2 * i] = a / c;
    uv[2 * i + 1] = b / c;
}

// ---------------------------------------------------------------------------
// Segmentation-mask pruning test (src/utils/gaussian_utils.py:35-47,101-147).
//   dilate_mask: conv2d of the {0,1} mask with an 11x11 box of ones, zero padding, > 0  ==  "any pixel of the
//     window is set"; done separably (rows, then columns) through a byte scratch image.
//   get_points_outside_mask: uv = project_points(xyz); x = int(clamp(u, 0, W-1)), y = int(clamp(v, 0, H-1))
//     (float clamp, then truncation like torch's .int()); value = !mask[y][x]; if ANY keypoint projects onto a
//     pixel outside the mask, every value is False (:125-131).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_dilate_rows(int H, int W, int r, const uint8_t* __restrict__ in,
                                                     uint8_t* __restrict__ out) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const uint8_t* row = in + (size_t)y * W;
    uint8_t a = 0;
    for (int k = max(0, x - r); k <= min(W - 1, x + r); ++k) a |= row[k];
    out[(size_t)y * W + x] = a ? 1 : 0;
}

__global__ __launch_bounds__(256) void k_dilate_cols(int H, int W, int r, const uint8_t* __restrict__ in,
                                                     uint8_t* __restrict__ out) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    uint8_t a = 0;
    for (int k = max(0, y - r); k <= min(H - 1, y + r); ++k) a |= in[(size_t)k * W + x];
    out[(size_t)y * W + x] = a ? 1 : 0;
}

__device__ __forceinline__ int mask_coord(float u, int n) {
    const float c = fminf(fmaxf(u, 0.0f), (float)(n - 1));  // NaN -> 0 (fmaxf returns the non-NaN operand)
    return (int)c;
}

__device__ __forceinline__ bool outside_mask(const float* Pm, float x, float y, float z, int H, int W,
                                             const uint8_t* __restrict__ mask) {
    const float a = Pm[0] * x + Pm[1] * y + Pm[2] * z + Pm[3];
    const float b = Pm[4] * x + Pm[5] * y + Pm[6] * z + Pm[7];
    const float c = Pm[8] * x + Pm[9] * y + Pm[10] * z + Pm[11];
    const int px = mask_coord(a / c, W), py = mask_coord(b / c, H);
    return mask[(size_t)py * W + px] == 0;
}

__global__ __launch_bounds__(256) void k_points_outside_mask(int N, const float* __restrict__ xyz,
                                                             const float* __restrict__ K, const float* __restrict__ E,
                                                             int H, int W, const uint8_t* __restrict__ mask, int n_key,
                                                             const float* __restrict__ keypts, uint8_t* __restrict__ out) {
    float Pm[12];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) Pm[4 * r + c] = K[3 * r] * E[c] + K[3 * r + 1] * E[4 + c] + K[3 * r + 2] * E[8 + c];
    // every workgroup re-derives the keypoint override (a few dozen points): no second launch, no global flag
    int key_out = 0;
    for (int k = threadIdx.x; k < n_key; k += 256)
        key_out |= outside_mask(Pm, keypts[3 * k], keypts[3 * k + 1], keypts[3 * k + 2], H, W, mask) ? 1 : 0;
    const bool override_all = __syncthreads_or(key_out) != 0;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const bool o = outside_mask(Pm, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], H, W, mask);
    out[i] = (o && !override_all) ? 1 : 0;
}

// mean_k |x_i - keypoint_k| > thresh: the keypoint-distance pruning test of on_after_backward
// (src/modules/hand_dynamic.py:210-218: torch.cdist(posed_xyz, keypoints).mean(1) > 0.2), direct differences
__global__ __launch_bounds__(256) void k_keypoint_far(int N, const float* __restrict__ xyz, int n_key,
                                                      const float* __restrict__ keypts, float thresh,
                                                      uint8_t* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    float acc = 0.f;
    for (int k = 0; k < n_key; ++k) {  // wave-uniform keypoint address: scalar loads
        const float dx = x - keypts[3 * k], dy = y - keypts[3 * k + 1], dz = z - keypts[3 * k + 2];
        acc += sqrtf(dx * dx + dy * dy + dz * dz);
    }
    out[i] = (acc / (float)n_key > thresh) ? 1 : 0;
}

__global__ __launch_bounds__(256) void k_l1_grad(int64_t count, const float4* __restrict__ a,
                                                 const float4* __restrict__ b, float scale,
                                                 float4* __restrict__ g, float* __restrict__ loss_sum,
                                                 const float* a1, const float* b1, float* g1) {
    __shared__ float s_part[4];
    const int64_t n4 = count >> 2;
    float acc = 0.f;
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n4; k += (int64_t)gridDim.x * 256) {
        const float4 x = a[k], y = b[k];
        const float d0 = x.x - y.x, d1 = x.y - y.y, d2 = x.z - y.z, d3 = x.w - y.w;
        acc += fabsf(d0) + fabsf(d1) + fabsf(d2) + fabsf(d3);
        float4 o;
        o.x = d0 > 0.f ? scale : (d0 < 0.f ? -scale : 0.f);
        o.y = d1 > 0.f ? scale : (d1 < 0.f ? -scale : 0.f);
        o.z = d2 > 0.f ? scale : (d2 < 0.f ? -scale : 0.f);
        o.w = d3 > 0.f ? scale : (d3 < 0.f ? -scale : 0.f);
        g[k] = o;
    }
    if (blockIdx.x == 0) {  // tail
        for (int64_t k = (n4 << 2) + threadIdx.x; k < count; k += 256) {
            const float d = a1[k] - b1[k];
            acc += fabsf(d);
            g1[k] = d > 0.f ? scale : (d < 0.f ? -scale : 0.f);
        }
    }
    acc = mgr_wave_sum63(acc);
    if ((threadIdx.x & 63) == 63) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0 && loss_sum) atomicAdd(loss_sum, (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]));
}

// ---------------------------------------------------------------------------
// host entries
// ---------------------------------------------------------------------------
extern "C" int mgr_skin_weights_fwd(int N, const float* xyz, const float* grid, int D, int H, int W,
                                    int B, int grid_stride, const float* center3, const float* scale3,
                                    float* out_w, void* stream_) {
    if (N < 0 || B <= 0 || B > MGR_MAX_BONES || D <= 0 || H <= 0 || W <= 0)
        return mgr_fail(MGR_EINVAL, "mgr_skin_weights_fwd: bad sizes");
    if (N == 0) return MGR_OK;
    if (!xyz || !grid || !center3 || !scale3 || !out_w) return mgr_fail(MGR_EINVAL, "mgr_skin_weights_fwd: null pointer");
    hipStream_t stream = (hipStream_t)stream_;
    if (grid_stride != B && grid_stride != SKIN_BP) return mgr_fail(MGR_EINVAL, "mgr_skin_weights_fwd: grid_stride must be B or 24");
    if (grid_stride == SKIN_BP && B <= SKIN_BP && ((uintptr_t)grid & 15) == 0) {
        static const bool one_thread = getenv("MGR_SKIN_FWD") && !strcmp(getenv("MGR_SKIN_FWD"), "thread");   // (A/B switch)
        if (one_thread) {
            MGR_PROF("k_skin_fwd24", stream);
            hipLaunchKernelGGL(k_skin_fwd24, dim3((N + 255) / 256), dim3(256), 0, stream, N, xyz, (const float4*)grid, D, H, W, B,
                               center3, scale3, out_w);
        } else {
            MGR_PROF("k_skin_fwd24x8", stream);
            hipLaunchKernelGGL(k_skin_fwd24x8, dim3((unsigned)(((size_t)N * 8 + 255) / 256)), dim3(256), 0, stream, N, xyz,
                               (const float4*)grid, D, H, W, B, center3, scale3, out_w);
        }
        MGR_LAUNCH_CHECK("k_skin_fwd24", stream, 0);
        return MGR_OK;
    }
    if (grid_stride != B) return mgr_fail(MGR_EINVAL, "mgr_skin_weights_fwd: padded grid must be 16-byte aligned with B <= 24");
    { MGR_PROF("k_skin_fwd", stream); hipLaunchKernelGGL(k_skin_fwd, dim3((N + 255) / 256), dim3(256), 0, stream, N, xyz, grid, D, H, W, B,
                       center3, scale3, out_w); }
    MGR_LAUNCH_CHECK("k_skin_fwd", stream, 0);
    return MGR_OK;
}

static int skin_weights_bwd_impl(int N, const float* xyz, const float* grid, int D, int H, int W, int B, int grid_stride,
                                 const float* center3, const float* scale3, const float* dL_dw, float* dL_dxyz,
                                 int accumulate, const uint32_t* index, const uint32_t* index_count, int max_count,
                                 void* stream_) {
    if (N < 0 || B <= 0 || B > MGR_MAX_BONES || D <= 0 || H <= 0 || W <= 0)
        return mgr_fail(MGR_EINVAL, "mgr_skin_weights_bwd: bad sizes");
    if (N == 0) return MGR_OK;
    if (!xyz || !grid || !center3 || !scale3 || !dL_dw || !dL_dxyz || (index && !index_count))
        return mgr_fail(MGR_EINVAL, "mgr_skin_weights_bwd: null pointer");
    hipStream_t stream = (hipStream_t)stream_;
    const int n_threads = index ? max_count : N;
    if (n_threads <= 0) return MGR_OK;
    if (grid_stride != B && grid_stride != SKIN_BP) return mgr_fail(MGR_EINVAL, "mgr_skin_weights_bwd: grid_stride must be B or 24");
    if (grid_stride == SKIN_BP && B <= SKIN_BP && ((uintptr_t)grid & 15) == 0) {
        { MGR_PROF("k_skin_bwd24", stream); hipLaunchKernelGGL(k_skin_bwd24, dim3((n_threads + 255) / 256), dim3(256), 0, stream, N, xyz, (const float4*)grid, D, H, W, B,
                           center3, scale3, dL_dw, dL_dxyz, accumulate, index, index_count); }
        MGR_LAUNCH_CHECK("k_skin_bwd24", stream, 0);
        return MGR_OK;
    }
    if (grid_stride != B) return mgr_fail(MGR_EINVAL, "mgr_skin_weights_bwd: padded grid must be 16-byte aligned with B <= 24");
    { MGR_PROF("k_skin_bwd", stream); hipLaunchKernelGGL(k_skin_bwd, dim3((n_threads + 255) / 256), dim3(256), 0, stream, N, xyz, grid, D, H, W, B,
                       center3, scale3, dL_dw, dL_dxyz, accumulate, index, index_count); }
    MGR_LAUNCH_CHECK("k_skin_bwd", stream, 0);
    return MGR_OK;
}

extern "C" int mgr_skin_weights_bwd(int N, const float* xyz, const float* grid, int D, int H, int W,
                                    int B, int grid_stride, const float* center3, const float* scale3,
                                    const float* dL_dw, float* dL_dxyz, int accumulate, void* stream_) {
    return skin_weights_bwd_impl(N, xyz, grid, D, H, W, B, grid_stride, center3, scale3, dL_dw, dL_dxyz, accumulate, nullptr,
                                 nullptr, 0, stream_);
}

extern "C" int mgr_skin_weights_bwd_indexed(int N, const float* xyz, const float* grid, int D, int H, int W, int B,
                                            int grid_stride, const float* center3, const float* scale3,
                                            const float* dL_dw, float* dL_dxyz, const uint32_t* index,
                                            const uint32_t* index_count, int max_count, void* stream_) {
    if (!index || !index_count || max_count < 0) return mgr_fail(MGR_EINVAL, "mgr_skin_weights_bwd_indexed: bad arguments");
    return skin_weights_bwd_impl(N, xyz, grid, D, H, W, B, grid_stride, center3, scale3, dL_dw, dL_dxyz, 1, index, index_count,
                                 max_count, stream_);
}

static bool tf_rows_ok(int tfr) { return tfr == 12 || tfr == 16; }

extern "C" int mgr_lbs_cov_fwd_rows(int P, int N, int B, const float* xyz, const float* log_scale,
                                    const float* rot, const float* skin_w, const float* transforms,
                                    float* posed_xyz, float* posed_cov, float* tf, int tf_row_floats, void* stream_) {
    if (P <= 0 || N < 0 || (skin_w && (B <= 0 || B > MGR_MAX_BONES)) || !tf_rows_ok(tf_row_floats))
        return mgr_fail(MGR_EINVAL, "mgr_lbs_cov_fwd: bad sizes");
    if (N == 0) return MGR_OK;
    if (!xyz || !log_scale || !rot || !posed_xyz || !posed_cov || (skin_w && !transforms))
        return mgr_fail(MGR_EINVAL, "mgr_lbs_cov_fwd: null pointer");
    hipStream_t stream = (hipStream_t)stream_;
    { MGR_PROF("k_lbs_fwd", stream); hipLaunchKernelGGL(k_lbs_fwd, dim3((N + 255) / 256, P), dim3(256), 0, stream, N, B, xyz, log_scale, rot,
                       skin_w, transforms, posed_xyz, posed_cov, tf, tf_row_floats); }
    MGR_LAUNCH_CHECK("k_lbs_fwd", stream, 0);
    return MGR_OK;
}
extern "C" int mgr_lbs_cov_fwd(int P, int N, int B, const float* xyz, const float* log_scale,
                               const float* rot, const float* skin_w, const float* transforms,
                               float* posed_xyz, float* posed_cov, float* tf, void* stream_) {
    return mgr_lbs_cov_fwd_rows(P, N, B, xyz, log_scale, rot, skin_w, transforms, posed_xyz, posed_cov, tf, 12, stream_);
}

extern "C" int mgr_lbs_cov_bwd_rows(int P, int N, int B, const float* xyz, const float* log_scale,
                                    const float* rot, const float* skin_w, const float* transforms,
                                    const float* dL_dposed_xyz, const float* dL_dposed_cov,
                                    const float* dL_dtf, int tf_row_floats, float* dL_dxyz, float* dL_dlog_scale,
                                    float* dL_drot, float* dL_dw, void* stream_) {
    if (P <= 0 || N < 0 || (skin_w && (B <= 0 || B > MGR_MAX_BONES)) || !tf_rows_ok(tf_row_floats))
        return mgr_fail(MGR_EINVAL, "mgr_lbs_cov_bwd: bad sizes");
    if (N == 0) return MGR_OK;
    if (!xyz || !log_scale || !rot || !dL_dposed_xyz || !dL_dposed_cov || !dL_dxyz || !dL_dlog_scale ||
        !dL_drot || (skin_w && (!transforms || !dL_dw)))
        return mgr_fail(MGR_EINVAL, "mgr_lbs_cov_bwd: null pointer");
    hipStream_t stream = (hipStream_t)stream_;
    { MGR_PROF("k_lbs_bwd", stream); hipLaunchKernelGGL(k_lbs_bwd, dim3((N + 255) / 256), dim3(256), 0, stream, P, N, B, xyz, log_scale, rot,
                       skin_w, transforms, dL_dposed_xyz, dL_dposed_cov, dL_dtf, dL_dxyz, dL_dlog_scale,
                       dL_drot, dL_dw, tf_row_floats); }
    MGR_LAUNCH_CHECK("k_lbs_bwd", stream, 0);
    return MGR_OK;
}
extern "C" int mgr_lbs_cov_bwd(int P, int N, int B, const float* xyz, const float* log_scale,
                               const float* rot, const float* skin_w, const float* transforms,
                               const float* dL_dposed_xyz, const float* dL_dposed_cov,
                               const float* dL_dtf, float* dL_dxyz, float* dL_dlog_scale,
                               float* dL_drot, float* dL_dw, void* stream_) {
    return mgr_lbs_cov_bwd_rows(P, N, B, xyz, log_scale, rot, skin_w, transforms, dL_dposed_xyz, dL_dposed_cov, dL_dtf, 12, dL_dxyz,
                                dL_dlog_scale, dL_drot, dL_dw, stream_);
}

extern "C" int mgr_sh_color_fwd_rows(int V, int N, const float* sh, const float* xyz, int64_t stride_xyz,
                                     const float* tf, int64_t stride_tf, int tf_row_floats, const float* cams, float* colors,
                                     void* stream_) {
    if (V <= 0 || N < 0 || !tf_rows_ok(tf_row_floats)) return mgr_fail(MGR_EINVAL, "mgr_sh_color_fwd: bad sizes");
    if (N == 0) return MGR_OK;
    if (!sh || !xyz || !cams || !colors) return mgr_fail(MGR_EINVAL, "mgr_sh_color_fwd: null pointer");
    hipStream_t stream = (hipStream_t)stream_;
    { MGR_PROF("k_sh_fwd", stream); hipLaunchKernelGGL(k_sh_fwd, dim3((N + 255) / 256, V), dim3(256), 0, stream, N, sh, xyz, stride_xyz, tf,
                       stride_tf, cams, colors, tf_row_floats); }
    MGR_LAUNCH_CHECK("k_sh_fwd", stream, 0);
    return MGR_OK;
}
extern "C" int mgr_sh_color_fwd(int V, int N, const float* sh, const float* xyz, int64_t stride_xyz,
                                const float* tf, int64_t stride_tf, const float* cams, float* colors,
                                void* stream_) {
    return mgr_sh_color_fwd_rows(V, N, sh, xyz, stride_xyz, tf, stride_tf, 12, cams, colors, stream_);
}

extern "C" int mgr_sh_color_bwd_rows(int V, int N, const float* sh, const float* xyz, int64_t stride_xyz,
                                     const float* tf, int64_t stride_tf, int tf_row_floats, const float* cams,
                                     const float* dL_dcolors, float* dL_dsh, float* dL_dxyz, float* dL_dtf,
                                     void* stream_) {
    if (V <= 0 || N < 0 || !tf_rows_ok(tf_row_floats)) return mgr_fail(MGR_EINVAL, "mgr_sh_color_bwd: bad sizes");
    if (N == 0) return MGR_OK;
    if (!sh || !xyz || !cams || !dL_dcolors || !dL_dsh || !dL_dxyz || (tf && !dL_dtf))
        return mgr_fail(MGR_EINVAL, "mgr_sh_color_bwd: null pointer");
    hipStream_t stream = (hipStream_t)stream_;
    { MGR_PROF("k_sh_bwd", stream); hipLaunchKernelGGL(k_sh_bwd, dim3((N + 255) / 256), dim3(256), 0, stream, V, N, sh, xyz, stride_xyz, tf,
                       stride_tf, cams, dL_dcolors, dL_dsh, dL_dxyz, dL_dtf, tf_row_floats); }
    MGR_LAUNCH_CHECK("k_sh_bwd", stream, 0);
    return MGR_OK;
}
extern "C" int mgr_sh_color_bwd(int V, int N, const float* sh, const float* xyz, int64_t stride_xyz,
                                const float* tf, int64_t stride_tf, const float* cams,
                                const float* dL_dcolors, float* dL_dsh, float* dL_dxyz, float* dL_dtf,
                                void* stream_) {
    return mgr_sh_color_bwd_rows(V, N, sh, xyz, stride_xyz, tf, stride_tf, 12, cams, dL_dcolors, dL_dsh, dL_dxyz, dL_dtf, stream_);
}

// ---------------------------------------------------------------------------
// Host-glue kernels of the reference-shaped route: what render_gaussians / TrainingModule.forward do with a dozen tiny
// torch launches each per step (camera table from the settings tuple; posed @ inv(rest) of the bone transforms).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_pack_camera(float tfx, float tfy, const float* __restrict__ view,
                                                    const float* __restrict__ proj, const float* __restrict__ campos,
                                                    float* __restrict__ out) {
    const int t = threadIdx.x;
    if (t >= MGR_CAM_FLOATS) return;
    float v = 0.f;
    if (t == 0) v = tfx;
    else if (t == 1) v = tfy;
    else if (t < 18) v = view ? view[t - 2] : 0.f;
    else if (t < 34) v = proj ? proj[t - 18] : 0.f;
    else if (t < 37) v = campos ? campos[t - 34] : 0.f;
    out[t] = v;
}

extern "C" int mgr_pack_camera(float tanfovx, float tanfovy, const float* view16, const float* proj16, const float* campos3,
                               float* out40, void* stream_) {
    if (!out40) return mgr_fail(MGR_EINVAL, "mgr_pack_camera: null pointer");
    hipStream_t stream = (hipStream_t)stream_;
    hipLaunchKernelGGL(k_pack_camera, dim3(1), dim3(64), 0, stream, tanfovx, tanfovy, view16, proj16, campos3, out40);
    MGR_LAUNCH_CHECK("k_pack_camera", stream, 0);
    return MGR_OK;
}

// One thread per bone: T_b = posed_b @ inv(rest_b), the inverse by Gauss-Jordan elimination with partial pivoting in fp32
// (the operation count and pivoting of the LU route torch.linalg.inv takes; results agree with it to fp32 roundoff).
// A singular rest matrix gives non-finite entries, like the reference's inverse raising would stop the step.
__global__ __launch_bounds__(64) void k_bone_transforms(int B, int background, const float* __restrict__ posed,
                                                        const float* __restrict__ rest, float* __restrict__ out) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B + (background ? 1 : 0)) return;
    float* o = out + (size_t)b * 16;
    if (b >= B) {
#pragma unroll
        for (int k = 0; k < 16; ++k) o[k] = (k % 5 == 0) ? 1.f : 0.f;
        return;
    }
    float a[4][8];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            a[r][c] = rest[(size_t)b * 16 + r * 4 + c];
            a[r][4 + c] = r == c ? 1.f : 0.f;
        }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        int piv = c;
        float best = fabsf(a[c][c]);
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r > c && fabsf(a[r][c]) > best) { best = fabsf(a[r][c]); piv = r; }
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r == piv && r != c) {
#pragma unroll
                for (int k = 0; k < 8; ++k) { const float t = a[c][k]; a[c][k] = a[r][k]; a[r][k] = t; }
            }
        const float inv = 1.0f / a[c][c];
#pragma unroll
        for (int k = 0; k < 8; ++k) a[c][k] *= inv;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r != c) {
                const float f = a[r][c];
#pragma unroll
                for (int k = 0; k < 8; ++k) a[r][k] -= f * a[c][k];
            }
    }
    const float* p = posed + (size_t)b * 16;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) acc += p[r * 4 + k] * a[k][4 + c];
            o[r * 4 + c] = acc;
        }
}

extern "C" int mgr_bone_transforms(int B, int background, const float* posed, const float* rest, float* out, void* stream_) {
    if (B < 0) return mgr_fail(MGR_EINVAL, "mgr_bone_transforms: bad size");
    const int n = B + (background ? 1 : 0);
    if (n == 0) return MGR_OK;
    if (!out || (B > 0 && (!posed || !rest))) return mgr_fail(MGR_EINVAL, "mgr_bone_transforms: null pointer");
    hipStream_t stream = (hipStream_t)stream_;
    hipLaunchKernelGGL(k_bone_transforms, dim3((n + 63) / 64), dim3(64), 0, stream, B, background, posed, rest, out);
    MGR_LAUNCH_CHECK("k_bone_transforms", stream, 0);
    return MGR_OK;
}

extern "C" int mgr_project_points(int N, const float* xyz, const float* K9, const float* E12, float* uv,
                                  void* stream_) {
    if (N < 0) return mgr_fail(MGR_EINVAL, "mgr_project_points: bad sizes");
    if (N == 0) return MGR_OK;
    if (!xyz || !K9 || !E12 || !uv) return mgr_fail(MGR_EINVAL, "mgr_project_points: null pointer");
    hipStream_t stream = (hipStream_t)stream_;
    { MGR_PROF("k_project", stream); hipLaunchKernelGGL(k_project, dim3((N + 255) / 256), dim3(256), 0, stream, N, xyz, K9, E12, uv); }
    MGR_LAUNCH_CHECK("k_project", stream, 0);
    return MGR_OK;
}

extern "C" int mgr_dilate_mask(int H, int W, int kernel_size, const uint8_t* mask, uint8_t* scratch, uint8_t* out,
                               void* stream_) {
    if (H <= 0 || W <= 0 || kernel_size < 1 || !(kernel_size & 1)) return mgr_fail(MGR_EINVAL, "mgr_dilate_mask: bad sizes");
    if (!mask || !scratch || !out) return mgr_fail(MGR_EINVAL, "mgr_dilate_mask: null pointer");
    hipStream_t stream = (hipStream_t)stream_;
    const dim3 grid((W + 255) / 256, H);
    hipLaunchKernelGGL(k_dilate_rows, grid, dim3(256), 0, stream, H, W, kernel_size / 2, mask, scratch);
    hipLaunchKernelGGL(k_dilate_cols, grid, dim3(256), 0, stream, H, W, kernel_size / 2, (const uint8_t*)scratch, out);
    MGR_LAUNCH_CHECK("k_dilate", stream, 0);
    return MGR_OK;
}

extern "C" int mgr_points_outside_mask(int N, const float* xyz, const float* K9, const float* E12, int H, int W,
                                       const uint8_t* mask, int n_keypoints, const float* keypoints, uint8_t* out,
                                       void* stream_) {
    if (N < 0 || H <= 0 || W <= 0 || n_keypoints < 0) return mgr_fail(MGR_EINVAL, "mgr_points_outside_mask: bad sizes");
    if (N == 0) return MGR_OK;
    if (!xyz || !K9 || !E12 || !mask || !out || (n_keypoints > 0 && !keypoints))
        return mgr_fail(MGR_EINVAL, "mgr_points_outside_mask: null pointer");
    hipStream_t stream = (hipStream_t)stream_;
    hipLaunchKernelGGL(k_points_outside_mask, dim3((N + 255) / 256), dim3(256), 0, stream, N, xyz, K9, E12, H, W, mask,
                       n_keypoints, keypoints, out);
    MGR_LAUNCH_CHECK("k_points_outside_mask", stream, 0);
    return MGR_OK;
}

extern "C" int mgr_keypoint_far_mask(int N, const float* xyz, int n_keypoints, const float* keypoints, float thresh,
                                     uint8_t* out, void* stream_) {
    if (N < 0 || n_keypoints <= 0) return mgr_fail(MGR_EINVAL, "mgr_keypoint_far_mask: bad sizes");
    if (N == 0) return MGR_OK;
    if (!xyz || !keypoints || !out) return mgr_fail(MGR_EINVAL, "mgr_keypoint_far_mask: null pointer");
    hipStream_t stream = (hipStream_t)stream_;
    hipLaunchKernelGGL(k_keypoint_far, dim3((N + 255) / 256), dim3(256), 0, stream, N, xyz, n_keypoints, keypoints, thresh, out);
    MGR_LAUNCH_CHECK("k_keypoint_far", stream, 0);
    return MGR_OK;
}

extern "C" int mgr_l1_loss_grad(int64_t count, const float* a, const float* b, float scale, float* dL_da,
                                float* loss_sum, void* stream_) {
    if (count < 0) return mgr_fail(MGR_EINVAL, "mgr_l1_loss_grad: bad sizes");
    if (count == 0) return MGR_OK;
    if (!a || !b || !dL_da) return mgr_fail(MGR_EINVAL, "mgr_l1_loss_grad: null pointer");
    if (((uintptr_t)a | (uintptr_t)b | (uintptr_t)dL_da) & 15)
        return mgr_fail(MGR_EINVAL, "mgr_l1_loss_grad: pointers must be 16-byte aligned");
    hipStream_t stream = (hipStream_t)stream_;
    int64_t n4 = count >> 2;
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > 1024) blocks = 1024;  // one same-address atomic per workgroup at the end
    if (blocks < 1) blocks = 1;
    { MGR_PROF("k_l1_grad", stream); hipLaunchKernelGGL(k_l1_grad, dim3(blocks), dim3(256), 0, stream, count, (const float4*)a, (const float4*)b,
                       scale, (float4*)dL_da, loss_sum, a, b, dL_da); }
    MGR_LAUNCH_CHECK("k_l1_grad", stream, 0);
    return MGR_OK;
}
