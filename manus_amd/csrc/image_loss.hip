// Fused image loss of the MANUS training step for gfx950: L1 + "SSIM" forward and backward in
// one pass over the rendered views.
//
// Replaces, for images kept in the rasterizer's (V,3,H,W) layout,
//     l1_loss(pred, gt)                         /root/reference/src/utils/loss_utils.py:22-27
//     1 - ssim(pred, gt)                        /root/reference/src/utils/loss_utils.py:39-97
// as they are called by loss_func               /root/reference/src/modules/base.py:323-365
// (weights 0.8 / 0.2, config/HAND_GAUSSIAN.yaml:22-23) and their autograd backward.
//
// The reference calls ssim() on HWC images: `channel = img1.size(-3)` (loss_utils.py:58) is then
// the image HEIGHT and F.conv2d(..., groups=channel) slides the 11x11 window over the (W, 3)
// plane of every image row, zero padded by 5 in both directions.  That is what is computed here
// (and what the oracle restates): per row h, per statistic s in {x, y, xx, yy, xy},
//     E_s[w][c] = sum_i sum_j g[i] g[j] s[w+i-5][c+j-5]            (zero outside the plane)
// = an 11-tap filter along w of the channel mix  sum_c' g[5+c'-c] s[w][c'].
// The backward is the same (symmetric) filter applied to three derivative maps
//     D1 = dS/dmu1 (total), D2 = dS/dE[xx], D3 = dS/dE[xy],
//     dS/dx[p] = F(D1)[p] + 2 x[p] F(D2)[p] + y[p] F(D3)[p].
//
// One 128-thread workgroup owns 246 consecutive w of TWO image rows.  The two rows are carried as
// the two halves of packed fp32 registers (v_pk_fma_f32: the rows are independent, so every
// filter tap is one packed FMA), each thread produces two neighbouring positions from a sliding
// window of 12 LDS reads instead of 2 x 11, and every intermediate stays in LDS: x, y are staged
// for w-10..w+265 (the derivative maps are needed 5 beyond the outputs, and they need the
// statistics 5 beyond that), only the gradient is written.  The two loss sums are written per
// workgroup and folded by a second tiny kernel: no float atomics, the loss value is reproducible.
#include "mgr_common.h"
#include "il_list.h"                // IL_T, IL_ND, IL_H1, IL_W, ILS_T, ILS_MAXW, ILM_R, il_blocks, the mapped list's workgroup
#include <cstdlib>

#define IL_ACC_SLOTS 64             // slot triples the workgroups of k_image_loss add their fixed-point sums to
#define IL_NX (IL_ND + 2 * IL_H1)   // 266 staged positions

struct IlWindow {
    float g[11];
};

typedef mgr_v2f v2f;
__device__ __forceinline__ v2f il_v2(float a) { v2f r = {a, a}; return r; }

__global__ __launch_bounds__(IL_T) void k_image_loss(int H, int W, const float* __restrict__ pred,
                                                     const float* __restrict__ target, IlWindow win, float w_l1,
                                                     float w_ssim, float grad_scale, float* __restrict__ dL_dpred,
                                                     float2* __restrict__ partial, const uint32_t* __restrict__ work_list,
                                                     const uint32_t* __restrict__ work_count, int gxb, int gyb,
                                                     unsigned long long* __restrict__ acc_slots) {
    // channel-mixed statistics (5 x 3 rows of positions) and, later, the channel-mixed derivative
    // maps (3 x 3); .x = image row h0, .y = image row h0 + 1
    __shared__ v2f s_mix[5][3][IL_NX];
    __shared__ float s_red[4];
    const int tid = threadIdx.x;
    const uint32_t n_work = *work_count;
    // The loss sums (round 6).  Thread 0 keeps the workgroup's sums of its spans as 64-bit FIXED-POINT integers (2^-32: integer
    // addition is associative, so the total does not depend on which workgroup took which span or on the order the atomics land
    // in -- the value is reproducible, as the fold over the per-span sums in index order was) and adds them to one of IL_ACC_SLOTS
    // slot triples at its end (no return value, no fence: the kernel boundary orders them); k_image_loss_fold adds the slots and
    // the spans the list pass finished (every pixel of those has SSIM 1 and no L1: all pixels minus the listed ones).  The fold
    // over 34 560 per-span sums took 12 us between this kernel and the backward blend; over 64 slots it is a minimal launch.
    long long acc_l1 = 0, acc_ss = 0;
    unsigned long long acc_px = 0;
  for (uint32_t wi = blockIdx.x; wi < n_work; wi += gridDim.x) {   // the workgroups k_image_loss_scan found different
    const uint32_t bid = work_list[wi];
    const int bxi = (int)(bid % (uint32_t)gxb), byi = (int)((bid / (uint32_t)gxb) % (uint32_t)gyb), v = (int)(bid / (uint32_t)(gxb * gyb));
    const int w0 = bxi * IL_W, h0 = byi * 2;
    const bool row1 = h0 + 1 < H;
    __syncthreads();  // LDS reuse across work items
    const size_t plane = (size_t)H * W;
    const float* px = pred + (size_t)v * 3 * plane + (size_t)h0 * W;
    const float* py = target + (size_t)v * 3 * plane + (size_t)h0 * W;
    // channel mix matrix M[c][c'] = g[5 + c' - c]; g is symmetric
    const v2f m0 = il_v2(win.g[5]), m1 = il_v2(win.g[4]), m2 = il_v2(win.g[3]);

    for (int t = tid; t < IL_NX; t += IL_T) {
        const int w = w0 - 2 * IL_H1 + t;
        v2f x[3], y[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) x[c] = y[c] = il_v2(0.f);
        if (w >= 0 && w < W) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                x[c].x = px[c * plane + w];
                y[c].x = py[c * plane + w];
                if (row1) {
                    x[c].y = px[c * plane + W + w];
                    y[c].y = py[c * plane + W + w];
                }
            }
        }
        const v2f q[5][3] = {{x[0], x[1], x[2]},
                             {y[0], y[1], y[2]},
                             {x[0] * x[0], x[1] * x[1], x[2] * x[2]},
                             {y[0] * y[0], y[1] * y[1], y[2] * y[2]},
                             {x[0] * y[0], x[1] * y[1], x[2] * y[2]}};
#pragma unroll
        for (int s5 = 0; s5 < 5; ++s5) {
            s_mix[s5][0][t] = m0 * q[s5][0] + m1 * q[s5][1] + m2 * q[s5][2];
            s_mix[s5][1][t] = m1 * q[s5][0] + m0 * q[s5][1] + m1 * q[s5][2];
            s_mix[s5][2][t] = m2 * q[s5][0] + m1 * q[s5][1] + m0 * q[s5][2];
        }
    }
    __syncthreads();

    // statistics -> SSIM value and derivative maps at the two positions u0, u0 + 1 (w = w0 - 5 + u)
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    const int u0 = 2 * tid;
    v2f ssim_acc = il_v2(0.f);
    v2f d[2][3][3];  // [position][map][channel]
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        v2f e[2][5];
#pragma unroll
        for (int s5 = 0; s5 < 5; ++s5) {
            v2f win12[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) win12[k] = s_mix[s5][c][u0 + k];
            v2f a = il_v2(0.f), b = il_v2(0.f);
#pragma unroll
            for (int i = 0; i < 11; ++i) {
                const v2f gi = il_v2(win.g[i]);
                a += gi * win12[i];
                b += gi * win12[i + 1];
            }
            e[0][s5] = a;
            e[1][s5] = b;
        }
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            const int w = w0 - IL_H1 + u0 + pp;
            const v2f mu1 = e[pp][0], mu2 = e[pp][1];
            const v2f s11 = e[pp][2] - mu1 * mu1, s22 = e[pp][3] - mu2 * mu2, s12 = e[pp][4] - mu1 * mu2;
            const v2f A = 2.f * mu1 * mu2 + C1, B = 2.f * s12 + C2;
            const v2f Cc = mu1 * mu1 + mu2 * mu2 + C1, Dd = s11 + s22 + C2;
            v2f iC, iD;
            iC.x = __builtin_amdgcn_rcpf(Cc.x); iC.y = __builtin_amdgcn_rcpf(Cc.y);  // 1 ulp; the bar is 1e-4 relative
            iD.x = __builtin_amdgcn_rcpf(Dd.x); iD.y = __builtin_amdgcn_rcpf(Dd.y);
            const v2f inv = iC * iD;
            const v2f S = A * B * inv;
            const bool okw = w >= 0 && w < W;
            v2f keep = {okw ? 1.f : 0.f, (okw && row1) ? 1.f : 0.f};   // positions outside the image have no SSIM value
            const bool own = u0 + pp >= IL_H1 && u0 + pp < IL_H1 + IL_W;
            if (own) ssim_acc += S * keep;
            d[pp][0][c] = keep * (2.f * mu2 * (B - A) * inv - S * 2.f * mu1 * iC + S * 2.f * mu1 * iD);
            d[pp][1][c] = keep * (-S * iD);
            d[pp][2][c] = keep * (2.f * A * inv);
        }
    }
    __syncthreads();  // everyone is done reading the statistics
    // channel mix of the derivative maps (M is symmetric), reusing s_mix[0..2]
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const v2f a = d[pp][k][0], b = d[pp][k][1], c = d[pp][k][2];
            s_mix[k][0][u0 + pp] = m0 * a + m1 * b + m2 * c;
            s_mix[k][1][u0 + pp] = m1 * a + m0 * b + m1 * c;
            s_mix[k][2][u0 + pp] = m2 * a + m1 * b + m0 * c;
        }
    }
    __syncthreads();

    // outputs o0, o0 + 1 (w = w0 + o), derivative index = o + 5, filter taps at o + i
    v2f l1_acc = il_v2(0.f);
    const int o0 = 2 * tid;
    if (o0 < IL_W) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            v2f f[2][3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                v2f win12[12];
#pragma unroll
                for (int j = 0; j < 12; ++j) win12[j] = s_mix[k][c][o0 + j];
                v2f a = il_v2(0.f), b = il_v2(0.f);
#pragma unroll
                for (int i = 0; i < 11; ++i) {
                    const v2f gi = il_v2(win.g[i]);
                    a += gi * win12[i];
                    b += gi * win12[i + 1];
                }
                f[0][k] = a;
                f[1][k] = b;
            }
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                const int w = w0 + o0 + pp;
                if (o0 + pp < IL_W && w < W) {
                    v2f x, y;
                    x.x = px[c * plane + w];
                    y.x = py[c * plane + w];
                    x.y = row1 ? px[c * plane + W + w] : 0.f;
                    y.y = row1 ? py[c * plane + W + w] : 0.f;
                    const v2f dS = f[pp][0] + 2.f * x * f[pp][1] + y * f[pp][2];
                    const v2f df = x - y;
                    v2f ad = {fabsf(df.x), fabsf(df.y)};
                    l1_acc += ad;
                    const v2f sgn = {df.x > 0.f ? 1.f : (df.x < 0.f ? -1.f : 0.f), df.y > 0.f ? 1.f : (df.y < 0.f ? -1.f : 0.f)};
                    const v2f gr = grad_scale * (w_l1 * sgn - w_ssim * dS);
                    float* go = dL_dpred + (size_t)v * 3 * plane + (size_t)h0 * W + c * plane + w;
                    go[0] = gr.x;
                    if (row1) go[W] = gr.y;
                }
            }
        }
    }
    // workgroup sums (fixed order)
    float l1_sum = mgr_wave_sum63(l1_acc.x + l1_acc.y);
    float ssim_sum = mgr_wave_sum63(ssim_acc.x + ssim_acc.y);
    if ((tid & 63) == 63) {
        s_red[tid >> 6] = l1_sum;
        s_red[2 + (tid >> 6)] = ssim_sum;
    }
    __syncthreads();
    if (tid == 0) {
        const float pa = s_red[0] + s_red[1], pb = s_red[2] + s_red[3];
        partial[bid] = make_float2(pa, pb);
        acc_l1 += __double2ll_rn((double)pa * 4294967296.0);
        acc_ss += __double2ll_rn((double)pb * 4294967296.0);
        acc_px += (unsigned long long)((min((bxi + 1) * IL_W, W) - bxi * IL_W) * 3 * (row1 ? 2 : 1));
    }
  }
    if (tid == 0 && acc_px) {
        unsigned long long* sl = acc_slots + 3 * (blockIdx.x & (IL_ACC_SLOTS - 1));
        atomicAdd(sl, (unsigned long long)acc_l1);
        atomicAdd(sl + 1, (unsigned long long)acc_ss);
        atomicAdd(sl + 2, acc_px);
    }
}

// Pass 1, almost no LDS, full occupancy.  Every (246 px x 2 rows) span whose rendered and target pixels are identical
// over the whole +-10 px staged range -- the shared background, most of a capture-like frame -- is finished here: all
// statistic pairs coincide, the SSIM map is 1 and at its maximum, so the gradient is 0 up to fp32 noise (the
// reference computes ~1e-9 there) and the L1 term vanishes; the span gets zeros and its output count.  The other
// spans are appended to the work list of k_image_loss.
// One workgroup takes a whole pair of image rows: 16-byte loads along w (W % 4 == 0; otherwise dwords) mark the
// pixels that differ in a bitmap, the spans are classified from the bitmap, zeros are written with 16-byte stores.
__global__ __launch_bounds__(ILS_T) void k_image_loss_scan(int H, int W, int gxb, const float* __restrict__ pred,
                                                           const float* __restrict__ target, float* __restrict__ dL_dpred,
                                                           float2* __restrict__ partial, uint32_t* __restrict__ work_list,
                                                           uint32_t* __restrict__ work_count) {
    __shared__ uint32_t s_diff[ILS_MAXW / 32];   // bit w: pixel column w differs in some channel / row
    __shared__ uint32_t s_ident[(ILS_MAXW / IL_W + 32) / 32];  // bit b: span b is identical
    const int tid = threadIdx.x;
    const int h0 = blockIdx.x * 2, v = blockIdx.y;
    const bool row1 = h0 + 1 < H;
    const size_t plane = (size_t)H * W;
    const float* px = pred + (size_t)v * 3 * plane + (size_t)h0 * W;
    const float* py = target + (size_t)v * 3 * plane + (size_t)h0 * W;
    float* pg = dL_dpred + (size_t)v * 3 * plane + (size_t)h0 * W;
    const int nwords = (W + 31) / 32;
    for (int k = tid; k < nwords; k += ILS_T) s_diff[k] = 0;
    if (tid < (int)(sizeof(s_ident) / 4)) s_ident[tid] = 0;
    __syncthreads();
    const int rows = row1 ? 2 : 1;
    const bool vec = (W & 3) == 0 && ((((uintptr_t)pred) | ((uintptr_t)target) | ((uintptr_t)dL_dpred)) & 15) == 0;
    if (vec) {
        const int n4 = W >> 2;
        for (int q = tid; q < n4; q += ILS_T) {
            uint32_t d = 0;
#pragma unroll
            for (int c = 0; c < 3; ++c)
                for (int r = 0; r < rows; ++r) {
                    const float4 a = *(const float4*)(px + c * plane + (size_t)r * W + 4 * q);
                    const float4 b = *(const float4*)(py + c * plane + (size_t)r * W + 4 * q);
                    d |= (a.x != b.x ? 1u : 0u) | (a.y != b.y ? 2u : 0u) | (a.z != b.z ? 4u : 0u) | (a.w != b.w ? 8u : 0u);
                }
            if (d) atomicOr(&s_diff[q >> 3], d << ((q & 7) * 4));
        }
    } else {
        for (int w = tid; w < W; w += ILS_T) {
            uint32_t d = 0;
#pragma unroll
            for (int c = 0; c < 3; ++c)
                for (int r = 0; r < rows; ++r) d |= px[c * plane + (size_t)r * W + w] != py[c * plane + (size_t)r * W + w] ? 1u : 0u;
            if (d) atomicOr(&s_diff[w >> 5], 1u << (w & 31));
        }
    }
    __syncthreads();
    // classify the spans: identical iff no differing pixel in [w0 - 10, w0 + 256) (clipped to the image)
    for (int b = tid; b < gxb; b += ILS_T) {
        const int lo = max(b * IL_W - 2 * IL_H1, 0), hi = min(b * IL_W + IL_ND, W);  // [lo, hi)
        uint32_t any = 0;
        for (int k = lo >> 5; k <= (hi - 1) >> 5; ++k) {
            uint32_t m = s_diff[k];
            const int base = k << 5;
            if (lo > base) m &= ~0u << (lo - base);
            if (hi < base + 32) m &= ~0u >> (base + 32 - hi);
            any |= m;
        }
        const uint32_t bid = ((uint32_t)v * gridDim.x + blockIdx.x) * (uint32_t)gxb + (uint32_t)b;
        if (any) {
            work_list[atomicAdd(work_count, 1u)] = bid;
        } else {
            atomicOr(&s_ident[b >> 5], 1u << (b & 31));
            const int wcnt = min((b + 1) * IL_W, W) - b * IL_W;
            partial[bid] = make_float2(0.f, (float)(wcnt * 3 * rows));
        }
    }
    __syncthreads();
    // zero gradient for the identical spans
    if (vec) {
        const int n4 = W >> 2;
        for (int q = tid; q < n4; q += ILS_T) {
            const int w = 4 * q, b0 = w / IL_W, b1 = (w + 3) / IL_W;
            const bool i0 = (s_ident[b0 >> 5] >> (b0 & 31)) & 1u, i1 = (s_ident[b1 >> 5] >> (b1 & 31)) & 1u;
            if (i0 && i1) {
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    for (int r = 0; r < rows; ++r) *(float4*)(pg + c * plane + (size_t)r * W + w) = make_float4(0.f, 0.f, 0.f, 0.f);
            } else if (i0 || i1) {
                for (int e = 0; e < 4; ++e) {
                    const int be = (w + e) / IL_W;
                    if ((s_ident[be >> 5] >> (be & 31)) & 1u)
                        for (int c = 0; c < 3; ++c)
                            for (int r = 0; r < rows; ++r) pg[c * plane + (size_t)r * W + w + e] = 0.f;
                }
            }
        }
    } else {
        for (int w = tid; w < W; w += ILS_T) {
            const int b = w / IL_W;
            if ((s_ident[b >> 5] >> (b & 31)) & 1u)
                for (int c = 0; c < 3; ++c)
                    for (int r = 0; r < rows; ++r) pg[c * plane + (size_t)r * W + w] = 0.f;
        }
    }
}

// Pass 1 when the rendered image comes from this library's rasterizer and its tile lists are at hand
// (mgr_image_loss_tiles): a span is "identical" when every 16x16 tile under its +-10 px staged range is EMPTY -- the
// rasterizer writes exactly the background colour there -- and the target equals that colour over the range.  The
// rendered image is not read at all, the target only under empty tiles, and nothing is written for identical spans:
// their gradient is never read either (the backward blend only visits tiles that hold Gaussians).  Spans under
// non-empty tiles go to the work list without being looked at.
__global__ __launch_bounds__(ILS_T) void k_image_loss_list(int H, int W, int gxb, const float* __restrict__ target,
                                                           const float* __restrict__ bg, const uint32_t* __restrict__ tile_start,
                                                           float2* __restrict__ partial, uint32_t* __restrict__ work_list,
                                                           uint32_t* __restrict__ work_count, const uint32_t* __restrict__ tmap,
                                                           uint32_t* __restrict__ tmap_out) {
    // tmap: the column masks below, computed by an earlier call for the same target and background (tmap_out) -- a target
    // image is a constant of its view, so a caller that renders the view again hands the masks back instead of having
    // 24 bytes per pixel of target read per step (mgr_image_loss_target_map / mgr_image_loss_tiles_list_mapped)
    __shared__ uint32_t s_diff[ILS_MAXW / 32];   // bit w: target column w differs from the background in some channel / row
    __shared__ uint32_t s_cand[(ILS_MAXW / IL_W + 32) / 32];   // bit b: every tile under span b is empty
    __shared__ uint32_t s_n, s_base;
    const int tid = threadIdx.x;
    if (tid == 0) s_n = 0;
    const int h0 = blockIdx.x * 2, v = blockIdx.y;
    const bool row1 = h0 + 1 < H;
    const size_t plane = (size_t)H * W;
    const float* py = target + (size_t)v * 3 * plane + (size_t)h0 * W;
    const int nwords = (W + 31) / 32;
    const int gxt = (W + 15) / 16, T = gxt * ((H + 15) / 16);
    const uint32_t* trow = tile_start + (size_t)v * T + (size_t)(h0 >> 4) * gxt;   // both rows of the pair lie in this tile row
    const size_t map_row = ((size_t)v * gridDim.x + blockIdx.x) * (size_t)nwords;
    for (int k = tid; k < nwords; k += ILS_T) s_diff[k] = tmap ? tmap[map_row + k] : 0u;
    if (tid < (int)(sizeof(s_cand) / 4)) s_cand[tid] = 0;
    __syncthreads();
    if (tile_start)
    for (int b = tid; b < gxb; b += ILS_T) {
        const int lo = max(b * IL_W - 2 * IL_H1, 0), hi = min(b * IL_W + IL_ND, W);  // [lo, hi)
        const int t0 = lo >> 4, t1 = (hi - 1) >> 4;
        if (trow[t1 + 1] == trow[t0]) atomicOr(&s_cand[b >> 5], 1u << (b & 31));   // consecutive tiles: one difference of the scan
    }
    // the target against the background colour, everywhere (ungated loads: the tile test above and these are one round
    // trip; 85 % of a capture-like frame lies under empty tiles anyway)
    const int rows = row1 ? 2 : 1;
    const float b0 = bg[0], b1 = bg[1], b2 = bg[2];
    const bool vec = (W & 3) == 0 && (((uintptr_t)target) & 15) == 0;
    if (tmap) {
    } else if (vec) {
        const int n4 = W >> 2;
        for (int q = tid; q < n4; q += ILS_T) {
            uint32_t d = 0;
            for (int rr = 0; rr < rows; ++rr) {
                const float4 c0 = *(const float4*)(py + (size_t)rr * W + 4 * q), c1 = *(const float4*)(py + plane + (size_t)rr * W + 4 * q),
                             c2 = *(const float4*)(py + 2 * plane + (size_t)rr * W + 4 * q);
                d |= ((c0.x != b0 || c1.x != b1 || c2.x != b2) ? 1u : 0u) | ((c0.y != b0 || c1.y != b1 || c2.y != b2) ? 2u : 0u) |
                     ((c0.z != b0 || c1.z != b1 || c2.z != b2) ? 4u : 0u) | ((c0.w != b0 || c1.w != b1 || c2.w != b2) ? 8u : 0u);
            }
            if (d) atomicOr(&s_diff[q >> 3], d << ((q & 7) * 4));
        }
    } else {
        for (int w = tid; w < W; w += ILS_T) {
            uint32_t d = 0;
            for (int rr = 0; rr < rows; ++rr)
                d |= (py[(size_t)rr * W + w] != b0 || py[plane + (size_t)rr * W + w] != b1 || py[2 * plane + (size_t)rr * W + w] != b2) ? 1u : 0u;
            if (d) atomicOr(&s_diff[w >> 5], 1u << (w & 31));
        }
    }
    __syncthreads();
    if (tmap_out) {
        for (int k = tid; k < nwords; k += ILS_T) tmap_out[map_row + k] = s_diff[k];
        if (!tile_start) return;     // (map only)
    }
    // classify; the listed spans of the row pair take ONE slot range of the global list (one returning atomic per
    // workgroup: thousands of them on a single counter cost more than reading the target)
    uint32_t my_rank[(ILS_MAXW / IL_W + ILS_T) / ILS_T];
    int nmine = 0;
    for (int b = tid; b < gxb; b += ILS_T, ++nmine) {
        const int lo = max(b * IL_W - 2 * IL_H1, 0), hi = min(b * IL_W + IL_ND, W);  // [lo, hi)
        uint32_t any = ((s_cand[b >> 5] >> (b & 31)) & 1u) ? 0u : 1u;
        for (int k = lo >> 5; k <= (hi - 1) >> 5 && !any; ++k) {
            uint32_t m = s_diff[k];
            const int base = k << 5;
            if (lo > base) m &= ~0u << (lo - base);
            if (hi < base + 32) m &= ~0u >> (base + 32 - hi);
            any |= m;
        }
        my_rank[nmine] = any ? atomicAdd(&s_n, 1u) : 0xFFFFFFFFu;
        if (!any) {
            const uint32_t bid = ((uint32_t)v * gridDim.x + blockIdx.x) * (uint32_t)gxb + (uint32_t)b;
            const int wcnt = min((b + 1) * IL_W, W) - b * IL_W;
            partial[bid] = make_float2(0.f, (float)(wcnt * 3 * rows));
        }
    }
    __syncthreads();
    if (tid == 0 && s_n) s_base = atomicAdd(work_count, s_n);
    __syncthreads();
    nmine = 0;
    for (int b = tid; b < gxb; b += ILS_T, ++nmine)
        if (my_rank[nmine] != 0xFFFFFFFFu)
            work_list[s_base + my_rank[nmine]] = ((uint32_t)v * gridDim.x + blockIdx.x) * (uint32_t)gxb + (uint32_t)b;
}

// The same classification from precomputed column masks (tmap: what k_image_loss_list leaves in tmap_out): one thread per
// (row pair, span), ILM_R row pairs per workgroup -- the list kernel above spends its time on one returning atomic per row
// pair, all on one counter (4320 of them at 8 views of 1080p: 31 us for a few kilobytes of work); here a workgroup
// collects its listed spans in LDS and takes ONE slot range.
__global__ __launch_bounds__(ILS_T) void k_image_loss_list_mapped(IlListArgs a) {
    __shared__ uint32_t s2[2];
    il_list_mapped_block(a, (int)blockIdx.x, (int)blockIdx.y, s2);
}

// the loss from the slot sums of k_image_loss: sums[0] = sum |pred - target|, sums[1] = sum of the SSIM map (the spans the list
// pass finished contribute one per pixel and channel: total_px3 minus the listed ones), sums[2] = the caller's loss value
__global__ __launch_bounds__(IL_ACC_SLOTS) void k_image_loss_fold(unsigned long long* __restrict__ acc_slots, long long total_px3,
                                                                  float* __restrict__ sums, float ca, float cb, float cc,
                                                                  uint32_t* __restrict__ work_count) {
    __shared__ unsigned long long s_fin[3 * IL_ACC_SLOTS];
    const int tid = threadIdx.x;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        s_fin[q * IL_ACC_SLOTS + tid] = acc_slots[3 * tid + q];
        acc_slots[3 * tid + q] = 0ull;                          // left zero for the next call (a kept workspace needs no memset)
    }
    __syncthreads();
    if (tid == 0) {
        long long t_l1 = 0, t_ss = 0;
        unsigned long long t_px = 0;
        for (int k = 0; k < IL_ACC_SLOTS; ++k) {
            t_l1 += (long long)s_fin[k];
            t_ss += (long long)s_fin[IL_ACC_SLOTS + k];
            t_px += s_fin[2 * IL_ACC_SLOTS + k];
        }
        const double ta = (double)t_l1 * (1.0 / 4294967296.0);
        const double tb = (double)t_ss * (1.0 / 4294967296.0) + (double)((unsigned long long)total_px3 - t_px);
        sums[0] = (float)ta;
        sums[1] = (float)tb;
        sums[2] = (float)((double)ca * ta + (double)cb * tb + (double)cc);  // the caller's loss value, no host-side arithmetic
        *work_count = 0u;   // the list is consumed: a caller that keeps the workspace builds the next one without a memset
    }
}


// byte offset of the sum slots (64-bit atomics: aligned) behind the per-span sums, the work list and the 256 bytes of counters
static size_t il_slots_offset(int64_t nb) { return (((size_t)nb * (sizeof(float2) + sizeof(uint32_t)) + 256) + 63) & ~(size_t)63; }
extern "C" size_t mgr_image_loss_workspace_bytes(int V, int H, int W) {
    if (V <= 0 || H <= 0 || W <= 0) return 0;
    return il_slots_offset(il_blocks(V, H, W)) + 3 * IL_ACC_SLOTS * sizeof(unsigned long long);  // per-span sums | work list | counters | sum slots
}

static int image_loss_impl(int V, int H, int W, const float* pred, const float* target, const float* bg3,
                           const uint32_t* tile_start, float w_l1, float w_ssim, float grad_scale, float loss_offset,
                           float* dL_dpred, float* sums, void* workspace, size_t workspace_bytes, void* stream_, int phase = 0,
                           const uint32_t* tmap = nullptr, bool clean = false) {
    // phase 0: everything; 1: the span list only (needs target, bg3, tile_start, workspace); 2: the rest, on a list already built
    if (V <= 0 || H <= 0 || W <= 0) return mgr_fail(MGR_EINVAL, "mgr_image_loss: bad sizes");
    if ((!target && !(phase == 1 && tmap)) || !workspace || (phase != 1 && (!pred || !dL_dpred || !sums))) return mgr_fail(MGR_EINVAL, "mgr_image_loss: null pointer");
    if (H > 65535 || V > 65535) return mgr_fail(MGR_EINVAL, "mgr_image_loss: H and V must fit a grid dimension");
    if (workspace_bytes < mgr_image_loss_workspace_bytes(V, H, W))
        return mgr_fail(MGR_ENOMEM, "mgr_image_loss: workspace too small");
    hipStream_t stream = (hipStream_t)stream_;
    // the reference's window: gaussian(11, 1.5) in fp32, normalised (loss_utils.py:39-47)
    IlWindow win;
    float sum = 0.f;
    for (int i = 0; i < 11; ++i) {
        win.g[i] = (float)exp(-(double)((i - 5) * (i - 5)) / (2.0 * 1.5 * 1.5));
        sum += win.g[i];
    }
    for (int i = 0; i < 11; ++i) win.g[i] /= sum;
    const dim3 grid((W + IL_W - 1) / IL_W, (H + 1) / 2, V);
    const int64_t nb = il_blocks(V, H, W);
    if (nb >= (1ll << 32)) return mgr_fail(MGR_EINVAL, "mgr_image_loss: image batch too large");
    float2* partial = (float2*)workspace;
    uint32_t* work_list = (uint32_t*)((char*)workspace + (size_t)nb * sizeof(float2));
    uint32_t* work_count = (uint32_t*)((char*)workspace + (size_t)nb * (sizeof(float2) + sizeof(uint32_t)) + 64);
    if (W > ILS_MAXW) return mgr_fail(MGR_EINVAL, "mgr_image_loss: image wider than 16384");
    unsigned long long* acc_slots = (unsigned long long*)((char*)workspace + il_slots_offset(nb));
    if (phase != 2 && !clean)      // (work counter and sum slots: a kept workspace leaves them zero itself)
        MGR_HIP(hipMemsetAsync(work_count, 0, (size_t)((char*)(acc_slots + 3 * IL_ACC_SLOTS) - (char*)work_count), stream));
    if (phase == 2) {
    } else if (tile_start && tmap) {
        MGR_PROF("k_image_loss_list", stream);
        const IlListArgs la = il_list_args(V, H, W, tmap, tile_start, workspace);
        hipLaunchKernelGGL(k_image_loss_list_mapped, dim3((unsigned)la.nbx, V), dim3(ILS_T), 0, stream, la);
    } else if (tile_start) {
        MGR_PROF("k_image_loss_list", stream);
        hipLaunchKernelGGL(k_image_loss_list, dim3(grid.y, V), dim3(ILS_T), 0, stream, H, W, (int)grid.x, target, bg3, tile_start,
                           partial, work_list, work_count, tmap, (uint32_t*)nullptr);
    } else {
        MGR_PROF("k_image_loss_scan", stream);
        hipLaunchKernelGGL(k_image_loss_scan, dim3(grid.y, V), dim3(ILS_T), 0, stream, H, W, (int)grid.x, pred, target, dL_dpred,
                           partial, work_list, work_count);
    }
    if (phase == 1) return MGR_OK;
    {
        MGR_PROF("k_image_loss", stream);
#ifndef IL_GRID
#define IL_GRID (256 * 5)
#endif
        const int64_t pb = nb < IL_GRID ? nb : IL_GRID;   // persistent: 5 workgroups of 32 KB LDS per CU are resident
        hipLaunchKernelGGL(k_image_loss, dim3((unsigned)pb), dim3(IL_T), 0, stream, H, W, pred, target, win, w_l1, w_ssim,
                           grad_scale, dL_dpred, partial, (const uint32_t*)work_list, (const uint32_t*)work_count,
                           (int)grid.x, (int)grid.y, acc_slots);
    }
    hipLaunchKernelGGL(k_image_loss_fold, dim3(1), dim3(IL_ACC_SLOTS), 0, stream, acc_slots, (long long)V * H * W * 3,
                       sums, grad_scale * w_l1, -grad_scale * w_ssim, loss_offset, work_count);
    MGR_LAUNCH_CHECK("k_image_loss", stream, 0);
    return MGR_OK;
}

extern "C" int mgr_image_loss(int V, int H, int W, const float* pred, const float* target, float w_l1, float w_ssim,
                              float grad_scale, float loss_offset, float* dL_dpred, float* sums, void* workspace,
                              size_t workspace_bytes, void* stream_) {
    return image_loss_impl(V, H, W, pred, target, nullptr, nullptr, w_l1, w_ssim, grad_scale, loss_offset, dL_dpred, sums,
                           workspace, workspace_bytes, stream_);
}

extern "C" int mgr_image_loss_tiles(int V, int H, int W, const float* pred, const float* target, const float* bg3,
                                    const uint32_t* tile_start, float w_l1, float w_ssim, float grad_scale, float loss_offset,
                                    float* dL_dpred, float* sums, void* workspace, size_t workspace_bytes, void* stream_) {
    if (!bg3 || !tile_start) return mgr_fail(MGR_EINVAL, "mgr_image_loss_tiles: null pointer");
    return image_loss_impl(V, H, W, pred, target, bg3, tile_start, w_l1, w_ssim, grad_scale, loss_offset, dL_dpred, sums,
                           workspace, workspace_bytes, stream_);
}

/* The two halves of mgr_image_loss_tiles: the span list needs the forward's tile offsets but not its image, so it can be
 * built on another stream while the forward blend runs (engine: mgr_views_forward split at the blend). */
extern "C" int mgr_image_loss_tiles_list(int V, int H, int W, const float* target, const float* bg3, const uint32_t* tile_start,
                                         void* workspace, size_t workspace_bytes, void* stream_) {
    if (!bg3 || !tile_start) return mgr_fail(MGR_EINVAL, "mgr_image_loss_tiles_list: null pointer");
    return image_loss_impl(V, H, W, nullptr, target, bg3, tile_start, 0.f, 0.f, 0.f, 0.f, nullptr, nullptr, workspace,
                           workspace_bytes, stream_, 1);
}

extern "C" int mgr_image_loss_tiles_finish(int V, int H, int W, const float* pred, const float* target, float w_l1, float w_ssim,
                                           float grad_scale, float loss_offset, float* dL_dpred, float* sums, void* workspace,
                                           size_t workspace_bytes, void* stream_) {
    return image_loss_impl(V, H, W, pred, target, nullptr, nullptr, w_l1, w_ssim, grad_scale, loss_offset, dL_dpred, sums,
                           workspace, workspace_bytes, stream_, 2);
}

/* The target-vs-background column masks of V target images (what the span list of mgr_image_loss_tiles derives from the
 * target every call): V x ceil(H / 2) x ceil(W / 32) words.  A target image is a constant of its view: a caller that renders
 * the same views again computes the masks once and builds the list from them (mgr_image_loss_tiles_list_mapped) without
 * reading the targets. */
extern "C" size_t mgr_image_loss_target_map_words(int V, int H, int W) {
    if (V <= 0 || H <= 0 || W <= 0) return 0;
    return (size_t)V * ((H + 1) / 2) * ((W + 31) / 32);
}

extern "C" int mgr_image_loss_target_map(int V, int H, int W, const float* target, const float* bg3, uint32_t* map, void* stream_) {
    if (V <= 0 || H <= 0 || W <= 0) return mgr_fail(MGR_EINVAL, "mgr_image_loss_target_map: bad sizes");
    if (!target || !bg3 || !map) return mgr_fail(MGR_EINVAL, "mgr_image_loss_target_map: null pointer");
    if (H > 65535 || V > 65535 || W > ILS_MAXW) return mgr_fail(MGR_EINVAL, "mgr_image_loss_target_map: image too large");
    hipLaunchKernelGGL(k_image_loss_list, dim3((H + 1) / 2, V), dim3(ILS_T), 0, (hipStream_t)stream_, H, W, (W + IL_W - 1) / IL_W,
                       target, bg3, (const uint32_t*)nullptr, (float2*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr,
                       (const uint32_t*)nullptr, map);
    MGR_LAUNCH_CHECK("k_image_loss_list", (hipStream_t)stream_, 0);
    return MGR_OK;
}

extern "C" int mgr_image_loss_tiles_list_mapped(int V, int H, int W, const uint32_t* map, const float* bg3, const uint32_t* tile_start,
                                                void* workspace, size_t workspace_bytes, int workspace_kept, void* stream_) {
    if (!map || !bg3 || !tile_start) return mgr_fail(MGR_EINVAL, "mgr_image_loss_tiles_list_mapped: null pointer");
    return image_loss_impl(V, H, W, nullptr, nullptr, bg3, tile_start, 0.f, 0.f, 0.f, 0.f, nullptr, nullptr, workspace,
                           workspace_bytes, stream_, 1, map, workspace_kept != 0);
}
