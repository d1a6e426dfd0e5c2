// Optimizer step and densification of the Gaussian parameter model for gfx950.
//
// Replaces, on the six leaf tensors of GaussianModel (/root/reference/src/models/gaussian.py),
//     torch.optim.Adam(l, lr=0, eps=1e-15).step()      gaussian.py:133-142 (six named groups)
//     densify_and_prune                                 gaussian.py:310-333
//       densify_and_clone / densify_and_split           gaussian.py:288-308 / 254-286
//       prune_points + optimizer-state surgery          gaussian.py:148-252
//     reset_opacity                                     gaussian.py:148-151
// The reference rebuilds every nn.Parameter and both Adam moments with boolean-mask indexing and
// torch.cat, group by group, three times per densification.  Here one kernel classifies every
// Gaussian, a two-phase scan turns the four keep flags into output positions in the reference's
// order  [kept originals | kept clones | kept split copies 0 | kept split copies 1],  and one kernel
// writes all new rows (parameters, both moments, skin weights) from a source map.
#include <hip/hip_fp16.h>

#include "mgr_common.h"

// ---------------------------------------------------------------------------
// fp16 storage copy of the higher-order SH coefficients (BASELINE config 5): _features_rest (N,45) fp32 -> (N,48)
// fp16 rows (16-byte aligned, 3 halves of padding).  The optimizer keeps the fp32 master copy; this refreshes the
// copy the render kernels read.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sh_to_half(int N, const float* __restrict__ f_rest, __half* __restrict__ out) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;   // one thread per output half
    if (e >= (size_t)N * 48) return;
    const size_t i = e / 48, k = e % 48;
    out[e] = __float2half_rn(k < 45 ? f_rest[i * 45 + k] : 0.0f);
}

extern "C" int mgr_sh_to_half(int N, const float* f_rest, void* out_half, void* stream_) {
    if (N < 0) return mgr_fail(MGR_EINVAL, "mgr_sh_to_half: bad size");
    if (N == 0) return MGR_OK;
    if (!f_rest || !out_half) return mgr_fail(MGR_EINVAL, "mgr_sh_to_half: null pointer");
    hipStream_t stream = (hipStream_t)stream_;
    const size_t total = (size_t)N * 48;
    { MGR_PROF("k_sh_to_half", stream); hipLaunchKernelGGL(k_sh_to_half, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, N, f_rest, (__half*)out_half); }
    MGR_LAUNCH_CHECK("k_sh_to_half", stream, 0);
    return MGR_OK;
}

// ---------------------------------------------------------------------------
// Adam: all groups in one launch
// ---------------------------------------------------------------------------
#define OPT_MAX_GROUPS 8
struct AdamGroups {
    float* p[OPT_MAX_GROUPS];
    const float* g[OPT_MAX_GROUPS];
    float* m[OPT_MAX_GROUPS];
    float* v[OPT_MAX_GROUPS];
    long long quad_end[OPT_MAX_GROUPS];  // running end, in 4-element chunks
    long long count[OPT_MAX_GROUPS];
    float step_size[OPT_MAX_GROUPS];     // lr / (1 - beta1^t), t = the group's own step count
    float inv_sqrt_bc2[OPT_MAX_GROUPS];  // 1 / sqrt(1 - beta2^t)
    int n;
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float omb1, float b2, float omb2,
                                         float inv_sqrt_bc2, float eps, float step_size) {
    // torch: exp_avg.lerp_(grad, 1-b1); exp_avg_sq.mul_(b2).addcmul_(grad, grad, value=1-b2);
    //        denom = exp_avg_sq.sqrt() / sqrt(bc2) + eps; param.addcdiv_(exp_avg, denom, value=-step_size)
    m = m + (g - m) * omb1;
    v = v * b2 + omb2 * (g * g);
    const float denom = sqrtf(v) * inv_sqrt_bc2 + eps;
    p = p - step_size * (m / denom);
}

__global__ __launch_bounds__(256) void k_adam(AdamGroups G, long long total_quads, float omb1, float b2, float omb2,
                                              float eps) {
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < total_quads; q += (long long)gridDim.x * 256) {
        int gi = 0;
        long long base = 0;
#pragma unroll
        for (int k = 0; k < OPT_MAX_GROUPS; ++k)
            if (k < G.n - 1 && q >= G.quad_end[k]) {
                gi = k + 1;
                base = G.quad_end[k];
            }
        const long long e0 = (q - base) * 4, cnt = G.count[gi];
        float* p = G.p[gi] + e0;
        const float* g = G.g[gi] + e0;
        float* m = G.m[gi] + e0;
        float* v = G.v[gi] + e0;
        const float ss = G.step_size[gi], inv_sqrt_bc2 = G.inv_sqrt_bc2[gi];
        const bool vec = e0 + 4 <= cnt && ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0;
        if (vec) {
            float4 P = *(float4*)p, M = *(float4*)m, V = *(float4*)v;
            const float4 Gd = *(const float4*)g;
            adam_one(P.x, Gd.x, M.x, V.x, omb1, b2, omb2, inv_sqrt_bc2, eps, ss);
            adam_one(P.y, Gd.y, M.y, V.y, omb1, b2, omb2, inv_sqrt_bc2, eps, ss);
            adam_one(P.z, Gd.z, M.z, V.z, omb1, b2, omb2, inv_sqrt_bc2, eps, ss);
            adam_one(P.w, Gd.w, M.w, V.w, omb1, b2, omb2, inv_sqrt_bc2, eps, ss);
            *(float4*)p = P; *(float4*)m = M; *(float4*)v = V;
        } else {
            for (int k = 0; k < 4 && e0 + k < cnt; ++k) {
                float P = p[k], M = m[k], V = v[k];
                adam_one(P, g[k], M, V, omb1, b2, omb2, inv_sqrt_bc2, eps, ss);
                p[k] = P; m[k] = M; v[k] = V;
            }
        }
    }
}

// Every group carries its own step count, like torch.optim.Adam's per-parameter state["step"]: the reference
// replaces a leaf (reset_opacity, densification, pruning) before optimizer.step() in on_after_backward, the new
// nn.Parameter has no .grad, and Adam skips it -- that group's count then lags the others (gaussian.py:153-165,
// hand_dynamic.py:193-224,259-277).  steps[k] <= 0 skips group k.
extern "C" int mgr_adam_step_groups(int n_groups, const int64_t* counts, float* const* params, const float* const* grads,
                                    float* const* exp_avg, float* const* exp_avg_sq, const double* lrs,
                                    const int64_t* steps, double beta1, double beta2, double eps, void* stream_) {
    if (n_groups <= 0 || n_groups > OPT_MAX_GROUPS) return mgr_fail(MGR_EINVAL, "mgr_adam_step: bad sizes");
    if (!counts || !params || !grads || !exp_avg || !exp_avg_sq || !lrs || !steps)
        return mgr_fail(MGR_EINVAL, "mgr_adam_step: null pointer");
    AdamGroups G;
    long long q = 0;
    G.n = n_groups;
    for (int k = 0; k < OPT_MAX_GROUPS; ++k) {
        const bool on = k < n_groups && steps[k] > 0;
        if (on && (counts[k] < 0 || (counts[k] > 0 && (!params[k] || !grads[k] || !exp_avg[k] || !exp_avg_sq[k]))))
            return mgr_fail(MGR_EINVAL, "mgr_adam_step: null group pointer");
        const double bc1 = on ? 1.0 - pow(beta1, (double)steps[k]) : 1.0, bc2 = on ? 1.0 - pow(beta2, (double)steps[k]) : 1.0;
        G.p[k] = on ? params[k] : nullptr;
        G.g[k] = on ? grads[k] : nullptr;
        G.m[k] = on ? exp_avg[k] : nullptr;
        G.v[k] = on ? exp_avg_sq[k] : nullptr;
        G.count[k] = on ? counts[k] : 0;
        q += on ? (counts[k] + 3) / 4 : 0;
        G.quad_end[k] = q;
        G.step_size[k] = on ? (float)(lrs[k] / bc1) : 0.f;
        G.inv_sqrt_bc2[k] = on ? (float)(1.0 / sqrt(bc2)) : 0.f;
    }
    if (q == 0) return MGR_OK;
    hipStream_t stream = (hipStream_t)stream_;
    long long blocks = (q + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    {
        MGR_PROF("k_adam", stream);
        hipLaunchKernelGGL(k_adam, dim3((unsigned)blocks), dim3(256), 0, stream, G, q, (float)(1.0 - beta1), (float)beta2,
                           (float)(1.0 - beta2), (float)eps);
    }
    MGR_LAUNCH_CHECK("k_adam", stream, 0);
    return MGR_OK;
}

extern "C" int mgr_adam_step(int n_groups, const int64_t* counts, float* const* params, const float* const* grads,
                             float* const* exp_avg, float* const* exp_avg_sq, const double* lrs, int64_t step,
                             double beta1, double beta2, double eps, void* stream_) {
    if (n_groups <= 0 || n_groups > OPT_MAX_GROUPS || step < 1) return mgr_fail(MGR_EINVAL, "mgr_adam_step: bad sizes");
    int64_t steps[OPT_MAX_GROUPS];
    for (int k = 0; k < OPT_MAX_GROUPS; ++k) steps[k] = step;
    return mgr_adam_step_groups(n_groups, counts, params, grads, exp_avg, exp_avg_sq, lrs, steps, beta1, beta2, eps, stream_);
}

// ---------------------------------------------------------------------------
// add_densification_stats (+ the max_radii2D update of density_update): one launch for the three read-modify-writes
// the reference does with four torch ops per step (gaussian.py:335-338, gaussian_utils.py:470-473)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_dens_stats(int N, const float* __restrict__ g2, const float* __restrict__ vis,
                                                    const int32_t* __restrict__ radii, float* __restrict__ accum,
                                                    float* __restrict__ denom, float* __restrict__ maxr) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    accum[i] += g2[i];
    denom[i] += vis[i];
    maxr[i] = fmaxf(maxr[i], (float)radii[i]);    // (torch.maximum propagates NaN; neither side is ever NaN here)
}

extern "C" int mgr_add_densification_stats(int N, const float* grad2d_sum, const float* vis_count, const int32_t* radii_max,
                                           float* xyz_gradient_accum, float* denom, float* max_radii2D, void* stream_) {
    if (N < 0) return mgr_fail(MGR_EINVAL, "mgr_add_densification_stats: bad size");
    if (N == 0) return MGR_OK;
    if (!grad2d_sum || !vis_count || !radii_max || !xyz_gradient_accum || !denom || !max_radii2D)
        return mgr_fail(MGR_EINVAL, "mgr_add_densification_stats: null pointer");
    hipStream_t stream = (hipStream_t)stream_;
    { MGR_PROF("k_dens_stats", stream); hipLaunchKernelGGL(k_dens_stats, dim3((N + 255) / 256), dim3(256), 0, stream, N, grad2d_sum, vis_count, radii_max, xyz_gradient_accum, denom, max_radii2D); }
    MGR_LAUNCH_CHECK("k_dens_stats", stream, 0);
    return MGR_OK;
}

// ---------------------------------------------------------------------------
// reset_opacity: opacity <- inverse_sigmoid(min(sigmoid(opacity), 0.01)); moments zeroed
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_reset_opacity(int N, float* __restrict__ op, float* __restrict__ m,
                                                       float* __restrict__ v) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const float s = fminf(1.0f / (1.0f + expf(-op[i])), 0.01f);
    op[i] = logf(s / (1.0f - s));
    m[i] = 0.f;
    v[i] = 0.f;
}

extern "C" int mgr_reset_opacity(int N, float* opacity_logit, float* exp_avg, float* exp_avg_sq, void* stream_) {
    if (N < 0) return mgr_fail(MGR_EINVAL, "mgr_reset_opacity: bad size");
    if (N == 0) return MGR_OK;
    if (!opacity_logit || !exp_avg || !exp_avg_sq) return mgr_fail(MGR_EINVAL, "mgr_reset_opacity: null pointer");
    hipStream_t stream = (hipStream_t)stream_;
    hipLaunchKernelGGL(k_reset_opacity, dim3((N + 255) / 256), dim3(256), 0, stream, N, opacity_logit, exp_avg, exp_avg_sq);
    MGR_LAUNCH_CHECK("k_reset_opacity", stream, 0);
    return MGR_OK;
}

// ---------------------------------------------------------------------------
// densify_and_prune, planning: flags -> scans -> source map
// ---------------------------------------------------------------------------
// flag bits of one Gaussian
#define DF_KEEP_ORIG 1u    // stays (not split, not pruned)
#define DF_KEEP_CLONE 2u   // cloned and the clone survives the prune
#define DF_SEL_SPLIT 4u    // selected for splitting (indexes the noise rows)
#define DF_KEEP_SPLIT 8u   // its two children survive the prune

__global__ __launch_bounds__(1024) void k_dens_flags(int N, const float* __restrict__ accum,
                                                     const float* __restrict__ denom,
                                                     const float* __restrict__ log_scale,
                                                     const float* __restrict__ op_logit, float max_grad,
                                                     float min_opacity, float dense_extent, float big_extent,
                                                     uint32_t* __restrict__ flags, uint4* __restrict__ block_sums) {
    __shared__ uint32_t s_scan[64];
    const int i = blockIdx.x * 1024 + threadIdx.x;
    uint32_t f = 0;
    if (i < N) {
        float gr = accum[i] / denom[i];
        if (gr != gr) gr = 0.0f;  // grads[grads.isnan()] = 0 (gaussian.py:312)
        const float l0 = log_scale[3 * i], l1 = log_scale[3 * i + 1], l2 = log_scale[3 * i + 2];
        const float s0 = expf(l0), s1 = expf(l1), s2 = expf(l2);
        const float smax = fmaxf(fmaxf(s0, s1), s2);            // NaN-ignoring like the comparisons below need
        const bool nan_row = (l0 != l0) || (l1 != l1) || (l2 != l2);
        // a NaN scale makes torch.max(...) NaN and both comparisons false
        const bool clone = !nan_row && fabsf(gr) >= max_grad && smax <= dense_extent;   // gaussian.py:290-293
        const bool split = !nan_row && gr >= max_grad && smax > dense_extent;           // gaussian.py:259-262
        const bool low_op = 1.0f / (1.0f + expf(-op_logit[i])) < min_opacity;
        // big_extent = 0.1 * extent only when max_screen_size is set (gaussian.py:316-320), +inf otherwise
        const bool prune_self = low_op || smax > big_extent || nan_row;     // gaussian.py:315-327
        // children: scaling = log(exp(s) / (0.8 * 2)) (gaussian.py:268), same opacity
        const float c0 = expf(logf(s0 / 1.6f)), c1 = expf(logf(s1 / 1.6f)), c2 = expf(logf(s2 / 1.6f));
        const bool prune_child = low_op || fmaxf(fmaxf(c0, c1), c2) > big_extent || nan_row;
        if (!split && !prune_self) f |= DF_KEEP_ORIG;
        if (clone && !prune_self) f |= DF_KEEP_CLONE;
        if (split) f |= DF_SEL_SPLIT;
        if (split && !prune_child) f |= DF_KEEP_SPLIT;
        flags[i] = f;
    }
    // block totals of the four flags (phase A of the scan)
    uint32_t t[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint32_t c = (f >> k) & 1u;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) c += (uint32_t)__shfl_xor((int)c, d, 64);
        t[k] = c;
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) s_scan[(threadIdx.x >> 6) * 4 + k + 0] = t[k];
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        uint32_t a = 0;
        for (int w = 0; w < 16; ++w) a += s_scan[w * 4 + threadIdx.x];
        ((uint32_t*)&block_sums[blockIdx.x])[threadIdx.x] = a;
    }
}

// phase B: every block re-derives its bases from the block sums, ranks its Gaussians and writes
// map[pos] = source | kind << 30 (kind 0 original, 1 clone, 2 / 3 split copy 0 / 1) and, for split
// children, aux[pos] = rank among the split-selected Gaussians (the noise row of copy c is c*n_sel + rank).
__global__ __launch_bounds__(1024) void k_dens_map(int N, int nblk, const uint32_t* __restrict__ flags,
                                                   const uint4* __restrict__ block_sums, uint32_t* __restrict__ map,
                                                   uint32_t* __restrict__ aux, uint32_t* __restrict__ counts) {
    __shared__ uint32_t s_base[4], s_tot[4];
    __shared__ uint32_t s_w[4][17];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 4) {
        uint32_t before = 0, all = 0;
        for (int b = 0; b < nblk; ++b) {
            const uint32_t c = ((const uint32_t*)&block_sums[b])[tid];
            if (b < (int)blockIdx.x) before += c;
            all += c;
        }
        s_base[tid] = before;
        s_tot[tid] = all;
    }
    const int i = blockIdx.x * 1024 + tid;
    const uint32_t f = i < N ? flags[i] : 0u;
    uint32_t rank[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned long long b = __ballot((f >> k) & 1u);
        rank[k] = (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
        if (lane == 0) s_w[k][wave] = (uint32_t)__popcll(b);
    }
    __syncthreads();
    if (tid < 4) {
        uint32_t run = 0;
        for (int w = 0; w < 16; ++w) {
            const uint32_t c = s_w[tid][w];
            s_w[tid][w] = run;
            run += c;
        }
    }
    __syncthreads();
    const uint32_t n_keep = s_tot[0], n_clone = s_tot[1], n_sel = s_tot[2], n_ks = s_tot[3];
    if (blockIdx.x == 0 && tid == 0) {
        counts[0] = n_keep; counts[1] = n_clone; counts[2] = n_sel; counts[3] = n_ks;
        counts[4] = n_keep + n_clone + 2u * n_ks;
    }
    if (i >= N) return;
    uint32_t r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = s_base[k] + s_w[k][wave] + rank[k];
    if (f & DF_KEEP_ORIG) map[r[0]] = (uint32_t)i;
    if (f & DF_KEEP_CLONE) map[n_keep + r[1]] = (uint32_t)i | (1u << 30);
    if (f & DF_KEEP_SPLIT) {
        const uint32_t p0 = n_keep + n_clone + r[3], p1 = p0 + n_ks;
        map[p0] = (uint32_t)i | (2u << 30);
        map[p1] = (uint32_t)i | (3u << 30);
        aux[p0] = r[2];
        aux[p1] = r[2];
    }
}

extern "C" size_t mgr_densify_workspace_bytes(int N) {
    if (N <= 0) return 0;
    const size_t nblk = ((size_t)N + 1023) / 1024;
    return mgr_align((size_t)N * 4) + mgr_align(nblk * 16) + 2 * mgr_align((size_t)N * 2 * 4) + 256;
}

struct DensLayout {
    size_t flags, sums, map, aux, counts;
};
static DensLayout dens_layout(int N) {
    DensLayout L;
    const size_t nblk = ((size_t)N + 1023) / 1024;
    size_t o = 0;
    L.flags = o;  o += mgr_align((size_t)N * 4);
    L.sums = o;   o += mgr_align(nblk * 16);
    L.map = o;    o += mgr_align((size_t)N * 2 * 4);
    L.aux = o;    o += mgr_align((size_t)N * 2 * 4);
    L.counts = o;
    return L;
}

extern "C" int mgr_densify_plan(int N, const float* grad_accum, const float* denom, const float* log_scale,
                                const float* opacity_logit, float max_grad, float min_opacity, float extent,
                                float percent_dense, float max_screen_size, void* workspace, size_t workspace_bytes,
                                int64_t* counts_host, void* stream_) {
    if (N <= 0) return mgr_fail(MGR_EINVAL, "mgr_densify_plan: bad size");
    if (!grad_accum || !denom || !log_scale || !opacity_logit || !workspace || !counts_host)
        return mgr_fail(MGR_EINVAL, "mgr_densify_plan: null pointer");
    if (workspace_bytes < mgr_densify_workspace_bytes(N)) return mgr_fail(MGR_ENOMEM, "mgr_densify_plan: workspace too small");
    hipStream_t stream = (hipStream_t)stream_;
    const DensLayout L = dens_layout(N);
    char* ws = (char*)workspace;
    const int nblk = (N + 1023) / 1024;
    hipLaunchKernelGGL(k_dens_flags, dim3(nblk), dim3(1024), 0, stream, N, grad_accum, denom, log_scale, opacity_logit,
                       max_grad, min_opacity, percent_dense * extent,
                       // `if max_screen_size:` (gaussian.py:316): None / 0 disables BOTH size tests; the screen-size
                       // half (max_radii2D > max_screen_size) can never fire anyway, densification_postfix has
                       // zeroed max_radii2D by then (:249-251)
                       max_screen_size > 0.0f ? 0.1f * extent : __builtin_huge_valf(), (uint32_t*)(ws + L.flags),
                       (uint4*)(ws + L.sums));
    hipLaunchKernelGGL(k_dens_map, dim3(nblk), dim3(1024), 0, stream, N, nblk, (const uint32_t*)(ws + L.flags),
                       (const uint4*)(ws + L.sums), (uint32_t*)(ws + L.map), (uint32_t*)(ws + L.aux),
                       (uint32_t*)(ws + L.counts));
    MGR_LAUNCH_CHECK("k_dens_map", stream, 0);
    uint32_t h[5];
    MGR_HIP(hipMemcpyAsync(h, ws + L.counts, sizeof(h), hipMemcpyDeviceToHost, stream));
    MGR_HIP(hipStreamSynchronize(stream));  // the caller allocates the new tensors from these
    for (int k = 0; k < 5; ++k) counts_host[k] = h[k];
    return MGR_OK;
}

// ---------------------------------------------------------------------------
// prune_points(mask) (gaussian.py:185-203): the same scan / map machinery with keep = !mask; the rows are then
// written by mgr_densify_apply (all kind 0: parameters and both moments copied) and the per-Gaussian statistics
// by mgr_gather_rows.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_prune_flags(int N, const uint8_t* __restrict__ mask, uint32_t* __restrict__ flags,
                                                      uint4* __restrict__ block_sums) {
    __shared__ uint32_t s_w[16];
    const int i = blockIdx.x * 1024 + threadIdx.x;
    const uint32_t f = (i < N && mask[i] == 0) ? DF_KEEP_ORIG : 0u;
    if (i < N) flags[i] = f;
    const unsigned long long b = __ballot(f != 0u);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = (uint32_t)__popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t a = 0;
        for (int w = 0; w < 16; ++w) a += s_w[w];
        block_sums[blockIdx.x] = make_uint4(a, 0u, 0u, 0u);
    }
}

extern "C" int mgr_prune_plan(int N, const uint8_t* prune_mask, void* workspace, size_t workspace_bytes,
                              int64_t* counts_host, void* stream_) {
    if (N <= 0) return mgr_fail(MGR_EINVAL, "mgr_prune_plan: bad size");
    if (!prune_mask || !workspace || !counts_host) return mgr_fail(MGR_EINVAL, "mgr_prune_plan: null pointer");
    if (workspace_bytes < mgr_densify_workspace_bytes(N)) return mgr_fail(MGR_ENOMEM, "mgr_prune_plan: workspace too small");
    hipStream_t stream = (hipStream_t)stream_;
    const DensLayout L = dens_layout(N);
    char* ws = (char*)workspace;
    const int nblk = (N + 1023) / 1024;
    hipLaunchKernelGGL(k_prune_flags, dim3(nblk), dim3(1024), 0, stream, N, prune_mask, (uint32_t*)(ws + L.flags),
                       (uint4*)(ws + L.sums));
    hipLaunchKernelGGL(k_dens_map, dim3(nblk), dim3(1024), 0, stream, N, nblk, (const uint32_t*)(ws + L.flags),
                       (const uint4*)(ws + L.sums), (uint32_t*)(ws + L.map), (uint32_t*)(ws + L.aux),
                       (uint32_t*)(ws + L.counts));
    MGR_LAUNCH_CHECK("k_prune_map", stream, 0);
    uint32_t h[5];
    MGR_HIP(hipMemcpyAsync(h, ws + L.counts, sizeof(h), hipMemcpyDeviceToHost, stream));
    MGR_HIP(hipStreamSynchronize(stream));
    for (int k = 0; k < 5; ++k) counts_host[k] = h[k];
    return MGR_OK;
}

// dst[o, :] = src[map[o] & 0x3FFFFFFF, :] for the M rows of the plan in `workspace` (rows of `width` 4-byte words)
__global__ __launch_bounds__(256) void k_gather_rows(uint32_t M, int width, const uint32_t* __restrict__ map,
                                                     const uint32_t* __restrict__ src, uint32_t* __restrict__ dst) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)M * width) return;
    const uint32_t o = (uint32_t)(e / width), c = (uint32_t)(e % width);
    dst[e] = src[(size_t)(map[o] & 0x3FFFFFFFu) * width + c];
}

extern "C" int mgr_gather_rows(int N, int64_t M, const void* workspace, const void* src, void* dst, int width,
                               void* stream_) {
    if (N <= 0 || M < 0 || M > 2ll * N || width <= 0) return mgr_fail(MGR_EINVAL, "mgr_gather_rows: bad sizes");
    if (M == 0) return MGR_OK;
    if (!workspace || !src || !dst) return mgr_fail(MGR_EINVAL, "mgr_gather_rows: null pointer");
    hipStream_t stream = (hipStream_t)stream_;
    const DensLayout L = dens_layout(N);
    const size_t total = (size_t)M * width;
    hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (uint32_t)M, width,
                       (const uint32_t*)((const char*)workspace + L.map), (const uint32_t*)src, (uint32_t*)dst);
    MGR_LAUNCH_CHECK("k_gather_rows", stream, 0);
    return MGR_OK;
}

// ---------------------------------------------------------------------------
// densify_and_prune, apply: one thread per new row
// ---------------------------------------------------------------------------
struct DensTensors {   // the six leaves in group order + moments, old and new
    const float* p[6];
    const float* m[6];
    const float* v[6];
    float* np[6];
    float* nm[6];
    float* nv[6];
};

template <int WIDTH>
__device__ __forceinline__ void dens_copy_row(const DensTensors& T, int k, uint32_t src, uint32_t dst, bool moments) {
#pragma unroll
    for (int e = 0; e < WIDTH; ++e) {
        T.np[k][(size_t)dst * WIDTH + e] = T.p[k][(size_t)src * WIDTH + e];
        T.nm[k][(size_t)dst * WIDTH + e] = moments ? T.m[k][(size_t)src * WIDTH + e] : 0.f;
        T.nv[k][(size_t)dst * WIDTH + e] = moments ? T.v[k][(size_t)src * WIDTH + e] : 0.f;
    }
}

__global__ __launch_bounds__(256) void k_dens_apply(uint32_t M, uint32_t n_sel, const uint32_t* __restrict__ map,
                                                    const uint32_t* __restrict__ aux, DensTensors T,
                                                    const float* __restrict__ skin, float* __restrict__ new_skin, int B,
                                                    const float* __restrict__ noise) {
    const uint32_t o = blockIdx.x * 256 + threadIdx.x;
    if (o >= M) return;
    const uint32_t w = map[o], src = w & 0x3FFFFFFFu, kind = w >> 30;
    const bool orig = kind == 0;
    // group order of training_setup (gaussian.py:133-140): xyz 3, f_dc 3, f_rest 45, opacity 1, scaling 3, rotation 4
    dens_copy_row<3>(T, 1, src, o, orig);
    dens_copy_row<45>(T, 2, src, o, orig);
    dens_copy_row<1>(T, 3, src, o, orig);
    dens_copy_row<4>(T, 5, src, o, orig);
    if (kind < 2) {
        dens_copy_row<3>(T, 0, src, o, orig);
        dens_copy_row<3>(T, 4, src, o, orig);
    } else {
        // split child (gaussian.py:264-268): xyz = R(q) (noise * exp(s)) + xyz, scaling = log(exp(s) / 1.6)
        const float* nz = noise + ((size_t)(kind - 2) * n_sel + aux[o]) * 3;
        const float s0 = expf(T.p[4][(size_t)src * 3]), s1 = expf(T.p[4][(size_t)src * 3 + 1]),
                    s2 = expf(T.p[4][(size_t)src * 3 + 2]);
        const float a0 = nz[0] * s0, a1 = nz[1] * s1, a2 = nz[2] * s2;
        const float* qr = T.p[5] + (size_t)src * 4;
        const float nrm = sqrtf(qr[0] * qr[0] + qr[1] * qr[1] + qr[2] * qr[2] + qr[3] * qr[3]);
        const float r = qr[0] / nrm, x = qr[1] / nrm, y = qr[2] / nrm, z = qr[3] / nrm;
        const float R[9] = {1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
                            2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                            2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)};
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            T.np[0][(size_t)o * 3 + e] = (R[3 * e] * a0 + R[3 * e + 1] * a1 + R[3 * e + 2] * a2) + T.p[0][(size_t)src * 3 + e];
            T.nm[0][(size_t)o * 3 + e] = 0.f;
            T.nv[0][(size_t)o * 3 + e] = 0.f;
        }
        T.np[4][(size_t)o * 3 + 0] = logf(s0 / 1.6f);
        T.np[4][(size_t)o * 3 + 1] = logf(s1 / 1.6f);
        T.np[4][(size_t)o * 3 + 2] = logf(s2 / 1.6f);
#pragma unroll
        for (int e = 0; e < 3; ++e) T.nm[4][(size_t)o * 3 + e] = T.nv[4][(size_t)o * 3 + e] = 0.f;
    }
    if (skin)
        for (int b = 0; b < B; ++b) new_skin[(size_t)o * B + b] = skin[(size_t)src * B + b];
}

extern "C" int mgr_densify_apply(int N, int64_t M, int64_t n_selected, const void* workspace, const float* const* params,
                                 const float* const* exp_avg, const float* const* exp_avg_sq, float* const* new_params,
                                 float* const* new_exp_avg, float* const* new_exp_avg_sq, const float* skin,
                                 float* new_skin, int B, const float* noise, void* stream_) {
    if (N <= 0 || M < 0 || n_selected < 0 || M > 2ll * N) return mgr_fail(MGR_EINVAL, "mgr_densify_apply: bad sizes");
    if (M == 0) return MGR_OK;
    if (!workspace || !params || !exp_avg || !exp_avg_sq || !new_params || !new_exp_avg || !new_exp_avg_sq)
        return mgr_fail(MGR_EINVAL, "mgr_densify_apply: null pointer");
    if (n_selected > 0 && !noise) return mgr_fail(MGR_EINVAL, "mgr_densify_apply: noise is required when Gaussians split");
    if ((skin != nullptr) != (new_skin != nullptr) || (skin && B <= 0))
        return mgr_fail(MGR_EINVAL, "mgr_densify_apply: skin / new_skin / B mismatch");
    DensTensors T;
    for (int k = 0; k < 6; ++k) {
        if (!params[k] || !exp_avg[k] || !exp_avg_sq[k] || !new_params[k] || !new_exp_avg[k] || !new_exp_avg_sq[k])
            return mgr_fail(MGR_EINVAL, "mgr_densify_apply: null tensor");
        T.p[k] = params[k]; T.m[k] = exp_avg[k]; T.v[k] = exp_avg_sq[k];
        T.np[k] = new_params[k]; T.nm[k] = new_exp_avg[k]; T.nv[k] = new_exp_avg_sq[k];
    }
    hipStream_t stream = (hipStream_t)stream_;
    const DensLayout L = dens_layout(N);
    const char* ws = (const char*)workspace;
    {
        MGR_PROF("k_dens_apply", stream);
        hipLaunchKernelGGL(k_dens_apply, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, stream, (uint32_t)M,
                           (uint32_t)n_selected, (const uint32_t*)(ws + L.map), (const uint32_t*)(ws + L.aux), T, skin,
                           new_skin, B, noise);
    }
    MGR_LAUNCH_CHECK("k_dens_apply", stream, 0);
    return MGR_OK;
}

// ---------------------------------------------------------------------------
// isotropic regulariser (src/modules/base.py:349-356):
//   L = mean_n (min_j s_nj / (max_j s_nj + 1e-8) - condition_number)^2,  s = exp(log_scale)
// value and gradient w.r.t. log_scale in one pass.  torch.max / torch.min send the gradient to the
// first index attaining the extreme; so does this.  Per-workgroup sums + a fold: reproducible value.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_isotropic_reg(int N, const float* __restrict__ log_scale, float cond,
                                                       float grad_scale, float* __restrict__ d_log_scale, int accumulate,
                                                       float* __restrict__ partial) {
    __shared__ float s_red[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    float term = 0.f;
    if (i < N) {
        const float s[3] = {expf(log_scale[3 * i]), expf(log_scale[3 * i + 1]), expf(log_scale[3 * i + 2])};
        int jmax = 0, jmin = 0;
#pragma unroll
        for (int j = 1; j < 3; ++j) {
            if (s[j] > s[jmax]) jmax = j;
            if (s[j] < s[jmin]) jmin = j;
        }
        const float den = s[jmax] + 1e-8f;
        const float r = s[jmin] / den - cond;
        term = r * r;
        const float dmin = 2.f * r / den, dmax = -2.f * r * s[jmin] / (den * den);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float g = grad_scale * ((j == jmin ? dmin : 0.f) + (j == jmax ? dmax : 0.f)) * s[j];
            d_log_scale[3 * i + j] = accumulate ? d_log_scale[3 * i + j] + g : g;
        }
    }
    term = mgr_wave_sum63(term);
    if ((threadIdx.x & 63) == 63) s_red[threadIdx.x >> 6] = term;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

__global__ __launch_bounds__(1024) void k_fold_sum(int n, const float* __restrict__ partial, float scale,
                                                   float* __restrict__ out) {
    __shared__ double s_a[16];
    double a = 0.0;
    for (int k = threadIdx.x; k < n; k += 1024) a += (double)partial[k];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) a += __shfl_xor(a, d, 64);
    if ((threadIdx.x & 63) == 0) s_a[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int k = 0; k < 16; ++k) t += s_a[k];
        out[0] = (float)(t * (double)scale);
    }
}

extern "C" size_t mgr_isotropic_reg_workspace_bytes(int N) { return N > 0 ? (size_t)((N + 255) / 256) * 4 : 0; }

extern "C" int mgr_isotropic_reg(int N, const float* log_scale, float condition_number, float weight, float* d_log_scale,
                                 int accumulate, float* loss, void* workspace, size_t workspace_bytes, void* stream_) {
    if (N <= 0) return mgr_fail(MGR_EINVAL, "mgr_isotropic_reg: bad size");
    if (!log_scale || !d_log_scale || !loss || !workspace) return mgr_fail(MGR_EINVAL, "mgr_isotropic_reg: null pointer");
    if (workspace_bytes < mgr_isotropic_reg_workspace_bytes(N)) return mgr_fail(MGR_ENOMEM, "mgr_isotropic_reg: workspace too small");
    hipStream_t stream = (hipStream_t)stream_;
    const int nblk = (N + 255) / 256;
    hipLaunchKernelGGL(k_isotropic_reg, dim3(nblk), dim3(256), 0, stream, N, log_scale, condition_number, weight / (float)N,
                       d_log_scale, accumulate, (float*)workspace);
    hipLaunchKernelGGL(k_fold_sum, dim3(1), dim3(1024), 0, stream, nblk, (const float*)workspace, weight / (float)N, loss);
    MGR_LAUNCH_CHECK("k_isotropic_reg", stream, 0);
    return MGR_OK;
}
