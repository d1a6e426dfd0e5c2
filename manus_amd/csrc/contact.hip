// Hand <-> object contact distance for gfx950: for every point of pt1 the distance to, and the index
// of, its nearest point of pt2.
//
// Replaces get_contact_dist (taichi kernel calculate_distances) and get_contact_map (chunked
// torch.cdist(...).min) of /root/reference/src/utils/gaussian_utils.py:514-549, used by get_cmap
// (:571-577) for the contact renders of src/modules/composite.py:143-214.
// Semantics of the reference loop, kept exactly: fp32, dist = sqrt(dx^2 + dy^2 + dz^2) summed in that
// order, a candidate replaces the running minimum only when its *rooted* distance is strictly
// smaller (so among equal distances the lowest index wins), min_dist starts at 1e9.
//
// Brute force, tiled: a workgroup holds 512 points of pt1 in registers (two per thread) and streams
// pt2 through LDS 1024 points at a time, every LDS read (one broadcast ds_read_b128) serving both
// points of all 256 threads.  The square root is only taken for candidates that already beat the
// running minimum of the squared distance (a necessary condition).  pt2 is split into S segments
// across workgroups to fill the chip; a second kernel merges the S partial results in segment
// order with the same strict comparison, which preserves the lowest-index rule.
#include "mgr_common.h"

#define CT_T 256
#define CT_TILE 1024

__device__ __forceinline__ void ct_try(float px, float py, float pz, const float4 q, uint32_t j, float& best2, float& best,
                                       uint32_t& idx) {
#pragma clang fp contract(off)   // the reference's sum is three rounded squares added in order, not FMAs
    const float dx = px - q.x, dy = py - q.y, dz = pz - q.z;
    const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
    const float d2 = (xx + yy) + zz;
    if (d2 < best2) {
        const float s = (float)sqrt((double)d2);  // correctly rounded fp32 root (v_sqrt_f32 alone is 1 ulp); rare path
        if (s < best) {
            best = s;
            idx = j;
        }
        best2 = d2;  // still necessary for any later strict improvement of the rooted distance
    }
}

__global__ __launch_bounds__(CT_T) void k_contact_partial(int N1, const float* __restrict__ pt1, int N2,
                                                          const float* __restrict__ pt2, int seg_len,
                                                          float* __restrict__ part_dist, uint32_t* __restrict__ part_idx) {
    __shared__ float4 s_pt[CT_TILE];
    const int tid = threadIdx.x;
    const int i0 = (blockIdx.x * CT_T + tid) * 2, i1 = i0 + 1;
    const int j_begin = blockIdx.y * seg_len, j_end = min(N2, j_begin + seg_len);
    float ax = 0.f, ay = 0.f, az = 0.f, bx = 0.f, by = 0.f, bz = 0.f;
    if (i0 < N1) { ax = pt1[3 * (size_t)i0]; ay = pt1[3 * (size_t)i0 + 1]; az = pt1[3 * (size_t)i0 + 2]; }
    if (i1 < N1) { bx = pt1[3 * (size_t)i1]; by = pt1[3 * (size_t)i1 + 1]; bz = pt1[3 * (size_t)i1 + 2]; }
    float a_best = 1e9f, a_best2 = 1e18f, b_best = 1e9f, b_best2 = 1e18f;
    uint32_t a_idx = 0, b_idx = 0;
    for (int j0 = j_begin; j0 < j_end; j0 += CT_TILE) {
        const int n = min(CT_TILE, j_end - j0);
        __syncthreads();
        for (int k = tid; k < n; k += CT_T) {
            const float* p = pt2 + 3 * (size_t)(j0 + k);
            s_pt[k] = make_float4(p[0], p[1], p[2], 0.f);
        }
        __syncthreads();
#pragma unroll 4
        for (int k = 0; k < n; ++k) {
            const float4 q = s_pt[k];
            ct_try(ax, ay, az, q, (uint32_t)(j0 + k), a_best2, a_best, a_idx);
            ct_try(bx, by, bz, q, (uint32_t)(j0 + k), b_best2, b_best, b_idx);
        }
    }
    const size_t row = (size_t)blockIdx.y * N1;
    if (i0 < N1) { part_dist[row + i0] = a_best; part_idx[row + i0] = a_idx; }
    if (i1 < N1) { part_dist[row + i1] = b_best; part_idx[row + i1] = b_idx; }
}

__global__ __launch_bounds__(256) void k_contact_merge(int N1, int S, const float* __restrict__ part_dist,
                                                       const uint32_t* __restrict__ part_idx, float* __restrict__ out_dist,
                                                       int32_t* __restrict__ out_idx) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N1) return;
    float best = 1e9f;
    uint32_t idx = 0;
    for (int s = 0; s < S; ++s) {  // ascending segments = ascending indices: strict < keeps the lowest index of a tie
        const float d = part_dist[(size_t)s * N1 + i];
        if (d < best) {
            best = d;
            idx = part_idx[(size_t)s * N1 + i];
        }
    }
    out_dist[i] = best;
    if (out_idx) out_idx[i] = (int32_t)idx;
}

static int ct_segments(int N1, int N2) {
    const long long blocks_i = ((long long)N1 + 2 * CT_T - 1) / (2 * CT_T);
    long long S = (2048 + blocks_i - 1) / blocks_i;            // aim at ~2k workgroups
    const long long max_s = ((long long)N2 + CT_TILE - 1) / CT_TILE;  // at least one LDS tile per segment
    if (S > max_s) S = max_s;
    if (S > 64) S = 64;
    if (S < 1) S = 1;
    return (int)S;
}

extern "C" size_t mgr_contact_workspace_bytes(int N1, int N2) {
    if (N1 <= 0 || N2 < 0) return 0;
    return (size_t)ct_segments(N1, N2) * (size_t)N1 * 8 + 256;
}

extern "C" int mgr_contact_dist(int N1, const float* pt1, int N2, const float* pt2, float* out_dist, int32_t* out_idx,
                                void* workspace, size_t workspace_bytes, void* stream_) {
    if (N1 < 0 || N2 < 0) return mgr_fail(MGR_EINVAL, "mgr_contact_dist: bad sizes");
    if (N1 == 0) return MGR_OK;
    if (!pt1 || !out_dist || (N2 > 0 && !pt2) || !workspace) return mgr_fail(MGR_EINVAL, "mgr_contact_dist: null pointer");
    if (workspace_bytes < mgr_contact_workspace_bytes(N1, N2)) return mgr_fail(MGR_ENOMEM, "mgr_contact_dist: workspace too small");
    hipStream_t stream = (hipStream_t)stream_;
    const int S = ct_segments(N1, N2);
    int seg_len = (N2 + S - 1) / S;
    seg_len = (seg_len + CT_TILE - 1) / CT_TILE * CT_TILE;
    if (seg_len < CT_TILE) seg_len = CT_TILE;
    float* pd = (float*)workspace;
    uint32_t* pi = (uint32_t*)((char*)workspace + (size_t)S * N1 * 4);
    {
        MGR_PROF("k_contact_partial", stream);
        hipLaunchKernelGGL(k_contact_partial, dim3((N1 + 2 * CT_T - 1) / (2 * CT_T), S), dim3(CT_T), 0, stream, N1, pt1, N2, pt2,
                           seg_len, pd, pi);
    }
    hipLaunchKernelGGL(k_contact_merge, dim3((N1 + 255) / 256), dim3(256), 0, stream, N1, S, (const float*)pd,
                       (const uint32_t*)pi, out_dist, out_idx);
    MGR_LAUNCH_CHECK("k_contact", stream, 0);
    return MGR_OK;
}
