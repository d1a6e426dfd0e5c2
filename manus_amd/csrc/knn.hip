// simple-knn replacement for gfx950: out[i] = mean of the three smallest squared
// distances from point i to the other points (exclusion by index).
//
// Replaces simple_knn._C.distCUDA2 (/root/reference/src/models/gaussian.py:4,110;
// contract: SURVEY.md Appendix B).  The upstream extension sorts Morton codes and
// prunes 1024-point boxes; here points are counting-sorted into a uniform grid
// sized from the bounding box (about 4 points per cell) and every query walks
// Chebyshev rings of cells until the third-best distance is provably final —
// the result equals brute force.
#include "mgr_common.h"

struct KnnParams {
    uint32_t bmin[3], bmax[3];  // ordered-uint encoded bbox
    float lo[3], h, inv_h;
    int dims[3];
    int ncells;
};

__device__ __forceinline__ uint32_t f2ord(float f) {
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t o) {
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

__global__ void k_knn_init(KnnParams* P) {
    for (int k = 0; k < 3; ++k) {
        P->bmin[k] = 0xFFFFFFFFu;
        P->bmax[k] = 0u;
    }
}

__global__ __launch_bounds__(256) void k_knn_bbox(int N, const float* __restrict__ xyz, KnnParams* P) {
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < N; i += gridDim.x * 256) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float v = xyz[3 * i + k];
            mn[k] = fminf(mn[k], v);
            mx[k] = fmaxf(mx[k], v);
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            mn[k] = fminf(mn[k], __shfl_xor(mn[k], d, 64));
            mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], d, 64));
        }
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (mn[k] <= mx[k]) {
                atomicMin(&P->bmin[k], f2ord(mn[k]));
                atomicMax(&P->bmax[k], f2ord(mx[k]));
            }
        }
    }
}

__global__ void k_knn_setup(int N, int max_cells, KnnParams* P) {
    float ext[3], vol = 1.f;
    for (int k = 0; k < 3; ++k) {
        P->lo[k] = ord2f(P->bmin[k]);
        ext[k] = ord2f(P->bmax[k]) - P->lo[k];
        if (!(ext[k] > 0.f)) ext[k] = 0.f;
    }
    const float emax = fmaxf(ext[0], fmaxf(ext[1], ext[2]));
    for (int k = 0; k < 3; ++k) vol *= fmaxf(ext[k], 1e-3f * emax);
    const float target = fmaxf(1.f, (float)N * 0.25f);
    float h = (vol > 0.f) ? cbrtf(vol / target) : 1.f;
    if (!(h > 0.f) || !isfinite(h)) h = 1.f;
    int d[3];
    for (int it = 0; it < 200; ++it) {
        long long tot = 1;
        for (int k = 0; k < 3; ++k) {
            float c = floorf(ext[k] / h) + 1.f;
            d[k] = (int)fminf(fmaxf(c, 1.f), 2048.f);
            tot *= d[k];
        }
        if (tot <= max_cells) break;
        h *= 1.2f;
    }
    // final safety: collapse to one cell
    if ((long long)d[0] * d[1] * d[2] > max_cells) d[0] = d[1] = d[2] = 1, h = fmaxf(emax, 1e-30f) * 2.f;
    P->h = h;
    P->inv_h = 1.f / h;
    for (int k = 0; k < 3; ++k) P->dims[k] = d[k];
    P->ncells = d[0] * d[1] * d[2];
}

__device__ __forceinline__ void knn_cell(const KnnParams* P, float x, float y, float z, int c[3]) {
    c[0] = min(P->dims[0] - 1, max(0, (int)((x - P->lo[0]) * P->inv_h)));
    c[1] = min(P->dims[1] - 1, max(0, (int)((y - P->lo[1]) * P->inv_h)));
    c[2] = min(P->dims[2] - 1, max(0, (int)((z - P->lo[2]) * P->inv_h)));
}

__global__ __launch_bounds__(256) void k_knn_count(int N, const float* __restrict__ xyz,
                                                   const KnnParams* P, uint32_t* __restrict__ cnt) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    int c[3];
    knn_cell(P, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], c);
    atomicAdd(&cnt[(c[2] * P->dims[1] + c[1]) * P->dims[0] + c[0]], 1u);
}

// single-block exclusive scan; start[ncells] = N; cursor zeroed
__global__ __launch_bounds__(1024) void k_knn_scan(const KnnParams* P, const uint32_t* __restrict__ cnt,
                                                   uint32_t* __restrict__ start,
                                                   uint32_t* __restrict__ cursor) {
    __shared__ uint32_t s_part[1024];
    const int n = P->ncells, tid = threadIdx.x;
    const int per = (n + 1023) / 1024;
    const int b = min(n, tid * per), e = min(n, b + per);
    uint32_t s = 0;
    for (int k = b; k < e; ++k) s += cnt[k];
    s_part[tid] = s;
    __syncthreads();
    if (tid == 0) {
        uint32_t run = 0;
        for (int k = 0; k < 1024; ++k) {
            const uint32_t t = s_part[k];
            s_part[k] = run;
            run += t;
        }
        start[n] = run;
    }
    __syncthreads();
    uint32_t run = s_part[tid];
    for (int k = b; k < e; ++k) {
        start[k] = run;
        cursor[k] = 0;
        run += cnt[k];
    }
}

__global__ __launch_bounds__(256) void k_knn_scatter(int N, const float* __restrict__ xyz,
                                                     const KnnParams* P,
                                                     const uint32_t* __restrict__ start,
                                                     uint32_t* __restrict__ cursor,
                                                     float4* __restrict__ sorted) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    int c[3];
    knn_cell(P, x, y, z, c);
    const int cell = (c[2] * P->dims[1] + c[1]) * P->dims[0] + c[0];
    const uint32_t pos = start[cell] + atomicAdd(&cursor[cell], 1u);
    sorted[pos] = make_float4(x, y, z, __int_as_float(i));
}

__global__ __launch_bounds__(256) void k_knn_query(int N, const float* __restrict__ xyz,
                                                   const KnnParams* P,
                                                   const uint32_t* __restrict__ start,
                                                   const float4* __restrict__ sorted,
                                                   float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    int c[3];
    knn_cell(P, x, y, z, c);
    const int dx_ = P->dims[0], dy_ = P->dims[1], dz_ = P->dims[2];
    const float h = P->h;
    float b0 = INFINITY, b1 = INFINITY, b2 = INFINITY;
    const int rmax = max(dx_, max(dy_, dz_));
    for (int r = 0; r <= rmax; ++r) {
        const int z0 = max(0, c[2] - r), z1 = min(dz_ - 1, c[2] + r);
        const int y0 = max(0, c[1] - r), y1 = min(dy_ - 1, c[1] + r);
        const int x0 = max(0, c[0] - r), x1 = min(dx_ - 1, c[0] + r);
        for (int cz = z0; cz <= z1; ++cz)
            for (int cy = y0; cy <= y1; ++cy) {
                const bool shell_zy = (abs(cz - c[2]) == r) || (abs(cy - c[1]) == r);
                // when (cz,cy) is on the shell the whole x-run belongs to ring r,
                // otherwise only its two end cells do
                const int step = shell_zy ? 1 : max(1, 2 * r);
                for (int cx = shell_zy ? x0 : c[0] - r; cx <= (shell_zy ? x1 : c[0] + r); cx += step) {
                    if (cx < 0 || cx >= dx_) continue;
                    const int cell = (cz * dy_ + cy) * dx_ + cx;
                    const uint32_t s = start[cell], e = start[cell + 1];
                    for (uint32_t k = s; k < e; ++k) {
                        const float4 q = sorted[k];
                        if (__float_as_int(q.w) == i) continue;
                        const float ddx = q.x - x, ddy = q.y - y, ddz = q.z - z;
                        const float d = ddx * ddx + ddy * ddy + ddz * ddz;
                        if (d < b2) {
                            if (d < b1) {
                                b2 = b1;
                                if (d < b0) { b1 = b0; b0 = d; } else { b1 = d; }
                            } else {
                                b2 = d;
                            }
                        }
                    }
                }
            }
        // every unvisited point lies at least r*h away along some axis
        const float bound = (float)r * h * 0.9999f;  // margin for cell-assignment rounding
        if (b2 <= bound * bound) break;
    }
    out[i] = (b0 + b1 + b2) / 3.0f;
}

static inline int knn_max_cells(int N) {
    int m = N;
    if (m < 64) m = 64;
    if (m > (1 << 21)) m = 1 << 21;
    return m;
}

extern "C" size_t mgr_knn3_workspace_bytes(int N) {
    const size_t mc = (size_t)knn_max_cells(N > 0 ? N : 1);
    return mgr_align(sizeof(KnnParams)) + mgr_align(mc * 4) + mgr_align((mc + 1) * 4) + mgr_align(mc * 4) +
           mgr_align((size_t)(N > 0 ? N : 1) * 16);
}

extern "C" int mgr_knn3_mean_dist2(int N, const float* xyz, float* out, void* workspace,
                                   size_t workspace_bytes, void* stream_) {
    if (N < 0) return mgr_fail(MGR_EINVAL, "mgr_knn3_mean_dist2: bad sizes");
    if (N == 0) return MGR_OK;
    if (!xyz || !out || !workspace) return mgr_fail(MGR_EINVAL, "mgr_knn3_mean_dist2: null pointer");
    if (workspace_bytes < mgr_knn3_workspace_bytes(N))
        return mgr_fail(MGR_ENOMEM, "mgr_knn3_mean_dist2: workspace too small");
    hipStream_t stream = (hipStream_t)stream_;
    const size_t mc = (size_t)knn_max_cells(N);
    char* ws = (char*)workspace;
    KnnParams* P = (KnnParams*)ws;
    size_t o = mgr_align(sizeof(KnnParams));
    uint32_t* cnt = (uint32_t*)(ws + o);     o += mgr_align(mc * 4);
    uint32_t* start = (uint32_t*)(ws + o);   o += mgr_align((mc + 1) * 4);
    uint32_t* cursor = (uint32_t*)(ws + o);  o += mgr_align(mc * 4);
    float4* sorted = (float4*)(ws + o);
    MGR_HIP(hipMemsetAsync(cnt, 0, mc * 4, stream));
    { MGR_PROF("k_knn_init", stream); hipLaunchKernelGGL(k_knn_init, dim3(1), dim3(1), 0, stream, P); }
    int blocks = (N + 255) / 256;
    { MGR_PROF("k_knn_bbox", stream); hipLaunchKernelGGL(k_knn_bbox, dim3(blocks > 1024 ? 1024 : blocks), dim3(256), 0, stream, N, xyz, P); }
    { MGR_PROF("k_knn_setup", stream); hipLaunchKernelGGL(k_knn_setup, dim3(1), dim3(1), 0, stream, N, (int)mc, P); }
    { MGR_PROF("k_knn_count", stream); hipLaunchKernelGGL(k_knn_count, dim3(blocks), dim3(256), 0, stream, N, xyz, P, cnt); }
    { MGR_PROF("k_knn_scan", stream); hipLaunchKernelGGL(k_knn_scan, dim3(1), dim3(1024), 0, stream, P, cnt, start, cursor); }
    { MGR_PROF("k_knn_scatter", stream); hipLaunchKernelGGL(k_knn_scatter, dim3(blocks), dim3(256), 0, stream, N, xyz, P, start, cursor, sorted); }
    { MGR_PROF("k_knn_query", stream); hipLaunchKernelGGL(k_knn_query, dim3(blocks), dim3(256), 0, stream, N, xyz, P, start, sorted, out); }
    MGR_LAUNCH_CHECK("knn3", stream, 0);
    return MGR_OK;
}
